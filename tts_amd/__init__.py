"""tts_amd -- MI355X-native implementation of the candidate-scoring loop of the reference's
``src/fuzzy_search.py``.  ``FuzzySearch`` is the drop-in class; ``Corpus`` the
handle-level API over ``libfuzzyhip.so``."""
from .config import FUZZY_SEARCH_THRESHOLD_FLOAT, TOP_N_MATCHES_INT
from .corpus import Corpus, distance_batch, merge_topk
from .fuzzy_search import FuzzySearch
from .weighted_search import WeightedFuzzySearch
from .advanced_matcher import AdvancedFuzzyMatcher

__all__ = ["FuzzySearch", "WeightedFuzzySearch", "AdvancedFuzzyMatcher", "Corpus", "merge_topk", "distance_batch", "TOP_N_MATCHES_INT", "FUZZY_SEARCH_THRESHOLD_FLOAT"]
