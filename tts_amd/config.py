"""Counterpart of the reference's ``src/config.py`` for the two constants the hot path imports
(reference ``src/fuzzy_search.py:1``).  That module is missing from the reference tree, so the values
are those of the direct ancestor of the hot loop (reference ``old_files/app_old_new.py:514,542``:
``threshold=0.35`` and ``[:5]``); SURVEY.md section 5 records the lineage.  Both are runtime parameters of
the C ABI, so an application that ships its own ``config`` simply passes its values through."""

TOP_N_MATCHES_INT = 5
FUZZY_SEARCH_THRESHOLD_FLOAT = 0.35
