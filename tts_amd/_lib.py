"""ctypes binding of ``libfuzzyhip.so`` (C ABI: ``include/fuzzyhip.h``).

The library is the product: there is no Python or CPU fallback.  Importing this module without the
built library, or calling a scoring entry point without a gfx950 device, raises."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FUZZYHIP_LIB: another build of the same library (kernel experiments, A/B runs on one box)
LIB_PATH = os.environ.get("FUZZYHIP_LIB") or os.path.join(_HERE, "libfuzzyhip.so")

TOKEN_SET, RATIO, TOKEN_SORT, PARTIAL, LEVENSHTEIN = 0, 1, 2, 3, 4
SCORERS = {"token_set": TOKEN_SET, "ratio": RATIO, "token_sort": TOKEN_SORT, "partial": PARTIAL,
           "levenshtein": LEVENSHTEIN}
ORDER_BY_INDEX = 0x100
MAX_K = 1024
IDX_BITS = 40
IDX_MASK = (1 << IDX_BITS) - 1
COMM_ID_BYTES = 128

c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_u8p = ctypes.POINTER(ctypes.c_uint8)


class FzInfo(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("base_index", ctypes.c_uint64), ("vocab_size", ctypes.c_uint64),
                ("alphabet_size", ctypes.c_uint32), ("n_tiles", ctypes.c_uint32),
                ("max_len", ctypes.c_uint32), ("max_tokens", ctypes.c_uint32),
                ("device_bytes", ctypes.c_uint64), ("total_symbols", ctypes.c_uint64),
                ("total_tokens", ctypes.c_uint64), ("device", ctypes.c_int32)]


class FzTiming(ctypes.Structure):
    _fields_ = [("score_ms", ctypes.c_float), ("select_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("score_launches", ctypes.c_uint32), ("select_launches", ctypes.c_uint32),
                ("pairs", ctypes.c_uint64), ("host_prep_ms", ctypes.c_double),
                ("fused", ctypes.c_uint32), ("fused_fallback_chunks", ctypes.c_uint32),
                ("exchange_ms", ctypes.c_float), ("reserved", ctypes.c_uint32)]


# every symbol include/fuzzyhip.h declares: name -> (restype, argtypes)
_SIGNATURES = {
    "fz_version": (ctypes.c_int, []),
    "fz_last_error": (ctypes.c_char_p, []),
    "fz_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "fz_corpus_create": (ctypes.c_int, [c_u32p, c_u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64,
                                        ctypes.POINTER(ctypes.c_void_p)]),
    "fz_corpus_save": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "fz_corpus_load": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]),
    "fz_corpus_destroy": (None, [ctypes.c_void_p]),
    "fz_corpus_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(FzInfo)]),
    "fz_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "fz_set_workspace_limit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64]),
    "fz_query_topk": (ctypes.c_int, [ctypes.c_void_p, c_u32p, c_u64p, ctypes.c_uint32, ctypes.c_int,
                                     ctypes.c_int32, ctypes.c_uint32, c_u64p, c_i32p, c_u32p]),
    "fz_query_topk_noalloc": (ctypes.c_int, [ctypes.c_void_p, c_u32p, c_u64p, ctypes.c_uint32, ctypes.c_int,
                                             ctypes.c_int32, ctypes.c_uint32, c_u64p, c_i32p, c_u32p]),
    "fz_query_topk_keys": (ctypes.c_int, [ctypes.c_void_p, c_u32p, c_u64p, ctypes.c_uint32, ctypes.c_int,
                                          ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_int]),
    "fz_query_topk_pair": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_u32p, c_u64p,
                                          ctypes.c_void_p, ctypes.c_int, c_u32p, c_u64p, ctypes.c_uint32,
                                          ctypes.POINTER(ctypes.c_uint16), ctypes.c_uint32, ctypes.c_uint32,
                                          c_u64p, c_i32p, c_u32p]),
    "fz_merge_topk": (ctypes.c_int, [c_u64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, c_u64p,
                                     c_i32p, c_u32p]),
    "fz_score_all": (ctypes.c_int, [ctypes.c_void_p, c_u32p, c_u64p, ctypes.c_uint32, ctypes.c_int, c_u8p]),
    "fz_distance_all": (ctypes.c_int, [ctypes.c_void_p, c_u32p, c_u64p, ctypes.c_uint32, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_uint16)]),
    "fz_distance_batch": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_u32p, c_u64p, c_u32p, c_u64p,
                                         ctypes.c_uint64, c_u32p]),
    "fz_last_timing": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(FzTiming)]),
    "fz_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "fz_comm_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_void_p)]),
    "fz_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "fz_query_topk_sharded": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_u32p, c_u64p, ctypes.c_uint32,
                                             ctypes.c_int, ctypes.c_int32, ctypes.c_uint32, c_u64p, c_i32p, c_u32p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libfuzzyhip.so and bind every entry point; raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C tts_amd/csrc` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`); tts_amd has no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


class FuzzyHipError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        msg = load().fz_last_error().decode("utf-8", "replace")
        raise FuzzyHipError(f"libfuzzyhip error {rc}: {msg}")
