"""ctypes binding of ``libfuzzyhip.so`` (C ABI: ``include/fuzzyhip.h``).

The library is the product: there is no Python or CPU fallback.  Importing this module without the
built library, or calling a scoring entry point without a gfx950 device, raises."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FUZZYHIP_LIB: another build of the same library (kernel experiments, A/B runs on one box)
LIB_PATH = os.environ.get("FUZZYHIP_LIB") or os.path.join(_HERE, "libfuzzyhip.so")

TOKEN_SET, RATIO, TOKEN_SORT, PARTIAL = 0, 1, 2, 3
SCORERS = {"token_set": TOKEN_SET, "ratio": RATIO, "token_sort": TOKEN_SORT, "partial": PARTIAL}
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
                ("total_tokens", ctypes.c_uint64), ("device", ctypes.c_int32),
                ("n_packed_pairs", ctypes.c_uint32), ("n_stream1_tiles", ctypes.c_uint32),
                ("n_stream2_tiles", ctypes.c_uint32), ("n_general_tiles", ctypes.c_uint32)]


class FzTiming(ctypes.Structure):
    _fields_ = [("score_ms", ctypes.c_float), ("select_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("score_launches", ctypes.c_uint32), ("select_launches", ctypes.c_uint32),
                ("pairs", ctypes.c_uint64), ("host_prep_ms", ctypes.c_double),
                ("fused", ctypes.c_uint32), ("fused_fallback_chunks", ctypes.c_uint32),
                ("exchange_ms", ctypes.c_float), ("fused_fallback_rows", ctypes.c_uint32)]


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
    "fz_query_topk_submit": (ctypes.c_int, [ctypes.c_void_p, c_u32p, c_u64p, ctypes.c_uint32, ctypes.c_int,
                                            ctypes.c_int32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int)]),
    "fz_query_topk_wait": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_u64p, c_i32p, c_u32p]),
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
    "fz_runtime_paths": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]),
    "fz_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "fz_comm_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_void_p)]),
    "fz_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "fz_query_topk_sharded": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_u32p, c_u64p, ctypes.c_uint32,
                                             ctypes.c_int, ctypes.c_int32, ctypes.c_uint32, c_u64p, c_i32p, c_u32p]),
}

_lib = None
_RUNTIME_LIBS = ("libamdhip64", "libhsa-runtime64", "librccl")


class FuzzyHipError(RuntimeError):
    pass


def mapped_runtimes() -> dict:
    """Real paths of the HIP / HSA / RCCL runtime libraries mapped into this process (/proc/self/maps)."""
    out = {k: set() for k in _RUNTIME_LIBS}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                parts = line.split(None, 5)
                if len(parts) < 6:
                    continue
                path = parts[5].strip()
                base = os.path.basename(path)
                for k in _RUNTIME_LIBS:
                    if base.startswith(k + ".so"):
                        out[k].add(os.path.realpath(path))
    except OSError:
        pass
    return {k: sorted(v) for k, v in out.items()}


_RULE = ("one HIP runtime per process: PyTorch wheels ship their own libamdhip64 / libhsa-runtime64 / librccl, and a "
         "process that loads libfuzzyhip first and torch afterwards maps both copies -- the second finds no device, and "
         "streams or device pointers of one mean nothing to the other.  Import torch BEFORE tts_amd touches the library "
         "(the loader then binds libfuzzyhip to torch's runtime by SONAME), set FUZZYHIP_HIP_RUNTIME=torch (tts_amd then "
         "maps torch's copy first, without importing torch), or keep torch out of the process")


def assert_single_runtime(where: str) -> None:
    """Raise when two copies of the HIP runtime are mapped (see _RULE); called when the library is loaded and
    wherever a stream or device pointer made by other code comes in (set_stream, torch tensors, communicators).
    A second libhsa-runtime64 alone is not an error: rocprofv3 maps /opt/rocm's copy into every process it
    profiles, and a copy no HIP runtime initialises is inert (mapped_runtimes() still lists it)."""
    rt = mapped_runtimes()
    for k in ("libamdhip64",):
        if len(rt[k]) > 1:
            raise FuzzyHipError(f"{where}: two copies of {k} are mapped in this process ({' and '.join(rt[k])}); {_RULE}")


def _torch_lib_dir():
    """torch/lib of an installed PyTorch, found without importing it (None when there is none)."""
    import importlib.util
    import sys
    mod = sys.modules.get("torch")
    if mod is not None and getattr(mod, "__file__", None):
        d = os.path.join(os.path.dirname(mod.__file__), "lib")
        return d if os.path.isdir(d) else None
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.origin:
        return None
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    return d if os.path.isdir(d) else None


def _preload_torch_runtime() -> None:
    """FUZZYHIP_HIP_RUNTIME=torch: map torch's own HIP runtime before libfuzzyhip so that the library binds to it
    (same SONAME) and a later `import torch` finds its runtime already there: one runtime whatever the order."""
    d = _torch_lib_dir()
    path = os.path.join(d, "libamdhip64.so") if d else None
    if not path or not os.path.exists(path):
        raise FuzzyHipError("FUZZYHIP_HIP_RUNTIME=torch, but no torch/lib/libamdhip64.so was found")
    ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def load() -> ctypes.CDLL:
    """Load libfuzzyhip.so and bind every entry point; raises if the library was not built, or if the process
    ends up with two HIP runtimes (assert_single_runtime)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C tts_amd/csrc` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`); tts_amd has no CPU fallback")
        want = os.environ.get("FUZZYHIP_HIP_RUNTIME", "auto")
        if want not in ("auto", "torch", "rocm"):
            raise FuzzyHipError("FUZZYHIP_HIP_RUNTIME must be auto, torch or rocm")
        if want == "torch":
            _preload_torch_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        assert_single_runtime("loading libfuzzyhip.so")
        if want == "rocm" and "torch" in os.path.realpath(runtime_paths(lib)["hip"]).split(os.sep):
            raise FuzzyHipError("FUZZYHIP_HIP_RUNTIME=rocm, but torch was imported first and the library is bound to its runtime")
        _lib = lib
    return _lib


def runtime_paths(lib=None) -> dict:
    """{'hip': the libamdhip64 the library is bound to, 'rccl': the librccl it opened ('' so far if none)}"""
    lib = lib or load()
    a, b = ctypes.create_string_buffer(4096), ctypes.create_string_buffer(4096)
    lib.fz_runtime_paths(a, 4096, b, 4096)
    return {"hip": a.value.decode("utf-8", "replace"), "rccl": b.value.decode("utf-8", "replace")}


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc: int) -> None:
    if rc != 0:
        msg = load().fz_last_error().decode("utf-8", "replace")
        raise FuzzyHipError(f"libfuzzyhip error {rc}: {msg}")
