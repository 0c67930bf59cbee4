"""One HIP runtime per process (tts_amd/_lib.py _RULE, include/fuzzyhip.h at fz_set_stream): PyTorch wheels ship
their own libamdhip64 / libhsa-runtime64 / librccl.  Each case is a process of its own (tests/dist_case.py order):
  torch_first  import torch, then the library: the library binds to torch's runtime; a torch stream and a torch
               tensor's data_ptr are usable (fz_set_stream, fz_query_topk_keys)
  lib_first    the library, then torch: two runtimes are mapped; the binding says so, with both paths and the rule
  preload      FUZZYHIP_HIP_RUNTIME=torch: the library first, on torch's runtime mapped by path; torch imported later
               finds its runtime in place and works"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _order(which, env=None):
    e = dict(os.environ)
    e.pop("FUZZYHIP_HIP_RUNTIME", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(HERE, "dist_case.py"), "order", which], env=e, capture_output=True,
                       text=True, timeout=280)
    assert r.returncode == 0 and "dist case ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_torch_then_library_shares_one_runtime():
    _order("torch_first")


def test_library_then_torch_is_reported():
    _order("lib_first")


def test_preloaded_torch_runtime_any_order():
    _order("preload", {"FUZZYHIP_HIP_RUNTIME": "torch"})
