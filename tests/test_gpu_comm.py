"""The exchange inside the library (fz_comm_*, fz_query_topk_sharded): local top-k, all-gather of
the packed keys, device merge.  Each case runs in its own process (tests/comm_case.py) because the library
opens one librccl per process: once with the in-process stand-in (ranks = threads sharing the one GPU of the
box: 2 and 3 shards, results equal to the single-corpus top-k on every rank) and once with the real librccl
(a communicator of size 1: RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(mode, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "comm_case.py"), mode], env=env, capture_output=True,
                       text=True, timeout=280)
    assert r.returncode == 0 and "comm case ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_sharded_exchange_ranks_as_threads():
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    assert os.path.exists(fake), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    _run("threads", {"FUZZYHIP_RCCL_LIB": fake})


def test_sharded_local_failure_does_not_hang_the_peers():
    """a rank whose local top-k step fails still enters the all-gather (zeroed keys, status word 1): it returns its
    own error, its peer returns FZ_ERR_COMM, and the next call on the same communicator works"""
    fake = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")
    _run("threads_fail", {"FUZZYHIP_RCCL_LIB": fake})


def test_sharded_exchange_real_rccl_one_rank():
    _run("real", {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
