"""The N>1 exchange step on CPU: world_size-2 gloo all-gather of fabricated per-shard key lists, then
the library's deterministic merge; every rank must obtain the single-shard answer.  The ranks are
processes of their own (tests/dist_case.py): the pytest process never imports torch."""
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def run_ranks(mode, world, timeout=280, env=None):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = dict(os.environ)
    e.update(env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_case.py"), mode, str(r), str(world), str(port)],
                              env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "dist case ok" in so, (so[-1000:], se[-3000:])


def test_two_rank_gather_merge():
    run_ranks("gloo_merge", 2)
