"""The N>1 exchange step on CPU: world_size-2 gloo all-gather of fabricated per-shard key lists, then
the library's deterministic merge; every rank must obtain the single-shard answer."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _key(score, idx):
    return (score << 40) | (((1 << 40) - 1) - idx)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from tts_amd.sharded import gather_and_merge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(11)
    nq, k, n = 5, 6, 300
    scores = rng.integers(40, 46, size=(nq, n))
    lo, hi = rank * n // world, (rank + 1) * n // world
    keys = np.zeros((nq, k), dtype=np.uint64)
    for q in range(nq):
        loc = sorted(range(lo, hi), key=lambda i: (-scores[q, i], i))[:k]
        for j, i in enumerate(loc):
            keys[q, j] = _key(int(scores[q, i]), i)
    idx, sc, cnt = gather_and_merge(torch.from_numpy(keys.view(np.int64)), nq, k)
    exp = [[(i, int(scores[q, i])) for i in sorted(range(n), key=lambda i: (-scores[q, i], i))[:k]] for q in range(nq)]
    got = [[(int(idx[q, j]), int(sc[q, j])) for j in range(int(cnt[q]))] for q in range(nq)]
    ret[rank] = (got == exp)
    dist.destroy_process_group()


def test_two_rank_gather_merge():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]
