"""Helper of the multi-process tests (run as a script, one process per rank, so that the pytest process itself
never imports torch: PyTorch wheels carry their own HIP runtime and the test process is to stay on the one the
library links, see tts_amd/_lib.py _RULE).  torch is imported FIRST here, before tts_amd loads the library.

    python tests/dist_case.py gloo_merge  <rank> <world> <port>   CPU: gloo all-gather of fabricated key lists + merge
    python tests/dist_case.py sharded_gpu <rank> <world> <port>   ranks share GPU 0, gloo carries the exchange
    python tests/dist_case.py order <torch_first|lib_first|preload>   one process, import-order rule on the GPU box"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _key(score, idx):
    return (score << 40) | (((1 << 40) - 1) - idx)


def gloo_merge(rank, world, port):
    import torch
    import torch.distributed as dist
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
    dist.destroy_process_group()
    assert got == exp


def sharded_gpu(rank, world, port):
    import torch  # noqa: F401 -- first: the library then binds to torch's HIP runtime
    import torch.distributed as dist
    from tts_amd import FuzzySearch, _lib, synth
    from tts_amd.sharded import ShardedFuzzySearch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cps, offs = synth.make_corpus(4001, max_chars=48, seed=21, vocab_size=40)
    items = [s.upper() if i % 3 == 0 else s for i, s in enumerate(synth.to_strings(cps, offs))]
    queries = synth.make_queries(cps, offs, 9, seed=22, max_chars=48)
    fs = ShardedFuzzySearch(items, items[:50], threshold=0.35, device=0, exchange="torch")
    got = fs.find_best_matches_many(queries, "product")
    single = FuzzySearch(items, items[:50], threshold=0.35)
    exp = single.find_best_matches_many(queries, "product")
    fs.close(); single.close()
    dist.destroy_process_group()
    assert "torch" in _lib.runtime_paths()["hip"], _lib.runtime_paths()
    assert got == exp


def order(which):
    from tts_amd import Corpus, _lib, synth
    cps, offs = synth.make_corpus(20_000, max_chars=64, seed=101)
    queries = synth.make_queries(cps, offs, 17, seed=102)
    if which == "torch_first":
        import torch
        with Corpus(cps, offs) as c:
            assert "torch" in _lib.runtime_paths()["hip"], _lib.runtime_paths()
            ref = c.topk(queries, 35, 10)
            st = torch.cuda.Stream()
            c.set_stream(st.cuda_stream)                    # a torch stream: same runtime, so it is usable
            got = c.topk(queries, 35, 10)
            keys = torch.zeros(len(queries) * 10, dtype=torch.int64, device="cuda")
            c.topk_keys_device(queries, 35, 10, keys.data_ptr(), 0, sync=True)
            c.set_stream(None)
        assert all((a == b).all() for a, b in zip(got, ref))
        host = keys.cpu().numpy().view(np.uint64).reshape(len(queries), 10)
        exp_keys = np.where(ref[1] > 0, (ref[1].astype(np.uint64) << np.uint64(40)) | (np.uint64((1 << 40) - 1) - ref[0]), np.uint64(0))
        assert (host == exp_keys).all()
    elif which == "lib_first":
        with Corpus(cps, offs) as c:
            assert "torch" not in _lib.runtime_paths()["hip"], _lib.runtime_paths()
            ref = c.topk(queries, 35, 10)
            import torch  # noqa: F401 -- maps a second HIP runtime
            try:
                c.set_stream(12345)
            except _lib.FuzzyHipError as e:
                msg = str(e)
                assert "two copies of libamdhip64" in msg and "torch" in msg and "Import torch BEFORE" in msg, msg
            else:
                raise AssertionError("two HIP runtimes in the process went unnoticed")
            # the library itself keeps working on its own runtime
            got = c.topk(queries, 35, 10)
        assert all((a == b).all() for a, b in zip(got, ref))
    elif which == "preload":
        assert os.environ.get("FUZZYHIP_HIP_RUNTIME") == "torch"
        with Corpus(cps, offs) as c:
            assert "torch" in _lib.runtime_paths()["hip"], _lib.runtime_paths()
            ref = c.topk(queries, 35, 10)
            import torch
            _lib.assert_single_runtime("after import torch")
            assert torch.cuda.is_available()
            st = torch.cuda.Stream()
            c.set_stream(st.cuda_stream)
            got = c.topk(queries, 35, 10)
            c.set_stream(None)
        assert all((a == b).all() for a, b in zip(got, ref))
    else:
        raise SystemExit("unknown order case")


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "order":
        order(sys.argv[2])
    else:
        {"gloo_merge": gloo_merge, "sharded_gpu": sharded_gpu}[mode](int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    print("dist case ok:", " ".join(sys.argv[1:]))
