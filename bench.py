#!/usr/bin/env python3
"""bench.py -- pairs/sec of the candidate-scoring path (FuzzySearch.find_best_matches, reference
src/fuzzy_search.py:16-38) on MI355X.

A step = one pass of the hot path over one batch: `--queries` prepared queries against this rank's
resident shard of `--candidates` candidates: query tables built on the host, uploaded, scored,
thresholded, top-k selected; for N > 1 the per-GPU top-k keys are all-gathered (RCCL) and merged.
Workload at N=1 = BASELINE.json configs[1] (C2: 1k queries x 1M candidates <= 64 chars, k=10).
Weak scaling: every rank holds its own `--candidates`-sized shard (contiguous global index range).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def measured_traffic(nq, n, max_chars, k):
    """HBM bytes per step of the scoring kernels from the committed PMC passes (rocprofv3 FETCH_SIZE /
    WRITE_SIZE, profiles/r01_traffic.json); None when this run's workload is not the profiled one."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            t = json.load(f)
    except OSError:
        return None, None
    w = t.get("workload", {})
    if (w.get("queries"), w.get("candidates_per_gpu"), w.get("max_chars"), w.get("k")) != (nq, n, max_chars, k):
        return None, None
    return t.get("score_kernels_bytes_per_step"), t


def cpu_baseline(cps, offs, queries, min_score, k, budget_s=12.0):
    """The oracle's restatement of the reference loop ("port"), 1 host thread, on a bounded sample
    of the same workload.  A reported baseline, not the target."""
    from oracle import fuzz_oracle as O
    O.lib().orc_set_bitpar(1)          # rapidfuzz itself runs the bit-parallel LCS
    n = len(offs) - 1
    ncand = min(n, 20000)
    sub_offs = np.ascontiguousarray(offs[:ncand + 1])
    t0 = time.perf_counter()
    O.topk(O.TOKEN_SET, queries[0], cps, sub_offs, min_score, k)
    dt = max(time.perf_counter() - t0, 1e-6)
    nqs = int(max(1, min(len(queries), budget_s / dt)))
    t0 = time.perf_counter()
    for q in queries[:nqs]:
        O.topk(O.TOKEN_SET, q, cps, sub_offs, min_score, k)
    dt = time.perf_counter() - t0
    out = {"value": nqs * ncand / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
           "sample": f"{nqs} queries x first {ncand} candidates of the same corpus, "
                     f"oracle/fuzz_oracle.c token_set_ratio + top-{k} loop, {dt:.1f} s"}
    # the same loop on the host cores this job may use (ctypes releases the GIL): SURVEY.md 8d, B2
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    qs = (queries * ((nqs * threads) // max(len(queries), 1) + 1))[:nqs * threads]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda q: O.topk(O.TOKEN_SET, q, cps, sub_offs, min_score, k), qs))
    dt = time.perf_counter() - t0
    out["all_cores"] = {"value": len(qs) * ncand / dt, "unit": "pairs/s", "cores": threads,
                        "sample": f"{len(qs)} queries x {ncand} candidates over {threads} threads, {dt:.1f} s"}
    O.lib().orc_set_bitpar(0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["C2", "C4"], default="C2",
                    help="C2 (default, the metric's configuration): names <= 64 chars; C4: long strings "
                         "64-512 chars (multi-word carry chain), an extra measurement, not the bench line")
    ap.add_argument("--candidates", type=int, default=None, help="per GPU (default 1e6; C4: 2e5)")
    ap.add_argument("--queries", type=int, default=None, help="default 1000; C4: 100")
    ap.add_argument("--max-chars", type=int, default=None)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--threshold", type=float, default=0.35)
    ap.add_argument("--latency-reps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tts_amd import Corpus, merge_topk, synth
    from tts_amd.preprocess import min_score_for_threshold

    long_strings = args.workload == "C4"
    if args.candidates is None:
        args.candidates = 200_000 if long_strings else 1_000_000
    if args.queries is None:
        args.queries = 100 if long_strings else 1000
    if args.max_chars is None:
        args.max_chars = 512 if long_strings else 64
    gen = dict(min_tokens=8, max_tokens=60) if long_strings else {}

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # FUZZYHIP_BENCH_REHEARSAL=1: several ranks share the visible GPUs and exchange over gloo -- a way
    # to run the N > 1 code path on a one-GPU box; never the measured configuration
    rehearsal = os.environ.get("FUZZYHIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # this rank's shard: its own seeded corpus, global indices [rank*n, (rank+1)*n)
    n = args.candidates
    cps, offs = synth.make_corpus(n, max_chars=args.max_chars, seed=synth.CORPUS_SEED + rank, **gen)
    # the same queries on every rank, drawn from rank 0's shard
    if rank == 0:
        q_src = (cps, offs)
    else:
        q_src = synth.make_corpus(n, max_chars=args.max_chars, seed=synth.CORPUS_SEED, **gen)
    queries = synth.make_queries(q_src[0], q_src[1], args.queries, seed=synth.QUERY_SEED,
                                 max_chars=args.max_chars)
    del q_src
    min_score = min_score_for_threshold(args.threshold)
    k = args.k

    corpus = Corpus(cps, offs, device=local_rank, base_index=rank * n)
    info = corpus.info()
    from tts_amd.preprocess import pack_prepared
    q_packed = pack_prepared(queries)
    nq = len(queries)

    keys_dev = gathered = None
    if world > 1:
        keys_dev = torch.zeros(nq * k, dtype=torch.int64, device="cuda")
        gathered = torch.zeros(world * nq * k, dtype=torch.int64, device="cpu" if rehearsal else "cuda")

    def step():
        if world == 1:
            return corpus.topk(q_packed, min_score, k)
        corpus.topk_keys_device(q_packed, min_score, k, keys_dev.data_ptr(), 0, sync=True)
        dist.all_gather_into_tensor(gathered, keys_dev.cpu() if rehearsal else keys_dev)
        host = gathered.cpu().numpy().view(np.uint64)
        return merge_topk(host, world, nq, k)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    score_ms = select_ms = prep_ms = 0.0
    launches = 0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = corpus.timing()
        score_ms += t["score_ms"]; select_ms += t["select_ms"]; prep_ms += t["host_prep_ms"]
        launches += t["score_launches"]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    pairs_per_step = float(nq) * n * world
    value = pairs_per_step * args.steps / elapsed

    out = None
    if rank == 0:
        # roofline of the dominant kernel (the scoring kernel): algorithmic bytes / launch duration,
        # both per launch; durations are hipEvents recorded by the library on its own stream.
        bytes_per_query = synth.algorithmic_bytes_per_query(info)
        alg_bytes = float(bytes_per_query) * nq * args.steps
        achieved = alg_bytes / (score_ms * 1e-3) / 1e9 if score_ms > 0 else 0.0
        traffic, prof = measured_traffic(nq, n, args.max_chars, k)
        out = {
            "metric": "query x candidate pairs/sec (token_set_ratio scoring + threshold + top-k)",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": (f"C4: {nq} queries x {n} candidates/GPU (8-60 tokens, <= {args.max_chars} chars, "
                                    f"multi-word patterns), top-{k}, threshold {args.threshold}") if long_strings else
                                   (f"C2: {nq} queries x {n} candidates/GPU (<= {args.max_chars} chars, "
                                    f"Zipf(1.2) tokens over 50k vocabulary), top-{k}, threshold {args.threshold}"),
                       "queries": nq, "candidates_per_gpu": n, "k": k, "min_score": min_score,
                       "parallelism": (f"shard{world}" if world > 1 else "single") + ("-rehearsal-gloo" if rehearsal else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per step)" if traffic else None,
                         "valu_busy_frac": prof.get("valu_busy_frac_stream1") if prof else None,
                         "co_limit": "integer VALU + SALU issue (corpus is Infinity-Cache resident; frac > 1 means cached re-reads, not HBM)",
                         "kernel": "fz_score_*", "launches_per_step": launches / max(args.steps, 1),
                         "algorithmic_bytes_per_step": alg_bytes / args.steps,
                         "kernel_ms_per_step": score_ms / args.steps},
            "kernel_ms_per_step": {"score": score_ms / args.steps, "select": select_ms / args.steps,
                                   "host_query_prep": prep_ms / args.steps},
            "kernel_pairs_per_s": float(nq) * n * args.steps / (score_ms * 1e-3) if score_ms > 0 else None,
        }
        if world == 1 and args.latency_reps > 0:
            lat = []
            one = pack_prepared(queries[:1])
            for i in range(args.latency_reps):
                one = pack_prepared([queries[i % nq]])
                t1 = time.perf_counter()
                corpus.topk(one, min_score, k)
                lat.append((time.perf_counter() - t1) * 1e3)
            out["p50_latency_ms"] = float(np.median(lat))
            out["p50_latency_config"] = f"1 query, k={k}, {n} resident candidates, {args.latency_reps} calls"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cps, offs, queries, min_score, k)
    corpus.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
