#!/usr/bin/env python3
"""bench.py -- pairs/sec of the candidate-scoring path (FuzzySearch.find_best_matches, reference
src/fuzzy_search.py:16-38) on MI355X.

A step = one pass of the hot path over one batch: the prepared queries against this rank's resident
shard: query tables built on the host, uploaded, scored, thresholded, top-k selected; for N > 1 the
per-GPU top-k keys are all-gathered (RCCL) and merged.

Workloads (BASELINE.json configs):
  C2  (default at --gpus 1, the configuration the metric is quoted on): 1k queries x 1M candidates <= 64
      chars, k = 10.  With --gpus N > 1 and --workload C2 every rank holds its own 1M shard (weak scaling).
  C3  (default at --gpus N > 1): ONE fixed corpus of --candidates (default 10^8) names cut into contiguous
      index ranges over the ranks, 10^4 queries, k = 10: strong scaling, the configuration north_star's
      ">= 7x at 8 GPUs" is quoted on.  The corpus is 64 seeded blocks, so every N in {1, 2, 4, 8, ...} sees
      the same candidates; blocks are statistically alike, so equal block counts are equal code points.
  C4  long strings 64-512 chars (multi-word carry chain), an extra measurement.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
C3_BLOCKS = 64           # seeded blocks of the C3 corpus


def source_hash() -> str:
    """Hash of the kernel and host sources the library is built from: PMC counts committed under
    profiles/ are only quoted when they were taken on these very sources."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "tts_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "tts_amd", "csrc", "*.cpp")) +
                   glob.glob(os.path.join(ROOT, "tts_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "include", "fuzzyhip.h")])
    for p in files:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def pmc_counts(wl, nq, n, max_chars, k):
    """Per-step instruction counts and HBM bytes of the score kernels from the committed rocprofv3 --pmc
    passes (profiles/pmc_current*.json, one per profiled workload, written by tools/publish_profile.py).
    Counts are a property of (sources, workload): None unless both match this run."""
    why = "no profiles/pmc_current*.json"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_current*.json"))):
        t = load_json(os.path.basename(path))
        if not t:
            continue
        w = t.get("workload", {})
        if (w.get("workload_id", "C2"), w.get("queries"), w.get("candidates_per_gpu"), w.get("max_chars"), w.get("k")) != (wl, nq, n, max_chars, k):
            why = "the committed PMC passes are for other workloads"
            continue
        if t.get("source_hash") != source_hash():
            return None, f"the committed PMC pass of this workload was taken on other sources ({t.get('source_hash')} != {source_hash()})"
        return t, None
    return None, why


SIMDS = 1024             # 256 CUs x 4 SIMD-32 (/opt/skills/guides/MI355X_MICROARCH.md)


def valu_peak():
    """Two ceilings for wave64 VALU instructions per second.  The chip's: a SIMD-32 issues a wave64 VALU instruction
    in 2 cycles (the guide's figure), so SIMDS x clock / 2, at the shader clock tools/valu_rate.hip measured inside
    its kernels.  And what that microbenchmark sustains with plain v_bitop3_b32 at the stream kernels' occupancy
    (8 waves per SIMD), which is higher than 2 cycles per instruction suggests (1.46)."""
    t = load_json("valu_rate_current.json")
    if not t:
        return None
    best = None
    for r in t.get("results", []):
        if r.get("waves_per_simd") == 8 and r["kind"].startswith("v_bitop3_b32"):
            best = r
    if not best:
        return None
    return {"clock_ghz": best["clock_ghz"], "chip_2cyc": SIMDS * best["clock_ghz"] / 2.0,
            "measured_at_occupancy": best["chip_ginstr_per_s"], "measured_cycles_per_instr": best["cycles_per_instr_per_simd"]}


def step_instructions(queries, info, max_chars):
    """Wave instructions per bench step that ARE the algorithm: one Hyyro step = 3 lane operations (and / add / or-xor)
    per pattern word per query symbol, token separators included, for every unit a wave takes a query through: a
    packed pair of tiles (one word for two candidates), a one-word tile, a two-word tile (2 x 3).  Padding steps,
    addressing, token-id compares, masks and look-ups are not in it."""
    steps = 0
    for q in queries:
        toks = sorted(set(q.split()))
        steps += sum(len(t) + 1 for t in toks)
    units = info.get("n_packed_pairs", 0) + info.get("n_stream1_tiles", 0) + 2 * info.get("n_stream2_tiles", 0)
    return 3.0 * steps * units


def cpu_baseline(cps, offs, queries, min_score, k, budget_s=10.0):
    """The oracle's restatement of the reference loop ("port"), on a bounded sample of the same
    workload: 1 host thread, then the host cores this job may use.  A reported baseline, not the target."""
    from concurrent.futures import ThreadPoolExecutor
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


def verify_rows(last, cps, offs, queries, min_score, k, rows=8):
    """Checker (not measured): `rows` rows of the LAST timed step's output against the oracle's loop over the whole
    corpus (reference src/fuzzy_search.py:22-38), spread over the batch; the oracle runs on the host threads."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import fuzz_oracle as O
    O.lib().orc_set_bitpar(1)
    idx, score, cnt = last
    nq = len(queries)
    pick = sorted({int(i) for i in np.linspace(0, nq - 1, rows)})
    threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    with ThreadPoolExecutor(threads) as ex:
        exp = list(ex.map(lambda qi: O.topk(O.TOKEN_SET, queries[qi], cps, offs, min_score, k), pick))
    O.lib().orc_set_bitpar(0)
    bad = [qi for qi, e in zip(pick, exp)
           if [(int(idx[qi, j]), int(score[qi, j])) for j in range(int(cnt[qi]))] != e]
    return {"results_checked": len(pick), "results_equal": not bad, "rows": pick, "rows_differing": bad}


def c1_leg(device, threshold, k_app=5):
    """BASELINE.json configs[0] (C1), 'as the app runs it': 1 query x 10^4 names <= 32 chars.  The CPU side
    is the loop of reference src/fuzzy_search.py:22-38 statement for statement over the oracle's
    pure-Python scorer (thefuzz itself is not importable here); the GPU side is the
    drop-in class on the same list."""
    from oracle import fuzz_oracle as O
    from tts_amd import FuzzySearch, synth
    cps, offs = synth.make_corpus(10_000, max_chars=32, seed=synth.CORPUS_SEED + 1)
    items = synth.to_strings(cps, offs)
    queries = synth.make_queries(cps, offs, 100, seed=synth.QUERY_SEED + 1, max_chars=32)
    t0 = time.perf_counter()
    exp = O.py_find_best_matches(items, queries[0], threshold, k_app)
    py_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    exp_c = O.c_find_best_matches(items, queries[0], threshold, k_app)
    c_s = time.perf_counter() - t0
    fs = FuzzySearch(items, items[:16], threshold=threshold, device=device, top_n=k_app)
    got = fs.find_best_matches(queries[0], "product")
    lat = []
    for q in queries:
        t1 = time.perf_counter()
        fs.find_best_matches(q, "product")
        lat.append((time.perf_counter() - t1) * 1e3)
    fs.close()
    return {"workload": "C1: 1 query x 10000 names <= 32 chars, threshold %.2f, top-%d" % (threshold, k_app),
            "python_loop": {"value": len(items) / py_s, "unit": "pairs/s", "cores": 1, "kind": "port",
                            "seconds_per_query": py_s,
                            "sample": "oracle.py_find_best_matches: reference loop over the pure-Python token_set_ratio, 1 query"},
            "c_loop_seconds_per_query": c_s,
            "gpu_p50_latency_ms": float(np.median(lat)),
            "gpu_p50_config": "FuzzySearch.find_best_matches (drop-in class, Python preprocessing included), 100 calls",
            "results_equal": bool(got == exp == exp_c)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 20 (C3: 2)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (C3: 1)")
    ap.add_argument("--workload", choices=["C2", "C3", "C4"], default=None,
                    help="default: C2 at --gpus 1, C3 (strong scaling over one 10^8-candidate corpus) at --gpus N > 1")
    ap.add_argument("--candidates", type=int, default=None,
                    help="C2/C4: per GPU (default 1e6 / 2e5); C3: the WHOLE corpus (default 1e8)")
    ap.add_argument("--queries", type=int, default=None, help="default 1000 (C3: 10000, C4: 100)")
    ap.add_argument("--max-chars", type=int, default=None)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--threshold", type=float, default=0.35)
    ap.add_argument("--latency-reps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", choices=["auto", "rccl", "torch"], default="auto",
                    help="N > 1: who carries the all-gather + merge.  rccl: inside the library (ncclAllGather on the corpus' "
                         "stream + device merge, fz_query_topk_sharded); torch: torch.distributed all-gather + host merge; "
                         "auto: rccl, torch if the communicator cannot be made.  At --gpus 1, rccl runs a communicator of size 1")
    args = ap.parse_args()

    # torch FIRST, then the library: PyTorch wheels ship their own HIP runtime, and the loader binds libfuzzyhip to
    # the copy that is already mapped (tts_amd/_lib.py _RULE: one HIP runtime per process).  torch is the launcher
    # here (rank / world, barrier, the max-over-ranks reduce -- all over gloo on the host) and the fence's
    # torch.cuda.synchronize(); the data path, RCCL all-gather included, is the library's.
    import torch
    import torch.distributed as dist
    from tts_amd import Corpus, _lib, merge_topk, synth
    from tts_amd.preprocess import min_score_for_threshold, pack_prepared

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.workload is None:
        args.workload = "C2" if world == 1 else "C3"
    wl = args.workload
    strong = wl == "C3"
    if args.candidates is None:
        args.candidates = {"C2": 1_000_000, "C3": 100_000_000, "C4": 200_000}[wl]
    if args.queries is None:
        args.queries = {"C2": 1000, "C3": 10_000, "C4": 100}[wl]
    if args.max_chars is None:
        args.max_chars = 512 if wl == "C4" else 64
    if args.steps is None:
        args.steps = 2 if strong else 20
    if args.warmup is None:
        args.warmup = 1 if strong else 3
    gen = dict(min_tokens=8, max_tokens=60) if wl == "C4" else {}

    # FUZZYHIP_BENCH_REHEARSAL=1: several ranks share the visible GPUs and exchange over gloo -- a way
    # to run the N > 1 code path on a one-GPU box; never the measured configuration
    rehearsal = os.environ.get("FUZZYHIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")        # one node; the container hostname may not resolve
        # control plane only: the one data-path collective is the library's ncclAllGather (no second RCCL
        # communicator inside torch)
        dist.init_process_group("gloo")

    t_setup = time.perf_counter()
    if strong:
        # one corpus for every N: C3_BLOCKS seeded blocks, this rank takes a contiguous run of them
        if args.candidates % C3_BLOCKS or C3_BLOCKS % world:
            raise SystemExit(f"C3: --candidates must be a multiple of {C3_BLOCKS} and --gpus must divide {C3_BLOCKS}")
        per_block = args.candidates // C3_BLOCKS
        b0, b1 = rank * C3_BLOCKS // world, (rank + 1) * C3_BLOCKS // world
        cps, offs = synth.make_corpus_blocks(b0, b1, per_block, max_chars=args.max_chars)
        base_index = b0 * per_block
        n_local = (b1 - b0) * per_block
        q_src = (cps, offs) if b0 == 0 else synth.make_corpus_blocks(0, 1, per_block, max_chars=args.max_chars)
    else:
        n_local = args.candidates
        cps, offs = synth.make_corpus(n_local, max_chars=args.max_chars, seed=synth.CORPUS_SEED + rank, **gen)
        base_index = rank * n_local
        q_src = (cps, offs) if rank == 0 else synth.make_corpus(n_local, max_chars=args.max_chars, seed=synth.CORPUS_SEED, **gen)
    # the same queries on every rank, drawn from the first shard / block
    queries = synth.make_queries(q_src[0], q_src[1], args.queries, seed=synth.QUERY_SEED, max_chars=args.max_chars)
    del q_src
    min_score = min_score_for_threshold(args.threshold)
    k = args.k

    corpus = Corpus(cps, offs, device=local_rank, base_index=base_index)
    info = corpus.info()
    setup_s = time.perf_counter() - t_setup
    q_packed = pack_prepared(queries)
    nq = len(queries)

    exch = {"gather_ms": 0.0, "merge_ms": 0.0}
    comm, comm_note = None, None
    if (world > 1 and not rehearsal and args.exchange in ("auto", "rccl")) or (world == 1 and args.exchange == "rccl"):
        from tts_amd._lib import FuzzyHipError
        from tts_amd.sharded import LibraryExchange, comm_id_via_torch
        try:
            comm = LibraryExchange(rank, world, local_rank, comm_id_via_torch() if world > 1 else (lambda my_id: my_id))
        except FuzzyHipError as e:
            if args.exchange == "rccl":
                raise
            comm_note = f"library communicator not made on rank {rank} ({e}); torch.distributed (gloo) carries the exchange"
        if world > 1:
            # every rank takes the same carrier: one flag, reduced over all ranks, before the first step
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm, comm_note = None, "library communicator not made on another rank; torch.distributed (gloo) carries the exchange"
            if int(ok.item()) == 0 and args.exchange == "rccl":
                raise SystemExit(comm_note)
    keys_dev = gathered = None
    if world > 1 and comm is None:
        keys_dev = torch.zeros(nq * k, dtype=torch.int64, device="cuda")
        gathered = torch.zeros(world * nq * k, dtype=torch.int64, device="cpu")

    def step():
        if comm is not None:
            out = corpus.topk_sharded(comm.handle, q_packed, min_score, k)
            exch["gather_ms"] += corpus.timing()["exchange_ms"]
            return out
        if world == 1:
            return corpus.topk(q_packed, min_score, k)
        corpus.topk_keys_device(q_packed, min_score, k, keys_dev.data_ptr(), 0, sync=True)
        t1 = time.perf_counter()
        dist.all_gather_into_tensor(gathered, keys_dev.cpu())
        host = gathered.numpy().view(np.uint64)
        t2 = time.perf_counter()
        out = merge_topk(host, world, nq, k)
        exch["gather_ms"] += (t2 - t1) * 1e3
        exch["merge_ms"] += (time.perf_counter() - t2) * 1e3
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    score_ms = select_ms = prep_ms = 0.0
    launches = fused_steps = fallback_chunks = fallback_rows = 0
    exch["gather_ms"] = exch["merge_ms"] = 0.0
    fence()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
        t = corpus.timing()
        score_ms += t["score_ms"]; select_ms += t["select_ms"]; prep_ms += t["host_prep_ms"]
        launches += t["score_launches"]
        fused_steps += t["fused"]; fallback_chunks += t["fused_fallback_chunks"]; fallback_rows += t["fused_fallback_rows"]
    fence()
    elapsed = time.perf_counter() - t0
    # The same K steps as a queue of batches: fz_query_topk_submit / fz_query_topk_wait with two batches in flight, so
    # that the host work of step i+1 (query tables, Python, decoding of step i-1) passes under the kernels of step i.
    # Every step is a whole call (tables, upload, kernels, copy back, decode) inside the fences; results are compared
    # with the synchronous loop's.
    piped = None
    if world == 1 and comm is None and args.steps >= 2 and not os.environ.get("FUZZYHIP_BENCH_NO_PIPELINE"):
        for _ in range(min(args.warmup, 2)):
            corpus.topk_wait(corpus.topk_submit(q_packed, min_score, k))
        fence()
        t0 = time.perf_counter()
        prev, last_p = None, None
        for _ in range(args.steps):
            tk = corpus.topk_submit(q_packed, min_score, k)
            if prev is not None:
                last_p = corpus.topk_wait(prev)
            prev = tk
        last_p = corpus.topk_wait(prev)
        fence()
        elapsed_p = time.perf_counter() - t0
        piped = {"elapsed": elapsed_p, "equal": bool(all((a == b).all() for a, b in zip(last, last_p)))}
    alg_local = float(synth.algorithmic_bytes_per_query(info))
    if world > 1:
        dev = "cpu"
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ab = torch.tensor([alg_local, float(n_local)], dtype=torch.float64, device=dev)
        dist.all_reduce(ab, op=dist.ReduceOp.SUM)
        alg_all, n_all = float(ab[0].item()), int(ab[1].item())
    else:
        alg_all, n_all = alg_local, n_local

    pairs_per_step = float(nq) * n_all
    value = pairs_per_step * args.steps / elapsed
    sync_line = None
    if piped and piped["equal"] and piped["elapsed"] < elapsed:
        # the headline is the job's throughput with two batches in flight; the one-call-at-a-time figures stay beside it
        sync_line = {"value": value, "ms_per_step": elapsed / args.steps * 1e3,
                     "calls": "fz_query_topk: one synchronous call per step (the reference's calling convention)"}
        elapsed = piped["elapsed"]
        value = pairs_per_step * args.steps / elapsed

    out = None
    if rank == 0:
        steps = max(args.steps, 1)
        score_s = score_ms * 1e-3 / steps                      # this rank's score phase per step (hipEvents)
        # ---- the ceiling that binds: integer issue (VALU), not HBM -------------------------------------------------
        pmc, pmc_why = pmc_counts(wl, nq, n_local, args.max_chars, k)
        peak = valu_peak()
        roof = {"bound": "valu-issue", "achieved": None, "peak": None, "unit": "G wave64 VALU instructions/s",
                "frac": None, "traffic": None,
                "kernel": "the score kernels of a step (fz_score_stream_pk + fz_score_stream<1> + fz_score_stream<2>; fz_score_general "
                          "for long strings): sample tiles first, then the rest",
                "kernel_ms_per_step": score_ms / steps, "launches_per_step": launches / steps,
                "how": "achieved = VALU instructions of the score kernels per step (rocprofv3 --pmc SQ_INSTS_VALU, "
                       "profiles/pmc_current*.json, quoted only when source hash and workload match this run) / the score "
                       "phase's duration measured live (hipEvents on the library's stream); peak = 1024 SIMDs x shader clock / "
                       "2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md; clock measured by tools/valu_rate.hip); "
                       "frac_at_kernel_occupancy = achieved / what that microbenchmark sustains with plain VALU at 8 waves per SIMD"}
        if peak:
            roof["peak"] = peak["chip_2cyc"]
            roof["peak_clock_ghz"] = peak["clock_ghz"]
            roof["peak_measured_at_kernel_occupancy"] = peak["measured_at_occupancy"]
        if pmc and score_s > 0:
            roof["achieved"] = pmc["score_insts_valu_per_step"] / score_s / 1e9
            roof["traffic"] = pmc["score_kernels_bytes_per_step"]
            roof["traffic_unit"] = "HBM bytes per step of the score kernels: FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, separate passes"
            roof["valu_insts_per_step"] = pmc["score_insts_valu_per_step"]
            roof["salu_insts_per_step"] = pmc.get("score_insts_salu_per_step")
            roof["all_insts_per_step"] = pmc.get("score_insts_all_per_step")
            roof["valu_issue_frac_2cyc_dominant_kernel_profiled"] = pmc.get("valu_issue_frac_2cyc_dominant")
            if roof["peak"]:
                roof["frac"] = roof["achieved"] / roof["peak"]
                roof["frac_at_kernel_occupancy"] = roof["achieved"] / peak["measured_at_occupancy"]
            if wl != "C4" and pmc.get("score_insts_all_per_step"):
                # how much of what issues is the algorithm proper (Hyyro steps on real symbols), by count
                roof["algorithm_instr_frac"] = step_instructions(queries, info, args.max_chars) / pmc["score_insts_all_per_step"]
            roof["wave_cycles_waiting_frac"] = pmc.get("score_wave_cycles_waiting_frac")
        else:
            roof["why_null"] = pmc_why
        # ---- the declared roofline: HBM.  Real traffic against the peak is the fraction;
        # algorithmic bytes (no credit for query batching) are a side figure, not a fraction of HBM ----------
        alg_bytes_step = alg_local * nq
        hbm = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "achieved": None, "frac": None,
               "algorithmic_bytes_per_step": alg_bytes_step,
               "algorithmic_GBps": alg_bytes_step / score_s / 1e9 if score_s > 0 else None,
               "algorithmic_over_peak": alg_bytes_step / score_s / 1e9 / HBM_PEAK_GBPS if score_s > 0 else None,
               "note": "SURVEY.md 8(d) counts 4*len + 4 + 4*tokens bytes per (query, candidate) pair with no credit for "
                       "query batching; a wave scores one tile against Q_b queries out of LDS and the shard is cache "
                       "resident, so algorithmic_over_peak may exceed 1 and is not a fraction of HBM; frac is measured "
                       "FETCH+WRITE bytes / score time / peak",
               "Q_b": "a work item takes one tile (or pair of tiles) through a group of up to 1024 queries out of LDS: the whole batch at C2"}
        if pmc and score_s > 0:
            hbm["achieved"] = pmc["score_kernels_bytes_per_step"] / score_s / 1e9
            hbm["frac"] = hbm["achieved"] / HBM_PEAK_GBPS
        out = {
            "metric": "query x candidate pairs/sec (token_set_ratio scoring + threshold + top-k)",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": {
                           "C2": f"C2: {nq} queries x {n_local} candidates/GPU (<= {args.max_chars} chars, Zipf(1.2) tokens over 50k vocabulary), top-{k}, threshold {args.threshold}",
                           "C3": f"C3: {nq} queries x {n_all} candidates in ONE corpus sharded over {world} GPU(s) by contiguous index ranges ({n_local}/GPU, <= {args.max_chars} chars), all-gather top-{k}, threshold {args.threshold}",
                           "C4": f"C4: {nq} queries x {n_local} candidates/GPU (8-60 tokens, <= {args.max_chars} chars, multi-word patterns), top-{k}, threshold {args.threshold}"}[wl],
                       "workload_id": wl, "max_chars": args.max_chars, "queries": nq, "candidates_per_gpu": n_local, "candidates_total": n_all, "k": k, "min_score": min_score,
                       "parallelism": (f"shard{world}" if world > 1 else "single") + ("-rehearsal-gloo" if rehearsal else ""),
                       "calls": ("fz_query_topk_submit / fz_query_topk_wait, two batches in flight (every step a whole call: tables, "
                                 "upload, kernels, copy back, decode)") if sync_line else "one synchronous call per step"},
            "roofline": roof,
            "roofline_hbm": hbm,
            "kernel_ms_per_step": {"score": score_ms / steps, "select": select_ms / steps, "host_query_prep": prep_ms / steps},
            "kernel_pairs_per_s": float(nq) * n_local / score_s if score_s > 0 else None,
            # fused: sample tiles -> per-row limit -> appended lists -> merge (no score matrix); matrix: u8 score matrix + select
            "topk_path": "fused" if fused_steps == args.steps else ("matrix" if fused_steps == 0 else "mixed"),
            # rows whose appended list overflowed (hits clustered where no sample tile fell) are redone through the matrix
            "fused_rows_redone_through_matrix": fallback_rows, "fused_chunks_with_such_rows": fallback_chunks,
            "setup_seconds": setup_s,
            "source_hash": source_hash(),
            "runtime": _lib.runtime_paths(),      # the HIP runtime the library is bound to, the librccl it opened
        }
        if sync_line:
            out["synchronous_calls"] = sync_line
            out["two_in_flight_results_equal_synchronous"] = True
        elif piped:
            out["two_in_flight"] = {"ms_per_step": piped["elapsed"] / args.steps * 1e3, "results_equal_synchronous": piped["equal"],
                                    "note": "not the headline: not faster than one call at a time, or results differed"}
        if world > 1:
            # the honest denominator of "x times 1 GPU": the SAME workload on one GPU, measured once and committed
            one = load_json("c3_1gpu_current.json")
            if one and (one.get("queries"), one.get("candidates_total"), one.get("k")) == (nq, n_all, k):
                out["value_1gpu_same_workload"] = one["value"]
                out["value_1gpu_source"] = one.get("source")
        if world > 1 or comm is not None:
            if comm is not None:
                out["exchange"] = {"carrier": "libfuzzyhip: ncclAllGather (RCCL) on the corpus' stream + fz_merge_gathered on the device",
                                   "bytes_per_rank_per_step": nq * k * 8,
                                   "all_gather_plus_device_merge_ms_per_step": exch["gather_ms"] / steps}
            else:
                out["exchange"] = {"carrier": "keys to the host, torch.distributed all_gather_into_tensor over gloo, fz_merge_topk on the host",
                                   "bytes_per_rank_per_step": nq * k * 8, "all_gather_ms_per_step": exch["gather_ms"] / steps,
                                   "host_merge_ms_per_step": exch["merge_ms"] / steps}
                if comm_note:
                    out["exchange"]["note"] = comm_note
        if world == 1 and args.latency_reps > 0:
            lat = []
            for i in range(args.latency_reps):
                one = pack_prepared([queries[i % nq]])
                t1 = time.perf_counter()
                corpus.topk(one, min_score, k)
                lat.append((time.perf_counter() - t1) * 1e3)
            out["p50_latency_ms"] = float(np.median(lat))
            out["p50_latency_config"] = f"1 query, k={k}, {n_local} resident candidates, {args.latency_reps} calls"
    if comm is not None:
        comm.close()
    corpus.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if last is not None and float(n_local) * args.max_chars <= 2.6e8:
            # the oracle as checker of what the timed region returned (after the timed region, never inside it)
            out.update(verify_rows(last, cps, offs, queries, min_score, k, rows=8 if float(n_local) * args.max_chars <= 1.3e8 else 2))
        out["cpu_baseline"] = cpu_baseline(cps, offs, queries, min_score, k)
        c1 = c1_leg(local_rank, args.threshold)
        out["cpu_baseline"]["python_loop"] = c1["python_loop"]
        out["p50_latency_ms_c1"] = c1["gpu_p50_latency_ms"]
        out["c1"] = c1
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
