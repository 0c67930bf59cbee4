#!/usr/bin/env python3
"""Derives the per-step PMC figures of one tools/prof_round.sh run (gpurun_out/<tag>_*) and, unless
--json-only, copies its summaries into profiles/ under a round-stage name and writes
profiles/pmc_current.json -- the file bench.py quotes instruction counts and HBM traffic from, stamped
with the hash of the sources the library was built from.
Usage: python tools/publish_profile.py r02a r02_a      (publish)
       python tools/publish_profile.py r02a --json-only (print the JSON; runs on the GPU box)"""
import csv
import json
import os
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, name = sys.argv[1], sys.argv[2]
json_only = name == "--json-only"
STEPS = int(os.environ.get("PROF_STEPS", "4"))      # bench steps the profiled command ran (warm-up + timed)
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def counters(db_path):
    db = sqlite3.connect(db_path)
    out = {}
    for kname, cname, n, total in db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                                             "group by kernel_name, counter_name"):
        out[(kname, cname)] = (n, total)
    return out


def derive():
    import bench
    fetch = counters(os.path.join(G, f"{tag}_pmc_fetch", "p_results.db"))
    write = counters(os.path.join(G, f"{tag}_pmc_write", "p_results.db"))
    sq1 = counters(os.path.join(G, f"{tag}_pmc_sq1", "p_results.db"))
    sq2 = counters(os.path.join(G, f"{tag}_pmc_sq2", "p_results.db"))
    grbm = counters(os.path.join(G, f"{tag}_pmc_grbm", "p_results.db"))
    kernels = {k for k, _ in fetch}
    score = sorted(k for k in kernels if "fz_score" in k)
    other = sorted(k for k in kernels if "fz_" in k and "fz_score" not in k)
    steps = STEPS                                               # warm-up + timed steps of the profiled command
    kib = 1024.0
    sf = sum(fetch[(k, "FETCH_SIZE")][1] for k in score) * kib * 2 / steps
    sw = sum(write[(k, "WRITE_SIZE")][1] for k in score) * kib / steps
    of = sum(fetch[(k, "FETCH_SIZE")][1] for k in other) * kib * 2 / steps
    ow = sum(write[(k, "WRITE_SIZE")][1] for k in other) * kib / steps
    per = {}
    for k in score + other:
        active = grbm[(k, "GRBM_GUI_ACTIVE")][1] / 8.0           # summed over 8 XCDs
        per[k] = {"dispatches_per_step": fetch[(k, "FETCH_SIZE")][0] / steps,
                  "insts_valu_per_step": sq2[(k, "SQ_INSTS_VALU")][1] / steps,
                  "insts_salu_per_step": sq2[(k, "SQ_INSTS_SALU")][1] / steps,
                  "insts_lds_per_step": sq2[(k, "SQ_INSTS_LDS")][1] / steps,
                  "insts_smem_per_step": sq2[(k, "SQ_INSTS_SMEM")][1] / steps,
                  "insts_vmem_per_step": sq2[(k, "SQ_INSTS_VMEM")][1] / steps,
                  "lds_bank_conflict_cycles_per_step": sq2[(k, "SQ_LDS_BANK_CONFLICT")][1] / steps,
                  "gpu_active_cycles_per_step": active / steps,
                  # wave64 VALU issues in 2 cycles on a SIMD-32 (MI355X_MICROARCH.md): busy = 2 x instructions / SIMD-cycles
                  "valu_issue_frac_2cyc": sq2[(k, "SQ_INSTS_VALU")][1] * 2.0 / (active * 1024.0),
                  "salu_insts_per_cu_cycle": sq2[(k, "SQ_INSTS_SALU")][1] / (active * 256.0),
                  "wave_cycles_waiting_frac": sq1[(k, "SQ_WAIT_ANY")][1] / max(sq1[(k, "SQ_WAVE_CYCLES")][1], 1.0),
                  "fetch_bytes_per_step": fetch[(k, "FETCH_SIZE")][1] * kib * 2 / steps,
                  "write_bytes_per_step": write[(k, "WRITE_SIZE")][1] * kib / steps}
    with open(os.path.join(G, f"{tag}_bench.json")) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    cfg = line["config"]
    main_k = max(score, key=lambda k: grbm[(k, "GRBM_GUI_ACTIVE")][1])
    return {
        "source": f"rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*, GRBM_* each in its own run) of the profiled "
                  f"command on MI355X, tag {tag}",
        "units": f"per step = per bench step (sums over the {STEPS} profiled steps / {STEPS}); FETCH_SIZE is KiB and is doubled "
                 "(gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE is KiB",
        "source_hash": bench.source_hash(),
        "workload": {"workload_id": cfg.get("workload_id", "C2"), "queries": cfg["queries"], "candidates_per_gpu": cfg["candidates_per_gpu"],
                     "max_chars": cfg.get("max_chars", 64), "k": cfg["k"]},
        "profiled_command": os.environ.get("PROF_CMD", "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-reps 0"),
        "score_kernels_bytes_per_step": int(sf + sw),
        "score_fetch_bytes_per_step": int(sf),
        "score_write_bytes_per_step": int(sw),
        "other_kernels_bytes_per_step": int(of + ow),
        "score_insts_valu_per_step": sum(per[k]["insts_valu_per_step"] for k in score),
        "score_insts_salu_per_step": sum(per[k]["insts_salu_per_step"] for k in score),
        "score_insts_lds_per_step": sum(per[k]["insts_lds_per_step"] for k in score),
        "score_insts_all_per_step": sum(per[k][f] for k in score for f in ("insts_valu_per_step", "insts_salu_per_step", "insts_lds_per_step",
                                                                            "insts_smem_per_step", "insts_vmem_per_step")),
        "score_wave_cycles_waiting_frac": sum(sq1[(k, "SQ_WAIT_ANY")][1] for k in score) / max(sum(sq1[(k, "SQ_WAVE_CYCLES")][1] for k in score), 1.0),
        "dominant_score_kernel": main_k,
        "salu_insts_per_cu_cycle_dominant": per[main_k]["salu_insts_per_cu_cycle"],
        "valu_issue_frac_2cyc_dominant": per[main_k]["valu_issue_frac_2cyc"],
        "per_kernel": per,
    }


out = derive()
if json_only:
    print(json.dumps(out, indent=1))
    sys.exit(0)

shutil.copy(os.path.join(G, f"{tag}_bench.json"), os.path.join(P, f"{name}_bench.json"))
shutil.copy(os.path.join(G, f"{tag}_summary.txt"), os.path.join(P, f"{name}_rocprofv3_summary.txt"))
db = sqlite3.connect(os.path.join(G, f"{tag}_trace", f"{tag}_results.db"))
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                  "group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
with open(os.path.join(P, f"{name}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / total, 2), r[4], r[5]])
wl_id = out["workload"]["workload_id"]
suffix = os.environ.get("PROF_SUFFIX", "" if wl_id == "C2" else "_" + wl_id)
for dst in (f"{name}_pmc.json", f"pmc_current{suffix}.json"):
    with open(os.path.join(P, dst), "w") as f:
        json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
