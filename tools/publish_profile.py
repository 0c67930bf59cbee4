#!/usr/bin/env python3
"""Copies the summaries of one tools/prof_round.sh run (gpurun_out/<tag>_*) into profiles/ under a
round-stage name and derives profiles/r01_traffic.json from its PMC passes.
Usage: python tools/publish_profile.py r01i r01_i"""
import csv
import json
import os
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def counters(db_path):
    db = sqlite3.connect(db_path)
    out = {}
    for kname, cname, n, total in db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                                             "group by kernel_name, counter_name"):
        out[(kname, cname)] = (n, total)
    return out


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

fetch = counters(os.path.join(G, f"{tag}_pmc_fetch", "p_results.db"))
write = counters(os.path.join(G, f"{tag}_pmc_write", "p_results.db"))
sq1 = counters(os.path.join(G, f"{tag}_pmc_sq1", "p_results.db"))
sq2 = counters(os.path.join(G, f"{tag}_pmc_sq2", "p_results.db"))
grbm = counters(os.path.join(G, f"{tag}_pmc_grbm", "p_results.db"))
score = [k for k in {k for k, _ in fetch} if "fz_score" in k]
sel = [k for k in {k for k, _ in fetch} if "fz_select_topk" in k]
disp = fetch[(sel[0], "FETCH_SIZE")][0]                      # dispatches = steps profiled (1 warm-up + 3)
kib = 1024.0
sf = sum(fetch[(k, "FETCH_SIZE")][1] for k in score) * kib * 2 / disp
sw = sum(write[(k, "WRITE_SIZE")][1] for k in score) * kib / disp
se = (fetch[(sel[0], "FETCH_SIZE")][1] * 2 + write[(sel[0], "WRITE_SIZE")][1]) * kib / disp
busy = {}
for k in score:
    active = grbm[(k, "GRBM_GUI_ACTIVE")][1] / 8.0           # summed over 8 XCDs
    busy[k] = {"valu_busy_frac": sq1[(k, "SQ_ACTIVE_INST_VALU")][1] * 4.0 / (active * 1024.0),
               "salu_insts_per_cu_cycle": sq2[(k, "SQ_INSTS_SALU")][1] / (active * 256.0),
               "insts_valu": sq2[(k, "SQ_INSTS_VALU")][1] / disp, "insts_salu": sq2[(k, "SQ_INSTS_SALU")][1] / disp,
               "insts_lds": sq2[(k, "SQ_INSTS_LDS")][1] / disp}
with open(os.path.join(G, f"{tag}_bench.json")) as f:
    cfg = json.loads(f.read().strip().splitlines()[-1])["config"]
main_k = max(score, key=lambda k: grbm[(k, "GRBM_GUI_ACTIVE")][1])
out = {
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --steps 3 --warmup 1, MI355X, round 1 (profiles/{name}_rocprofv3_summary.txt)",
    "units": "counters are KiB, summed over the dispatches of one run (1 warm-up + 3 steps); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; confirmed here on fz_select_topk, whose two passes read the 1e9-byte score matrix twice)",
    "workload": {"queries": cfg["queries"], "candidates_per_gpu": cfg["candidates_per_gpu"], "max_chars": 64, "k": cfg["k"]},
    "score_kernels_bytes_per_step": int(sf + sw),
    "score_fetch_bytes_per_step": int(sf),
    "score_write_bytes_per_step": int(sw),
    "select_bytes_per_step": int(se),
    "valu_busy_frac_stream1": busy[main_k]["valu_busy_frac"],
    "salu_insts_per_cu_cycle_stream1": busy[main_k]["salu_insts_per_cu_cycle"],
    "dominant_score_kernel": main_k,
    "per_kernel": busy,
    "note": "valu_busy = SQ_ACTIVE_INST_VALU x 4 cycles / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs); salu = SQ_INSTS_SALU per CU per cycle (one scalar issue per CU-cycle); the *_stream1 keys name the dominant score kernel of the run",
}
with open(os.path.join(P, "r01_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
