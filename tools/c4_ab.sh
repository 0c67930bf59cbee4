# C4 (long strings, general kernel) A/B of library builds: bash tools/c4_ab.sh <tag> <lib> [<lib> ...]
set -e
T=$1; shift
O=gpurun_out
for rep in 1 2; do
  for L in "$@"; do
    N=$(basename $L .so)
    FUZZYHIP_LIB=$PWD/$L timeout -k 10 200 python bench.py --workload C4 --no-cpu-baseline --steps 5 --warmup 2 --latency-reps 20 > $O/${T}_${N}_$rep.json 2> $O/${T}_${N}_$rep.err
    python - <<PY
import json
d=json.loads(open("$O/${T}_${N}_$rep.json").read().strip().splitlines()[-1])
print("$N rep $rep: C4 step %.3f ms  score %.3f  select %.3f  prep %.3f  p50 %.3f  value %.3e" % (d["ms_per_step"], d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["kernel_ms_per_step"]["host_query_prep"], d.get("p50_latency_ms",0), d["value"]))
PY
  done
done
