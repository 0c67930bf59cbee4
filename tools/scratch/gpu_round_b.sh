# tests, then matrix vs fused at C2 with the current library: bash tools/gpu_round_b.sh <tag>
set -e
T=${1:-r02p}; O=gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/${T}_tests.log 2>&1
tail -3 $O/${T}_tests.log
for rep in 1 2; do
for spec in matrix:FUZZYHIP_NO_FUSED=1 fused:FUZZYHIP_FUSED_MIN_MB=0 default:X=1; do
  N=${spec%%:*}; E=${spec#*:}
  env $E timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --latency-reps 50 > $O/${T}_${N}_$rep.json 2> $O/${T}_${N}_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/${T}_${N}_$rep.json").read().strip().splitlines()[-1])
print("$N rep $rep: step %.3f ms  score %.3f  select %.3f  prep %.3f  p50 %.3f  value %.3e" % (d["ms_per_step"], d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["kernel_ms_per_step"]["host_query_prep"], d.get("p50_latency_ms",0), d["value"]))
PY
done
done
