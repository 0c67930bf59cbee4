# bench lines under environment variants on one box: bash tools/sweep_env.sh <tag> "VAR=VAL VAR2=VAL" "..." ...
T=$1; shift
O=gpurun_out
i=0
for E in "$@"; do
  i=$((i+1))
  for rep in 1 2; do
    env $E timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --latency-reps 30 > $O/${T}_${i}_$rep.json 2> $O/${T}_${i}_$rep.err || { echo "FAILED: $E"; tail -3 $O/${T}_${i}_$rep.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$O/${T}_${i}_$rep.json").read().strip().splitlines()[-1])
s=d.get("synchronous_calls") or {}
print("[$E] rep $rep: step %.3f ms (sync %.3f)  score %.3f  select %.3f  prep %.3f  p50 %.3f  value %.3e" % (d["ms_per_step"], s.get("ms_per_step",0), d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["kernel_ms_per_step"]["host_query_prep"], d.get("p50_latency_ms",0), d["value"]))
PY
  done
done
