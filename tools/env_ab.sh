# bench.py at C2 under different environments, interleaved twice: bash tools/env_ab.sh <tag> name:ENV=..,ENV=.. [...]
set -e
O=gpurun_out; T=$1; shift
for rep in 1 2; do
for spec in "$@"; do
  N=${spec%%:*}; E=${spec#*:}; [ "$E" = "$spec" ] && E="FZ_NONE=1"
  env $(echo $E | tr ',' ' ') timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --latency-reps 50 > $O/${T}_${N}_$rep.json 2> $O/${T}_${N}_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/${T}_${N}_$rep.json").read().strip().splitlines()[-1])
print("$N rep $rep: step %.3f ms  score %.3f  select %.3f  prep %.3f  p50 %.3f  value %.3e" % (d["ms_per_step"], d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["kernel_ms_per_step"]["host_query_prep"], d.get("p50_latency_ms",0), d["value"]))
PY
done
done
