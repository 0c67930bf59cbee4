# single-query p50 under different environments: bash tools/p50_ab.sh <tag> name:ENV=..,.. [...]
set -e
O=gpurun_out; T=$1; shift
for rep in 1 2; do
for spec in "$@"; do
  N=${spec%%:*}; E=${spec#*:}; [ "$E" = "$spec" ] && E="FZ_NONE=1"
  env $(echo $E | tr ',' ' ') timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --latency-reps 200 > $O/${T}_${N}_$rep.json 2> $O/${T}_${N}_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/${T}_${N}_$rep.json").read().strip().splitlines()[-1])
print("$N rep $rep: p50 %.4f ms   step %.3f" % (d.get("p50_latency_ms",0), d["ms_per_step"]))
PY
done
done
