# shard-size study: bash tools/big_ab.sh <tag> <candidates> <queries> name:ENV=..,ENV=.. [...]
set -e
O=gpurun_out; T=$1; NC=$2; NQ=$3; shift 3
for spec in "$@"; do
  N=${spec%%:*}; E=${spec#*:}; [ "$E" = "$spec" ] && E=""
  env $(echo $E | tr ',' ' ') timeout -k 10 400 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --latency-reps 0 --candidates $NC --queries $NQ > $O/${T}_$N.json 2> $O/${T}_$N.err
  python - <<PY
import json
d=json.loads(open("$O/${T}_$N.json").read().strip().splitlines()[-1])
print("$NC x $NQ $N: step %.2f ms  score %.2f  select %.2f  prep %.2f  value %.3e" % (d["ms_per_step"], d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["kernel_ms_per_step"]["host_query_prep"], d["value"]))
PY
done
