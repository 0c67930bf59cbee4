# the C3 shard sizes of 2, 4 and 8 ranks on ONE GPU (10^4 queries each): what a rank's local step costs
set -e
O=gpurun_out; T=${1:-c3s}
for N in 50000000 25000000 12500000; do
  timeout -k 10 500 python bench.py --workload C3 --gpus 1 --candidates $N --no-cpu-baseline --exchange rccl > $O/${T}_$N.json 2> $O/${T}_$N.err
  python - <<PY
import json
d=json.loads(open("$O/${T}_$N.json").read().strip().splitlines()[-1])
print("$N: step %.1f ms  score %.1f  select %.1f  value %.3e  exchange %s" % (d["ms_per_step"], d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["value"], d.get("exchange",{}).get("all_gather_plus_device_merge_ms_per_step")))
PY
done
