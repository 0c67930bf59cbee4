# C3 (10^4 queries, k = 10) on ONE GPU at the whole corpus (10^8) and at the shard sizes of 2, 4 and 8 ranks, each through
# the sharded entry point with a size-1 RCCL communicator: what a rank's local step + the exchange cost.  -> gpurun_out/<tag>_<n>.json
set -e
O=gpurun_out; T=${1:-c3study}
for N in 12500032 25000000 50000000 100000000; do
  timeout -k 10 900 python bench.py --workload C3 --gpus 1 --candidates $N --no-cpu-baseline --exchange rccl > $O/${T}_$N.json 2> $O/${T}_$N.err
  python - <<PY
import json
d=json.loads(open("$O/${T}_$N.json").read().strip().splitlines()[-1])
print("$N: step %.1f ms  score %.1f  select %.1f  value %.3e  redo %s exchange %s" % (d["ms_per_step"], d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["value"], d.get("fused_rows_redone_through_matrix"), d.get("exchange",{}).get("all_gather_plus_device_merge_ms_per_step")))
PY
done
