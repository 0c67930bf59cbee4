# parity files that exercise the stream kernels, then the bench line.  Usage: bash tools/gpu_quick.sh <tag>
set -e
O=gpurun_out; T=${1:-q}
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_scorers.py tests/test_gpu_fullsize.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread > $O/${T}_tests.log 2>&1
timeout -k 10 300 python bench.py --steps 10 > $O/${T}_bench.json 2> $O/${T}_bench.err
python - <<PY
import json
d=json.loads(open("$O/${T}_bench.json").read().strip().splitlines()[-1])
print("step %.3f ms  score %.3f  select %.3f  prep %.3f  p50 %.3f  value %.3e  equal %s" % (d["ms_per_step"], d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["kernel_ms_per_step"]["host_query_prep"], d.get("p50_latency_ms",0), d["value"], d.get("results_equal")))
PY
