set -e
O=gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py -x -q --timeout 300 --timeout-method thread > $O/r3u_tests.log 2>&1
timeout -k 10 500 python bench.py --workload C3 --gpus 1 --candidates 12500032 --no-cpu-baseline --exchange rccl > $O/r3u_12500032.json 2> $O/r3u_12500032.err
python - <<PY
import json
d=json.loads(open("$O/r3u_12500032.json").read().strip().splitlines()[-1])
print("12.5M: step %.1f ms  score %.1f  select %.1f  value %.3e  redo rows %s exchange %s" % (d["ms_per_step"], d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["value"], d["fused_rows_redone_through_matrix"], d.get("exchange",{}).get("all_gather_plus_device_merge_ms_per_step")))
PY
timeout -k 10 300 python bench.py --steps 10 > $O/r3u_bench.json 2> $O/r3u_bench.err
python - <<PY
import json
d=json.loads(open("$O/r3u_bench.json").read().strip().splitlines()[-1])
print("C2 step %.3f ms  score %.3f  select %.3f  prep %.3f  p50 %.3f  value %.3e  equal %s" % (d["ms_per_step"], d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["kernel_ms_per_step"]["host_query_prep"], d.get("p50_latency_ms",0), d["value"], d.get("results_equal")))
PY
