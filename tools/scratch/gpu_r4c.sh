set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > $O/r4c_tests.log 2>&1 || { tail -30 $O/r4c_tests.log; exit 1; }
tail -2 $O/r4c_tests.log
for rep in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline > $O/r4c_bench$rep.json 2> $O/r4c_bench$rep.err
python - <<PY
import json
d=json.loads(open("$O/r4c_bench$rep.json").read().strip().splitlines()[-1])
print("C2 step %.3f ms value %.3e | sync %s | two_in_flight %s | score %.3f select %.3f prep %.3f" % (d["ms_per_step"], d["value"], d.get("synchronous_calls"), d.get("two_in_flight"), d["kernel_ms_per_step"]["score"], d["kernel_ms_per_step"]["select"], d["kernel_ms_per_step"]["host_query_prep"]))
PY
done
