set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py -x -q --timeout 300 --timeout-method thread > $O/r4b_tests.log 2>&1
tail -2 $O/r4b_tests.log
B="python3 bench.py --workload C4 --candidates 1000000 --queries 1000 --steps 2 --warmup 1 --no-cpu-baseline --latency-reps 0"
timeout -k 10 300 $B > $O/r4b_c4.json 2> $O/r4b_c4.err
python - <<PY
import json
d=json.loads(open("$O/r4b_c4.json").read().strip().splitlines()[-1])
print("C4 step %.2f ms score %.2f value %.3e" % (d["ms_per_step"], d["kernel_ms_per_step"]["score"], d["value"]))
PY
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/r4b_trace -o t -- $B > $O/r4b_trace.log 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect("$O/r4b_trace/t_results.db")
for r in db.execute("select name,count(*),sum(end-start)/1e6 from kernels group by name order by 3 desc limit 8"): print("%-50s %4d %9.2f ms"%(r[0][:50],r[1],r[2]))
PY
