set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 120 python tools/pipe_probe.py 20
echo "-- no overlap"
FUZZYHIP_NO_OVERLAP=1 timeout -k 10 120 python tools/pipe_probe.py 20
timeout -s KILL 120 rocprofv3 --kernel-trace -d $O/r4d_t -o t -- python3 tools/pipe_probe.py 4 > $O/r4d_t.log 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect("$O/r4d_t/t_results.db")
rows=db.execute("select name,start,end from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if "fz_merge_lists" in r[0]]
# last two calls (pipelined)
a=idx[-3]
t0=rows[a][2]
for r in rows[a:]:
    if "fz_" in r[0] or "copyBuffer" in r[0]: print("%-40s start %8.1f dur %8.1f end %8.1f"%(r[0][:40],(r[1]-t0)/1e3,(r[2]-r[1])/1e3,(r[2]-t0)/1e3))
PY
