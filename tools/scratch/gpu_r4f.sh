set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "-- torch runtime, no overlap"; FUZZYHIP_NO_OVERLAP=1 FUZZYHIP_HIP_RUNTIME=torch timeout -k 10 120 python tools/pipe_probe.py 20
echo "-- torch runtime, 1 prep thread"; FUZZYHIP_PREP_THREADS=1 FUZZYHIP_HIP_RUNTIME=torch timeout -k 10 120 python tools/pipe_probe.py 20
O=gpurun_out
FUZZYHIP_HIP_RUNTIME=torch timeout -s KILL 120 rocprofv3 --kernel-trace -d $O/r4f_t -o t -- python3 tools/pipe_probe.py 4 > $O/r4f_t.log 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect("$O/r4f_t/t_results.db")
rows=db.execute("select name,start,end from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if "fz_merge_lists" in r[0]]
a=idx[-2]
t0=rows[a][2]
for r in rows[a:idx[-1]+1]:
    print("%-40s start %8.1f dur %8.1f end %8.1f"%(r[0][:40],(r[1]-t0)/1e3,(r[2]-r[1])/1e3,(r[2]-t0)/1e3))
PY
