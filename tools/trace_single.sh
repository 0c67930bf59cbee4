set -e
T=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --kernel-trace -d gpurun_out/${T}_trace -o t -- python3 tools/trace_single.py 8 > gpurun_out/${T}_trace.log 2>&1
python3 - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("gpurun_out/${T}_trace/*.db")[0])
rows=db.execute("select name, start, end from kernels order by start").fetchall()
last=rows[-14:]
t0=last[0][1]
for n,s,e in last: print("%-58s start %9.1f us  dur %8.1f us"%(n[:58],(s-t0)/1e3,(e-s)/1e3))
PY
