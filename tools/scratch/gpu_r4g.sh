set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
FUZZYHIP_HIP_RUNTIME=torch timeout -s KILL 120 rocprofv3 --kernel-trace -d $O/r4g_t -o t -- python3 tools/pipe_probe.py 8 > $O/r4g_t.log 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect("$O/r4g_t/t_results.db")
rows=db.execute("select name,start,end from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if "fz_merge_lists" in r[0]]
print("merge-to-merge periods (us):", [round((rows[idx[i+1]][2]-rows[idx[i]][2])/1e3) for i in range(len(idx)-1)])
a=idx[-5]
t0=rows[a][2]
for r in rows[a:idx[-3]+1]:
    print("%-40s start %8.1f dur %8.1f end %8.1f"%(r[0][:40],(r[1]-t0)/1e3,(r[2]-r[1])/1e3,(r[2]-t0)/1e3))
PY
