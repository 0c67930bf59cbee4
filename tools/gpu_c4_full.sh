# C4 at BASELINE.md's size once: 1000 queries x 10^7 candidates of 64-512 chars on one GPU (corpus synthesis takes minutes:
# a ticker keeps the run visibly alive)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
( while true; do date >> $O/c4_full_tick.log; sleep 30; done ) &
TICK=$!
timeout -k 10 1000 python bench.py --workload C4 --candidates 10000000 --queries 1000 --steps 2 --warmup 1 --no-cpu-baseline --latency-reps 0 > $O/c4_full.json 2> $O/c4_full.err
RC=$?
kill $TICK
python - <<PY
import json
d=json.loads(open("$O/c4_full.json").read().strip().splitlines()[-1])
print("C4 1e7: step %.1f ms value %.3e setup %.0f s score %.1f" % (d["ms_per_step"], d["value"], d["setup_seconds"], d["kernel_ms_per_step"]["score"]))
PY
exit $RC
