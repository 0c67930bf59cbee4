set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python bench.py --workload C4 --steps 3 --warmup 1 --no-cpu-baseline --latency-reps 0 > $O/r01c_c4_bench.json 2> $O/r01c_c4_bench.err
B="python3 bench.py --workload C4 --steps 2 --warmup 1 --no-cpu-baseline --latency-reps 0"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/r01c_c4_trace -o c4 -- $B > $O/r01c_c4_trace.log 2>&1
python tools/rocpd_summary.py $O/r01c_c4_trace/c4_results.db > $O/r01c_c4_summary.txt 2>&1
echo done
