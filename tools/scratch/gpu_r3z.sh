set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-reps 0"
timeout -s KILL 60 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d $O/r3z_pmc_a -o p -- $B > $O/r3z_a.log 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES -d $O/r3z_pmc_b -o p -- $B > $O/r3z_b.log 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES -d $O/r3z_pmc_c -o p -- $B > $O/r3z_c.log 2>&1
echo done
