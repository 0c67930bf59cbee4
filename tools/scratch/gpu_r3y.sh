set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-reps 0"
timeout -s KILL 60 rocprofv3 -L > $O/avail.txt 2>&1 || true
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC -d $O/r3y_pmc_a -o p -- $B > $O/r3y_a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_CYCLES -d $O/r3y_pmc_b -o p -- $B > $O/r3y_b.log 2>&1
echo done
