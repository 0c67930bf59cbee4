# One profiling round on the GPU box: default bench line, kernel trace, PMC passes (each its own run).
# Usage: [BENCH_ARGS="--workload C4 ..."] [PROF_STEPS_TIMED=3] bash tools/prof_round.sh <tag> [notests]     -> gpurun_out/<tag>_*
set -e
TAG=${1:-r02x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/${TAG}_tests.log 2>&1
fi
touch $O/${TAG}_stamp
NS=${PROF_STEPS_TIMED:-3}
timeout -k 10 600 python bench.py $BENCH_ARGS > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
B="python3 bench.py $BENCH_ARGS --steps $NS --warmup 1 --no-cpu-baseline --latency-reps 0"
export PROF_STEPS=$((NS+1)) PROF_CMD="$B"
export FUZZYHIP_BENCH_NO_PIPELINE=1      # the profiled runs make warm-up + NS synchronous steps and nothing else (the counters are divided by that)
timeout -s KILL ${PROF_TIMEOUT:-120} rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -o ${TAG} -- $B > $O/${TAG}_trace.log 2>&1
timeout -s KILL ${PROF_TIMEOUT:-120} rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o p -- $B > $O/${TAG}_pmc_fetch.log 2>&1
timeout -s KILL ${PROF_TIMEOUT:-120} rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o p -- $B > $O/${TAG}_pmc_write.log 2>&1
timeout -s KILL ${PROF_TIMEOUT:-120} rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/${TAG}_pmc_sq1 -o p -- $B > $O/${TAG}_pmc_sq1.log 2>&1
timeout -s KILL ${PROF_TIMEOUT:-120} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU -d $O/${TAG}_pmc_sq2 -o p -- $B > $O/${TAG}_pmc_sq2.log 2>&1
timeout -s KILL ${PROF_TIMEOUT:-120} rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/${TAG}_pmc_grbm -o p -- $B > $O/${TAG}_pmc_grbm.log 2>&1
find $O -name "*.db" -newer $O/${TAG}_stamp | sort > $O/${TAG}_dbs.txt
python tools/rocpd_summary.py $(cat $O/${TAG}_dbs.txt) > $O/${TAG}_summary.txt 2>&1
python tools/publish_profile.py ${TAG} --json-only > $O/${TAG}_pmc.json 2> $O/${TAG}_pmc.err || true
echo done
