# the three profiling rounds of a release: C2 (the headline), C4 at 10^6 x 1000, the C3 shard of 8 ranks at 2000 queries
set -e
T=${1:-rel}
bash tools/prof_round.sh ${T} notests
BENCH_ARGS="--workload C4 --candidates 1000000 --queries 1000" PROF_STEPS_TIMED=2 PROF_TIMEOUT=300 bash tools/prof_round.sh ${T}c4 notests
BENCH_ARGS="--workload C3 --gpus 1 --candidates 12500032 --queries 2000 --exchange rccl" PROF_STEPS_TIMED=2 PROF_TIMEOUT=300 bash tools/prof_round.sh ${T}c3 notests
