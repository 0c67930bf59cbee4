set -e
bash tools/prof_round.sh r4m notests
BENCH_ARGS="--workload C4 --candidates 1000000 --queries 1000" PROF_STEPS_TIMED=2 PROF_TIMEOUT=300 bash tools/prof_round.sh r4mc4 notests
BENCH_ARGS="--workload C3 --gpus 1 --candidates 12500032 --queries 2000 --exchange rccl" PROF_STEPS_TIMED=2 PROF_TIMEOUT=300 bash tools/prof_round.sh r4mc3 notests
