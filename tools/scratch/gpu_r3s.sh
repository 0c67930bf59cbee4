set -e
O=gpurun_out
timeout -k 10 500 python bench.py --workload C3 --gpus 1 --candidates 12500032 --no-cpu-baseline --exchange rccl > $O/r3r_12500032.json 2> $O/r3r_12500032.err
BENCH_ARGS="--workload C4 --candidates 1000000 --queries 1000" PROF_TIMEOUT=400 bash tools/prof_round.sh r3s_c4 notests
BENCH_ARGS="--workload C3 --gpus 1 --candidates 12500032 --queries 2000" PROF_TIMEOUT=300 bash tools/prof_round.sh r3s_c3shard notests
