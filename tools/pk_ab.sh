# GPU tests, then the bench with and without the packed kernel.  Usage: bash tools/pk_ab.sh <tag>
set -e
O=gpurun_out; T=${1:-pk}
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/${T}_tests.log 2>&1
FUZZYHIP_TRACE=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --latency-reps 20 > $O/${T}_bench_pack.json 2> $O/${T}_bench_pack.err
FUZZYHIP_NO_PACK=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --latency-reps 20 > $O/${T}_bench_nopack.json 2> $O/${T}_bench_nopack.err
