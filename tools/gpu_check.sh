# GPU tests, then the default bench line (no profiler).  Usage: bash tools/gpu_check.sh <tag>
set -e
O=gpurun_out; T=${1:-chk}
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/${T}_tests.log 2>&1
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/${T}_bench.json 2> $O/${T}_bench.err
