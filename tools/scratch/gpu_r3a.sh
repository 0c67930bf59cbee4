# round 3, first GPU call: the new tests by file, then the whole GPU suite, then the default bench line
set -e
O=gpurun_out; T=${1:-r3a}
timeout -k 10 400 python -u -m pytest tests/test_gpu_fused.py tests/test_gpu_runtime.py tests/test_gpu_comm.py -x -v --timeout 300 --timeout-method thread > $O/${T}_new.log 2>&1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/${T}_tests.log 2>&1
timeout -k 10 300 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
