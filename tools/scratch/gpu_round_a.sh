# valu microbenchmark, GPU tests, C3 two-rank rehearsal (gloo, one GPU).  Usage: bash tools/gpu_round_a.sh <tag>
set -e
O=gpurun_out; T=${1:-a}
timeout -k 10 120 tools/bin/valu_rate > $O/${T}_valu_rate.json 2> $O/${T}_valu_rate.err
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=8 > $O/${T}_tests.log 2>&1
FUZZYHIP_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --candidates 2000000 --queries 2000 --steps 2 --warmup 1 > $O/${T}_c3_rehearsal.json 2> $O/${T}_c3_rehearsal.err
echo done
