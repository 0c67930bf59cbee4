# sweep of the query-group size of the stream kernels (FUZZYHIP_STREAM_BLOCKS sets how many groups a launch is cut into)
set -e
for b in 512 1024 2048 4096 8192 16384 65536; do
  export FUZZYHIP_STREAM_BLOCKS=$b
  timeout -k 5 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --latency-reps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocks=$b', round(d['ms_per_step'],3), d['kernel_ms_per_step'])"
done
