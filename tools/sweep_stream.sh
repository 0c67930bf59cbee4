set -e
for w in 14 12 6; do for b in 1024 4096 8192 32768; do for o in 1; do
  export FUZZYHIP_STREAM_WAVES=$w FUZZYHIP_STREAM_BLOCKS=$b FUZZYHIP_STREAM_OVERSUB=$o
  timeout -k 5 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-reps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves=$w blocks=$b oversub=$o', round(d['ms_per_step'],3), d['kernel_ms_per_step'])"
done; done; done
