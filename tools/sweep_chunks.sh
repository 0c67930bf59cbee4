set -e
for c in 1 2 3 4 8; do
  export FUZZYHIP_CHUNKS=$c
  timeout -k 5 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --latency-reps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunks=$c', round(d['ms_per_step'],3), d['kernel_ms_per_step'])"
done
