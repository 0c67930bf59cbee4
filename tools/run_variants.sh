# benches each variants/var_*/libfuzzyhip.so in turn (copied over tts_amd/libfuzzyhip.so in the box's scratch copy)
set -e
O=gpurun_out; T=${1:-var}; shift
cp tts_amd/libfuzzyhip.so /tmp/lib_orig.so
for v in "$@"; do
  cp variants/var_$v/libfuzzyhip.so tts_amd/libfuzzyhip.so
  for e in ""; do
    env $e timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --latency-reps 0 > $O/${T}_$v.json 2> $O/${T}_$v.err
    python -c "
import json,sys
d=json.loads(open('$O/${T}_$v.json').read()); print('$v', '$e', round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
  done
done
cp /tmp/lib_orig.so tts_amd/libfuzzyhip.so
