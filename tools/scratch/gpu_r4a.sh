set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
i=0
for E in "FUZZYHIP_TRACE_X=0" "FUZZYHIP_STREAM_BLOCKS=768" "FUZZYHIP_STREAM_BLOCKS=1536" "FUZZYHIP_STREAM_BLOCKS=3072"; do
  i=$((i+1))
  export $E
  timeout -s KILL 120 rocprofv3 --kernel-trace -d $O/r4a_t$i -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-reps 0 > $O/r4a_t$i.log 2>&1
  echo "== $E"
  python tools/timeline.py $O/r4a_t$i/t_results.db
  unset ${E%%=*}
done
