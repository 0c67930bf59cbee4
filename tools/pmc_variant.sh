# instruction counters of the score kernels for the in-tree library and for one variants/var_<name> build
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-reps 0"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU -d $O/pv_base -o p -- $B > $O/pv_base.log 2>&1
cp tts_amd/libfuzzyhip.so /tmp/lib_orig.so
cp variants/var_$1/libfuzzyhip.so tts_amd/libfuzzyhip.so
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU -d $O/pv_$1 -o p -- $B > $O/pv_$1.log 2>&1
cp /tmp/lib_orig.so tts_amd/libfuzzyhip.so
python tools/rocpd_summary.py $(find $O/pv_base $O/pv_$1 -name "*.db" | sort) | grep "==\|stream_pk" 
