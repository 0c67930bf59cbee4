# experiment builds: bash tools/build_x.sh NAME "-DFLAG ..."  -> variants/libfuzzyhip_NAME.so
set -e
N=$1; F=$2
D=/tmp/fzx_$N; mkdir -p $D
H=/opt/rocm/bin/hipcc
FL="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -I/root/repo/include -I/root/repo/tts_amd/csrc -Wno-unused-function $F"
$H $FL -c /root/repo/tts_amd/csrc/fz_kernels.hip -o $D/k.o &
$H $FL -c /root/repo/tts_amd/csrc/fz_host.cpp -o $D/h.o &
wait
$H --offload-arch=gfx950 -shared -fPIC -o /root/repo/variants/libfuzzyhip_$N.so $D/k.o $D/h.o
echo built $N
