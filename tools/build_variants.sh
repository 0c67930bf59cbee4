# builds experiment variants of libfuzzyhip.so into variants/var_<name>/ (usage: bash tools/build_variants.sh name "-DX=..." ...)
set -e
cd "$(dirname "$0")/../tts_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -I../../include -I. -Wall -Wno-unused-function"
make -s build/fz_host.o
while [ $# -gt 1 ]; do
  name=$1; defs=$2; shift 2
  mkdir -p ../../variants/var_$name
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c fz_kernels.hip -o ../../variants/var_$name/k.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/var_$name/libfuzzyhip.so ../../variants/var_$name/k.o build/fz_host.o && rm ../../variants/var_$name/k.o && echo built $name ) &
done
wait
