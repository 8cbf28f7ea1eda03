#!/bin/bash
# One-file A/B build: tools/ab_one.sh NAME FILE(.hip, without extension) FLAGS...  -> tools/diag/libnerf_NAME.so
# = the shipped objects of csrc/build/ with FILE recompiled under the extra FLAGS (timing-only ablations, scheduling experiments).
# Selected at run time with NERF_HIP_LIB=$PWD/tools/diag/libnerf_NAME.so; never the shipped library.
set -e
name=$1; file=$2; shift 2
cd "$(dirname "$0")/../nerf_meets_mlx_amd/csrc"
mkdir -p build_ab_$name ../../tools/diag
split="-mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -DNERF_DMA_CLOBBER_M0=1 -Wno-inline-asm -mllvm -amdgpu-atomic-optimizer-strategy=None"
case $file in mlp22) if [ -n "$NO_MAXILP" ]; then extra="$split"; else extra="$split -mllvm -amdgpu-sched-strategy=max-ilp"; fi;; mlp_s16|mlp_s16x) extra=$split;; mlp_dww) extra="-fno-slp-vectorize -DNERF_DMA_CLOBBER_M0=1 -Wno-inline-asm";; mlp) extra="-mllvm -amdgpu-atomic-optimizer-strategy=None";; *) extra="";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $extra "$@" -save-temps=obj -c $file.hip -o build_ab_$name/$file.o
objs=""
for o in build/*.o; do b=$(basename $o); case $b in *-hip-amdgcn-*) continue;; esac; if [ "$b" = "$file.o" ]; then objs="$objs build_ab_$name/$file.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/diag/libnerf_$name.so $objs -ldl
rm -f build_ab_$name/*-hip-amdgcn-*.o build_ab_$name/*.bc build_ab_$name/*.hipi build_ab_$name/*-host-* build_ab_$name/*.out* build_ab_$name/*.hipfb
echo "built tools/diag/libnerf_$name.so"
