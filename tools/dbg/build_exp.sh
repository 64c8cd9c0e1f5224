#!/bin/bash
# build libvidseg_hip_exp<N>.so = the product library with gemm_conv.hip compiled under -DPH_EXP=<N> (kernel experiments;
# select with VIDSEG_LIB=libvidseg_hip_exp<N>.so).  usage: tools/dbg/build_exp.sh N [extra hipcc flags]
set -e
cd "$(dirname "$0")/../.."
n=$1; shift
C=vidseg_diffusion_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DPH_EXP=$n "$@" -I include -c $C/gemm_conv.hip -o /tmp/gemm_conv.exp$n.o
objs=$(ls $C/*.o | grep -v "\.bf16\.o" | grep -v gemm_conv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vidseg_diffusion_amd/libvidseg_hip_exp$n.so /tmp/gemm_conv.exp$n.o $objs
echo built exp$n
