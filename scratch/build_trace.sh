#!/bin/bash
# scratch: a second libhugs build with -DHUGS_TRACE (phase timestamps in k_gemm_nt_bf16_big) -> scratch/libhugs_trace.so
set -e
cd "$(dirname "$0")/../nerf-hugs_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -munsafe-fp-atomics -I../../scratch -I."
mkdir -p ../../scratch/_obj
$HIPCC $FLAGS -DHUGS_TRACE $EXTRA -c hugs_gemm.hip -o ../../scratch/_obj/hugs_gemm_trace.o
objs=$(ls _obj/*.o | grep -v hugs_gemm.o)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libhugs_trace.so ../../scratch/_obj/hugs_gemm_trace.o $objs
echo built scratch/libhugs_trace.so
