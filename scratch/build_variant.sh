#!/bin/bash
# scratch: build libhugs variant: $1 = output name, $2 = gemm source file (default current), EXTRA = extra flags
set -e
cd "$(dirname "$0")/../nerf-hugs_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -munsafe-fp-atomics -I."
mkdir -p ../../scratch/_obj
SRC=${2:-hugs_gemm.hip}
$HIPCC $FLAGS $EXTRA -c $SRC -o ../../scratch/_obj/$1.o 2>&1 | grep -v "warning\|^ \|note\|\^\|generated" || true
objs=$(ls _obj/*.o | grep -v hugs_gemm.o)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../scratch/$1.so ../../scratch/_obj/$1.o $objs
echo built scratch/$1.so
