#!/bin/bash
# Builds libhugs_hip.so for gfx950 (cross-compiles without a GPU).  hugs_stepfun.hip is built with
# -ffp-contract=off: its outputs are bit-exact against oracle/stepfun_ref.c.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -munsafe-fp-atomics"
mkdir -p _obj
pids=()
for f in hugs_*.hip; do
  extra=""
  [ "$f" = "hugs_stepfun.hip" ] && extra="-ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
  dep="$f"
  [ "$f" = "hugs_gemm_f16.hip" ] && dep="hugs_gemm.hip"      # (it is hugs_gemm.hip compiled with half operands)
  if [ ! -f "_obj/${f%.hip}.o" ] || [ "$f" -nt "_obj/${f%.hip}.o" ] || [ "$dep" -nt "_obj/${f%.hip}.o" ] || [ hugs_common.h -nt "_obj/${f%.hip}.o" ]; then
    $HIPCC $FLAGS $extra -c "$f" -o "_obj/${f%.hip}.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libhugs_hip.so _obj/*.o
echo "built $(pwd)/libhugs_hip.so"
