#!/bin/bash
# Builds libhugs_hip.so for gfx950 (cross-compiles without a GPU).  hugs_stepfun.hip is built with
# -ffp-contract=off: its outputs are bit-exact against oracle/stepfun_ref.c.
# Every object is compiled with -Rpass-analysis=kernel-resource-usage; the remarks (registers, scratch bytes per lane, LDS,
# occupancy of every kernel) are kept next to it as _obj/<file>.res, which tests/test_cpu_kernel_resources.py reads.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage"
mkdir -p _obj
pids=(); names=()
for f in hugs_*.hip; do
  extra=""
  [ "$f" = "hugs_stepfun.hip" ] && extra="-ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
  dep="$f"
  [ "$f" = "hugs_gemm_f16.hip" ] && dep="hugs_gemm.hip"      # (it is hugs_gemm.hip compiled with half operands)
  o="_obj/${f%.hip}.o"
  inc=""
  case "$f" in hugs_gemm.hip|hugs_gemm_f16.hip) inc="hugs_gemm_chain.inc hugs_gemm_p64.inc hugs_gemm_w4.inc hugs_gemm_dq.inc";; hugs_hashgrid.hip) inc="hugs_hashgrid_binned.inc";; esac      # (textually included)
  newer=0; for i_ in $inc; do [ "$i_" -nt "$o" ] && newer=1; done
  if [ ! -f "$o" ] || [ ! -f "${o%.o}.res" ] || [ "$f" -nt "$o" ] || [ "$dep" -nt "$o" ] || [ hugs_common.h -nt "$o" ] || [ build.sh -nt "$o" ] || [ $newer -eq 1 ]; then
    ( $HIPCC $FLAGS $extra -c "$f" -o "$o" 2> "${o%.o}.res" || { grep -v "remark:" "${o%.o}.res" >&2; rm -f "$o" "${o%.o}.res"; exit 1; } ) &
    pids+=($!); names+=("$f")
  fi
done
rc=0
for i in "${!pids[@]}"; do wait ${pids[$i]} || { echo "build.sh: ${names[$i]} failed" >&2; rc=1; }; done
[ $rc -eq 0 ] || exit 1
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libhugs_hip.so _obj/*.o
echo "built $(pwd)/libhugs_hip.so"
