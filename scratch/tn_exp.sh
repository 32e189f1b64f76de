#!/bin/bash
# scratch: builds TN-kernel measurement variants (HUGS_TN_EXP bit mask) -> scratch/libtnN.so
cd "$(dirname "$0")/.."
for e in 0 1 2 3 4 5 6; do EXTRA="-DHUGS_TN_EXP=$e" scratch/build_variant.sh libtn$e & done; wait
