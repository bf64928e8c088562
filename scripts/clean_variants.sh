#!/bin/bash
# Remove experiment builds (scripts/exp/f4s_flags.py, scripts/build_variant.sh): they are git-ignored but would travel to the GPU box with every
# gpurun snapshot (103 MB per lease).  The product library edvr_amd/lib/libedvr_amd.so and its objects stay.
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$R"/edvr_amd/lib/variants "$R"/edvr_amd/build/flagv_* "$R"/edvr_amd/build/variant_* "$R"/scripts/micro/f4s_loop "$R"/scripts/micro/mfma16_prices \
       "$R"/scripts/micro/mfma_prices "$R"/scripts/micro/mfma_valu "$R"/scripts/micro/mfma_boundary "$R"/scripts/micro/atomics "$R"/scripts/micro/lds_atomics
du -sh "$R"/edvr_amd/lib "$R"/edvr_amd/build 2>/dev/null
