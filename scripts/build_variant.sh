#!/bin/bash
# build_variant.sh NAME "-DFLAG=..."  -> edvr_amd/lib/variants/libedvr_amd_NAME.so (A/B kernel experiments)
# Every translation unit is compiled with -DEDVR_VARIANT=NAME: edvr_version() of the result ends in "variant:NAME", which
# tests/conftest.py refuses and bench.py records - a variant loaded through EDVR_AMD_LIB cannot pass for the product.
set -e
cd "$(dirname "$0")/.."
mkdir -p edvr_amd/lib/variants edvr_amd/build/var_$1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -DEDVR_VARIANT=$1 $2"
for s in $(python -c "from edvr_amd.build import SOURCES; print(' '.join(x[:-4] for x in SOURCES))"); do
  hipcc $F -c edvr_amd/csrc/$s.hip -o edvr_amd/build/var_$1/$s.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o edvr_amd/lib/variants/libedvr_amd_$1.so edvr_amd/build/var_$1/*.o
echo built edvr_amd/lib/variants/libedvr_amd_$1.so
