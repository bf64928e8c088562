#!/bin/bash
# build_variant.sh NAME "-DFLAG=..."  -> edvr_amd/lib/variants/libedvr_amd_NAME.so (A/B kernel experiments)
set -e
cd "$(dirname "$0")/.."
mkdir -p edvr_amd/lib/variants edvr_amd/build/var_$1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast $2"
for s in api pack conv2d dcn dcn_any elementwise wgrad backward dcn_fused dcn_bwd_fused winograd winograd_f4 winograd_wgrad optim metrics conv_small conv1x1 data; do
  hipcc $F -c edvr_amd/csrc/$s.hip -o edvr_amd/build/var_$1/$s.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o edvr_amd/lib/variants/libedvr_amd_$1.so edvr_amd/build/var_$1/*.o
echo built edvr_amd/lib/variants/libedvr_amd_$1.so
