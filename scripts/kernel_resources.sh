#!/bin/bash
# Register / scratch / LDS usage of every kernel of one translation unit (hipcc remarks), one line per kernel:
#   scripts/kernel_resources.sh edvr_amd/csrc/winograd.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -c "$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys
cur = {}
def flush():
    if cur: print("{name}: VGPR {VGPRs} AGPR {AGPRs} SGPR {TotalSGPRs} scratch {ScratchSize} B/lane, LDS {LDS Size} B, occupancy {Occupancy}".format(**cur))
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1).split(" [")[0], m.group(2)
    if k == "Function Name":
        flush(); cur = {"name": v}
    else: cur[k] = v
flush()' | c++filt
