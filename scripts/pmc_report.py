"""Average rocprofv3 --pmc counter values per launch of one kernel: pmc_report.py KERNEL_SUBSTRING OUT.json pass1.csv [pass2.csv ...]"""
import csv
import json
import sys
from collections import defaultdict

def source_hash():
    """hash of the kernel sources this measurement was taken on (edvr_amd/build.py::source_hash)"""
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from edvr_amd.build import source_hash as h
    return h()


key, out = sys.argv[1], sys.argv[2]
acc, cnt = defaultdict(float), defaultdict(int)
for path in sys.argv[3:]:
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if key in row['Kernel_Name']:
                acc[row['Counter_Name']] += float(row['Counter_Value'])
                cnt[row['Counter_Name']] += 1
rep = {k: acc[k] / cnt[k] for k in sorted(acc)}
rep['_launches_per_counter'] = max(cnt.values()) if cnt else 0
rep['_kernel'] = key
rep['csrc_sha16'] = source_hash()
if 'SQ_BUSY_CYCLES' in rep and 'SQ_VALU_MFMA_BUSY_CYCLES' in rep and 'GRBM_GUI_ACTIVE' in rep:
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs in units of 4 cycles... report the ratio the guide uses
    rep['_mfma_busy_frac'] = rep["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * rep["GRBM_GUI_ACTIVE"] / 8)
json.dump(rep, open(out, 'w'), indent=1)
print(json.dumps(rep, indent=1))
