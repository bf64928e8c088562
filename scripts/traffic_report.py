"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv files into per-kernel average bytes per launch.
usage: traffic_report.py OUT.json fetch.csv write.csv [calib_fetch.csv calib_write.csv]"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(?:void )?([\w:]+(?:<[^(]{0,60})?)', name)
    return (m.group(1) if m else name)[:90]


def load(path, counter):
    per = defaultdict(list)
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] == counter:
                per[short(row['Kernel_Name'])].append((float(row['Counter_Value']), int(row['Grid_Size'])))
    return per


def source_hash():
    """hash of the kernel sources this measurement was taken on (edvr_amd/build.py::source_hash)"""
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from edvr_amd.build import source_hash as h
    return h()


CALIB_BYTES = 20 * 128 * 180 * 320 * 4  # scripts/traffic_calib.py: every copy reads and writes this many bytes


def finalize(rep):
    """Calibrate the counters on the known-byte-count copies (MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half the bytes of a
    streaming read on gfx950; WRITE_SIZE is in KiB) and add corrected bytes per launch to every kernel."""
    cal = next((v for k, v in rep.get('calibration', {}).items() if 'copyBuffer' in k), None)
    if not cal or not cal['fetch'] or not cal['write']:
        return rep
    nf = max(1, len(cal['fetch']) // 2)  # first half of the copies: aligned 16 B/lane; second half: misaligned 4 B/lane
    f_fetch = CALIB_BYTES / (sum(cal['fetch'][:nf]) / nf)
    f_write = CALIB_BYTES / (sum(cal['write'][:nf]) / nf)
    rep['bytes_per_counter_unit'] = {'FETCH_SIZE': f_fetch, 'WRITE_SIZE': f_write, 'calibrated_on': f'{CALIB_BYTES} B device copies',
                                     'fetch_unit_misaligned_4B_lanes': CALIB_BYTES / (sum(cal['fetch'][nf:]) / max(1, len(cal['fetch'][nf:])))}
    for v in rep['kernels'].values():
        v['fetch_bytes_per_launch'] = v['fetch_avg'] * f_fetch
        v['write_bytes_per_launch'] = None if v['write_avg'] is None else v['write_avg'] * f_write
        v['hbm_bytes_per_launch'] = v['fetch_bytes_per_launch'] + (v['write_bytes_per_launch'] or 0.0)
    return rep


def main():
    if sys.argv[1] == '--refinalize':
        rep = finalize(json.load(open(sys.argv[2])))
        json.dump(rep, open(sys.argv[3], 'w'), indent=1)
        return
    out, fetch, write = sys.argv[1:4]
    rep = {'unit': 'fetch_avg / write_avg are raw rocprofv3 counter units per launch (FETCH_SIZE and WRITE_SIZE collected in separate '
                   '--pmc passes); *_bytes_per_launch are calibrated on known-size copies, see bytes_per_counter_unit',
           'kernels': {}}
    fp, wp = load(fetch, 'FETCH_SIZE'), load(write, 'WRITE_SIZE')
    for k in sorted(fp, key=lambda k: -sum(v for v, _ in fp[k])):
        f = [v for v, _ in fp[k]]
        w = [v for v, _ in wp.get(k, [])]
        rep['kernels'][k] = {'launches': len(f), 'fetch_avg': sum(f) / len(f), 'fetch_total': sum(f),
                             'write_avg': (sum(w) / len(w)) if w else None, 'write_total': sum(w) if w else None}
    if len(sys.argv) >= 6:
        cf, cw = load(sys.argv[4], 'FETCH_SIZE'), load(sys.argv[5], 'WRITE_SIZE')
        rep['calibration'] = {k: {'fetch': [v for v, _ in cf[k]], 'write': [v for v, _ in cw.get(k, [])], 'grid': [g for _, g in cf[k]]}
                              for k in cf if 'copy' in k.lower() or 'elementwise' in k.lower()}
    finalize(rep)
    rep['csrc_sha16'] = source_hash()
    with open(out, 'w') as f:
        json.dump(rep, f, indent=1)
    for k, v in list(rep['kernels'].items())[:12]:
        print(f"{k[:70]:70s} n={v['launches']:4d} fetch_avg={v['fetch_avg']:.4g} write_avg={v['write_avg'] if v['write_avg'] is None else round(v['write_avg'], 1)}")
    print(json.dumps(rep.get('calibration', {}))[:1500])


main()
