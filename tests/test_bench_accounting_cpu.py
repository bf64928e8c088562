"""CPU: the flop / traffic accounting helpers of bench.py (no GPU work)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_executed_flops_per_algorithm():
    b = _bench()
    assert b._executed('conv3x3_winograd_f4_kernel', 144.0) == 36.0        # F(4x4,3x3): 36 of 144 multiplies per 4x4 tile
    assert b._executed('conv3x3_winograd_kernel<true, false>', 36.0) == 16.0  # F(2x2,3x3): 16 of 36
    assert b._executed('conv3x3_winograd_wgrad_kernel', 36.0) == 16.0
    assert b._executed('conv2d_mfma_kernel<3, 1, 4, 32, 1>', 10.0) == 10.0
    assert b._is_mfma('conv3x3_winograd_f4_kernel') and not b._is_mfma('upsample2x')


def test_roofline_fraction_is_useful_executed_over_peak():
    b = _bench()
    flops, secs = 4.0 * 157.3e12, 2.0                                       # algorithmic flops of an F(4x4) kernel over 2 s
    per = {'conv3x3_winograd_f4_kernel': [10, flops, secs, 0.0]}
    r = b.roofline_object(per, 1, 4.0, 'no_such_workload', False)
    assert r['kernel'] == 'conv3x3_winograd_f4_kernel' and r['bound'] == 'mfma'
    assert abs(r['frac'] - 0.5) < 1e-3 and r['frac'] <= 1.0                 # executed = algorithmic / 4
    assert abs(r['algorithmic_over_peak'] - 2.0) < 1e-3 and abs(r['algorithmic_frac_winograd'] - 0.5) < 1e-3
    assert r['traffic'] is None
    # with the C side's count of the MFMAs actually issued (2 % padded tiles): the fraction stays on the USEFUL flops - padded
    # tiles are not an achievement - and the issued figure is reported beside it
    per = {'conv3x3_winograd_f4_kernel': [10, flops, secs, 0.0, 1.02 * flops / 4.0]}
    r = b.roofline_object(per, 1, 4.0, 'no_such_workload', False)
    assert abs(r['frac'] - 0.5) < 1e-3 and abs(r['incl_padding_frac'] - 0.51) < 1e-3 and abs(r['padding_overhead'] - 0.02) < 1e-3
    assert abs(r['executed_without_padding_tflops'] - 0.5 * 157.3) < 0.1 and r['achieved'] == r['executed_without_padding_tflops']
    assert 'NOT counted' in r['definition']
    tab = b.kernel_table(per, 1, 4.0)['conv3x3_winograd_f4_kernel']
    assert abs(tab['frac_of_mfma_peak'] - 0.5) < 1e-3 and abs(tab['frac_incl_padding'] - 0.51) < 1e-3


def test_executed_flops_of_the_c_side_equal_the_pmc_count():
    """edvr_conv2d_executed_flops on the micro layer of profiles/r2/winograd_f4_micro_pmc.json (n = 20, 128 -> 128, 180 x 320):
    rocprofv3 counted SQ_INSTS_VALU_MFMA_MOPS_F32 = 169 574 400 over 8 launches of that layer, one count per 512 flops."""
    import ctypes
    from edvr_amd import _lib
    d = _lib.ConvDesc()
    d.c1, d.n, d.h, d.w, d.co, d.ks, d.stride, d.algo = 128, 20, 180, 320, 128, 3, 1, _lib.CONV_AUTO
    d.wpk_f4 = 16  # any non-null, 16-byte aligned address: only the eligibility rules look at it
    ex = ctypes.c_double(0.0)
    assert _lib.lib().edvr_conv2d_executed_flops(ctypes.byref(d), ctypes.byref(ex)) == 0
    pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r2', 'winograd_f4_micro_pmc.json')))
    assert ex.value == pmc['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512.0, (ex.value, pmc['SQ_INSTS_VALU_MFMA_MOPS_F32'])
    assert abs(ex.value / (2.0 * 9 * 128 * 128 * 20 * 180 * 320 / 4.0) - 1.0) < 0.03  # ~2 % tile padding over algorithmic / 4
    d.wpk_f4 = None  # without the F(4x4) weights the launch is the F(2x2) kernel: 8 x 32 pixel items, 16 positions
    assert _lib.lib().edvr_conv2d_executed_flops(ctypes.byref(d), ctypes.byref(ex)) == 0
    assert ex.value == 20 * 2 * (10 * 23) * 128 * (64.0 * 64 * 16 * 2)
    d.algo = _lib.CONV_DIRECT
    assert _lib.lib().edvr_conv2d_executed_flops(ctypes.byref(d), ctypes.byref(ex)) == 0
    assert ex.value == 2.0 * 9 * 128 * 128 * 20 * 180 * 320


def test_measured_traffic_aggregates_template_instantiations():
    b = _bench()
    tr, src = b.measured_traffic('edvr_l_x4_t5_180x320', 'conv3x3_winograd_f4_kernel')
    rep = json.load(open(os.path.join(ROOT, src)))
    hits = [v for k, v in rep['kernels'].items() if 'conv3x3_winograd_f4_kernel' in k]
    assert len(hits) >= 1 and tr['launches'] == sum(v['launches'] for v in hits)
    want = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in hits) / tr['launches']
    assert abs(tr['hbm_bytes_per_launch'] - want) < 1.0


# ------------------------------------------------------------------ the ONE line the driver parses (round 5's 24 KB line did not)
def _full_result(b):
    """A full report of the shape the default run produces: round 5's committed one with every optional leg present."""
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r5', 'bench_default_run.json')))
    full['dtype'] = b.DTYPE
    return full


def test_compact_line_is_small_strict_json_with_the_contract_fields():
    b = _bench()
    full = _full_result(b)
    line = b.compact_line(full)
    assert '\n' not in line and len(line.encode()) < 6000, len(line)
    got = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))  # NaN / Infinity would raise
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in got, k
    assert got['value'] == full['value'] and got['ms_per_step'] == full['ms_per_step'] and got['vs_baseline'] is None
    assert len(got['dtype']) <= 100 and 'workload' in got['config'] and 'model' not in got['config']
    r = got['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert 'traffic' in r and r['kernel'] == full['roofline']['kernel']
    c = got['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    assert got['parity']['ok'] is True and got['fp32_mfma']['value'] == full['fp32_mfma']['value']
    assert got['train']['iters_per_sec'] == full['train']['iters_per_sec'] and got['train']['parity']['ok'] is True
    assert got['train']['fp32_mfma']['iters_per_sec'] == full['train']['fp32_mfma']['iters_per_sec']
    assert got['target_4k']['value'] == full['target_4k']['value']


def test_compact_line_survives_non_finite_numbers_and_oversized_legs():
    b = _bench()
    full = _full_result(b)
    full['parity']['max_rel_err'] = float('nan')
    full['train']['parity']['grad_rel_err_max'] = float('inf')
    full['configs'] = {f'configs[{i}] a very long description of a configuration, number {i}': {'clips_per_sec': float(i)} for i in range(400)}
    line = b.compact_line(full)
    assert len(line.encode()) < 6000
    got = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert got['parity']['max_rel_err'] is None and got['train']['parity']['grad_rel_err_max'] is None
    assert 'configs' not in got and 'roofline' in got and 'cpu_baseline' in got  # optional legs go first, never the contract


def test_emit_prints_the_compact_line_last_and_writes_the_full_report(tmp_path, capsys):
    b = _bench()
    full = _full_result(b)
    full['train']['parity']['grad_rel_err_max'] = float('nan')
    path = str(tmp_path / 'bench_full.json')
    b.emit(full, 1, path)
    assert capsys.readouterr().out == '' and not os.path.exists(path)  # ranks other than 0 print nothing
    b.emit(full, 0, path)
    out = capsys.readouterr().out
    lines = out.strip().split('\n')
    assert len(lines) == 1 and json.loads(lines[-1])['value'] == full['value']
    rep = json.load(open(path), parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert rep['kernels'] and rep['train']['parity']['grad_rel_err_max'] is None


def test_training_mode_line_carries_iters_per_sec():
    b = _bench()
    full = _full_result(b)
    full.update(metric='EDVR-L x4 training clips/sec (= iters/sec x global batch)', iters_per_sec=9.44)
    got = json.loads(b.compact_line(full))
    assert got['iters_per_sec'] == 9.44
