"""TEST INFRASTRUCTURE ONLY - generate tests/golden/*.pt (run in the authoring container).

    python -m oracle.make_golden

Golden vectors come from the REFERENCE ITSELF wherever it can execute here:
  dcn_*.pt   - inputs + outputs/gradients of the reference's own DCNv2 kernels
  dcn1_*.pt  - the same for DCNv1 (DeformConv), from the reference's own deformable_im2col / col2im / col2im_coord kernels
               (deform_conv_cuda_kernel.cu compiled serially for the CPU: oracle/_ref), fp64
  edvr_*.pt  - seeded input + output of the reference's own Python network (basicsr/models/archs/edvr_arch.py
               imported unchanged, oracle/ref_import.py) with the DCN op supplied by oracle/_ref
Small shapes only: the fixtures are committed.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import dcn_oracle as O, ref_import  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')

DCN_CASES = {
    # name: (B, C, H, W, Co, k, stride, pad, dil, groups, dg, sigma, mode)
    'edvr_like': (1, 16, 10, 12, 16, 3, 1, 1, 1, 1, 8, 1.5, 'rand'),
    'zero_offset': (1, 16, 8, 8, 16, 3, 1, 1, 1, 1, 8, 0.0, 'rand'),
    'integer_taps': (1, 16, 8, 8, 8, 3, 1, 1, 1, 1, 8, 2.0, 'int'),
    'half_taps': (1, 8, 7, 9, 8, 3, 1, 1, 1, 1, 4, 2.0, 'half'),
    'out_of_bounds': (1, 8, 6, 6, 8, 3, 1, 1, 1, 1, 2, 8.0, 'rand'),
    'stride2_groups2': (2, 8, 7, 9, 6, 3, 2, 1, 1, 2, 2, 1.5, 'rand'),
    'dilation2': (1, 8, 10, 10, 4, 3, 1, 2, 2, 1, 4, 2.0, 'rand'),
}
EDVR_CASES = ('M_T5', 'M_noTSA', 'L_deblur_hr')


def dcn_case(name):
    B, C, H, W, Co, k, stride, pad, dil, groups, dg, sigma, mode = DCN_CASES[name]
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    Ho, Wo = O._out_hw(H, W, k, k, stride, pad, dil)
    dt = torch.float64
    x = torch.randn(B, C, H, W, generator=g, dtype=dt)
    w = torch.randn(Co, C // groups, k, k, generator=g, dtype=dt) * 0.1
    b = torch.randn(Co, generator=g, dtype=dt)
    off = torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g, dtype=dt) * sigma
    if mode == 'int':
        off = off.round()
    elif mode == 'half':
        off = off.round() + 0.5
    m = torch.rand(B, dg * k * k, Ho, Wo, generator=g, dtype=dt)
    dy = torch.randn(B, Co, Ho, Wo, generator=g, dtype=dt)
    cfg = (stride, pad, dil, groups, dg)
    y = O.ref_forward(x, off, m, w, b, *cfg)
    dx, doff, dm, dw, db = O.ref_backward(x, off, m, w, dy, True, *cfg)
    return dict(cfg=cfg, x=x, offset=off, mask=m, weight=w, bias=b, dy=dy, y=y, dx=dx, doffset=doff, dmask=dm,
                dweight=dw, dbias=db)


DCN1_CASES = ('edvr_like', 'half_taps', 'stride2_groups2', 'dilation2')  # same shapes; DCNv1 = no mask, no bias


def dcn1_rect_case():
    """DCNv1 with RECTANGULAR stride / padding / dilation pairs (deform_conv.py:33-36,56-58: h first), groups 2, dg 2."""
    B, C, H, W, Co, k, groups, dg = 2, 8, 9, 11, 6, 3, 2, 2
    stride, pad, dil = (2, 1), (1, 2), (1, 2)
    g = torch.Generator().manual_seed(sum(map(ord, 'dcn1_rect')))
    Ho, Wo = O._out_hw(H, W, k, k, stride, pad, dil)
    dt = torch.float64
    x = torch.randn(B, C, H, W, generator=g, dtype=dt)
    w = torch.randn(Co, C // groups, k, k, generator=g, dtype=dt) * 0.1
    off = torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g, dtype=dt) * 1.5
    dy = torch.randn(B, Co, Ho, Wo, generator=g, dtype=dt)
    cfg = (stride, pad, dil, groups, dg)
    y = O.ref_dcn1_forward(x, off, w, *cfg)
    dx, doff, dw = O.ref_dcn1_backward(x, off, w, dy, *cfg)
    return dict(cfg=cfg, x=x, offset=off, weight=w, dy=dy, y=y, dx=dx, doffset=doff, dweight=dw)


def dcn1_case(name):
    d = dcn_case(name)  # same seeded inputs
    cfg = d['cfg']
    y = O.ref_dcn1_forward(d['x'], d['offset'], d['weight'], *cfg)
    dx, doff, dw = O.ref_dcn1_backward(d['x'], d['offset'], d['weight'], d['dy'], *cfg)
    return dict(cfg=cfg, x=d['x'], offset=d['offset'], weight=d['weight'], dy=d['dy'], y=y, dx=dx, doffset=doff, dweight=dw)


def edvr_case(name):
    from util_edvr import CONFIGS, randomize_offsets
    kwargs, shape = CONFIGS[name]
    ea, _ = ref_import.load(dcn=lambda *a: O.ref_forward(*a))
    torch.manual_seed(10)
    net = randomize_offsets(ea.EDVR(**kwargs)).eval()
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        y = net(x)
    # weights are reproducible from the seed; tests rebuild them and compare `param_checksum`
    checksum = float(sum(p.double().abs().sum() for p in net.parameters()))
    return dict(kwargs=kwargs, x=x, y=y, param_checksum=checksum, n_keys=len(net.state_dict()),
                n_params=sum(p.numel() for p in net.parameters()))


def psnr_case():
    """PSNR of the reference's own calculate_psnr (basicsr/metrics/psnr_ssim.py) on tensor2img-converted random tensors
    (uint8, HWC, BGR as tensor2img produces them), for crop_border / test_y_channel combinations."""
    import sys
    import types
    import importlib.util
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))  # psnr_ssim.py imports cv2 for SSIM only
    # the package __init__ files pull in torchvision / lmdb: load the three files the PSNR path needs under bare parent packages
    for pkg in ('basicsr', 'basicsr.utils', 'basicsr.metrics'):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join('/root/reference', rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    load('basicsr.utils.matlab_functions', 'basicsr/utils/matlab_functions.py')
    load('basicsr.metrics.metric_util', 'basicsr/metrics/metric_util.py')
    calculate_psnr = load('basicsr.metrics.psnr_ssim', 'basicsr/metrics/psnr_ssim.py').calculate_psnr
    from oracle import edvr_oracle as EO
    g = torch.Generator().manual_seed(77)
    pred = torch.rand(3, 3, 24, 36, generator=g) * 1.2 - 0.1  # some values outside [0, 1]: tensor2img clamps
    gt = (pred + 0.05 * torch.randn(3, 3, 24, 36, generator=g)).clamp(-0.2, 1.2)
    gt[2] = pred[2]                                           # identical image: inf
    gray_p, gray_g = pred[:, :1].contiguous(), gt[:, :1].contiguous()
    def hwc_bgr(t):  # what tensor2img returns for a 3-channel RGB tensor (rgb2bgr=True), CHW -> HWC
        u = EO.tensor2img_uint8(t).numpy()
        return u[::-1].transpose(1, 2, 0) if u.shape[0] == 3 else u.transpose(1, 2, 0)
    cases = []
    for crop in (0, 2):
        for ych in (False, True):
            cases.append(dict(crop_border=crop, test_y_channel=ych, gray=False,
                              psnr=[float(calculate_psnr(hwc_bgr(pred[i]), hwc_bgr(gt[i]), crop, 'HWC', ych)) for i in range(3)]))
    cases.append(dict(crop_border=1, test_y_channel=False, gray=True,
                      psnr=[float(calculate_psnr(hwc_bgr(gray_p[i]), hwc_bgr(gray_g[i]), 1, 'HWC', False)) for i in range(3)]))
    torch.save(dict(pred=pred, gt=gt, cases=cases), os.path.join(OUT, 'psnr.pt'))


def ssim_case():
    """SSIM of the reference's own calculate_ssim (basicsr/metrics/psnr_ssim.py:54-141) on the tensors of psnr.pt.  cv2 is absent:
    its two calls are supplied from their documented definitions - getGaussianKernel(11, 1.5) = normalised exp(-(i-5)^2 / 2 sigma^2),
    filter2D = correlation with BORDER_REFLECT_101 (scipy.ndimage.correlate, mode 'mirror'; only the valid region is used)."""
    import sys
    import types
    import importlib.util
    import numpy as np
    from scipy import ndimage
    cv2 = types.ModuleType('cv2')

    def gk(ksize, sigma):
        x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
        k = np.exp(-0.5 / (sigma * sigma) * x * x)
        return (k / k.sum()).reshape(-1, 1)
    cv2.getGaussianKernel = gk
    cv2.filter2D = lambda img, ddepth, kernel: ndimage.correlate(img, kernel, mode='mirror')
    saved = sys.modules.get('cv2')
    sys.modules['cv2'] = cv2
    for pkg in ('basicsr', 'basicsr.utils', 'basicsr.metrics'):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join('/root/reference', rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    load('basicsr.utils.matlab_functions', 'basicsr/utils/matlab_functions.py')
    load('basicsr.metrics.metric_util', 'basicsr/metrics/metric_util.py')
    calculate_ssim = load('basicsr.metrics.psnr_ssim', 'basicsr/metrics/psnr_ssim.py').calculate_ssim
    from oracle import edvr_oracle as EO
    d = torch.load(os.path.join(OUT, 'psnr.pt'))
    pred, gt = d['pred'], d['gt']

    def hwc_bgr(t):
        u = EO.tensor2img_uint8(t).numpy()
        return u[::-1].transpose(1, 2, 0) if u.shape[0] == 3 else u.transpose(1, 2, 0)
    cases = []
    for crop in (0, 2):
        for ych in (False, True):
            cases.append(dict(crop_border=crop, test_y_channel=ych, gray=False,
                              ssim=[float(calculate_ssim(hwc_bgr(pred[i]), hwc_bgr(gt[i]), crop, 'HWC', ych)) for i in range(3)]))
    cases.append(dict(crop_border=1, test_y_channel=False, gray=True,
                      ssim=[float(calculate_ssim(hwc_bgr(pred[i, :1]), hwc_bgr(gt[i, :1]), 1, 'HWC', False)) for i in range(3)]))
    torch.save(dict(cases=cases, note='inputs: pred / gt of psnr.pt'), os.path.join(OUT, 'ssim.pt'))
    if saved is not None:
        sys.modules['cv2'] = saved


def frame_indices_case():
    """generate_frame_indices of the reference (basicsr/data/data_util.py:35-88), executed from its source text (the module
    itself imports cv2 / lmdb helpers): every centre index of sequences of 7, 10 and 100 frames, 5- and 7-frame windows."""
    import json
    src = open('/root/reference/basicsr/data/data_util.py').read()
    ns = {}
    exec(src[src.index('def generate_frame_indices('):src.index('def paired_paths_from_lmdb(')], ns)
    ref = ns['generate_frame_indices']
    gold = {f'{pad}/{nf}/{T}': [ref(i, T, nf, pad) for i in range(T)]
            for pad in ('replicate', 'reflection', 'reflection_circle', 'circle') for nf in (5, 7) for T in (7, 10, 100)}
    with open(os.path.join(OUT, 'frame_indices.json'), 'w') as f:
        json.dump(gold, f)


def lr_sched_case():
    """Learning rates of the reference's own schedulers (basicsr/models/lr_scheduler.py) for two parameter groups."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('ref_lr_scheduler', '/root/reference/basicsr/models/lr_scheduler.py')
    ls = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ls)
    out = {}
    for name, make in (('cosine', lambda o: ls.CosineAnnealingRestartLR(o, periods=[50, 50, 30, 70], restart_weights=[1, 0.5, 0.5, 0.25], eta_min=1e-7)),
                       ('multistep', lambda o: ls.MultiStepRestartLR(o, milestones=[20, 40, 90, 120], gamma=0.5, restarts=[0, 80], restart_weights=[1, 0.5]))):
        w = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.SGD([{'params': [w[0]], 'lr': 4e-4}, {'params': [w[1]], 'lr': 1e-4}], lr=4e-4)
        sch = make(opt)
        lrs = []
        for _ in range(200):
            lrs.append([g['lr'] for g in opt.param_groups])
            opt.step()
            sch.step()
        out[name] = lrs
    with open(os.path.join(OUT, 'lr_sched.json'), 'w') as f:
        json.dump(out, f)


def data_case():
    """Input pipeline: outputs of the reference's own REDSDataset.__getitem__ (basicsr/data/reds_dataset.py), transforms.py,
    img_util.py (imfrombytes / img2tensor) and EnlargedSampler (data_sampler.py), executed from /root/reference.  cv2 is not
    installed here: a numpy stand-in provides the three calls on this path (flip in place, cvtColor BGR2RGB, imdecode of the
    raw container below); the storage backend is an in-memory client serving oracle.data_oracle.synthetic_frame."""
    import importlib.util
    import logging
    import random
    import struct
    import sys
    import types
    import numpy as np
    from oracle import data_oracle as DO

    cv2 = types.ModuleType('cv2')
    cv2.IMREAD_COLOR, cv2.IMREAD_GRAYSCALE, cv2.IMREAD_UNCHANGED, cv2.COLOR_BGR2RGB = 1, 0, -1, 4

    def flip(src, code, dst=None):
        out = src[:, ::-1].copy() if code == 1 else src[::-1].copy()
        if dst is not None:
            dst[...] = out
            return dst
        return out

    def imdecode(buf, flag):  # b'RAW0' + <h, w> + BGR bytes
        raw = buf.tobytes()
        assert raw[:4] == b'RAW0' and flag == cv2.IMREAD_COLOR
        h, w = struct.unpack('<ii', raw[4:12])
        return np.frombuffer(raw[12:], np.uint8).reshape(h, w, 3).copy()

    cv2.flip, cv2.imdecode = flip, imdecode
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[:, :, ::-1])
    saved = {k: sys.modules.get(k) for k in ('cv2', 'torchvision', 'torchvision.utils', 'basicsr', 'basicsr.utils', 'basicsr.data',
                                             'basicsr.utils.flow_util', 'basicsr.data.transforms')}
    sys.modules['cv2'] = cv2
    tv, tvu = types.ModuleType('torchvision'), types.ModuleType('torchvision.utils')
    tvu.make_grid = None
    sys.modules['torchvision'], sys.modules['torchvision.utils'] = tv, tvu

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join('/root/reference', rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    for pkg in ('basicsr', 'basicsr.utils', 'basicsr.data'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    img_util = load('basicsr.utils.img_util', 'basicsr/utils/img_util.py')
    LQ_HW, SCALE = (20, 28), 4

    class FileClient:  # stands for basicsr/utils/file_client.py (disk backend): path -> bytes
        def __init__(self, backend, **kw):
            assert backend == 'disk'

        def get(self, path, client_key):
            clip, frame = str(path).split('/')[-2:]
            h, w = LQ_HW if client_key == 'lq' else (LQ_HW[0] * SCALE, LQ_HW[1] * SCALE)
            return b'RAW0' + struct.pack('<ii', h, w) + DO.synthetic_frame(client_key, clip, frame[:-4], h, w).tobytes()

    u = sys.modules['basicsr.utils']
    u.FileClient, u.get_root_logger = FileClient, lambda: logging.getLogger('ref')
    u.imfrombytes, u.img2tensor = img_util.imfrombytes, img_util.img2tensor
    fl = types.ModuleType('basicsr.utils.flow_util')
    fl.dequantize_flow = None
    sys.modules['basicsr.utils.flow_util'] = fl
    load('basicsr.data.transforms', 'basicsr/data/transforms.py')
    reds = load('basicsr.data.reds_dataset', 'basicsr/data/reds_dataset.py')
    sampler = load('ref_data_sampler', 'basicsr/data/data_sampler.py')

    meta = [f'{c:03d} 100 (80,112,3)\n' for c in (0, 1, 2, 11, 15, 20, 240, 250, 269)]
    meta_path = '/tmp/edvr_meta_info_golden.txt'
    with open(meta_path, 'w') as f:
        f.writelines(meta)
    out = dict(meta=meta, lq_hw=LQ_HW, scale=SCALE, samples=[], keys={}, sampler=[])
    variants = [dict(num_frame=5, interval_list=[1], random_reverse=False, use_flip=True, use_rot=True, val_partition='REDS4'),
                dict(num_frame=7, interval_list=[1, 2, 3], random_reverse=True, use_flip=True, use_rot=True, val_partition='official'),
                dict(num_frame=5, interval_list=[2], random_reverse=True, use_flip=False, use_rot=True, val_partition='REDS4'),
                dict(num_frame=3, interval_list=[1, 5], random_reverse=False, use_flip=True, use_rot=False, val_partition='official'),
                dict(num_frame=5, interval_list=[1], random_reverse=False, use_flip=False, use_rot=False, val_partition='REDS4')]
    for vi, v in enumerate(variants):
        opt = dict(v, dataroot_gt='/data/gt', dataroot_lq='/data/lq', dataroot_flow=None, meta_info_file=meta_path,
                   io_backend=dict(type='disk'), gt_size=32, scale=SCALE)
        ds = reds.REDSDataset(opt)
        out['keys'][v['val_partition']] = (len(ds), ds.keys[:3], ds.keys[-3:])
        for seed, index in ((1, 0), (2, 1), (3, 98), (4, 99), (5, 150), (6, 297), (7, len(ds) - 1), (8, 251)):
            random.seed(seed * 1000 + vi)
            item = ds[index]
            nxt = random.random()  # position of the random stream after the sample
            lq8, gt8 = (item['lq'] * 255).round().to(torch.uint8), (item['gt'] * 255).round().to(torch.uint8)
            assert torch.equal(lq8.float() / 255., item['lq']) and torch.equal(gt8.float() / 255., item['gt'])
            out['samples'].append(dict(variant=vi, opt=v, seed=seed * 1000 + vi, index=index, key=item['key'], lq_u8=lq8, gt_u8=gt8,
                                       next_random=nxt))

    class Sized:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    for n, rep, ratio in ((37, 2, 1), (37, 3, 4), (100, 8, 200), (5, 2, 1)):
        for epoch in (0, 3):
            for rank in range(rep):
                smp = sampler.EnlargedSampler(Sized(n), rep, rank, ratio)
                smp.set_epoch(epoch)
                idx = list(smp)
                out['sampler'].append(dict(n=n, replicas=rep, rank=rank, ratio=ratio, epoch=epoch, first=idx[:16], last=idx[-4:],
                                           count=len(idx), checksum=int(sum((i + 1) * (k + 1) for k, i in enumerate(idx)) % (1 << 61))))
    # imfrombytes: all 256 byte values (the float32 division the device kernel has to reproduce bit for bit)
    ramp = np.arange(256, dtype=np.uint8).reshape(1, 256, 1).repeat(3, 2)
    out['div255'] = torch.from_numpy(img_util.imfrombytes(b'RAW0' + struct.pack('<ii', 1, 256) + ramp.tobytes(), float32=True)[0, :, 0].copy())
    # read_img_seq's conversion (data_util.py:26-31) on two whole frames, executed from the reference's img2tensor
    fr = [DO.synthetic_frame('lq', '000', f'{i:08d}', 18, 30) for i in range(2)]
    out['read_img_seq'] = torch.stack(img_util.img2tensor([f.astype(np.float32) / 255. for f in fr], bgr2rgb=True, float32=True), 0)
    torch.save(out, os.path.join(OUT, 'data_pipeline.pt'))
    for k, m in saved.items():
        if m is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = m


def vimeo_train_case():
    """Vimeo90KDataset (training; basicsr/data/vimeo90k_dataset.py:10-134) of the reference, executed from its source on synthetic
    7-frame sequences through the same numpy stand-ins for cv2 as data_case.  The class REVERSES ITS NEIGHBOUR LIST IN PLACE
    (:82-83): the state carries over from one sample to the next, so the samples are drawn from one dataset object in sequence and
    the fixture records that order."""
    import importlib.util
    import logging
    import random
    import struct
    import sys
    import types
    import numpy as np
    from oracle import data_oracle as DO

    cv2 = types.ModuleType('cv2')
    cv2.IMREAD_COLOR, cv2.IMREAD_GRAYSCALE, cv2.IMREAD_UNCHANGED, cv2.COLOR_BGR2RGB = 1, 0, -1, 4

    def flip(src, code, dst=None):
        out = src[:, ::-1].copy() if code == 1 else src[::-1].copy()
        if dst is not None:
            dst[...] = out
            return dst
        return out

    def imdecode(buf, flag):  # b'RAW0' + <h, w> + BGR bytes
        raw = buf.tobytes()
        assert raw[:4] == b'RAW0' and flag == cv2.IMREAD_COLOR
        h, w = struct.unpack('<ii', raw[4:12])
        return np.frombuffer(raw[12:], np.uint8).reshape(h, w, 3).copy()

    cv2.flip, cv2.imdecode = flip, imdecode
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[:, :, ::-1])
    names = ('cv2', 'torchvision', 'torchvision.utils', 'basicsr', 'basicsr.utils', 'basicsr.data', 'basicsr.utils.img_util',
             'basicsr.data.transforms', 'basicsr.data.vimeo90k_dataset')
    saved = {k: sys.modules.get(k) for k in names}
    sys.modules['cv2'] = cv2
    tv, tvu = types.ModuleType('torchvision'), types.ModuleType('torchvision.utils')
    tvu.make_grid = None
    sys.modules['torchvision'], sys.modules['torchvision.utils'] = tv, tvu

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join('/root/reference', rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    for pkg in ('basicsr', 'basicsr.utils', 'basicsr.data'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    img_util = load('basicsr.utils.img_util', 'basicsr/utils/img_util.py')
    LQ_HW, SCALE = (16, 28), 4

    class FileClient:  # disk backend: <root>/<clip>/<seq>/im<n>.png -> bytes of synthetic_frame(kind, '<clip>/<seq>', 'im<n>')
        def __init__(self, backend, **kw):
            assert backend == 'disk'

        def get(self, path, client_key):
            clip, seq, frame = str(path).split('/')[-3:]
            h, w = LQ_HW if client_key == 'lq' else (LQ_HW[0] * SCALE, LQ_HW[1] * SCALE)
            return b'RAW0' + struct.pack('<ii', h, w) + DO.synthetic_frame(client_key, f'{clip}/{seq}', frame[:-4], h, w).tobytes()

    u = sys.modules['basicsr.utils']
    u.FileClient, u.get_root_logger = FileClient, lambda: logging.getLogger('ref')
    u.imfrombytes, u.img2tensor = img_util.imfrombytes, img_util.img2tensor
    load('basicsr.data.transforms', 'basicsr/data/transforms.py')
    vim = load('basicsr.data.vimeo90k_dataset', 'basicsr/data/vimeo90k_dataset.py')

    meta = [f'{c:05d}/{q:04d} 7 (64,112,3)\n' for c, q in ((1, 1), (1, 2), (1, 266), (2, 1), (3, 7), (96, 1000))]
    meta_path = '/tmp/edvr_meta_info_vimeo_golden.txt'
    with open(meta_path, 'w') as f:
        f.writelines(meta)
    out = dict(meta=meta, lq_hw=LQ_HW, scale=SCALE, runs=[])
    variants = [dict(num_frame=7, random_reverse=False, use_flip=True, use_rot=True),
                dict(num_frame=5, random_reverse=True, use_flip=True, use_rot=True),
                dict(num_frame=3, random_reverse=True, use_flip=False, use_rot=True),
                dict(num_frame=5, random_reverse=True, use_flip=True, use_rot=False)]  # (num_frame = 1 crashes in the reference: :116)
    for vi, v in enumerate(variants):
        opt = dict(v, dataroot_gt='/data/gt', dataroot_lq='/data/lq', meta_info_file=meta_path, io_backend=dict(type='disk'), gt_size=32,
                   scale=SCALE)
        ds = vim.Vimeo90KDataset(opt)
        assert len(ds) == len(meta)
        random.seed(500 + vi)
        samples = []
        for index in (0, 5, 2, 2, 1, 4, 3, 0):  # one dataset object, in sequence: the neighbour list's orientation persists
            item = ds[index]
            lq8, gt8 = (item['lq'] * 255).round().to(torch.uint8), (item['gt'] * 255).round().to(torch.uint8)
            assert torch.equal(lq8.float() / 255., item['lq']) and torch.equal(gt8.float() / 255., item['gt'])
            samples.append(dict(index=index, key=item['key'], lq_u8=lq8, gt_u8=gt8, neighbors=list(ds.neighbor_list)))
        out['runs'].append(dict(opt=v, seed=500 + vi, samples=samples, next_random=random.random()))
    torch.save(out, os.path.join(OUT, 'vimeo90k_train.pt'))
    for k, m in saved.items():
        if m is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = m


def video_test_case():
    """VideoTestDataset (basicsr/data/video_test_dataset.py) of the reference, executed from its source on a small PNG tree
    (2 folders x 7 frames of synthetic_frame, a dot file that scandir has to skip), for cache_data on / off and two padding
    modes.  Its helpers are the reference's own source texts (read_img_seq, generate_frame_indices from data_util.py, scandir
    from utils/misc.py, img2tensor / mod_crop); cv2.imread - cv2 is absent - is PIL's PNG decoder returning BGR."""
    import importlib.util
    import logging
    import shutil
    import sys
    import types
    import numpy as np
    from PIL import Image
    from oracle import data_oracle as DO

    root = '/tmp/edvr_vt_golden'
    shutil.rmtree(root, ignore_errors=True)
    spec = dict(folders=['000', '011'], frames=7, lq_hw=(10, 14), scale=4)
    for kind in ('lq', 'gt'):
        h, w = spec['lq_hw'] if kind == 'lq' else (spec['lq_hw'][0] * 4, spec['lq_hw'][1] * 4)
        for folder in spec['folders']:
            os.makedirs(os.path.join(root, kind, folder))
            for f in range(spec['frames']):
                Image.fromarray(np.ascontiguousarray(DO.synthetic_frame(kind, folder, f'{f:08d}', h, w)[:, :, ::-1])).save(
                    os.path.join(root, kind, folder, f'{f:08d}.png'))
            open(os.path.join(root, kind, folder, '.hidden'), 'w').close()
    saved = {k: sys.modules.get(k) for k in ('cv2', 'torchvision', 'torchvision.utils', 'basicsr', 'basicsr.utils', 'basicsr.data',
                                             'basicsr.data.data_util', 'basicsr.data.transforms')}
    cv2 = types.ModuleType('cv2')
    cv2.COLOR_BGR2RGB = 4
    cv2.imread = lambda path: np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1])
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[:, :, ::-1])
    sys.modules['cv2'] = cv2
    tv, tvu = types.ModuleType('torchvision'), types.ModuleType('torchvision.utils')
    tvu.make_grid = None
    sys.modules['torchvision'], sys.modules['torchvision.utils'] = tv, tvu
    for pkg in ('basicsr', 'basicsr.utils', 'basicsr.data'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m

    def load(name, rel):
        sp = importlib.util.spec_from_file_location(name, os.path.join('/root/reference', rel))
        mod = importlib.util.module_from_spec(sp)
        sys.modules[name] = mod
        sp.loader.exec_module(mod)
        return mod

    img_util = load('basicsr.utils.img_util', 'basicsr/utils/img_util.py')
    misc_src = open('/root/reference/basicsr/utils/misc.py').read()
    ns = {'os': os, 'osp': os.path}
    exec(misc_src[misc_src.index('def scandir('):misc_src.index('def check_resume(')], ns)
    u = sys.modules['basicsr.utils']
    u.get_root_logger, u.scandir, u.img2tensor = (lambda: logging.getLogger('ref')), ns['scandir'], img_util.img2tensor
    tr = load('basicsr.data.transforms', 'basicsr/data/transforms.py')
    du_src = open('/root/reference/basicsr/data/data_util.py').read()
    du = types.ModuleType('basicsr.data.data_util')
    du.__dict__.update(cv2=cv2, np=np, torch=torch, osp=os.path, mod_crop=tr.mod_crop, img2tensor=img_util.img2tensor, scandir=ns['scandir'])
    exec(du_src[du_src.index('def read_img_seq('):du_src.index('def paired_paths_from_lmdb(')], du.__dict__)
    du.duf_downsample = None
    sys.modules['basicsr.data.data_util'] = du
    vt = load('basicsr.data.video_test_dataset', 'basicsr/data/video_test_dataset.py')
    out = dict(spec=spec, runs=[])
    for cache, padding, nf in ((True, 'reflection_circle', 5), (False, 'replicate', 3)):
        opt = dict(name='REDS4', dataroot_gt=os.path.join(root, 'gt'), dataroot_lq=os.path.join(root, 'lq'), io_backend=dict(type='disk'),
                   cache_data=cache, num_frame=nf, padding=padding)
        ds = vt.VideoTestDataset(opt)
        rel = lambda pth: os.path.relpath(pth, root)
        info = {k: ([rel(v) for v in vals] if k.endswith('path') else list(vals)) for k, vals in ds.data_info.items()}
        items = []
        for index in (0, 1, 6, 7, 13):
            it = ds[index]
            lq8, gt8 = (it['lq'] * 255).round().to(torch.uint8), (it['gt'] * 255).round().to(torch.uint8)
            assert torch.equal(lq8.float() / 255., it['lq']) and torch.equal(gt8.float() / 255., it['gt'])
            items.append(dict(index=index, lq_u8=lq8, gt_u8=gt8, folder=it['folder'], idx=it['idx'], border=it['border'], lq_path=rel(it['lq_path'])))
        out['runs'].append(dict(cache_data=cache, padding=padding, num_frame=nf, length=len(ds), data_info=info, items=items))
    # VideoTestVimeo90KDataset of the same file: index lists only (its items are read_img_seq of those paths)
    meta = os.path.join(root, 'vimeo_meta.txt')
    with open(meta, 'w') as f:
        f.writelines(f'{a:05d}/{b:04d} 7 (256,448,3)\n' for a, b in ((1, 266), (1, 268), (2, 10), (96, 1)))
    out['vimeo'] = []
    for nf in (7, 5, 3):
        ds = vt.VideoTestVimeo90KDataset(dict(name='Vimeo90K-Test', dataroot_gt='/data/v/gt', dataroot_lq='/data/v/lq', meta_info_file=meta,
                                              io_backend=dict(type='disk'), cache_data=False, num_frame=nf, padding='reflection_circle'))
        out['vimeo'].append(dict(num_frame=nf, length=len(ds), data_info={k: list(v) for k, v in ds.data_info.items()}))
    out['vimeo_meta'] = open(meta).read()
    torch.save(out, os.path.join(OUT, 'video_test.pt'))
    shutil.rmtree(root, ignore_errors=True)
    for k, m in saved.items():
        if m is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = m


def main():
    assert ref_import.available() and O.have_ref(), 'needs /root/reference and oracle/_ref (make -C oracle ref)'
    os.makedirs(OUT, exist_ok=True)
    for name in DCN_CASES:
        torch.save(dcn_case(name), os.path.join(OUT, f'dcn_{name}.pt'))
    for name in DCN1_CASES:
        torch.save(dcn1_case(name), os.path.join(OUT, f'dcn1_{name}.pt'))
    torch.save(dcn1_rect_case(), os.path.join(OUT, 'dcn1_rect.pt'))
    lr_sched_case()
    psnr_case()
    ssim_case()
    frame_indices_case()
    data_case()
    vimeo_train_case()
    video_test_case()
    for name in EDVR_CASES:
        torch.save(edvr_case(name), os.path.join(OUT, f'edvr_{name}.pt'))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
