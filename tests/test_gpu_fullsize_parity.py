"""-m gpu: BASELINE.json configurations at their REAL sizes against the CPU oracle (same clip, same weights).

  configs[0]  EDVR-M x4, 5 frames, 64x64 LR crop, 1 clip            (the reference's own CPU-runnable plumbing case)
  configs[1]  EDVR-M x4, 5 frames, 180x320 LR -> 720x1280, ONE clip of the batch (the oracle needs a few seconds per clip)

Pass criteria (north_star: outputs match within 1e-3 dB PSNR in fp32): max |ours - oracle| / max|oracle| <= 2e-4 and
|PSNR(ours, gt) - PSNR(oracle, gt)| <= 1e-3 dB on a synthetic ground truth (lq seed 0, gt seed 1).
bench.py emits the same comparison for the headline EDVR-L workload in its `parity` object.

EDVR-L (128 channels, 40 reconstruction blocks - the network of the headline metric) is compared here as well, INCLUDING the
intermediates `aligned` / `fused` / `trunk` (the final output is close to bilinear(x_center) and nearly blind to PCD / TSA
errors):
  EDVR-L x4, 5 frames, 180x320, one clip                 vs the CPU oracle (fp32 torch ops + C DCNv2)
  configs[2]  EDVR-L x4, 7 frames, 180x320, one clip     \
  configs[4]  EDVR-L deblur (hr_in, predeblur), 720x1280   > vs the same functional oracle (oracle/edvr_oracle.py) executed on the
  north_star  EDVR-L x4, 5 frames, 720x1280 -> 4K        /   GPU in stock PyTorch-ROCm fp32 ops (convs as F.unfold + rocBLAS GEMM
                                                             or MIOpen, pure-torch DCNv2) - the CPU oracle would need minutes
                                                             per clip at these sizes
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
EDVR_M = dict(num_feat=64, num_frame=5, num_reconstruct_block=10, center_frame_idx=2)


@pytest.mark.parametrize('hw', [(64, 64), (180, 320)])
def test_edvr_m_full_size_output_and_psnr_match_the_oracle(gpu, hw):
    from edvr_amd import EDVR
    from oracle import dcn_oracle as O, edvr_oracle as EO
    from util_edvr import randomize_offsets
    torch.manual_seed(10)
    net = randomize_offsets(EDVR(**EDVR_M)).eval()
    x = torch.rand(1, 5, 3, *hw, generator=torch.Generator().manual_seed(0))
    gt = torch.rand(1, 3, 4 * hw[0], 4 * hw[1], generator=torch.Generator().manual_seed(1))
    taps_ref, taps = {}, {}
    with torch.no_grad():
        ref = EO.edvr_forward(net.state_dict(), x, center=2, dcn=O.dcnv2_c, taps=taps_ref)
        net = net.to(gpu)
        net.taps = taps
        out = net(x.to(gpu)).cpu()
    for k in ('aligned', 'fused', 'trunk'):
        e = ((taps[k].cpu() - taps_ref[k]).abs().max() / taps_ref[k].abs().max()).item()
        assert e < 2e-4, (k, e)
    e = ((out - ref).abs().max() / ref.abs().max()).item()
    assert e < 2e-4, e
    d = abs(EO.psnr(out, gt) - EO.psnr(ref, gt))
    assert d <= 1e-3, d
    # and the clamp-round-uint8 images tensor2img would write differ in at most a handful of +-1 pixels
    a, b = EO.tensor2img_uint8(out), EO.tensor2img_uint8(ref)
    assert (a.int() - b.int()).abs().max().item() <= 1 and (a != b).float().mean().item() < 1e-3


EDVR_L = dict(num_feat=128, num_frame=5, num_reconstruct_block=40, center_frame_idx=None)


def _compare(out, ref, taps, taps_ref, gt, EO):
    for k in ('aligned', 'fused', 'trunk'):
        a, b = taps[k], taps_ref[k].to(taps[k].device)
        e = ((a - b).abs().max() / b.abs().max()).item()
        assert e < 2e-4, (k, e)
    e = ((out - ref).abs().max() / ref.abs().max()).item()
    assert e < 2e-4, e
    d = abs(EO.psnr(out, gt) - EO.psnr(ref, gt))
    assert d <= 1e-3, d
    a, b = EO.tensor2img_uint8(out), EO.tensor2img_uint8(ref)
    assert (a.int() - b.int()).abs().max().item() <= 1 and (a != b).float().mean().item() < 1e-3


def test_edvr_l_one_clip_with_intermediates_matches_the_cpu_oracle(gpu):
    """The headline network on one 180x320 clip of the headline workload: every conv of this run is a full-size launch of the
    kernels bench.py times (F(4x4) Winograd on 128 channels, the fused DCN on the L1 / L2 / L3 levels)."""
    from edvr_amd import EDVR
    from oracle import dcn_oracle as O, edvr_oracle as EO
    from util_edvr import randomize_offsets
    torch.manual_seed(10)
    net = randomize_offsets(EDVR(**EDVR_L)).eval()
    x = torch.rand(1, 5, 3, 180, 320, generator=torch.Generator().manual_seed(0))
    gt = torch.rand(1, 3, 720, 1280, generator=torch.Generator().manual_seed(1))
    taps_ref, taps = {}, {}
    with torch.no_grad():
        ref = EO.edvr_forward(net.state_dict(), x, dcn=O.dcnv2_c, taps=taps_ref)
        net = net.to(gpu)
        net.taps = taps
        out = net(x.to(gpu)).cpu()
    _compare(out, ref, {k: v.cpu() for k, v in taps.items()}, taps_ref, gt, EO)


BIG = {
    # name: (ctor kwargs, clip shape, output scale, convolution of the oracle arm)
    'cfg2_L_T7_180x320': (dict(num_feat=128, num_frame=7, num_reconstruct_block=40, center_frame_idx=None), (7, 3, 180, 320), 4, 'conv2d'),
    'cfg4_L_deblur_720x1280': (dict(EDVR_L, hr_in=True, with_predeblur=True), (5, 3, 720, 1280), 1, 'conv2d'),
    # (MIOpen compiles ~20 kernels for the new shapes of this one on a fresh box: 4 minutes; im2col + GEMM needs no compilation)
    'north_star_L_720x1280_to_4k': (EDVR_L, (5, 3, 720, 1280), 4, 'unfold'),
}


@pytest.mark.parametrize('name', list(BIG))
def test_edvr_l_big_configs_match_stock_rocm_ops_with_intermediates(gpu, name):
    """One clip of BASELINE.json's configs[2], configs[4] and the north_star's 720p -> 4K target at their REAL sizes.  Oracle =
    oracle/edvr_oracle.py (the restatement pinned against the reference's own Python, tests/test_golden.py) executed on the GPU
    in stock fp32 ops: F.conv2d -> MIOpen (or F.unfold + torch.addmm -> rocBLAS, oracle/edvr_oracle.py::conv_unfold), DCNv2 = the
    floor/gather restatement oracle/dcn_oracle.py::dcnv2_torch.  Nothing of
    edvr_amd runs in that arm.  4K exercises what smaller runs do not: planes beyond 2^31 bytes per image (the 1x1 stream
    kernel's channel segments, csrc/conv1x1.hip), the 32-bit buffer-offset eligibility of the F(4x4) and fused-DCN kernels."""
    from edvr_amd import EDVR
    from oracle import dcn_oracle as O, edvr_oracle as EO
    from util_edvr import randomize_offsets
    cfg, shape, scale, conv_impl = BIG[name]
    torch.manual_seed(10)
    net = randomize_offsets(EDVR(**cfg)).eval().to(gpu)
    x = torch.rand(1, *shape, generator=torch.Generator().manual_seed(0)).to(gpu)
    gt = torch.rand(1, 3, scale * shape[2], scale * shape[3], generator=torch.Generator().manual_seed(1)).to(gpu)
    taps_ref, taps = {}, {}
    net.taps = taps
    with torch.no_grad():
        out = net(x)
        torch.cuda.synchronize()
        sd = net.state_dict()
        ref = EO.edvr_forward(sd, x, center=cfg.get('center_frame_idx'), hr_in=cfg.get('hr_in', False),
                              with_predeblur=cfg.get('with_predeblur', False), dcn=O.dcnv2_torch, taps=taps_ref, conv_impl=conv_impl)
    _compare(out, ref, taps, taps_ref, gt, EO)
