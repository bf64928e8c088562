"""-m gpu: BASELINE.json configurations at their REAL sizes against the CPU oracle (same clip, same weights).

  configs[0]  EDVR-M x4, 5 frames, 64x64 LR crop, 1 clip            (the reference's own CPU-runnable plumbing case)
  configs[1]  EDVR-M x4, 5 frames, 180x320 LR -> 720x1280, ONE clip of the batch (the oracle needs a few seconds per clip)

Pass criteria (north_star: outputs match within 1e-3 dB PSNR in fp32): max |ours - oracle| / max|oracle| <= 2e-4 and
|PSNR(ours, gt) - PSNR(oracle, gt)| <= 1e-3 dB on a synthetic ground truth (lq seed 0, gt seed 1).
bench.py emits the same comparison for the headline EDVR-L workload in its `parity` object.
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
