"""-m gpu: edvr_amd.graphs.GraphedEDVR - one no-grad forward captured into a hipGraph and replayed (small-clip serving, where the
host's per-launch work is the latency).  Replays must be bit-identical to the eager launches they recorded."""
import logging
import time

import pytest
import torch

from util_edvr import build

pytestmark = pytest.mark.gpu


def _latency(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


@pytest.mark.parametrize('name', ['M_T5', 'L_deblur_hr'])
def test_graph_replay_equals_eager(gpu, name):
    from edvr_amd.graphs import GraphedEDVR
    net, x, _ = build(name)
    net = net.to(gpu)
    xg = x.to(gpu)
    with torch.no_grad():
        want = net(xg).clone()
        net.check_offsets()
    g = GraphedEDVR(net, xg)
    assert torch.equal(g(xg), want)
    x2 = torch.rand(x.shape, generator=torch.Generator().manual_seed(5)).to(gpu)
    got2 = g(x2).clone()
    with torch.no_grad():
        assert torch.equal(got2, net(x2))
    assert torch.equal(g(xg), want)  # and back: nothing of the second clip is left in the static buffers
    with pytest.raises(ValueError):
        g(xg[:, :3])


def test_graph_refuses_stale_weights_and_recaptures(gpu):
    from edvr_amd.graphs import GraphedEDVR
    net, x, _ = build('M_T5')
    net = net.to(gpu)
    xg = x.to(gpu)
    g = GraphedEDVR(net, xg)
    before = g(xg).clone()
    with torch.no_grad():
        net.conv_first.weight.mul_(1.5)  # as an optimizer step / load_state_dict would: the version counter moves
    with pytest.raises(RuntimeError):
        g(xg)
    g.refresh()
    with torch.no_grad():
        want = net(xg)
    assert torch.equal(g(xg), want) and not torch.equal(want, before)


def test_graph_offset_check_on_demand(gpu, caplog):
    from edvr_amd.graphs import GraphedEDVR
    net, x, _ = build('M_T5')
    with torch.no_grad():
        net.pcd_align.cas_dcnpack.conv_offset.bias.fill_(80.0)
    net = net.to(gpu)
    g = GraphedEDVR(net, x.to(gpu))
    with caplog.at_level(logging.WARNING, logger='basicsr'):
        caplog.clear()
        g(x.to(gpu))
        assert not [r for r in caplog.records if 'larger than 50' in r.getMessage()]  # a replay runs no Python
        g.check_offsets()
        assert len([r for r in caplog.records if 'larger than 50' in r.getMessage()]) == x.shape[1]


def test_graph_replay_latency_on_a_small_clip(gpu):
    """BASELINE.json configs[0]: EDVR-M, one 64x64 clip, ~170 short launches.  Measured: 3.01 ms eager vs 3.02 ms replayed - the
    latency of a small clip is the GPU's serial chain of short kernels (>= 15 us each), not the host's ~5 us per ctypes launch; the
    replay removes the host work (it matters when the host is busy), it cannot shorten that chain.  Asserted: identical results
    and no slowdown."""
    from edvr_amd import EDVR
    from edvr_amd.graphs import GraphedEDVR
    from util_edvr import randomize_offsets
    torch.manual_seed(10)
    net = randomize_offsets(EDVR(num_feat=64, num_frame=5, num_reconstruct_block=10, center_frame_idx=2)).eval().to(gpu)
    x = torch.rand(1, 5, 3, 64, 64, generator=torch.Generator().manual_seed(0)).to(gpu)

    def eager():
        with torch.no_grad():
            return net(x)
    t_eager = _latency(eager)
    g = GraphedEDVR(net, x)
    t_graph = _latency(lambda: g(x))
    print(f'EDVR-M 64x64 b1: eager {t_eager * 1e3:.2f} ms, graph replay {t_graph * 1e3:.2f} ms per clip')
    assert torch.equal(g(x), eager())
    assert t_graph < 1.15 * t_eager


def test_graph_keeps_its_packed_weights_through_a_training_step(gpu):
    """The captured launches read the packed conv-weight layouts of capture time.  The training path re-packs every layout after
    an optimizer step, IN PLACE where the cache is the buffer's only owner (ops.prepack_conv_weights): the graph holds and pins
    its layouts (ops.pin_packed_weights), so training steps leave them bit-for-bit as captured and pack into fresh buffers; once
    the graph is gone the in-place rewriting resumes.  (Biases and the DCN weights are read from the live parameters by the
    captured launches, which is why `check_weights=True` - the default - refuses a replay after any parameter change.)"""
    import gc
    from edvr_amd import ops
    from edvr_amd.graphs import GraphedEDVR
    net, x, _ = build('M_T5')
    net = net.to(gpu)
    xg = x.to(gpu)
    g = GraphedEDVR(net, xg)
    before = g(xg).clone()
    assert g._pinned and {b.data_ptr() for b in g._pinned} <= ops._PINNED_PACKED
    snap = [(b, b.clone()) for b in g._pinned]
    w = net.conv_l2_2.weight
    ptr_captured = ops.pack_conv_weight(w).data_ptr()
    net.train()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    for _ in range(2):  # forward (packs the current versions in one launch) + backward + step (new versions)
        opt.zero_grad()
        net(xg).square().mean().backward()
        opt.step()
    net.eval()
    with torch.no_grad():
        after = net(xg).clone()
    assert not torch.equal(after, before)                                  # the eager path runs on the updated weights
    assert all(torch.equal(b, c) for b, c in snap)                         # the captured layouts were not rewritten
    assert ops.pack_conv_weight(w).data_ptr() != ptr_captured              # ... the new versions went to fresh buffers
    with pytest.raises(RuntimeError):
        g(xg)                                                              # parameters changed: refresh() first
    g.refresh()
    assert torch.equal(g(xg), after)
    del g, snap
    gc.collect()
    assert not ops._PINNED_PACKED
    net.train()
    ptr_now = ops.pack_conv_weight(w).data_ptr()
    opt.zero_grad()
    net(xg).square().mean().backward()
    opt.step()
    net(xg)                                                                # training forward: re-packs, in place again
    assert ops.pack_conv_weight(w).data_ptr() == ptr_now
