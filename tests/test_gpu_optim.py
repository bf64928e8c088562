"""-m gpu: FusedAdam (edvr_adam_multi_f32, one launch for all tensors) against torch.optim.Adam, step by step."""
import copy

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _params(gpu, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(128, 128, 3, 3), (128,), (216, 128, 3, 3), (3, 64, 3, 3), (70001,), (1,)]  # 147456 and 248832 elements: > 1 chunk
    return [nn.Parameter(torch.randn(*s, generator=g).to(gpu)) for s in shapes]


@pytest.mark.parametrize('wd', [0.0, 0.01])
def test_fused_adam_matches_torch_adam(gpu, wd):
    from edvr_amd.optim import FusedAdam
    a, b = _params(gpu, 1), _params(gpu, 1)
    groups = lambda ps: [{'params': ps[:3], 'lr': 4e-4}, {'params': ps[3:], 'lr': 1e-4, 'weight_decay': wd}]
    ref = torch.optim.Adam(groups(a), lr=4e-4, betas=(0.9, 0.99), foreach=False)
    ours = FusedAdam(groups(b), lr=4e-4, betas=(0.9, 0.99))
    g = torch.Generator().manual_seed(2)
    for step in range(6):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).to(gpu)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if step == 3:  # a frozen parameter (no gradient) is skipped, and its step count does not advance
            a[1].grad = b[1].grad = None
        if step == 4:
            for o in (ref, ours):
                o.param_groups[0]['lr'] = 2e-4  # what an LR scheduler does
        ref.step()
        ours.step()
        for pa, pb in zip(a, b):
            assert (pa - pb).abs().max().item() <= 1e-6 * max(1.0, pa.abs().max().item()), step
    sa, sb = ref.state_dict()['state'], ours.state_dict()['state']
    for k in sa:
        assert float(sa[k]['step']) == float(sb[k]['step'])
        for name in ('exp_avg', 'exp_avg_sq'):  # a few fp32 ulps of the largest element, accumulated over the steps (fma contraction differs)
            ra, rb = sa[k][name], sb[k][name]
            assert (ra - rb).abs().max().item() <= 2e-6 * max(1.0, ra.abs().max().item()), (k, name)
    # the reference's resume path: a torch.optim.Adam state dict loads into FusedAdam and training continues identically
    c = [nn.Parameter(p.detach().clone()) for p in a]
    resumed = FusedAdam(groups(c), lr=4e-4, betas=(0.9, 0.99))
    resumed.load_state_dict(copy.deepcopy(ref.state_dict()))  # (load_state_dict keeps references to same-device tensors)
    for pa, pc in zip(a, c):
        gr = torch.randn(pa.shape, generator=g).to(gpu)
        pa.grad, pc.grad = gr.clone(), gr.clone()
    ref.step()
    resumed.step()
    for pa, pc in zip(a, c):
        assert (pa - pc).abs().max().item() <= 1e-6 * max(1.0, pa.abs().max().item())


def test_fused_adam_refuses_what_it_does_not_implement(gpu):
    from edvr_amd.optim import FusedAdam
    with pytest.raises(NotImplementedError):
        FusedAdam([nn.Parameter(torch.zeros(2, device=gpu))], amsgrad=True)
    p = nn.Parameter(torch.zeros(2))
    p.grad = torch.ones(2)
    with pytest.raises(NotImplementedError):
        FusedAdam([p]).step()  # CPU parameter


def test_fused_adam_step_is_seen_by_the_next_forward(gpu):
    """FusedAdam writes the parameters through raw pointers; the packed-weight cache of the conv launches is keyed on the
    parameter version, so the step has to bump it.  After a step the module must compute with the NEW weights."""
    import torch.nn.functional as F
    from edvr_amd import functional as F_
    from edvr_amd.optim import FusedAdam
    torch.manual_seed(3)
    conv = nn.Conv2d(64, 64, 3, 1, 1).to(gpu)
    x = torch.randn(2, 64, 20, 24, device=gpu)
    opt = FusedAdam(conv.parameters(), lr=1e-1, betas=(0.9, 0.99))  # large step: the outputs move far beyond rounding
    for _ in range(3):
        v0 = conv.weight._version
        y = F_.conv(conv, x)
        ref = F.conv2d(x, conv.weight.detach(), conv.bias.detach(), padding=1)
        assert (y.detach() - ref).abs().max().item() < 1e-3 * ref.abs().max().item()
        opt.zero_grad(set_to_none=True)
        y.square().sum().backward()
        before = conv.weight.detach().clone()
        opt.step()
        assert conv.weight._version > v0 and not torch.equal(before, conv.weight.detach())


def test_fused_adam_follows_replaced_storage(gpu):
    """The chunk table holds raw device pointers (round-1 advisor finding): replacing a parameter's storage while the Parameter
    object survives (net.to(), .float(), `p.data = ...`, load_state_dict of the optimizer) must not leave stale pointers behind."""
    from edvr_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(257, 33, generator=g).to(gpu)), torch.nn.Parameter(torch.randn(70000, generator=g).to(gpu))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, ropt = FusedAdam(ps, lr=1e-2), torch.optim.Adam(ref, lr=1e-2)

    def step():
        for p, r in zip(ps, ref):
            gr = torch.randn(p.shape, generator=g).to(gpu)
            p.grad, r.grad = gr.clone(), gr.clone()
        opt.step()
        ropt.step()

    step()
    old = [p.data for p in ps]
    old_vals = [o.clone() for o in old]
    for p in ps:
        p.data = p.data.clone()  # same Parameter object, new storage (what net.to() / .float() do)
    step()
    for o, ov in zip(old, old_vals):
        assert torch.equal(o, ov)  # the abandoned storage was not written
    sd = opt.state_dict()
    opt.load_state_dict(sd)  # replaces the moment buffers
    step()
    for p, r in zip(ps, ref):
        assert (p - r).abs().max().item() < 1e-6
