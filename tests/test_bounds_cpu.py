"""CPU: the host logic behind the magnitude bounds of the split-operand kernels (edvr_amd/ops.py) - no kernel runs here."""
import torch

from edvr_amd import ops


def test_bounds_work_on_inference_tensors():
    """torch.inference_mode() tensors have no version counter (reading `_version` raises): a bound still attaches and is found."""
    with torch.inference_mode():
        t = torch.zeros(4)
        b = torch.ones(1)
        ops.set_bound(t, b)
        assert ops.get_bound(t) is b
    u = torch.zeros(4)
    ops.set_bound(u, b)
    assert ops.get_bound(u) is b
    u.add_(1.0)  # an in-place torch op moves the version counter: the bound is void
    assert ops.get_bound(u) is None


def test_void_bound_and_carry_bound():
    a, b = torch.zeros(3), torch.zeros(3)
    ops.set_bound(a, torch.tensor([2.0]))
    ops.set_bound(b, torch.tensor([3.0]), 1)
    y = ops.carry_bound(torch.zeros(3), a, b, scale=2.0)
    assert float(ops.get_bound(y)) == 10.0 and ops._bound_depth(y) == 1
    ops.void_bound(a, None, y)
    assert ops.get_bound(a) is None and ops.get_bound(y) is None and ops.get_bound(b) is not None
    assert ops.get_bound(ops.carry_bound(torch.zeros(3), a, b)) is None  # a source without a bound: nothing is handed on


def test_invalidate_packed_weights_drops_the_weight_norms_too():
    """linear_bound() derives x_amax from cached weight norms keyed like the packed weights; a `.data` write needs both
    caches dropped, or the next split conv scales by the OLD norm (too small a bound overflows f16)."""
    w = torch.nn.Parameter(torch.ones(2, 2, 3, 3))
    assert float(ops.weight_l1max(w)) == 18.0
    w.data.mul_(10.0)  # behind the version counter's back
    assert float(ops.weight_l1max(w)) == 18.0  # stale by construction ...
    ops.invalidate_packed_weights()
    assert float(ops.weight_l1max(w)) == 180.0  # ... until the documented call


def test_arena_hands_out_distinct_zeroed_slots_and_tracks_what_the_guard_has_not_seen():
    a = ops._AmaxArena()
    dev = torch.device('cpu')
    s = [a.slot(dev) for _ in range(5)]
    assert len({t.data_ptr() for t in s}) == 5 and all(float(t) == 0.0 for t in s)
    part = a.take_unexamined()
    assert len(part) == 1 and part[0].numel() == 5 and a.take_unexamined() == []
    a.slot(dev)
    a.fresh(dev)  # (before a hipGraph capture) the unexamined tail of the old block is kept for the guard
    a.slot(dev)
    parts = a.take_unexamined()
    assert [p.numel() for p in parts] == [1, 1]
    # a block rolls over without losing slots
    for _ in range(ops._AmaxArena.BLOCK + 3):
        a.slot(dev)
    assert sum(p.numel() for p in a.take_unexamined()) == ops._AmaxArena.BLOCK + 3


def test_guard_flag_semantics(monkeypatch):
    """split_guard_check raises on a non-finite flag, or (EDVR_SPLIT_GUARD=fallback) warns once and switches the split kernels off."""
    import pytest

    class Done:
        def query(self):
            return True

        def synchronize(self):
            pass
    monkeypatch.setattr(ops, '_GUARD_PENDING', [(torch.tensor([3.5]), Done())])
    ops.split_guard_check()
    assert ops._GUARD_PENDING == []
    monkeypatch.setattr(ops, '_GUARD_PENDING', [(torch.tensor([float('nan')]), Done())])
    with pytest.raises(ops.SplitOperandOverflow):
        ops.split_guard_check()
    monkeypatch.setattr(ops, 'SPLIT_GUARD', 'fallback')
    monkeypatch.setattr(ops, '_GUARD_PENDING', [(torch.tensor([float('inf')]), Done())])
    prev = (ops.F4S_INFERENCE, ops.F4S_TRAINING)
    try:
        with pytest.warns(UserWarning, match='fp32 kernels'):
            ops.split_guard_check()
        assert (ops.F4S_INFERENCE, ops.F4S_TRAINING) == (False, False)
    finally:
        ops.set_f4s(*prev)
