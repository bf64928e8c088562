"""Optimizer / scheduler / checkpoint glue of the EDVR training loop, host side in Python like the
reference, the optimizer step as ONE HIP launch (csrc/optim.hip).

Mirrors xinntao/EDVR (BasicSR v1.2.0):
  EDVRModel.setup_optimizers   basicsr/models/edvr_model.py:21-53     -> make_optimizer (dcn_lr_mul parameter groups)
  torch.optim.Adam(...)        edvr_model.py:47-49, sr_model.py:112   -> FusedAdam (same arguments, param_groups, state_dict format)
  CosineAnnealingRestartLR     basicsr/models/lr_scheduler.py:70-118  -> CosineAnnealingRestartLR
  MultiStepRestartLR           lr_scheduler.py:6-48                   -> MultiStepRestartLR
  EDVRModel.optimize_parameters TSA warm-up  edvr_model.py:55-69      -> tsa_freeze_schedule
  save_network / load_network / save_training_state / resume_training  base_model.py:171-201,223-263,265-304
"""
import math
import os
from collections import Counter

import numpy as np
import torch
from torch.optim.lr_scheduler import _LRScheduler

from . import _lib

ADAM_CHUNK = 65536  # elements per table record (include/edvr_amd.h)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) on edvr_adam_multi_f32: one launch for all tensors of all groups.
    `param_groups` and `state_dict()` have torch.optim.Adam's layout ('step', 'exp_avg', 'exp_avg_sq' per parameter), so LR
    schedulers and the reference's `.state` resume files work unchanged.  fp32 CUDA parameters only; no amsgrad / maximize."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError('FusedAdam: amsgrad is not supported (the reference never enables it)')
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError('FusedAdam: invalid hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        b = {tuple(g['betas']) for g in self.param_groups}
        e = {g['eps'] for g in self.param_groups}
        if len(b) != 1 or len(e) != 1:
            raise NotImplementedError('FusedAdam: one (betas, eps) for all groups (only lr / weight_decay differ per group)')
        self._rec = np.dtype([('p', '<u8'), ('g', '<u8'), ('m', '<u8'), ('v', '<u8'), ('n', '<i4'), ('lr', '<f4'), ('wd', '<f4'),
                              ('c1', '<f4'), ('c2', '<f4'), ('pad', '<f4', (3,))])
        assert self._rec.itemsize == _lib.lib().edvr_adam_chunk_bytes()
        self._static = None  # (key, record array with the per-parameter constant fields, chunk -> parameter index)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._static = None  # the moment buffers were replaced
        for st in self.state.values():  # a resume file loaded with map_location=device puts 'step' on the GPU: float(step) in
            if torch.is_tensor(st.get('step')) and st['step'].is_cuda:  # step() would then cost one host sync per parameter
                st['step'] = st['step'].cpu()

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = torch.tensor(0.0, dtype=torch.float32)  # torch.optim.Adam keeps a (CPU) tensor step
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        items = []  # (param, group) of every parameter that has a gradient this step (frozen ones are skipped like torch does)
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                    raise NotImplementedError('FusedAdam: dense fp32 CUDA parameters only')
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise NotImplementedError('FusedAdam: contiguous parameters and gradients only')
                items.append((p, group))
        if not items:
            return loss
        dev = items[0][0].device
        # The table holds RAW device pointers.  The key therefore carries them too: net.to() / .float() / `p.data = ...` /
        # load_state_dict replace a parameter's (or a moment's) storage while the Parameter object - and id(p) - survive, and a
        # table keyed on ids alone would keep writing through the stale pointers.
        for p, _ in items:
            self._init_state(p)
        key = tuple((id(p), p.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr()) for p, _ in items)
        if self._static is None or self._static[0] != key:
            recs, owner = [], []
            for idx, (p, _) in enumerate(items):
                st = self._init_state(p)
                n, base_p, base_m, base_v = p.numel(), p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
                for off in range(0, n, ADAM_CHUNK):
                    recs.append((base_p + 4 * off, 0, base_m + 4 * off, base_v + 4 * off, min(ADAM_CHUNK, n - off), 0, 0, 0, 0, (0, 0, 0)))
                    owner.append((idx, off))
            tab = np.array(recs, dtype=self._rec)
            self._static = (key, tab, np.array([o[0] for o in owner]), np.array([o[1] for o in owner], dtype=np.uint64))
        _, tab, owner, offs = self._static
        beta1, beta2 = self.param_groups[0]['betas']
        gptr = np.empty(len(items), dtype=np.uint64)
        lr = np.empty(len(items), dtype=np.float32)
        wd = np.empty(len(items), dtype=np.float32)
        c1 = np.empty(len(items), dtype=np.float32)
        c2 = np.empty(len(items), dtype=np.float32)
        for i, (p, group) in enumerate(items):
            st = self.state[p]
            st['step'] += 1
            t = float(st['step'])
            gptr[i] = p.grad.data_ptr()
            lr[i], wd[i] = group['lr'], group['weight_decay']
            c1[i] = 1.0 / (1.0 - beta1 ** t)
            c2[i] = 1.0 / math.sqrt(1.0 - beta2 ** t)
        tab['g'] = gptr[owner] + 4 * offs
        tab['lr'], tab['wd'], tab['c1'], tab['c2'] = lr[owner], wd[owner], c1[owner], c2[owner]
        table = torch.from_numpy(tab.view(np.uint8).reshape(-1)).to(dev, non_blocking=False)
        _lib.check(_lib.lib().edvr_adam_multi_f32(table.data_ptr(), len(tab), float(beta1), float(beta2), float(self.param_groups[0]['eps']),
                                                  torch.cuda.current_stream(dev).cuda_stream), 'edvr_adam_multi_f32')
        self._keep = table  # the launch is asynchronous: keep the table alive until the next step replaces it
        # The kernel wrote the parameters through raw pointers: tell autograd, like an in-place torch op would.  Everything keyed
        # on a parameter's version - the packed-weight cache of ops.pack_conv_weight above all - has to see the update.
        torch.autograd.graph.increment_version([p for p, _ in items])
        return loss


def make_optimizer(net, lr=4e-4, dcn_lr_mul=1, betas=(0.9, 0.99), fused=True, **kw):
    """EDVRModel.setup_optimizers (edvr_model.py:21-53): parameters whose name contains 'dcn' get lr * dcn_lr_mul
    (normal parameters first, as in the reference)."""
    cls = FusedAdam if fused else torch.optim.Adam
    if dcn_lr_mul == 1:
        return cls(net.parameters(), lr=lr, betas=betas, **kw)
    normal = [p for n, p in net.named_parameters() if 'dcn' not in n]
    dcn = [p for n, p in net.named_parameters() if 'dcn' in n]
    return cls([{'params': normal, 'lr': lr}, {'params': dcn, 'lr': lr * dcn_lr_mul}], lr=lr, betas=betas, **kw)


class MultiStepRestartLR(_LRScheduler):
    """lr_scheduler.py:6-48."""

    def __init__(self, optimizer, milestones, gamma=0.1, restarts=(0,), restart_weights=(1,), last_epoch=-1):
        self.milestones = Counter(milestones)
        self.gamma = gamma
        self.restarts = restarts
        self.restart_weights = restart_weights
        assert len(self.restarts) == len(self.restart_weights), 'restarts and their weights do not match.'
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        if self.last_epoch in self.restarts:
            weight = self.restart_weights[self.restarts.index(self.last_epoch)]
            return [group['initial_lr'] * weight for group in self.optimizer.param_groups]
        if self.last_epoch not in self.milestones:
            return [group['lr'] for group in self.optimizer.param_groups]
        return [group['lr'] * self.gamma ** self.milestones[self.last_epoch] for group in self.optimizer.param_groups]


def get_position_from_periods(iteration, cumulative_period):
    """lr_scheduler.py:51-67: index of the cycle `iteration` falls into."""
    for i, period in enumerate(cumulative_period):
        if iteration <= period:
            return i


class CosineAnnealingRestartLR(_LRScheduler):
    """lr_scheduler.py:70-118: cosine annealing with restarts (periods, restart_weights, eta_min)."""

    def __init__(self, optimizer, periods, restart_weights=(1,), eta_min=0, last_epoch=-1):
        self.periods = periods
        self.restart_weights = restart_weights
        self.eta_min = eta_min
        assert len(self.periods) == len(self.restart_weights), 'periods and restart_weights should have the same length.'
        self.cumulative_period = [sum(self.periods[0:i + 1]) for i in range(0, len(self.periods))]
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        idx = get_position_from_periods(self.last_epoch, self.cumulative_period)
        current_weight = self.restart_weights[idx]
        nearest_restart = 0 if idx == 0 else self.cumulative_period[idx - 1]
        current_period = self.periods[idx]
        return [self.eta_min + current_weight * 0.5 * (base_lr - self.eta_min) *
                (1 + math.cos(math.pi * ((self.last_epoch - nearest_restart) / current_period))) for base_lr in self.base_lrs]


def tsa_freeze_schedule(net, current_iter, tsa_iter):
    """EDVRModel.optimize_parameters (edvr_model.py:55-69): at iteration 1 freeze everything whose name does not contain
    'fusion' (only the TSA module trains), at iteration `tsa_iter` unfreeze everything.  Returns True when the set of trainable
    parameters changed.

    DDP: wrap the net BEFORE iteration 1 (all parameters trainable, as base_model.py:63-69 does at construction) with
    find_unused_parameters=True; the frozen ones are then "unused" and skipped by the reducer.  At `tsa_iter` the reference
    sets `net_g.find_unused_parameters = False` on the live wrapper (edvr_model.py:66-68).  That relied on the reducer of its
    PyTorch generation returning early for an empty output list; in PyTorch 2.x the C++ reducer keeps the constructor's flag,
    receives no outputs to search, declares EVERY parameter unused and raises "Expected to mark a variable ready only once" at
    the first gradient hook (reproduced with two nn.Linear layers over gloo).  So the wrapper is left untouched here: keep it
    (costs one autograd-graph walk per iteration) or re-wrap with `edvr_amd.dist.rewrap_ddp(model)` when this returns True."""
    if not tsa_iter:
        return False
    bare = net.module if hasattr(net, 'module') else net
    if current_iter == 1:
        for name, p in bare.named_parameters():
            if 'fusion' not in name:
                p.requires_grad = False
        return True
    if current_iter == tsa_iter:
        for p in bare.parameters():
            p.requires_grad = True
        return True
    return False


def save_network(net, path, param_key='params'):
    """base_model.py:171-201: {'params': state_dict} with 'module.' prefixes removed and tensors on the CPU."""
    bare = net.module if hasattr(net, 'module') else net
    sd = {(k[7:] if k.startswith('module.') else k): v.cpu() for k, v in bare.state_dict().items()}
    torch.save({param_key: sd}, path)


def load_network(net, path, strict=True, param_key='params'):
    """base_model.py:223-263: loads official EDVR checkpoints ({'params': ...} or a bare state_dict), strips 'module.'."""
    bare = net.module if hasattr(net, 'module') else net
    load_net = torch.load(path, map_location='cpu')
    if isinstance(load_net, dict) and param_key in load_net and isinstance(load_net[param_key], dict):
        load_net = load_net[param_key]
    load_net = {(k[7:] if k.startswith('module.') else k): v for k, v in load_net.items()}
    bare.load_state_dict(load_net, strict=strict)


def save_training_state(path, epoch, current_iter, optimizers, schedulers):
    """base_model.py:265-286 (`<iter>.state`)."""
    torch.save({'epoch': epoch, 'iter': current_iter, 'optimizers': [o.state_dict() for o in optimizers],
                'schedulers': [s.state_dict() for s in schedulers]}, path)


def resume_training(resume_state, optimizers, schedulers):
    """base_model.py:288-304."""
    if isinstance(resume_state, (str, os.PathLike)):
        resume_state = torch.load(resume_state, map_location='cpu')
    assert len(resume_state['optimizers']) == len(optimizers), 'Wrong lengths of optimizers'
    assert len(resume_state['schedulers']) == len(schedulers), 'Wrong lengths of schedulers'
    for o, s in zip(optimizers, resume_state['optimizers']):
        o.load_state_dict(s)
    for o, s in zip(schedulers, resume_state['schedulers']):
        o.load_state_dict(s)
    return resume_state['epoch'], resume_state['iter']
