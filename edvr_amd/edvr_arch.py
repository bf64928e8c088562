"""EDVR on MI355X: PCDAlignment / TSAFusion / PredeblurModule / EDVR with the reference's API.

Drop-in for basicsr/models/archs/edvr_arch.py of xinntao/EDVR (PCDAlignment :9-117, TSAFusion
:120-214, PredeblurModule :217-269, EDVR :272-420): same class names, constructor keywords,
sub-module / parameter names (state_dict keys and shapes, so official checkpoints load with
strict=True), parameter creation order (same seed -> same init), input/output contract.

What differs is how the graph is executed:
  * every convolution is one launch of the fp32 MFMA implicit-GEMM kernel with its LeakyReLU /
    ReLU / residual add / PixelShuffle / channel-concat fused (no torch.cat, no separate
    activation or add kernels);
  * PCD alignment runs ONCE on all b*t frames (the reference loops over t in Python with batch b,
    :396-402); the reference-frame features are read in place through an image-index map instead
    of being cloned t times (:392-401);
  * conv_offset's output is consumed as zero-copy channel slices with the mask sigmoid fused;
    the `offset abs mean > 50` check (arch_util.py:248-253) is evaluated once per forward instead
    of forcing 4*t host synchronisations - and WITHOUT any host synchronisation in no-grad
    (inference) mode: the statistics travel to pinned memory asynchronously and are examined when
    the next forward starts (or on `EDVR.check_offsets()`), so a b = 1 streaming loop keeps
    running ahead of the GPU;
  * TSA temporal attention (t dot-products + sigmoid + broadcast multiply, :171-184) is one
    bandwidth-bound kernel; max+avg pooling and their concat are one kernel.
"""
import torch
from torch import nn

from . import functional as F_
from .arch_util import DCNv2Pack, ResidualBlockNoBN, make_layer, warn_offset_absmean

LRELU = F_.ACT_LRELU


def _conv3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1)


class PCDAlignment(nn.Module):
    """Pyramid, cascading and deformable alignment (3 pyramid levels + a cascade DCN)."""

    def __init__(self, num_feat=64, deformable_groups=8):
        super().__init__()
        self.offset_conv1, self.offset_conv2, self.offset_conv3 = nn.ModuleDict(), nn.ModuleDict(), nn.ModuleDict()
        self.dcn_pack, self.feat_conv = nn.ModuleDict(), nn.ModuleDict()
        for lv in (3, 2, 1):  # L3 = 1/4 size, L2 = 1/2, L1 = full
            key = f'l{lv}'
            self.offset_conv1[key] = _conv3(2 * num_feat, num_feat)
            if lv == 3:
                self.offset_conv2[key] = _conv3(num_feat, num_feat)
            else:
                self.offset_conv2[key] = _conv3(2 * num_feat, num_feat)
                self.offset_conv3[key] = _conv3(num_feat, num_feat)
            self.dcn_pack[key] = DCNv2Pack(num_feat, num_feat, 3, padding=1, deformable_groups=deformable_groups)
            if lv < 3:
                self.feat_conv[key] = _conv3(2 * num_feat, num_feat)
        self.cas_offset_conv1 = _conv3(2 * num_feat, num_feat)
        self.cas_offset_conv2 = _conv3(num_feat, num_feat)
        self.cas_dcnpack = DCNv2Pack(num_feat, num_feat, 3, padding=1, deformable_groups=deformable_groups)
        self.upsample = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    def dcn_modules(self):
        return [self.dcn_pack['l3'], self.dcn_pack['l2'], self.dcn_pack['l1'], self.cas_dcnpack]

    def align(self, nbr, ref, ref_map=None):
        """nbr[l]: (n, C, h_l, w_l).  ref[l]: reference features, image i of nbr pairs with image
        (i // div) * mul + add of ref when ref_map = (div, mul, add), else with image i."""
        up_off = up_feat = feat = None
        for lv in (3, 2, 1):
            key, x = f'l{lv}', nbr[lv - 1]
            off = F_.conv(self.offset_conv1[key], x, x2=ref[lv - 1], x2_map=ref_map, act=LRELU)
            if lv == 3:
                off = F_.conv(self.offset_conv2[key], off, act=LRELU)
            else:
                off = F_.conv(self.offset_conv2[key], off, x2=up_off, act=LRELU)
                off = F_.conv(self.offset_conv3[key], off, act=LRELU)
            feat = self.dcn_pack[key](x, off, act=LRELU if lv == 3 else F_.ACT_NONE)  # LeakyReLU of :103-104 fused
            if lv < 3:
                feat = F_.conv(self.feat_conv[key], feat, x2=up_feat, act=LRELU if lv > 1 else F_.ACT_NONE)
            if lv > 1:
                up_off = F_.upsample2x(off, 2.0)  # offsets double with the resolution (:109)
                up_feat = F_.upsample2x(feat)
        off = F_.conv(self.cas_offset_conv1, feat, x2=ref[0], x2_map=ref_map, act=LRELU)
        off = F_.conv(self.cas_offset_conv2, off, act=LRELU)
        return self.cas_dcnpack(feat, off, act=LRELU)

    def forward(self, nbr_feat_l, ref_feat_l):
        return self.align(nbr_feat_l, ref_feat_l)


class TSAFusion(nn.Module):
    """Temporal + spatial attention fusion."""

    def __init__(self, num_feat=64, num_frame=5, center_frame_idx=2):
        super().__init__()
        self.center_frame_idx = center_frame_idx
        self.temporal_attn1 = _conv3(num_feat, num_feat)
        self.temporal_attn2 = _conv3(num_feat, num_feat)
        self.feat_fusion = nn.Conv2d(num_frame * num_feat, num_feat, 1, 1)
        self.max_pool = nn.MaxPool2d(3, stride=2, padding=1)
        self.avg_pool = nn.AvgPool2d(3, stride=2, padding=1)
        self.spatial_attn1 = nn.Conv2d(num_frame * num_feat, num_feat, 1)
        self.spatial_attn2 = nn.Conv2d(num_feat * 2, num_feat, 1)
        self.spatial_attn3 = _conv3(num_feat, num_feat)
        self.spatial_attn4 = nn.Conv2d(num_feat, num_feat, 1)
        self.spatial_attn5 = _conv3(num_feat, num_feat)
        self.spatial_attn_l1 = nn.Conv2d(num_feat, num_feat, 1)
        self.spatial_attn_l2 = _conv3(num_feat * 2, num_feat)
        self.spatial_attn_l3 = _conv3(num_feat, num_feat)
        self.spatial_attn_add1 = nn.Conv2d(num_feat, num_feat, 1)
        self.spatial_attn_add2 = nn.Conv2d(num_feat, num_feat, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)
        self.upsample = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)

    def forward(self, aligned_feat):
        b, t, c, h, w = aligned_feat.shape
        alias = F_.ops.carry_bound  # (a view of a tensor inherits its magnitude bound: the split-operand convs read it, ops.input_bound)
        emb_ref = F_.conv(self.temporal_attn1, alias(aligned_feat[:, self.center_frame_idx], aligned_feat))  # strided view, no clone
        emb = F_.conv(self.temporal_attn2, alias(aligned_feat.reshape(b * t, c, h, w), aligned_feat)).view(b, t, -1, h, w)
        mod = F_.tsa_temporal(emb, emb_ref, aligned_feat)
        mod = alias(mod.view(b, t * c, h, w), mod)

        feat = F_.conv(self.feat_fusion, mod, act=LRELU)
        attn = F_.conv(self.spatial_attn1, mod, act=LRELU)
        attn = F_.conv(self.spatial_attn2, F_.pool_maxavg(attn), act=LRELU)
        lvl = F_.conv(self.spatial_attn_l1, attn, act=LRELU)
        lvl = F_.conv(self.spatial_attn_l2, F_.pool_maxavg(lvl), act=LRELU)
        lvl = F_.upsample2x(F_.conv(self.spatial_attn_l3, lvl, act=LRELU))
        attn = F_.conv(self.spatial_attn3, attn, act=LRELU, res1=lvl)
        attn = F_.upsample2x(F_.conv(self.spatial_attn4, attn, act=LRELU))
        attn = F_.conv(self.spatial_attn5, attn)
        attn_add = F_.conv(self.spatial_attn_add2, F_.conv(self.spatial_attn_add1, attn, act=LRELU))
        return F_.tsa_combine(feat, attn, attn_add)  # feat * sigmoid(attn) * 2 + attn_add


class PredeblurModule(nn.Module):
    """Pre-deblur pyramid used by the deblurring configurations."""

    def __init__(self, num_in_ch=3, num_feat=64, hr_in=False):
        super().__init__()
        self.hr_in = hr_in
        self.conv_first = _conv3(num_in_ch, num_feat)
        if hr_in:
            self.stride_conv_hr1 = _conv3(num_feat, num_feat, 2)
            self.stride_conv_hr2 = _conv3(num_feat, num_feat, 2)
        self.stride_conv_l2 = _conv3(num_feat, num_feat, 2)
        self.stride_conv_l3 = _conv3(num_feat, num_feat, 2)
        self.resblock_l3 = ResidualBlockNoBN(num_feat=num_feat)
        self.resblock_l2_1 = ResidualBlockNoBN(num_feat=num_feat)
        self.resblock_l2_2 = ResidualBlockNoBN(num_feat=num_feat)
        self.resblock_l1 = nn.ModuleList([ResidualBlockNoBN(num_feat=num_feat) for _ in range(5)])
        self.upsample = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)

    def forward(self, x):
        f1 = F_.conv(self.conv_first, x, act=LRELU)
        if self.hr_in:
            f1 = F_.conv(self.stride_conv_hr2, F_.conv(self.stride_conv_hr1, f1, act=LRELU), act=LRELU)
        f2 = F_.conv(self.stride_conv_l2, f1, act=LRELU)
        f3 = F_.conv(self.stride_conv_l3, f2, act=LRELU)
        f3 = F_.upsample2x(self.resblock_l3(f3))
        rb = self.resblock_l2_1  # resblock(f2) + f3 with the second add fused into conv2's epilogue
        f2 = F_.conv(rb.conv2, F_.conv(rb.conv1, f2, act=F_.ACT_RELU), res1=f2, res2=f3)
        f2 = F_.upsample2x(self.resblock_l2_2(f2))
        f1 = self.resblock_l1[0](f1)
        rb = self.resblock_l1[1]
        f1 = F_.conv(rb.conv2, F_.conv(rb.conv1, f1, act=F_.ACT_RELU), res1=f1, res2=f2)
        for i in range(2, 5):
            f1 = self.resblock_l1[i](f1)
        return f1


class EDVR(nn.Module):
    """EDVR video restoration network (x4 SR, or deblurring with hr_in=True)."""

    def __init__(self, num_in_ch=3, num_out_ch=3, num_feat=64, num_frame=5, deformable_groups=8, num_extract_block=5,
                 num_reconstruct_block=10, center_frame_idx=2, hr_in=False, with_predeblur=False, with_tsa=True):
        super().__init__()
        self.center_frame_idx = num_frame // 2 if center_frame_idx is None else center_frame_idx
        self.hr_in, self.with_predeblur, self.with_tsa = hr_in, with_predeblur, with_tsa
        if with_predeblur:
            self.predeblur = PredeblurModule(num_feat=num_feat, hr_in=hr_in)
            self.conv_1x1 = nn.Conv2d(num_feat, num_feat, 1, 1)
        else:
            self.conv_first = _conv3(num_in_ch, num_feat)
        self.feature_extraction = make_layer(ResidualBlockNoBN, num_extract_block, num_feat=num_feat)
        self.conv_l2_1 = _conv3(num_feat, num_feat, 2)
        self.conv_l2_2 = _conv3(num_feat, num_feat)
        self.conv_l3_1 = _conv3(num_feat, num_feat, 2)
        self.conv_l3_2 = _conv3(num_feat, num_feat)
        self.pcd_align = PCDAlignment(num_feat=num_feat, deformable_groups=deformable_groups)
        if with_tsa:
            self.fusion = TSAFusion(num_feat=num_feat, num_frame=num_frame, center_frame_idx=self.center_frame_idx)
        else:
            self.fusion = nn.Conv2d(num_frame * num_feat, num_feat, 1, 1)
        self.reconstruction = make_layer(ResidualBlockNoBN, num_reconstruct_block, num_feat=num_feat)
        self.upconv1 = _conv3(num_feat, num_feat * 4)
        self.upconv2 = _conv3(num_feat, 64 * 4)
        self.pixel_shuffle = nn.PixelShuffle(2)
        self.conv_hr = _conv3(64, 64)
        self.conv_last = _conv3(64, 3)
        self.lrelu = nn.LeakyReLU(negative_slope=0.1, inplace=True)
        self.taps = None  # set to a dict to collect intermediates (parity tests)
        self._conv_weights = None         # conv weight Parameters (collected at the first training forward)
        self._conv_meta = None
        self._captured_offset_stats = None

    # Offset statistics of earlier no-grad forwards still on their way to the host: (pinned host tensor, copy-done event, per-layer
    # records) per forward.  Kept OUTSIDE the module's __dict__ (a torch.cuda.Event can neither be pickled nor deep-copied:
    # copy.deepcopy(net) for an EMA copy or torch.save(net) right after an eval forward must keep working) in a registry keyed by
    # the module; a copy starts with an empty queue.
    _PENDING = __import__('weakref').WeakKeyDictionary()

    @property
    def _pending_offset_stats(self):
        q = EDVR._PENDING.get(self)
        if q is None:
            q = EDVR._PENDING[self] = []
        return q

    def train(self, mode=True):
        """Switching between train() and eval() first evaluates what the forwards so far have queued (one wait for the last copy):
        the `Offset abs mean ... larger than 50` warning of the final clip of an inference run is then logged at the latest when the
        caller leaves eval mode - or call check_offsets() right after the last clip."""
        self.check_offsets(wait=True)
        return super().train(mode)

    def forward(self, x):
        b, t, c, h, w = x.shape
        if self.hr_in:
            assert h % 16 == 0 and w % 16 == 0, 'The height and width must be multiple of 16.'
        else:
            assert h % 4 == 0 and w % 4 == 0, 'The height and width must be multiple of 4.'
        x = x.contiguous()
        ctr = self.center_frame_idx
        self.check_offsets(wait=False)  # offset statistics of earlier forwards whose copy has landed: no synchronisation
        if x.is_cuda:
            F_.ops.split_guard_check(wait=False)  # overflow flags of earlier forwards (split-operand kernels), likewise
        if torch.is_grad_enabled() and x.is_cuda:
            # training: the optimizer has rewritten every weight - all packed layouts of all conv layers in one launch (ops.py)
            if self._conv_weights is None:
                convs = [m for m in self.modules() if isinstance(m, nn.Conv2d)]
                self._conv_weights = [m.weight for m in convs]
                # (F(4x4) forward layout: stride-1 convs only; flipped layouts: not for the convs that read the input frames)
                first = self.predeblur.conv_first if self.with_predeblur else self.conv_first
                self._conv_meta = [(m.stride[0] == 1, m is not first) for m in convs]
            F_.ops.prepack_conv_weights(self._conv_weights, self._conv_meta)
        frames = x.view(b * t, c, h, w)
        if F_.ops.F4S_INFERENCE or F_.ops.F4S_TRAINING:
            F_.ops.input_bound(frames)  # max |input| (one pass over 3-channel frames): the first link of the chain of magnitude bounds
        if self.with_predeblur:
            f1 = F_.conv(self.conv_1x1, self.predeblur(frames))
            if self.hr_in:
                h, w = h // 4, w // 4
        else:
            f1 = F_.conv(self.conv_first, frames, act=LRELU)
        f1 = self.feature_extraction(f1)
        f2 = F_.conv(self.conv_l2_2, F_.conv(self.conv_l2_1, f1, act=LRELU), act=LRELU)
        f3 = F_.conv(self.conv_l3_2, F_.conv(self.conv_l3_1, f2, act=LRELU), act=LRELU)

        # all b*t frames aligned in one pass; frame i pairs with the centre frame of its clip
        sink = []
        dcns = self.pcd_align.dcn_modules()
        for m in dcns:
            m.stats_sink = sink
        try:
            aligned = self.pcd_align.align([f1, f2, f3], [f1, f2, f3], ref_map=(t, t, ctr))
        finally:
            for m in dcns:
                m.stats_sink = None
        aligned = F_.ops.carry_bound(aligned.view(b, t, -1, h, w), aligned)
        taps = self.taps
        if taps is not None:
            taps['aligned'] = aligned
        feat = self.fusion(aligned) if self.with_tsa else F_.conv(self.fusion, F_.ops.carry_bound(aligned.view(b, -1, h, w), aligned))
        if taps is not None:
            taps['fused'] = feat
        out = self.reconstruction(feat)
        if taps is not None:
            taps['trunk'] = out
        out = F_.conv(self.upconv1, out, act=LRELU, out_mode=F_.OUT_PIXEL_SHUFFLE2)
        out = F_.conv(self.upconv2, out, act=LRELU, out_mode=F_.OUT_PIXEL_SHUFFLE2)
        out = F_.conv(self.conv_hr, out, act=LRELU)
        x_center = x[:, ctr]
        if self.hr_in:
            out = F_.conv(self.conv_last, out, res1=x_center)
        else:
            out = F_.upsample4x_add(F_.conv(self.conv_last, out), x_center)
        self._queue_offset_check(sink, b, t)
        F_.ops.split_guard_submit(x.device)
        return out

    def _queue_offset_check(self, sink, b, t):
        """arch_util.py:248-253, evaluated per (DCN layer, frame) like the reference's per-call check, with ONE device->host
        copy per forward.  No forward waits for the GPU: the copy goes to pinned memory behind an event and is examined by
        `check_offsets` (called when the next forward starts, or by the user).  In training the backward of the same iteration
        picks its dX STRATEGY (never a value) from the statistics that have arrived by then - normally the previous iteration's,
        whose offsets differ by one optimizer step; EDVR_DCN_HINT_WAIT=1 evaluates them right away instead (one host
        synchronisation per iteration, deterministic kernel choice)."""
        if not sink:
            return
        import torch
        sums = torch.stack([e[0] for e in sink]).view(len(sink), 2, b, t).sum(2)  # (layers, {|offset|, roughness}, t), on the device
        recs = [(per_img * b, module) for _, per_img, module in sink]
        if torch.cuda.is_current_stream_capturing():  # hipGraph capture (edvr_amd/graphs.py): the sums are outputs of the graph,
            self._captured_offset_stats = (sums, recs)  # read back on demand by GraphedEDVR.check_offsets
            return
        if torch.is_grad_enabled() and F_.ops.HINT_WAIT:
            self._examine_offsets(sums.cpu(), recs)
            return
        host = torch.empty(sums.shape, dtype=sums.dtype, pin_memory=True)
        host.copy_(sums, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self._pending_offset_stats.append((host, done, recs))
        if len(self._pending_offset_stats) > 64:  # a caller that never lets the GPU catch up: bound the queue
            self.check_offsets(wait=True)

    def check_offsets(self, wait=True):
        """Evaluate the `Offset abs mean is ..., larger than 50` check of the no-grad forwards issued so far.  wait=False only
        looks at forwards whose statistics have already arrived on the host (never blocks).  A no-grad forward never waits for the
        GPU, so its own check is evaluated by the NEXT forward, by train() / eval(), or by this method: call it after the final
        clip of an inference run (the reference logs inside every call, arch_util.py:248-253).  The per-layer kernel hints
        (last_offset_absmean / last_offset_rough) are updated at the same moment, i.e. one forward late in a streaming loop."""
        while self._pending_offset_stats:
            host, done, recs = self._pending_offset_stats[0]
            if wait:
                done.synchronize()
            elif not done.query():
                return
            self._pending_offset_stats.pop(0)
            self._examine_offsets(host, recs)

    @staticmethod
    def _examine_offsets(sums, recs):
        for li, (count, module) in enumerate(recs):
            per_frame = (sums[li, 0] / count).tolist()
            module.last_offset_absmean = sum(per_frame) / len(per_frame)
            rough = float(sums[li, 1].sum())
            module.last_offset_rough = rough / (0.75 * count * len(per_frame)) if rough >= 0 else None
            for v in per_frame:
                warn_offset_absmean(v)
