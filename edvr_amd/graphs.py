"""Whole-forward hipGraph replay for fixed-shape inference (serving small clips one at a time).

An EDVR forward is ~140-450 kernel launches issued from Python through ctypes (~5 us of host work each).  The launch sequence of one
no-grad forward can be captured once into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed with one host call per clip: the
host thread is then free (a serving process that decodes / encodes frames on the same cores), and a loaded host no longer shows up
as GPU idle time.  What it does NOT do, measured on BASELINE.json's configs[0] (EDVR-M, one 64x64 clip: 3.01 ms eager, 3.02 ms
replayed, tests/test_gpu_graphs.py): shorten the latency of a small clip on an idle host - that latency is the GPU's serial chain
of ~170 short kernels (>= 15 us each), which a graph replays as it is.

    g = GraphedEDVR(net, example_clip)      # warm-up (packs weights, sizes workspaces, settles the kernels' performance hints) + capture
    out = g(clip)                           # copy into the static input, replay, returns the static output tensor (see `clone`)
    g.check_offsets()                       # the `Offset abs mean ... larger than 50` check of the LAST replay (arch_util.py:248-253)

What a replay freezes: shapes, the packed weights (replays keep using the layouts packed at capture time - the graph holds those
buffers and pins them against the training path's in-place re-packing: call `refresh()` after changing parameters - `__call__`
checks the parameters' version counters and refuses to replay stale weights), and the per-layer
performance hints (halo class of the fused DCN kernel: a replay keeps the class of capture time, so replays are bit-identical
to each other; the classes agree with one another to fp32 rounding, not bit for bit - include/edvr_amd.h `halo_hint`).
"""
import torch


class GraphedEDVR:
    def __init__(self, net, example, warmup=2, clone=False, check_weights=True):
        if not example.is_cuda:
            raise NotImplementedError('edvr_amd runs on the GPU only')
        self.net, self.clone, self.check_weights = net, clone, check_weights
        self.static_in = example.detach().clone()
        self._stream = torch.cuda.Stream(device=example.device)
        self._pinned = []
        self._capture(warmup)

    def __del__(self):
        from . import ops
        ops.unpin_packed_weights(getattr(self, '_pinned', []))

    def _versions(self):
        return sum(p._version for p in self.net.parameters())

    def _capture(self, warmup):
        from . import ops
        net, s = self.net, self._stream
        was_training = net.training
        net.eval()
        ops.unpin_packed_weights(self._pinned)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(max(1, warmup)):  # on the capture stream: its workspace (ops.workspace is per stream) exists before the capture
                net(self.static_in)
            net.check_offsets()  # flush: nothing pending may be examined inside the capture
            ops.split_guard_check(wait=True)
            ops.reserve_amax_slots(self.static_in.device)  # a fresh block of bound slots: no allocation / zero-fill inside the graph
            arena = ops._arena(self.static_in.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=s):
                self.static_out = net(self.static_in)
            # the slots the captured launches write: folded into the overflow guard after every replay (nothing inside the graph)
            self._guard_slots = arena.take_unexamined()
        torch.cuda.current_stream().wait_stream(s)
        # the captured launches read the packed weight layouts of THIS moment: hold them and keep the training path's in-place
        # re-packing (ops.prepack_conv_weights) off them - a later optimizer step then packs into fresh buffers
        self._pinned = ops.pin_packed_weights([p for p in net.parameters() if p.dim() == 4])
        self._offset_stats = getattr(net, '_captured_offset_stats', None)
        net._captured_offset_stats = None
        self._ver = self._versions()
        net.train(was_training)

    def refresh(self, warmup=1):
        """Re-capture (after load_state_dict / an optimizer step: the replayed launches read the packed weights of capture time)."""
        self._capture(warmup)

    def __call__(self, x):
        if tuple(x.shape) != tuple(self.static_in.shape) or x.dtype != self.static_in.dtype:
            raise ValueError(f'graph captured for {tuple(self.static_in.shape)} {self.static_in.dtype}, got {tuple(x.shape)} {x.dtype}')
        if self.check_weights and self._versions() != self._ver:
            raise RuntimeError('parameters changed since the capture: call refresh() (replays would use stale packed weights)')
        from . import ops
        ops.split_guard_check(wait=False)
        self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        ops.split_guard_submit(self.static_in.device, list(self._guard_slots))
        return self.static_out.clone() if self.clone else self.static_out

    def check_offsets(self):
        """The reference's per-call `Offset abs mean is ..., larger than 50` warning for the last replay: the per-layer sums are
        outputs of the graph (conv_offset's epilogue); this reads them back (one host synchronisation) and logs like EDVR.forward."""
        if self._offset_stats is None:
            return
        sums, recs = self._offset_stats
        self.net._examine_offsets(sums.cpu(), recs)
