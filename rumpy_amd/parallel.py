"""Data parallelism for the SR hot path: one process per GPU, replicas hold identical fp32 master weights and Adam
state, and the ONLY exchange per step is a mean all-reduce of the flat gradient buffer over RCCL/xGMI
(torch.distributed backend "nccl" on ROCm), issued in buckets on a side HIP stream.

Replaces nn.DataParallel (rumpy/shared_framework/models/base_architecture.py:70-77), which re-broadcasts every
parameter each step from one Python process.  Equal per-rank shards + mean of per-rank mean-L1 gradients equals the
reference's global-batch mean-L1 gradient.  Works with any backend (tests use gloo on CPU tensors).
"""
import os

import torch
import torch.distributed as dist


def bucket_bounds(n, bucket_elems):
    """Split [0, n) into contiguous buckets, LAST parameters first (tail/upsampler gradients are ready first)."""
    bounds = []
    hi = n
    while hi > 0:
        lo = max(0, hi - bucket_elems)
        bounds.append((lo, hi))
        hi = lo
    return bounds


class GradientAverager:
    def __init__(self, net=None, flat_grad=None, bucket_elems=None, group=None):
        if bucket_elems is None:      # collectives per step = host time per step (a torch.distributed call costs the host 30-50 us): few, large ones
            bucket_elems = int(os.environ.get('RUMPY_DP_BUCKET', 1 << 23))
        self.flat_g = flat_grad if flat_grad is not None else net.flat_g
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # test hook for 1-GPU boxes: RUMPY_DP_FORCE=1 sends the gradients of a ONE-rank group through the collectives too, so that the real
        # RCCL communicator, the side stream and the bucket slicing run where only one GPU exists (sum over one rank = identity)
        self.active = self.world_size > 1 or (os.environ.get('RUMPY_DP_FORCE') == '1' and dist.is_available() and dist.is_initialized())
        self.buckets = bucket_bounds(self.flat_g.numel(), bucket_elems)
        self.bucket_elems = bucket_elems
        self.inline = os.environ.get('RUMPY_DP_INLINE', '1') == '1'
        self.early_lo = None                 # start of the part whose all-reduce was launched early by begin()
        # which form average() takes (reported by bench.py as distributed.allreduce_form): 'inline' = ONE blocking-form collective on the
        # caller's stream after the backward pass; 'side' = asynchronous buckets on the side stream after the backward pass; 'early' = the
        # upper half on the side stream under the remaining weight gradients (set by BaseModel.set_multi_gpu when it installs begin())
        self.form = 'inline' if (self.inline and len(self.buckets) == 1) else 'side'
        self.side = torch.cuda.Stream(self.flat_g.device) if self.flat_g.is_cuda else None
        self.pending = []

    def launch_bucket(self, idx):
        """All-reduce bucket idx asynchronously (call as soon as its gradients are final)."""
        if not self.active:
            return
        lo, hi = self.buckets[idx]
        view = self.flat_g[lo:hi]
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.flat_g.device))
            with torch.cuda.stream(self.side):
                self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def begin(self, ptr):
        """Engine hook (SREngine.backward(on_ready=...)): every gradient at device address >= ptr is final on the current stream ->
        start the all-reduce of that upper part of the flat buffer on the side stream; the launches that follow on the main stream (the
        remaining weight gradients) run next to it.  average() then covers the lower part."""
        if not self.active or self.early_lo is not None:
            return
        lo = (int(ptr) - self.flat_g.data_ptr()) // self.flat_g.element_size()
        if lo <= 0 or lo >= self.flat_g.numel():
            return
        self.early_lo = lo
        self._launch_range(lo, self.flat_g.numel())

    def _launch_range(self, lo, hi):
        step = max(1, self.bucket_elems)
        spans = []
        while hi > lo:                                   # LAST parameters first, like bucket_bounds
            spans.append((max(lo, hi - step), hi))
            hi = spans[-1][0]
        for a, b in spans:
            view = self.flat_g[a:b]
            if self.side is not None:
                self.side.wait_stream(torch.cuda.current_stream(self.flat_g.device))
                with torch.cuda.stream(self.side):
                    self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.side is not None:
            torch.cuda.current_stream(self.flat_g.device).wait_stream(self.side)

    def average_sum(self):
        """SUM over ranks of every bucket, left in flat_g -> returns the factor that turns the sum into the mean (1 / world).  For the fused
        optimizer only, which takes the factor as `grad_mult` (rumpy_adam_step multiplies every gradient by it in the same pass: no extra
        kernel).  Anything that reads flat_g / p.grad afterwards sees world-times-larger values: everybody else calls average()."""
        if not self.active:
            return 1.0
        if self.early_lo is not None:        # the upper part is already in flight (begin()): only the rest is launched here
            self._launch_range(0, self.early_lo)
            self.early_lo = None
        elif self.inline and len(self.buckets) == 1:
            # nothing left to run beside it (the whole backward pass is behind us, the optimizer waits for the result): ONE blocking-form
            # collective issued from the CURRENT stream - no side stream, no work handle; c10d's process group enqueues a sync op on the
            # caller's stream (torch >= 2.8), which removes the event hops main -> side -> communicator stream -> main around it
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.group)
        else:
            for i in range(len(self.buckets)):
                self.launch_bucket(i)
        self.finish()
        return 1.0 / self.world_size

    def average(self):
        """the mean over ranks, in place: flat_g (and every p.grad viewing it) holds the averaged gradient afterwards"""
        mult = self.average_sum()
        if mult != 1.0:
            self.flat_g.mul_(mult)


class ParameterGradientAverager:
    """Mean over the ranks of the ``.grad`` of a list of parameters that do NOT live in the generator's flat gradient buffer - the
    trainable encoder of the blind pipeline's joint losses (1.3 M elements: conv / BatchNorm / mlp-head gradients, some of them views of
    the encoder's own flat buffer, the head's plain autograd tensors).  One coalesced all-reduce per step: the gradients are gathered
    into one staging buffer (``torch._foreach_copy_``: a handful of launches), summed over the ranks, and written back scaled by 1 / world."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world_size > 1 or (os.environ.get('RUMPY_DP_FORCE') == '1' and dist.is_available() and dist.is_initialized())
        self._stage = self._views = None

    def average(self):
        if not self.active:
            return
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not self.params:
            return
        dev = self.params[0].device
        n = len(self.params)
        if self._stage is None or self._stage.device != dev:
            sizes = [p.numel() for p in self.params]
            self._stage = torch.empty(sum(sizes) + n, dtype=torch.float32, device=dev)     # + one "has a gradient on this rank" flag per parameter
            self._views = [v.view(p.shape) for v, p in zip(self._stage[:sum(sizes)].split(sizes), self.params)]
            self._have = self._stage[sum(sizes):]
        if len(live) != n:
            # a parameter without a gradient on THIS rank may have one on another: the collective's size must not depend on the rank - it
            # enters the sum as zero
            self._stage.zero_()
        flags = torch.zeros(n, dtype=torch.float32)
        flags[live] = 1.0
        self._have.copy_(flags, non_blocking=True)
        grads = [self.params[i].grad for i in live]
        views = [self._views[i] for i in live]
        if grads:
            torch._foreach_copy_(views, grads)
        dist.all_reduce(self._stage, op=dist.ReduceOp.SUM, group=self.group)
        if self.world_size > 1:
            self._stage[:self._stage.numel() - n].mul_(1.0 / self.world_size)
        if grads:
            torch._foreach_copy_(grads, views)
        if len(live) != n:
            # ... and where another rank had one, the mean becomes this rank's gradient too: every replica takes the same optimizer step
            # (a parameter stepped on some ranks only would let the replicas drift apart)
            have = self._have.tolist()
            for i in range(n):
                if self.params[i].grad is None and have[i] > 0:
                    self.params[i].grad = self._views[i].clone()


def broadcast_parameters(net, src=0, group=None):
    """Make every replica start from rank src's weights (one flat broadcast)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get('RUMPY_DP_FORCE') == '1'):
        hip = getattr(net, 'hip_generator', net)        # a pipeline trains its generator; its frozen encoder is replicated as well
        dist.broadcast(hip.flat_p, src=src, group=group)
        hip._packed_version = None
        if hip is net and hasattr(net, 'state_dict'):
            # whatever a flat-protocol module holds beside its trainable flat buffer (MoCo: the key encoder, the queue and its pointer, the
            # BatchNorm statistics of both encoders) - tensors inside flat_p's storage have just travelled
            own = hip.flat_p.untyped_storage().data_ptr()
            rest = [t for t in net.state_dict().values() if t.untyped_storage().data_ptr() != own]
            for t in rest:
                dist.broadcast(t, src=src, group=group)
            if rest and hasattr(net, 'mark_weights_updated'):
                net.mark_weights_updated()
        enc = getattr(net, 'E', None) if hip is not net else None
        if enc is not None:
            for t in enc.state_dict().values():
                dist.broadcast(t, src=src, group=group)
            for mod in enc.modules():           # the encoder itself, or the query / key encoders of a MoCo module (joint losses)
                if hasattr(mod, 'weights_rewritten'):
                    mod.weights_rewritten()
