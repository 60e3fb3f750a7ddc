"""Per-bucket gradient all-reduce for the DGL step (one process per GPU, RCCL over xGMI).

Stock DistributedDataParallel is wrong for this step: fc_out receives gradients in
both backward passes of the reference, so DDP either raises or silently skips the fusion head's
reduction.  Here the gradients live in one flat arena with disjoint buckets (DGLTrainer)

    fusion              : the trained fusion-head tensors       (6 150 elements for CREMA-D / concat)
    audio_l4, audio_rest   : audio_net.layer4.* (8.39 M) / the rest of audio_net (2.78 M)
    visual_l4, visual_rest : the same split of visual_net

and each bucket is summed across ranks exactly once per step, as soon as it is final: the
fusion bucket right after the head backward, an encoder's layer4 bucket after the first
phase of its backward (gdl_encoder_backward_phase) -- three quarters of the bytes travel
while layer3 .. stem are still being differentiated -- and the rest after the second.  The
1/world averaging is folded into the clip / SGD kernels (grad_scale).  `fc_auxi` never has a
gradient and is not part of any bucket.  BatchNorm statistics stay per rank, as with the
reference's nn.DataParallel replicas.

The class only needs torch.distributed, so the bucket logic is testable on CPU with gloo.
"""
import torch
import torch.distributed as dist


class BucketReducer:
    def __init__(self, flat_grads, buckets, process_group=None):
        """flat_grads: 1-D tensor; buckets: {name: (lo, hi)} element ranges into it."""
        self.flat = flat_grads
        self.buckets = dict(buckets)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.pending = {}
        spans = sorted(self.buckets.values())
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            if b0 < a1:
                raise ValueError("BucketReducer: buckets overlap")

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def launch(self, name):
        """Start the sum-all-reduce of one bucket on the current stream's dependency chain.
        Each bucket may be launched once per step."""
        if name in self.pending:
            raise RuntimeError(f"BucketReducer: bucket {name!r} reduced twice in one step")
        lo, hi = self.buckets[name]
        if self.world == 1:
            self.pending[name] = None
            return
        self.pending[name] = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def wait_all(self):
        """Block the current stream until every launched bucket is reduced; all buckets must have
        been launched (a missing bucket is the DDP failure mode this class exists to prevent)."""
        missing = set(self.buckets) - set(self.pending)
        if missing:
            raise RuntimeError(f"BucketReducer: buckets never reduced this step: {sorted(missing)}")
        for w in self.pending.values():
            if w is not None:
                w.wait()
        self.pending = {}

    def broadcast_buffers(self, tensors, src=0):
        """Mirror 'replica 0 persists' for BatchNorm running statistics before eval / checkpoint."""
        if self.world == 1:
            return
        for t in tensors:
            dist.broadcast(t, src=src, group=self.pg)
