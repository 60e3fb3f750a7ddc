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
    def __init__(self, flat_grads, buckets, process_group=None, backend="torch", force_comm=False):
        """flat_grads: 1-D tensor; buckets: {name: (lo, hi)} element ranges into it.
        backend "torch": torch.distributed all_reduce on the group (nccl = RCCL on ROCm; gloo on CPU).
        backend "abi":   the library's own communicator (include/gdl_hip.h gdl_comm_*: RCCL bound by the extension),
                         bootstrapped through the process group -- rank 0's unique id is broadcast as an object -- and
                         issued on a dedicated stream behind the producing stream's event.
        force_comm: run the collectives even in a world of one (tests)."""
        self.flat = flat_grads
        self.buckets = dict(buckets)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.backend = backend
        self.force_comm = bool(force_comm)
        self.comm = None
        if backend == "abi" and (self.world > 1 or self.force_comm):
            import ctypes

            from . import _lib as L

            if not flat_grads.is_cuda or flat_grads.dtype != torch.float32:
                raise ValueError("BucketReducer: the 'abi' backend needs a float32 gradient arena on the GPU")
            rank = dist.get_rank(process_group) if dist.is_initialized() else 0
            ident = [None]
            if rank == 0:
                buf = ctypes.create_string_buffer(128)
                L.call("gdl_comm_unique_id", buf)
                ident[0] = bytes(buf.raw)
            if dist.is_initialized() and self.world > 1:
                dist.broadcast_object_list(ident, src=0, group=process_group)
            h = ctypes.c_void_p()
            with torch.cuda.device(flat_grads.device):
                L.call("gdl_comm_init", ctypes.byref(h), rank, self.world, ctypes.create_string_buffer(ident[0], 128))
                self.cstream = torch.cuda.Stream(device=flat_grads.device)
            self.comm, self._L = h, L
        elif backend not in ("torch", "abi"):
            raise ValueError(f"BucketReducer: unknown backend {backend!r}")
        self.bytes_sent = 0  # bytes handed to all-reduce calls since construction (describe())
        self.calls = 0
        self.pending = {}
        self.enabled = True  # False: launch() / wait_all() keep their bookkeeping but move no data (bench.py times the
        #                      step without its collectives to report how much of them the backward hides)
        spans = sorted(self.buckets.values())
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            if b0 < a1:
                raise ValueError("BucketReducer: buckets overlap")

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def describe(self):
        """Who moves the buckets, and between how many ranks AS THE COMMUNICATOR REPORTS IT: ncclCommCount through gdl_comm_world for
        the 'abi' backend, torch.distributed's world size + backend name otherwise (bench.py's comm.nranks)."""
        if self.comm is not None:
            return {"backend": "abi (gdl_comm_*: RCCL bound by the extension)", "nranks": int(self._L.load().gdl_comm_world(self.comm)),
                    "nranks_source": "ncclCommCount"}
        if dist.is_initialized():
            return {"backend": f"torch.distributed/{dist.get_backend(self.pg)}", "nranks": int(dist.get_world_size(self.pg)),
                    "nranks_source": "dist.get_world_size"}
        return {"backend": "none", "nranks": 1, "nranks_source": "no process group"}

    def launch(self, name):
        """Start the sum-all-reduce of one bucket on the current stream's dependency chain.
        Each bucket may be launched once per step."""
        if name in self.pending:
            raise RuntimeError(f"BucketReducer: bucket {name!r} reduced twice in one step")
        lo, hi = self.buckets[name]
        if (self.world == 1 and not self.force_comm) or not self.enabled or hi <= lo:
            self.pending[name] = None
            return
        self.bytes_sent += (hi - lo) * self.flat.element_size()
        self.calls += 1
        if self.comm is not None:
            cur = torch.cuda.current_stream(self.flat.device)
            self.cstream.wait_stream(cur)
            seg = self.flat[lo:hi]
            self._L.call("gdl_comm_allreduce_bucket", self.comm, seg.data_ptr(), hi - lo, self.cstream.cuda_stream)
            self.pending[name] = self.cstream.record_event()
            return
        self.pending[name] = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def wait_all(self):
        """Block the current stream until every launched bucket is reduced; all buckets must have
        been launched (a missing bucket is the DDP failure mode this class exists to prevent)."""
        missing = set(self.buckets) - set(self.pending)
        if missing:
            raise RuntimeError(f"BucketReducer: buckets never reduced this step: {sorted(missing)}")
        for w in self.pending.values():
            if w is None:
                continue
            if self.comm is not None:
                torch.cuda.current_stream(self.flat.device).wait_event(w)
            else:
                w.wait()
        self.pending = {}

    def close(self):
        if self.comm is not None:
            torch.cuda.synchronize(self.flat.device)
            self._L.call("gdl_comm_destroy", self.comm)
            self.comm = None

    def broadcast_buffers(self, tensors, src=0):
        """Mirror 'replica 0 persists' for BatchNorm running statistics before eval / checkpoint
        (the reference's nn.DataParallel keeps replica 0's buffers, main_dgl.py:244)."""
        if self.world == 1:
            return
        for t in tensors:
            dist.broadcast(t, src=src, group=self.pg)

    def sync_state(self, tensors, src=0):
        """Make every rank's replica state (flat parameter / momentum arenas, BatchNorm buffers) equal to rank
        `src`'s: one broadcast per tensor.  Called once when the trainer is built and after a checkpoint load, so a
        rank-0-only load or a seed that differs between ranks cannot diverge silently."""
        self.broadcast_buffers(tensors, src)

    def time_buckets(self, sync, barrier=None, reps=5):
        """Stand-alone duration of each bucket's all-reduce (ms, best of `reps`; nothing else on the device):
        what the step would pay if none of it were overlapped.  `sync()` must drain the device."""
        import time

        out = {}
        for name, (lo, hi) in self.buckets.items():
            if self.world == 1:
                out[name] = 0.0
                continue
            scratch = torch.zeros_like(self.flat[lo:hi])
            best = float("inf")
            for _ in range(reps + 1):
                (barrier or (lambda: dist.barrier(group=self.pg)))()
                sync()
                t0 = time.perf_counter()
                dist.all_reduce(scratch, op=dist.ReduceOp.SUM, group=self.pg)
                sync()
                best = min(best, (time.perf_counter() - t0) * 1e3)
            out[name] = round(best, 4)
        return out
