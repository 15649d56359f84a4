"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce overlapped with backward.

The reference gets this implicitly from `strategy: ddp` (configs/autoencoder/pose/autoencoder_kl_16x16x16.yaml:137,
train.py:162; torch DDP over NCCL inside loss.backward()).  Here the optimizer's gradients live in one contiguous
arena (optim.FusedAdam), cut into buckets at parameter boundaries.  A post-accumulate hook per parameter counts a
bucket down; when its last gradient lands, the bucket (a plain slice of the arena: no packing copy) is all-reduced
asynchronously -- RCCL runs it on its own stream, ordered after the producing kernels -- while backward continues
towards the encoder.  xGMI is point-to-point, so buckets are sized large (default 32 MB: ~9 collectives for the
284 MB generator payload) to stay bandwidth- rather than latency-bound.  finish() waits and averages.

Works with any backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU tensors for the tests.
Not synchronised across ranks, as in the reference: BatchNorm statistics, _rescale min/max, RNG streams.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, optimizer, process_group=None, bucket_mb=32.0, prescaled=False, comm_dtype=None, f32_accumulate=False, first_bucket_mb=1.0):
        """prescaled=True: the caller back-propagates `loss * inv_world` (trainer.Trainer does), so the summed buckets
        already ARE the mean and finish() needs no averaging pass over the arena (284 MB for optimizer 0).
        comm_dtype=torch.bfloat16: a bucket travels as bf16 -- cast into a staging buffer when its last gradient lands, all-reduced
        there, written back to the f32 arena in finish() -- half the bytes per xGMI link (142 MB instead of 284 MB per generator
        step).  For the bf16 step, whose backward is a third as long as the f32 one: with the default f32 buckets the
        exchange is 3-4 % of that step unless fully hidden.  The mean of world <= 8 bf16-rounded, pre-scaled gradients carries a
        relative error of about 2^-9 per element, the size of the rounding the bf16 activations already put into them; master
        weights, Adam moments and the clip norm stay f32.
        f32_accumulate=True (with comm_dtype bf16; SURVEY.md 5 "reduce-scatter with f32 accumulation"): instead of one bf16 all-reduce
        (whose partial sums are rounded to bf16 at every hop of the ring), a bucket is exchanged in two halves of the same traffic --
        all-to-all of its world-many shards (rank r receives shard r of every rank, as bf16), the world shards summed in f32 on the rank
        that owns them, ONE rounding of the sum to bf16, all-gather of the reduced shards.  The second half of a bucket is issued when the
        next bucket starts (or in finish()), so both halves overlap with the rest of the backward pass.  Backends without all-to-all
        (gloo, in the CPU tests) take the same arithmetic through all-gathers.
        first_bucket_mb: size of the bucket at the FRONT of the arena -- the first parameters of the model (encoder.conv_in, down.0 ...), whose
        gradients the backward produces LAST.  Its collective cannot hide behind anything, so it is kept small, as torch DDP keeps its first
        bucket at 1 MB: measured on the f32 step (bench.py's RCCL world-size-1 side run, `dp_bucket_timeline`) seven of eight 32 MB buckets are
        ready 88-97 ms into a 152 ms backward and the front one -- 35 MB before this split -- at 152 ms, i.e. its whole all-reduce was exposed."""
        self.group = process_group
        self.comm_dtype = comm_dtype if comm_dtype not in (None, torch.float32) else None
        self.f32_accumulate = bool(f32_accumulate) and self.comm_dtype is not None
        self._staging = {}
        self._stage2 = []        # f32_accumulate: (bucket, work of its all-to-all) waiting for the local sum + all-gather
        self.world = dist.get_world_size(process_group)
        self.inv_world = 1.0 / self.world
        self.prescaled = bool(prescaled)
        self.optimizer = optimizer
        if hasattr(optimizer, "param_slices"):
            slices = optimizer.param_slices()
            self.arena = optimizer.flat_grad
            self.own_arena = False
        else:  # generic torch optimizer: build the gradient arena here
            params = [p for g in optimizer.param_groups for p in g["params"]]
            slices, total = [], 0
            for p in params:
                slices.append((p, total, p.numel()))
                total += (p.numel() + 63) // 64 * 64
            self.arena = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
            self.own_arena = True
        self.slices = slices
        # buckets: runs of consecutive parameters, ~bucket_mb each
        limit = int(bucket_mb * 1024 * 1024 / 4)
        first_limit = min(limit, max(1, int(first_bucket_mb * 1024 * 1024 / 4)))
        self.buckets = []       # (start, end) in arena elements
        self.param_bucket = {}
        start, count, members = None, 0, []
        for i, (p, off, n) in enumerate(slices):
            if start is None:
                start = off
            members.append(i)
            end = slices[i + 1][1] if i + 1 < len(slices) else self.arena.numel()
            if end - start >= (limit if self.buckets else first_limit) or i + 1 == len(slices):
                b = len(self.buckets)
                self.buckets.append((start, end))
                for m in members:
                    self.param_bucket[m] = b
                start, members = None, []
        self.bucket_size = [0] * len(self.buckets)
        self.bucket_members = [[] for _ in self.buckets]      # parameter indices (arena order) per bucket
        for i in range(len(slices)):
            self.bucket_size[self.param_bucket[i]] += 1
            self.bucket_members[self.param_bucket[i]].append(i)
        self._pending = None
        self._works = []
        self._launched = set()
        self._touched_buckets = set()
        self.launch_order = []   # bucket ids in the order their collectives were issued (tests look at this)
        # record_timeline = True (a diagnostic, off by default): one event on the compute stream when the backward starts, one per bucket at the
        # moment its collective is issued (= its last gradient's kernel is queued), one when finish() returns -- `timeline()` turns them into
        # "bucket b (n MB) ready k ms into a backward of K ms", the schedule an N-rank run overlaps its collectives with
        self.record_timeline = False
        self._tl = None
        for i, (p, _, _) in enumerate(slices):
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i))

    # ---- per-backward state ---------------------------------------------------------------------------------------
    def prepare_for_backward(self):
        if self.own_arena:
            self.arena.zero_()
            for p, off, n in self.slices:
                view = self.arena[off:off + n].view(p.shape)
                if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                    p.grad = view
        # FusedAdam's arena: `.grad` is whatever the trainer left (None after zero_grad(set_to_none=True)); a bucket's
        # gradients are gathered into the arena right before its collective (`_launch`)
        self._pending = list(self.bucket_size)
        self._works = []
        self._stage2 = []
        self._launched = set()
        self._touched_buckets = set()
        self.launch_order = []
        if self.record_timeline and self.arena.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._tl = {"start": ev, "buckets": [], "end": None}

    def timeline(self):
        """After finish() of a backward recorded with `record_timeline`: {"backward_ms", "buckets": [{"bucket", "mbytes", "ready_ms"}]} (one
        device synchronisation).  ready_ms counts from prepare_for_backward() on the compute stream."""
        tl = self._tl
        if not tl or tl["end"] is None:
            return None
        tl["end"].synchronize()
        return {"backward_ms": tl["start"].elapsed_time(tl["end"]),
                "buckets": [{"bucket": b, "mbytes": (self.buckets[b][1] - self.buckets[b][0]) * 4 / 1e6, "ready_ms": tl["start"].elapsed_time(ev)}
                            for b, ev in tl["buckets"]]}

    def _on_grad(self, i):
        if self._pending is None:
            return  # backward outside a prepare/finish window (e.g. torch.autograd.grad probes)
        b = self.param_bucket[i]
        self._touched_buckets.add(b)
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        s, e = self.buckets[b]
        if not self.own_arena:
            self.optimizer.gather_grads(self.bucket_members[b])
        self._launched.add(b)
        self.launch_order.append(b)
        if self._tl is not None and self.record_timeline:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._tl["buckets"].append((b, ev))
        if self.comm_dtype is None:
            self._works.append(dist.all_reduce(self.arena[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            if self.f32_accumulate:
                self._second_halves()                   # the buckets issued before this one: sum + all-gather, under the backward still running
                self._first_half(b)
                return
            buf = self._staging.get(b)
            if buf is None:
                buf = self._staging[b] = torch.empty(e - s, dtype=self.comm_dtype, device=self.arena.device)
            buf.copy_(self.arena[s:e])                  # f32 -> bf16, one pass over the bucket
            self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ---- bf16 on the wire, f32 sums (f32_accumulate) ----------------------------------------------------------------------
    def _shards(self, b):
        s, e = self.buckets[b]
        shard = (e - s + self.world - 1) // self.world
        return s, e, shard

    def _first_half(self, b):
        s, e, shard = self._shards(b)
        st = self._staging.get(("a2a", b))
        if st is None:
            dev = self.arena.device
            st = self._staging[("a2a", b)] = {"send": torch.zeros(self.world * shard, dtype=self.comm_dtype, device=dev),
                                              "recv": torch.empty(self.world * shard, dtype=self.comm_dtype, device=dev),
                                              "red": torch.empty(shard, dtype=self.comm_dtype, device=dev),
                                              "out": torch.empty(self.world * shard, dtype=self.comm_dtype, device=dev)}
        st["send"][:e - s].copy_(self.arena[s:e])       # f32 -> bf16 (the padding past the bucket stays 0)
        if dist.get_backend(self.group) == "gloo":      # no all-to-all there: every rank gets everything and keeps its own shard of each
            parts = [torch.empty_like(st["send"]) for _ in range(self.world)]
            w = dist.all_gather(parts, st["send"], group=self.group, async_op=True)
            st["parts"] = parts
        else:
            w = dist.all_to_all_single(st["recv"], st["send"], group=self.group, async_op=True)
        self._stage2.append((b, w))

    def _second_halves(self):
        rank = dist.get_rank(self.group)
        for b, w in self._stage2:
            w.wait()                                    # (a stream dependency on the collective, not a host wait, under RCCL)
            s, e, shard = self._shards(b)
            st = self._staging[("a2a", b)]
            if "parts" in st and dist.get_backend(self.group) == "gloo":
                mine = torch.stack([q[rank * shard:(rank + 1) * shard] for q in st["parts"]])
            else:
                mine = st["recv"].view(self.world, shard)
            st["red"].copy_(mine.float().sum(dim=0))    # world shards summed in f32, rounded to bf16 once
            self._works.append(dist.all_gather_into_tensor(st["out"], st["red"], group=self.group, async_op=True))
        self._stage2 = []

    def finish(self):
        """Issue the collectives of buckets that only some of their parameters reached (same set on every rank,
        since control flow depends on global_step only) and wait for all.  The mean comes for free when the loss was
        pre-scaled by 1/world (`prescaled`); otherwise the reduced buckets are averaged here."""
        for b in sorted(self._touched_buckets - self._launched, reverse=True):
            self._launch(b)
        if self.f32_accumulate:
            self._second_halves()
        for w in self._works:
            w.wait()
        if self.comm_dtype is not None:
            for b in self.launch_order:
                s, e = self.buckets[b]
                src = self._staging[("a2a", b)]["out"][:e - s] if self.f32_accumulate else self._staging[b]
                self.arena[s:e].copy_(src)                # bf16 -> f32 back into the arena
        if not self.prescaled:
            for b in self.launch_order:
                s, e = self.buckets[b]
                self.arena[s:e].mul_(self.inv_world)
        self._pending = None
        if self._tl is not None and self.record_timeline:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._tl["end"] = ev

    # ---- start-up ------------------------------------------------------------------------------------------------------
    def broadcast_parameters(self, module):
        """Rank 0's weights and buffers everywhere (DDP does this at construction)."""
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)
        # raw .data writes bump no version counter: conv weight packs cached from an earlier forward are stale now
        from . import ops
        ops.PACK_CACHE.bump()


def all_reduce_mean(value, group=None):
    """Mean over ranks of a scalar (python number or 0-d tensor): the `sync_dist=True` of `self.log`
    (src/models/autoencoder.py:359).  Identity when no process group is active."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value
    t = value.detach().clone().float() if torch.is_tensor(value) else torch.tensor(float(value))
    if dist.get_backend(group) == "nccl" and not t.is_cuda:
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t / dist.get_world_size(group)
