"""torch.autograd.Function wrappers around the C ABI kernels (include/odvae_hip.h).

Activations are logical NCHW tensors in channels_last memory, i.e. NHWC in HBM; conv weights stay OIHW
parameters (checkpoint interchange with the reference) and are repacked per call.
torch supplies device memory, the current stream and autograd bookkeeping; every FLOP and every byte
of the hot path moves through libodvae_hip.so.  No CPU fallback exists.
"""
import contextlib
import os
import weakref

import torch
from torch.autograd import Function

from . import lib as _lib

CL = torch.channels_last


def _L():
    return _lib.load()


BF16 = torch.bfloat16


def _cl(t, dtype=torch.float32):
    """NHWC-in-memory view of a logical NCHW tensor (copy only if the caller handed us another layout)."""
    _lib.require_device(t, dtypes=(dtype,))
    if t.dim() != 4:
        raise ValueError("expected a 4-d NCHW tensor, got shape %s" % (tuple(t.shape),))
    n, c, h, w = t.shape
    if t.stride() == (h * w * c, 1, w * c, c):
        return t
    out = torch.empty((n, c, h, w), dtype=t.dtype, device=t.device, memory_format=CL)
    out.copy_(t)
    if out.stride() != (h * w * c, 1, w * c, c):  # degenerate sizes: force the canonical NHWC strides
        out = out.as_strided((n, c, h, w), (h * w * c, 1, w * c, c))
    return out


def _new_cl(n, c, h, w, like, dtype=torch.float32):
    t = torch.empty((n, h, w, c), dtype=dtype, device=like.device)
    return t.permute(0, 3, 1, 2)


def _ws(nbytes, like):
    return _lib.workspace.get(nbytes, like.device)


class _KernelEvents:
    """HIP-event brackets around individual kernel launches on torch's current stream (bench.py's live roofline).
    Off by default; when on, every launch of a tagged kernel records (algorithmic flops, start, stop)."""

    def __init__(self):
        self.on = False
        self.extra = False    # also bracket the secondary kernel families (bench.py: one extra step after the timed region)
        self.rec = {}
        self.issued = 0.0
        self.sample, self._n = 1, 0

    def enable(self):
        self.on, self.rec, self.issued = True, {}, 0.0

    def count(self, issued_flops):
        """Multiply-add work actually issued to the MFMA pipe by one launch (Winograd: 16/36 of the direct form); host
        arithmetic only, summed over the timed steps for bench.py's whole-step MFMA fraction."""
        if self.on:
            self.issued += issued_flops

    def disable(self):
        self.on = False

    def begin(self, secondary=False):
        if not self.on or (secondary and not self.extra):
            return None
        if not secondary and self.sample > 1:   # bracket one launch in `sample` of the dominant family (see bench.py)
            self._n += 1
            if self._n % self.sample:
                return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, name, flops, ev0, nbytes=0.0, issued=None, variant=None):
        """flops = ALGORITHMIC work of the launch (direct form); issued = multiply-add work actually sent to the MFMA pipe
        (defaults to flops).  variant: which instantiation of the family's kernel ran (its epilogue); the same record is then also
        listed under "name|variant" (summary("name|variant"), variants(name)): the family figure aggregates kernels that do different
        amounts of non-matrix work per launch."""
        if self.on:
            self.issued += flops if issued is None else issued
        if ev0 is None:
            return
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        item = (flops, ev0, ev1, nbytes, flops if issued is None else issued)
        self.rec.setdefault(name, []).append(item)
        if variant is not None:
            self.rec.setdefault(name + "|" + variant, []).append(item)

    def variants(self, name):
        return sorted(k.split("|", 1)[1] for k in self.rec if k.startswith(name + "|"))

    def summary(self, name):
        torch.cuda.synchronize()
        items = self.rec.get(name, [])
        if not items:
            return None
        total_ms = sum(it[1].elapsed_time(it[2]) for it in items)
        flops = sum(it[0] for it in items)
        nbytes = sum(it[3] for it in items)
        issued = sum(it[4] for it in items)
        return {"launches": len(items), "total_ms": total_ms, "avg_ms": total_ms / len(items),
                "gflop_per_launch": flops / len(items) / 1e9, "tflops": flops / total_ms / 1e9,
                "issued_gflop_per_launch": issued / len(items) / 1e9, "issued_tflops": issued / total_ms / 1e9,
                "bytes_per_launch": nbytes / len(items), "tbytes_per_s": nbytes / total_ms / 1e9}


KERNEL_EVENTS = _KernelEvents()


# ------------------------------------------------------------------------------------------------------
# 3x3 convolution family
# ------------------------------------------------------------------------------------------------------
class _PackCache:
    """Weight packs keyed by (storage pointer, tensor version, optimizer epoch): a conv's forward and data-gradient packs
    are built together, once per weight update, instead of once per call.  FusedAdam updates parameters from a raw
    kernel (no torch version bump), so it advances `epoch` itself.
    A stale entry is REPACKED INTO ITS OWN BUFFERS (same weight object, same pack kind, hence same sizes): the per-step churn of
    ~180 small allocations and frees kept the caching allocator's small pool fragmenting, and a training step in steady state asked the
    driver for fresh 2 MB segments once or twice (hipMalloc: a device synchronisation -- DESIGN.md 4, the bf16 256x256 side run).
    A graph built before a weight update must not be back-propagated after it: its data-gradient pack now holds the NEW weights
    (torch would raise its "modified by an inplace operation" error there; `check_epoch` raises the same way)."""

    INPLACE = os.environ.get("ODVAE_PACK_INPLACE", "1") != "0"

    def __init__(self):
        self.epoch = 0
        self._trainable = set()
        self.store = {}
        self._tables = {}

    def bump(self):
        self.epoch += 1

    def _epoch_of(self, weight):
        """The optimizer epoch a pack of this weight is valid for, or -1 for a weight that has never been trainable while this cache saw it
        (a tensor whose requires_grad is only switched off for a while -- the discriminator during the generator phase -- is still updated
        by its optimizer without a version bump)."""
        if weight.requires_grad:
            self._trainable.add(id(weight))
            return self.epoch
        return self.epoch if id(weight) in self._trainable else -1

    def get(self, weight, want_dgrad, up=False):
        key = (id(weight), up)
        # (a frozen weight -- the LPIPS-style VGG stack -- is in no optimizer: its packs do not go stale with the optimizer epoch, only with a
        # write to the tensor itself (load_state_dict: version counter))
        tag = (weight.data_ptr(), weight._version, self._epoch_of(weight))
        hit = self.store.get(key)
        same = hit is not None and hit[0]() is weight
        if same and hit[1] == tag and (hit[3] is not None or not want_dgrad):
            return hit[2], hit[3]
        reuse = None
        if same and self.INPLACE and hit[4] == (tuple(weight.shape), weight.device) and (hit[3] is not None or not want_dgrad):
            if self.BATCH and up in _BATCH_KINDS and self._refresh_kind(up, weight.device):
                hit = self.store.get(key)
                if hit is not None and hit[1] == tag:
                    return hit[2], hit[3]
            reuse = (hit[2], hit[3])          # refill the buffers this weight's packs already live in
            want_dgrad = hit[3] is not None
        fwd, dgr = _pack_conv3x3_now(weight, True, want_dgrad, up, into=reuse)
        # An entry dies with its weight tensor.  AttnBlock builds its [3C, C] q/k/v weight with torch.cat on every forward: each of
        # those temporaries left its two packs behind (round 3 purged dead entries only past 4 096 of them), 14 per step in the bf16
        # path, 6-10 MB that the allocator had to find -- a few fresh hipMalloc calls per step, each a device synchronisation; the
        # f32 path never saw it because its 1x1 convs read the weight as it is.
        store = self.store
        ref = weakref.ref(weight, lambda r, k=key: store.pop(k, None) if (store.get(k) or (None,))[0] is r else None)
        store[key] = (ref, tag, fwd, dgr, (tuple(weight.shape), weight.device))
        return fwd, dgr

    # One launch per pack KIND instead of one per weight: the first stale hit after a weight update refills every stale entry of that
    # kind (the f32 training step repacks 64 Winograd weight packs after every optimizer step: 15-30 us launches, mostly overhead; 234.7 -> 234.4 ms).
    # The device table of (weight, pack, pack, Cout, Cin, taps) records is rebuilt only when the set of entries changes: parameters live
    # in the optimizer's arena and the packs are refilled in place, so the pointers are the same step after step.
    BATCH = os.environ.get("ODVAE_PACK_BATCH", "1") != "0"

    def _refresh_kind(self, kind, device):
        import numpy as np
        L = _L()
        live = []
        for key, ent in self.store.items():
            if key[1] != kind or ent[2] is None:
                continue
            w = ent[0]()
            if w is None or w.device != device or ent[4] != (tuple(w.shape), w.device) or not w.is_contiguous() or w.dtype != torch.float32:
                continue
            tag = (w.data_ptr(), w._version, self._epoch_of(w))
            if ent[1] == tag or not _batchable(kind, w, L):
                continue
            live.append((key, ent, w, tag))
        if len(live) < 2:
            return False
        sig = tuple((k, e[2].data_ptr(), 0 if e[3] is None else e[3].data_ptr(), w.data_ptr()) for k, e, w, _ in live)
        cached = self._tables.get((kind, device))
        if cached is None or cached[0] != sig:
            rec = np.zeros(len(live), dtype=np.dtype([("w", "<u8"), ("fwd", "<u8"), ("dgr", "<u8"), ("cout", "<i4"), ("cin", "<i4"), ("taps", "<i4"), ("pad", "<i4")]))
            for i, (_, e, w, _) in enumerate(live):
                rec[i] = (w.data_ptr(), e[2].data_ptr(), 0 if e[3] is None else e[3].data_ptr(), w.shape[0], w.shape[1], w.shape[2] * w.shape[3], 0)
            table = _lib.upload(torch.from_numpy(rec.view(np.uint8).copy()), device)
            cached = self._tables[(kind, device)] = (sig, table)
        fn = {"wino": L.odvae_conv3x3_pack_wino_batch, "wino4": L.odvae_conv3x3_pack_wino4_batch}[kind]
        _lib.check(fn(cached[1].data_ptr(), len(live), _lib.stream_ptr()), "pack batch (%s)" % kind)
        for key, e, w, tag in live:
            self.store[key] = (e[0], tag, e[2], e[3], e[4])
        return True

    def check_epoch(self, epoch, what):
        if epoch != self.epoch:
            raise RuntimeError("%s: the weights were updated (optimizer step) between this graph's forward and its backward; the "
                               "cached weight packs now hold the new weights" % what)


_BATCH_KINDS = ("wino", "wino4")      # (bf16 packs: batching measured 0.3-0.65 ms per step slower, see conv_bf16.hip)


def _batchable(kind, w, L):
    """Shapes the batched pack kernels take: the Winograd packs without padding in either direction (the single-weight launcher zero-fills
    padding with a memset first)."""
    cout, cin = w.shape[0], w.shape[1]
    if tuple(w.shape[2:]) != (3, 3):
        return False
    rp, op = ((L.odvae_conv3x3_wino_reduce_pad, L.odvae_conv3x3_wino_out_pad) if kind == "wino"
              else (L.odvae_conv3x3_wino4_reduce_pad, L.odvae_conv3x3_wino4_out_pad))
    return rp(cin) == cin and op(cout) == cout and rp(cout) == cout and op(cin) == cin


PACK_CACHE = _PackCache()


def _pack_conv3x3_now(weight, want_fwd=True, want_dgrad=False, up=False, into=None):
    """up=True: the 16-tap packs of an Upsample conv (taps that hit the same low-res pixel pre-summed, modes 5 / 6);
    up="wino": the Winograd F(2x2,3x3) packs U = G g G^T of a stride-1 conv (conv3x3_wino_f32.hip).
    into=(fwd, dgr): refill these pack tensors (of the same weight and kind) instead of allocating."""
    L = _L()
    ifwd, idgr = into if into is not None else (None, None)
    w = weight.detach().contiguous()
    _lib.require_device(w)
    cout, cin = w.shape[0], w.shape[1]
    if up == "bf16":   # bf16 MFMA-fragment packs of a 3x3 or 1x1 conv (conv_bf16.hip); master weights stay f32
        taps = w.shape[2] * w.shape[3]
        fwd = (ifwd if ifwd is not None else torch.empty(L.odvae_conv_bf16_pack_elems(cin, cout, taps), dtype=BF16, device=w.device)) if want_fwd else None
        dgr = (idgr if idgr is not None else torch.empty(L.odvae_conv_bf16_pack_elems(cout, cin, taps), dtype=BF16, device=w.device)) if want_dgrad else None
        _lib.check(L.odvae_conv_pack_bf16(w.data_ptr(), cout, cin, taps, _lib.ptr(fwd), _lib.ptr(dgr), _lib.stream_ptr()), "conv_pack_bf16")
        return fwd, dgr
    floats = (L.odvae_conv3x3_wino_pack_floats if up == "wino" else L.odvae_conv3x3_wino4_pack_floats if up == "wino4" else
              L.odvae_conv3x3_up_pack_floats if up else L.odvae_conv3x3_pack_floats)
    fwd = dgr = None
    if want_fwd:
        fwd = ifwd if ifwd is not None else torch.empty(floats(cin, cout), dtype=torch.float32, device=w.device)
    if want_dgrad:
        dgr = idgr if idgr is not None else torch.empty(floats(cout, cin), dtype=torch.float32, device=w.device)
    pack = (L.odvae_conv3x3_pack_wino_f32 if up == "wino" else L.odvae_conv3x3_pack_wino4_f32 if up == "wino4" else
            L.odvae_conv3x3_pack_up_f32 if up else L.odvae_conv3x3_pack_f32)
    _lib.check(pack(w.data_ptr(), cout, cin, _lib.ptr(fwd), _lib.ptr(dgr), _lib.stream_ptr()), "conv3x3_pack")
    return fwd, dgr


def pack_conv3x3(weight, want_fwd=True, want_dgrad=False, up=False):
    """OIHW parameter -> kernel packs (see conv3x3_f32.hip), cached per weight update."""
    fwd, dgr = PACK_CACHE.get(weight, want_dgrad, up)
    return (fwd if want_fwd else None), (dgr if want_dgrad else None)


# Upsample + conv3x3: 1 = per output parity class with pre-summed taps (modes 5 / 6, 16 instead of 36 tap-products per
# input pixel); ODVAE_UPCONV_DENSE=1 keeps the dense form (mode 2, data gradient = mode 0 + 2x2 sum-pool) for A/B runs
UPCONV_BY_PARITY = os.environ.get("ODVAE_UPCONV_DENSE", "0") != "1"
# Upsample conv forward / data gradient on the F(4x4,3x3) kernel where the OUTPUT shape qualifies (2.25 instead of 4 multiply-adds per output
# pixel; the weight gradient stays on the parity-class kernel); ODVAE_UPCONV_WINOGRAD4=0 keeps modes 5 / 6
UPCONV_WINOGRAD4 = os.environ.get("ODVAE_UPCONV_WINOGRAD4", "1") != "0"
# ... and its data gradient 2x2-summed inside the F(4x4) output transform (odvae_conv3x3_wino4_pool_f32); ODVAE_UPCONV_POOLED_DGRAD=0: two steps
UPCONV_POOLED_DGRAD = os.environ.get("ODVAE_UPCONV_POOLED_DGRAD", "1") != "0"


def _conv3x3_raw(mode, x, pack, cin, cout, bias, residual, act=0):
    L = _L()
    n, _, hi, wi = x.shape
    if mode == 0:
        ho, wo = hi, wi
    elif mode in (1, 6):
        ho, wo = hi // 2, wi // 2
    else:
        ho, wo = 2 * hi, 2 * wi
    y = _new_cl(n, cout, ho, wo, x)
    tag = KERNEL_EVENTS.begin() if (mode == 0 and cout > 32 and cin > 3) else None   # the 128-wide v2 kernel only
    _lib.check(L.odvae_conv3x3_f32(mode, x.data_ptr(), n, hi, wi, cin, pack.data_ptr(), cout,
                                   _lib.ptr(bias), _lib.ptr(residual), y.data_ptr(), ho, wo, int(act), _lib.stream_ptr()),
               "conv3x3(mode=%d)" % mode)
    KERNEL_EVENTS.end("conv3x3_128x128", 2.0 * 9 * cin * cout * n * ho * wo, tag,
                      4.0 * (n * hi * wi * cin + n * ho * wo * cout * (2 if residual is not None else 1) + 9 * cin * cout),
                      issued=_conv_issued(mode, n, hi, wi, ho, wo, cin, cout))
    return y


def _conv_issued(mode, n, hi, wi, ho, wo, cin, cout):
    """Multiply-add FLOP the direct kernels issue: modes 0/1/2 nine taps per output pixel; mode 3 (transposed stride-2
    data gradient, inserted zeros skipped) nine per LOW-res pixel; modes 5/6 (Upsample conv by parity class) sixteen
    pre-summed taps per low-res pixel."""
    if mode in (5, 6, 3):
        lo = n * min(hi, ho) * min(wi, wo)
        return 2.0 * (9 if mode == 3 else 16) * cin * cout * lo
    return 2.0 * 9 * cin * cout * n * ho * wo


# Stride-1 3x3 convs by Winograd F(2x2, 3x3) (conv3x3_wino_f32.hip) when the shape allows; ODVAE_CONV_WINOGRAD=0 keeps
# the direct implicit-GEMM kernel everywhere
# True inside `weight_gradient_only()`: the 3x3 conv Functions skip their data gradient.  torch fixes ctx.needs_input_grad at FORWARD time, so a
# torch.autograd.grad(y, weight, grad_outputs=g) probe would otherwise also run (and throw away) the layer's data-gradient launch: the two
# last-layer probes of the adaptive weight (losses.adaptive_weight_one_pass) need decoder.conv_out's weight gradient only.
WEIGHT_GRADIENT_ONLY = False


@contextlib.contextmanager
def weight_gradient_only():
    global WEIGHT_GRADIENT_ONLY
    prev, WEIGHT_GRADIENT_ONLY = WEIGHT_GRADIENT_ONLY, True
    try:
        yield
    finally:
        WEIGHT_GRADIENT_ONLY = prev


WINOGRAD = os.environ.get("ODVAE_CONV_WINOGRAD", "1") != "0"
# Weight gradient of those convs in the Winograd domain (conv3x3_wgrad_wino_f32.hip) where the shape allows (even H, W,
# channel counts in multiples of 128); ODVAE_WGRAD_WINOGRAD=0 keeps the direct weight-gradient kernel everywhere
WGRAD_WINOGRAD = os.environ.get("ODVAE_WGRAD_WINOGRAD", "1") != "0"


def _wino_ok(h, w, cin, cout):
    # both the forward (reduce over cin) and the data gradient (reduce over cout) need channel counts in whole quads
    return WINOGRAD and h % 2 == 0 and w % 2 == 0 and cin % 4 == 0 and cout % 4 == 0 and cin >= 16 and cout >= 16


# F(4x4, 3x3) (conv3x3_wino4_f32.hip) where the shape allows: 36 products per 4x4 tile instead of 64.  ODVAE_CONV_WINOGRAD4=0/1.
WINOGRAD4 = os.environ.get("ODVAE_CONV_WINOGRAD4", "1") != "0"


def _wino4_ok(h, w, cin, cout):
    return WINOGRAD and WINOGRAD4 and bool(_L().odvae_conv3x3_wino4_supported(h, w, cin, cout))


# GroupNorm statistics of a conv's output from its own epilogue (conv3x3_wino4_f32.hip): the F(4x4) forward convs leave (sum, sum of
# squares) per output tile and channel group, and the GroupNorm that reads the tensor skips its statistics pass.  ODVAE_GN_FUSED_STATS=0
# turns it off.
GN_FUSED_STATS = os.environ.get("ODVAE_GN_FUSED_STATS", "1") != "0"
GN_GROUPS = 32      # Normalize() of the reference model: GroupNorm(32, C, eps 1e-6)


# First pass of a GroupNorm's backward (per-channel sums of du * xhat and du) from the epilogue of the data-gradient launch that produces
# its dy: the conv that reads a = swish(GroupNorm(x)) finds the link the GroupNorm left on `a`, its backward runs
# odvae_conv3x3_wino4_gnbwd_f32 and leaves the sums in the link; the GroupNorm's backward then skips gn_bwd_reduce_kernel (one read of x
# and of dy less).  ODVAE_GN_FUSED_BWD=0 turns it off.
GN_FUSED_BWD = os.environ.get("ODVAE_GN_FUSED_BWD", "1") != "0"
GN_FUSED_BWD_HITS = 0      # GroupNorm backwards that ran without their reduce pass (diagnostics, tests)
GN_FUSED_BWD_SUSPENDED = 0  # > 0 inside an activation-checkpointed unit (modules.Decoder): a saved tensor of such a unit may be unpacked once per
                            # backward, and the link would read the GroupNorm's from the conv's backward before the GroupNorm's own does


class _GnBwdLink:
    """What the consumer conv's data gradient needs of the GroupNorm in front of it, and where it leaves the sums.  `out` identifies the
    GroupNorm's output the way _gn_partials identifies a conv's: (data_ptr, version, shape).  The GroupNorm's operands are NOT held here:
    `node` is a weak reference to its autograd context, whose saved tensors (x, gamma, beta, mean, rstd) the conv's backward reads -- through
    the saved-tensor hooks, so an activation-checkpointed unit recomputes them instead of this link keeping the first forward's alive."""
    __slots__ = ("node", "groups", "out", "sums")

    def __init__(self):
        self.node = self.out = self.sums = None
        self.groups = 0

    def operands(self):
        """(x, mean, rstd, gamma, beta) of the GroupNorm, or None if its node is gone."""
        ctx = self.node() if self.node is not None else None
        if ctx is None:
            return None
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        return x, mean, rstd, gamma.detach().contiguous(), beta.detach().contiguous()

    def take_sums(self, dy):
        """The sums, if they were made from exactly this gradient tensor (else None); single use."""
        sums, self.sums = self.sums, None
        if sums is None:
            return None
        p, ptr, version, shape = sums
        return p if (ptr == dy.data_ptr() and version == dy._version and shape == tuple(dy.shape)) else None


def _gn_bwd_link_of(a, cin):
    """The link a swish-GroupNorm left on this very tensor (f32, 32 groups), if `a` still holds that GroupNorm's output."""
    link = getattr(a, "_gn_bwd_link", None)
    if link is None or not GN_FUSED_BWD or link.out != (a.data_ptr(), a._version, tuple(a.shape)) or a.shape[1] != cin:
        return None
    return link


def _gn_stats_ok(cout):
    cpg = cout // GN_GROUPS
    return GN_FUSED_STATS and cout % GN_GROUPS == 0 and 1 <= cpg <= 32 and (cpg & (cpg - 1)) == 0


def _conv3x3_wino_raw(x, pack, cin, cout, bias, residual, act=0, f4=False, stats=False, up=False):
    """stats=True (F(4x4) only): returns (y, partials [N][tiles][32][2]) -- the GroupNorm statistics of y per output tile.
    up=True (F(4x4) only): x is the low-resolution input of an Upsample conv, y has twice its height and width."""
    L = _L()
    n, _, h, w = x.shape
    if up:
        h, w = 2 * h, 2 * w
    y = _new_cl(n, cout, h, w, x)
    tag = KERNEL_EVENTS.begin() if cout > 32 else None
    partial = None
    if up:
        if stats:
            partial = torch.empty(n, L.odvae_conv3x3_wino4_stats_chunks(h, w), GN_GROUPS, 2, dtype=torch.float32, device=x.device)
        _lib.check(L.odvae_conv3x3_wino4_up_f32(x.data_ptr(), n, h, w, cin, pack.data_ptr(), cout, _lib.ptr(bias), _lib.ptr(residual),
                                                y.data_ptr(), _lib.ptr(partial), GN_GROUPS if stats else 0, _lib.stream_ptr()), "conv3x3_wino4_up")
    elif stats:
        partial = torch.empty(n, L.odvae_conv3x3_wino4_stats_chunks(h, w), GN_GROUPS, 2, dtype=torch.float32, device=x.device)
        _lib.check(L.odvae_conv3x3_wino4_stats_f32(x.data_ptr(), n, h, w, cin, pack.data_ptr(), cout, _lib.ptr(bias), _lib.ptr(residual),
                                                   y.data_ptr(), partial.data_ptr(), GN_GROUPS, _lib.stream_ptr()), "conv3x3_wino4_stats")
    else:
        fn = L.odvae_conv3x3_wino4_f32 if f4 else L.odvae_conv3x3_wino_f32
        _lib.check(fn(x.data_ptr(), n, h, w, cin, pack.data_ptr(), cout, _lib.ptr(bias), _lib.ptr(residual),
                      y.data_ptr(), int(act), _lib.stream_ptr()), "conv3x3_wino4" if f4 else "conv3x3_wino")
    # issued multiply-adds per output pixel and (ci, co): F(2x2,3x3) 16 per 2x2 tile = 4, F(4x4,3x3) 36 per 4x4 tile = 2.25
    KERNEL_EVENTS.end("conv3x3_wino4" if f4 else "conv3x3_128x128", 2.0 * 9 * cin * cout * n * h * w, tag,
                      4.0 * (n * h * w * cin // (4 if up else 1) + n * h * w * cout * (2 if residual is not None else 1) + 9 * cin * cout),
                      issued=2.0 * (2.25 if f4 else 4.0) * cin * cout * n * h * w,
                      variant=(("upsample" if up else "plain") + (" + GroupNorm statistics" if stats else "")) if f4 else None)
    return (y, partial) if stats else y


class _Conv3x3(Function):
    """mode 0: stride 1 pad 1; mode 1: Downsample (pad (0,1,0,1), stride 2); mode 2: Upsample (nearest 2x) + conv."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, mode, relu, gn_stats=False, gn_link=None):
        """gn_stats=True: returns (y, partials) when the F(4x4) kernel takes the layer (partials: GroupNorm statistics of y per output
        tile, not differentiable), (y, None) otherwise.  gn_link: x is swish(GroupNorm(.)) and this is its link (_GnBwdLink)."""
        x = _cl(x)
        res = _cl(residual) if residual is not None else None
        cout, cin = weight.shape[0], weight.shape[1]
        up = mode == 2 and UPCONV_BY_PARITY
        if up and UPCONV_WINOGRAD4 and not relu and _wino4_ok(2 * x.shape[2], 2 * x.shape[3], cin, cout):
            # the Upsample conv on the F(4x4) kernel: the halo of the (never formed) upsampled image is read from x[iy >> 1][ix >> 1]
            up = "wino4up"
        if mode == 0 and _wino_ok(x.shape[2], x.shape[3], cin, cout):
            # (not with a fused ReLU, i.e. not in the frozen VGG stack of the perceptual loss: F(4x4)'s ~1e-5 output error flips ten times
            # more ReLU masks than F(2x2)'s ~1e-6, and the input gradient of LPIPS then leaves the 5e-3 the parity tests hold it to)
            up = "wino4" if (not relu and _wino4_ok(x.shape[2], x.shape[3], cin, cout)) else "wino"
        fwd_pack, _ = pack_conv3x3(weight, True, bool(ctx.needs_input_grad[0]), "wino4" if up == "wino4up" else up)  # both packs in one launch
        b = bias.detach().contiguous() if bias is not None else None
        partial = None
        if up == "wino4up":
            want = bool(gn_stats and _gn_stats_ok(cout))
            out = _conv3x3_wino_raw(x, fwd_pack, cin, cout, b, res, f4=True, stats=want, up=True)
            y, partial = out if want else (out, None)
        elif up == "wino4" and gn_stats and _gn_stats_ok(cout):
            y, partial = _conv3x3_wino_raw(x, fwd_pack, cin, cout, b, res, f4=True, stats=True)
        elif up in ("wino", "wino4"):
            y = _conv3x3_wino_raw(x, fwd_pack, cin, cout, b, res, act=1 if relu else 0, f4=up == "wino4")
        else:
            y = _conv3x3_raw(5 if up else mode, x, fwd_pack, cin, cout, b, res, act=1 if relu else 0)
        ctx.mode, ctx.up = mode, up
        ctx.gn_link = gn_link if (up == "wino4" and gn_link is not None and ctx.needs_input_grad[0]) else None
        ctx.pack_epoch = PACK_CACHE.epoch
        ctx.relu = bool(relu)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        if gn_stats:
            if partial is not None:
                ctx.mark_non_differentiable(partial)
            ctx.set_materialize_grads(False)   # the statistics output has no gradient: not a tensor of zeros per backward (a fill launch each)
            return y, partial
        return y

    @staticmethod
    def backward(ctx, dy, _dpartial=None):
        L = _L()
        if dy is None:
            return None, None, None, None, None, None, None, None
        x, weight, y_act = ctx.saved_tensors
        mode = ctx.mode
        dy = _cl(dy)
        if ctx.relu:  # gradient through the fused ReLU: dy * (y > 0)
            g = _new_cl(*[dy.shape[i] for i in (0, 1, 2, 3)], dy)
            _lib.check(L.odvae_leaky_relu_bwd_f32(y_act.data_ptr(), dy.data_ptr(), g.data_ptr(), 0.0, dy.numel(),
                                                  _lib.stream_ptr()), "relu_bwd")
            dy = g
        cout, cin = weight.shape[0], weight.shape[1]
        n, _, hi, wi = x.shape
        _, _, ho, wo = dy.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0] and not WEIGHT_GRADIENT_ONLY:
            PACK_CACHE.check_epoch(ctx.pack_epoch, "conv3x3 backward")
            _, dgr = pack_conv3x3(weight, False, True, "wino4" if ctx.up == "wino4up" else ctx.up)
            if ctx.up == "wino4up" and UPCONV_POOLED_DGRAD:
                # gradient w.r.t. the upsampled image on the F(4x4) kernel, 2x2-summed in its output transform (never stored at full size)
                dx = _new_cl(n, cin, hi, wi, x)
                tag = KERNEL_EVENTS.begin()
                _lib.check(L.odvae_conv3x3_wino4_pool_f32(dy.data_ptr(), n, ho, wo, cout, dgr.data_ptr(), cin, dx.data_ptr(), _lib.stream_ptr()),
                           "conv3x3_wino4_pool")
                KERNEL_EVENTS.end("conv3x3_wino4", 2.0 * 9 * cin * cout * n * ho * wo, tag, 4.0 * (n * ho * wo * cout + n * hi * wi * cin + 9 * cin * cout),
                                  issued=2.0 * 2.25 * cin * cout * n * ho * wo, variant="data gradient, 2x2-summed (Upsample)")
            elif ctx.up == "wino4up":   # the same in two steps: full-resolution gradient, then its 2x2 sum-pool
                du = _conv3x3_wino_raw(dy, dgr, cout, cin, None, None, f4=True)
                dx = _new_cl(n, cin, hi, wi, x)
                _lib.check(L.odvae_upsample2x_bwd_f32(du.data_ptr(), dx.data_ptr(), n, hi, wi, cin, _lib.stream_ptr()), "upsample2x_bwd")
            elif ctx.up == "wino4" and ctx.gn_link is not None and (gn_ops := ctx.gn_link.operands()) is not None:
                # da and, from the same output transform, the first pass of the backward of the GroupNorm that produced this conv's input
                lk = ctx.gn_link
                gx, gmean, grstd, ggamma, gbeta = gn_ops
                dx = _new_cl(n, cin, hi, wi, x)
                sums = torch.empty(n, L.odvae_conv3x3_wino4_stats_chunks(hi, wi), 2, cin, dtype=torch.float32, device=x.device)
                tag = KERNEL_EVENTS.begin()
                _lib.check(L.odvae_conv3x3_wino4_gnbwd_f32(dy.data_ptr(), n, hi, wi, cout, dgr.data_ptr(), cin, dx.data_ptr(), _cl(gx).data_ptr(),
                                                           gmean.data_ptr(), grstd.data_ptr(), ggamma.data_ptr(), gbeta.data_ptr(),
                                                           lk.groups, sums.data_ptr(), _lib.stream_ptr()), "conv3x3_wino4_gnbwd")
                KERNEL_EVENTS.end("conv3x3_wino4", 2.0 * 9 * cin * cout * n * hi * wi, tag, 4.0 * (n * hi * wi * (2 * cin + cout) + 9 * cin * cout),
                                  issued=2.0 * 2.25 * cin * cout * n * hi * wi, variant="data gradient + GroupNorm-backward sums")
                lk.sums = (sums, dx.data_ptr(), dx._version, tuple(dx.shape))
            elif ctx.up in ("wino", "wino4"):
                dx = _conv3x3_wino_raw(dy, dgr, cout, cin, None, None, f4=ctx.up == "wino4")
            elif mode == 0:
                dx = _conv3x3_raw(0, dy, dgr, cout, cin, None, None)
            elif mode == 1:
                dx = _conv3x3_raw(3, dy, dgr, cout, cin, None, None)
            elif ctx.up:
                dx = _conv3x3_raw(6, dy, dgr, cout, cin, None, None)   # 4x4-tap stride-2 conv over dy, pre-summed weights
            else:
                du = _conv3x3_raw(0, dy, dgr, cout, cin, None, None)  # gradient w.r.t. the upsampled image
                dx = _new_cl(n, cin, hi, wi, x)
                _lib.check(L.odvae_upsample2x_bwd_f32(du.data_ptr(), dx.data_ptr(), n, hi, wi, cin, _lib.stream_ptr()),
                           "upsample2x_bwd")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            wmode = 5 if ctx.up in (True, "wino4up") else mode
            if mode == 0 and WGRAD_WINOGRAD and L.odvae_conv3x3_wgrad_wino_supported(n, hi, wi, cin, cout):
                need = L.odvae_conv3x3_wgrad_wino_workspace_bytes(n, hi, wi, cin, cout)
                wp, wn = _ws(need, x)
                tag = KERNEL_EVENTS.begin(secondary=True)
                _lib.check(L.odvae_conv3x3_wgrad_wino_f32(x.data_ptr(), dy.data_ptr(), n, hi, wi, cin, cout,
                                                          dw.data_ptr(), _lib.ptr(db), wp, wn, _lib.stream_ptr()),
                           "conv3x3_wgrad_wino")
                KERNEL_EVENTS.end("conv3x3_wgrad_wino", 2.0 * 9 * cin * cout * n * hi * wi, tag,
                                  4.0 * (n * hi * wi * (cin + cout) + 9 * cin * cout),
                                  issued=2.0 * 4 * cin * cout * n * hi * wi)
            else:
                need = L.odvae_conv3x3_wgrad_workspace_bytes(wmode, n, ho, wo, cin, cout)
                wp, wn = _ws(need, x)
                _lib.check(L.odvae_conv3x3_wgrad_f32(wmode, x.data_ptr(), dy.data_ptr(), n, hi, wi, cin, ho, wo, cout,
                                                     dw.data_ptr(), _lib.ptr(db), wp, wn, _lib.stream_ptr()),
                           "conv3x3_wgrad(mode=%d)" % wmode)
                KERNEL_EVENTS.end("conv3x3_wgrad", 2.0 * 9 * cin * cout * n * ho * wo, None,
                                  issued=_conv_issued(wmode, n, hi, wi, ho, wo, cin, cout))
        dres = dy if ctx.has_res and ctx.needs_input_grad[3] else None
        return dx, dw, db, dres, None, None, None, None


def conv3x3(x, weight, bias=None, residual=None, mode=0, relu=False, out_f32=False, gn_stats=False):
    """bf16 activations take the mixed-precision kernels (conv_bf16.hip); out_f32 makes that path hand back f32 (the f32 ends of
    the network: encoder.conv_out -> moments, decoder.conv_out -> reconstruction).
    gn_stats=True: the caller says a GroupNorm(32) reads the result next.  Where the F(4x4) kernel takes the layer, its output transform
    then also writes the GroupNorm statistics of y per tile; they travel as the attribute `_gn_partials` of the returned tensor (the
    Python object, so a copy or a view does not carry them) and `group_norm` / `group_norm_skip` use them instead of a statistics pass."""
    if x.dtype == BF16:
        if relu:
            raise NotImplementedError("fused ReLU is only on the f32 path (the LPIPS-style VGG stack stays f32)")
        if gn_stats and mode == 0 and not out_f32:
            y, partial = _ConvB.apply(x, weight, bias, residual, mode, False, True)
            if partial is not None:
                y._gn_partials = (partial, y.data_ptr(), y._version, tuple(y.shape))
            return y
        return _ConvB.apply(x, weight, bias, residual, mode, bool(out_f32))
    link = _gn_bwd_link_of(x, weight.shape[1]) if (mode == 0 and not relu) else None
    if gn_stats and not relu and mode in (0, 2):
        y, partial = _Conv3x3.apply(x, weight, bias, residual, mode, relu, True, link)
        if partial is not None:
            # the statistics are valid for exactly these values: the tag (storage, version counter, shape) lets the consumer tell
            # whether anything wrote to the tensor in between (an in-place op, a hook, a checkpoint wrapper's copy)
            y._gn_partials = (partial, y.data_ptr(), y._version, tuple(y.shape))
        return y
    return _Conv3x3.apply(x, weight, bias, residual, mode, relu, False, link)


# ------------------------------------------------------------------------------------------------------
# GEMM-backed ops
# ------------------------------------------------------------------------------------------------------
def gemm(ta, tb, m, n, k, alpha, a, lda, sa, b, ldb, sb, c, ldc, sc, bias=None, residual=None, batch=1):
    """Raw batched GEMM on device pointers held by tensors a, b, c (see gemm_f32.hip)."""
    L = _L()
    need = L.odvae_gemm_f32_workspace_bytes(m, n, k, batch)
    wp, wn = _ws(need, c) if need else (None, 0)
    tag = KERNEL_EVENTS.begin(secondary=True)
    _lib.check(L.odvae_gemm_f32(int(ta), int(tb), m, n, k, float(alpha), a.data_ptr(), lda, sa, b.data_ptr(), ldb, sb,
                                c.data_ptr(), ldc, sc, _lib.ptr(bias), _lib.ptr(residual), batch, wp, wn,
                                _lib.stream_ptr()), "gemm_f32")
    KERNEL_EVENTS.end("gemm_f32", 2.0 * m * n * k * batch, tag, 4.0 * batch * (m * k + k * n + m * n))


def _colsum(x2d_ptr_tensor, rows, c):
    L = _L()
    out = torch.empty(c, dtype=torch.float32, device=x2d_ptr_tensor.device)
    need = L.odvae_colsum_workspace_bytes(rows, c)
    wp, wn = _ws(need, out)
    _lib.check(L.odvae_colsum_f32(x2d_ptr_tensor.data_ptr(), rows, c, out.data_ptr(), wp, wn, _lib.stream_ptr()), "colsum")
    return out


class _Conv1x1(Function):
    """y[M][Cout] = x[M][Cin] . W[Cout][Cin]^T + b (+ residual), M = N*H*W pixels (NHWC)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        x = _cl(x)
        res = _cl(residual) if residual is not None else None
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        wt = weight.detach().reshape(cout, cin).contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        y = _new_cl(n, cout, h, w, x)
        m = n * h * w
        gemm(0, 1, m, cout, cin, 1.0, x, cin, 0, wt, cin, 0, y, cout, 0, b, res)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _cl(dy)
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        m = n * h * w
        wt = weight.detach().reshape(cout, cin).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _new_cl(n, cin, h, w, x)
            gemm(0, 0, m, cin, cout, 1.0, dy, cout, 0, wt, cin, 0, dx, cin, 0)
        if ctx.needs_input_grad[1]:
            dw2 = torch.empty(cout, cin, dtype=torch.float32, device=x.device)
            gemm(1, 0, cout, cin, m, 1.0, dy, cout, 0, x, cin, 0, dw2, cin, 0)
            dw = dw2.view(cout, cin, 1, 1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy, m, cout)
        dres = dy if ctx.has_res and ctx.needs_input_grad[3] else None
        return dx, dw, db, dres


def conv1x1(x, weight, bias=None, residual=None):
    if x.dtype == BF16:
        return _ConvB.apply(x, weight, bias, residual, 4, False)
    return _Conv1x1.apply(x, weight, bias, residual)


# softmax backward folded into the dP product's epilogue; ODVAE_ATTN_UNFUSED=1 keeps the separate row-wise pass
FUSED_SOFTMAX_BWD = os.environ.get("ODVAE_ATTN_UNFUSED", "0") != "1"


# Row softmax folded into the two products of the forward (include/odvae_hip.h, odvae_attn_row_bound_f32 ...): the QK^T epilogue writes
# E = exp(scale (s - bound_i)) against the Cauchy-Schwarz bound of the row's scores, the PV product sums the rows of E beside its multiplies
# and normalises its result; the backward takes P = E / l through row factors.  No softmax pass over the T x T tensor.  A bound too loose
# for f32 (a row whose exponentials all underflow) raises a device flag and predicated fallback launches redo the block with the row
# maximum.  ODVAE_ATTN_FOLDED_SOFTMAX=0: bmm -> softmax pass -> bmm.
ATTN_FOLDED_SOFTMAX = os.environ.get("ODVAE_ATTN_FOLDED_SOFTMAX", "1") != "0"
_ATTN_LAST_FLAG = None


class _Attention(Function):
    """Single-head attention over T = H*W tokens from a packed qkv tensor [N, 3C, H, W]:
    softmax(q k^T * C^-0.5) v  ([UPSTREAM] AttnBlock.forward).  Scores live in HBM (T x T per image)."""

    @staticmethod
    def forward(ctx, qkv):
        L = _L()
        qkv = _cl(qkv)
        n, c3, h, w = qkv.shape
        c = c3 // 3
        t = h * w
        scale = float(c) ** -0.5
        p = torch.empty(n, t, t, dtype=torch.float32, device=qkv.device)
        o = _new_cl(n, c, h, w, qkv)
        q = qkv.as_strided((1,), (1,), qkv.storage_offset())
        k = qkv.as_strided((1,), (1,), qkv.storage_offset() + c)
        v = qkv.as_strided((1,), (1,), qkv.storage_offset() + 2 * c)
        sq = t * c3
        rinv = None
        if ATTN_FOLDED_SOFTMAX and L.odvae_gemm_f32_workspace_bytes(t, t, c, n) == 0 and L.odvae_gemm_f32_workspace_bytes(t, c, t, n) == 0:
            st = _lib.stream_ptr()
            bound = torch.empty(2, n * t, dtype=torch.float32, device=qkv.device)      # [0]: the bounds, [1]: scratch (|k_j|)
            rinv = torch.empty(n * t, dtype=torch.float32, device=qkv.device)
            flag = torch.empty(1, dtype=torch.int32, device=qkv.device)
            _lib.check(L.odvae_attn_row_bound_f32(q.data_ptr(), n, t, c, bound.data_ptr(), bound[1].data_ptr(), flag.data_ptr(), st), "attn_row_bound")
            tag = KERNEL_EVENTS.begin(secondary=True)
            _lib.check(L.odvae_gemm_exp_bound_f32(t, t, c, scale, q.data_ptr(), c3, sq, k.data_ptr(), c3, sq, bound.data_ptr(), t,
                                                  p.data_ptr(), t, t * t, n, st), "gemm_exp_bound")
            KERNEL_EVENTS.end("gemm_f32", 2.0 * t * t * c * n, tag, 4.0 * n * (2 * t * c + t * t))
            tag = KERNEL_EVENTS.begin(secondary=True)
            _lib.check(L.odvae_gemm_rownorm_f32(t, c, t, p.data_ptr(), t, t * t, v.data_ptr(), c3, sq, o.data_ptr(), c, t * c,
                                                rinv.data_ptr(), t, flag.data_ptr(), n, st), "gemm_rownorm")
            KERNEL_EVENTS.end("gemm_f32", 2.0 * t * t * c * n, tag, 4.0 * n * (2 * t * c + t * t))
            # fallback under the device-side flag (three launches that return at once when it is 0): exact softmax, row factors 1
            _lib.check(L.odvae_gemm_pred_f32(0, 1, t, t, c, 1.0, q.data_ptr(), c3, sq, k.data_ptr(), c3, sq, p.data_ptr(), t, t * t, n,
                                             flag.data_ptr(), st), "gemm_pred(QK^T)")
            _lib.check(L.odvae_softmax_rows_pred_f32(p.data_ptr(), p.data_ptr(), n * t, t, scale, flag.data_ptr(), rinv.data_ptr(), st), "softmax_rows_pred")
            _lib.check(L.odvae_gemm_pred_f32(0, 0, t, c, t, 1.0, p.data_ptr(), t, t * t, v.data_ptr(), c3, sq, o.data_ptr(), c, t * c, n,
                                             flag.data_ptr(), st), "gemm_pred(PV)")
            global _ATTN_LAST_FLAG
            _ATTN_LAST_FLAG = flag      # (diagnostics / tests: 1 after a forward whose bound underflowed and whose fallback ran)
        else:
            gemm(0, 1, t, t, c, 1.0, q, c3, sq, k, c3, sq, p, t, t * t, batch=n)
            _lib.check(L.odvae_softmax_rows_f32(p.data_ptr(), p.data_ptr(), n * t, t, scale, _lib.stream_ptr()), "softmax_rows")
            gemm(0, 0, t, c, t, 1.0, p, t, t * t, v, c3, sq, o, c, t * c, batch=n)
        ctx.folded = rinv is not None
        if rinv is not None:
            ctx.save_for_backward(qkv, p, o, rinv)
        else:
            ctx.save_for_backward(qkv, p, o)
        return o

    @staticmethod
    def backward(ctx, do):
        L = _L()
        if ctx.folded:
            qkv, p, o, rinv = ctx.saved_tensors      # p holds E = exp(scale (s - bound)), P = E * rinv[row]
        else:
            (qkv, p, o), rinv = ctx.saved_tensors, None
        do = _cl(do)
        n, c3, h, w = qkv.shape
        c = c3 // 3
        t = h * w
        scale = float(c) ** -0.5
        sq = t * c3
        q = qkv.as_strided((1,), (1,), qkv.storage_offset())
        k = qkv.as_strided((1,), (1,), qkv.storage_offset() + c)
        v = qkv.as_strided((1,), (1,), qkv.storage_offset() + 2 * c)
        dqkv = _new_cl(n, c3, h, w, qkv)
        dq = dqkv.as_strided((1,), (1,), dqkv.storage_offset())
        dk = dqkv.as_strided((1,), (1,), dqkv.storage_offset() + c)
        dv = dqkv.as_strided((1,), (1,), dqkv.storage_offset() + 2 * c)
        if rinv is not None:
            # D_i = dO_i . O_i and dO_i / l_i in one pass over dO; dV = E^T (dO / l); dS = scale E rinv (dO V^T - D)
            drow = torch.empty(n * t, dtype=torch.float32, device=p.device)
            dos = torch.empty_like(do)
            _lib.check(L.odvae_rowdot_scale_f32(do.data_ptr(), o.data_ptr(), rinv.data_ptr(), n * t, c, drow.data_ptr(), dos.data_ptr(),
                                                _lib.stream_ptr()), "rowdot_scale")
            gemm(1, 0, t, c, t, 1.0, p, t, t * t, dos, c, t * c, dv, c3, sq, batch=n)
            del dos
            dp = torch.empty_like(p)
            tag = KERNEL_EVENTS.begin(secondary=True)
            _lib.check(L.odvae_gemm_softmax_bwd_scaled_f32(t, t, c, scale, do.data_ptr(), c, t * c, v.data_ptr(), c3, sq, p.data_ptr(),
                                                           drow.data_ptr(), rinv.data_ptr(), t, dp.data_ptr(), t, t * t, n, _lib.stream_ptr()),
                       "gemm_softmax_bwd_scaled")
            KERNEL_EVENTS.end("gemm_f32", 2.0 * t * t * c * n, tag, 4.0 * n * (2 * t * c + 2 * t * t))
            gemm(0, 0, t, c, t, 1.0, dp, t, t * t, k, c3, sq, dq, c3, sq, batch=n)
            gemm(1, 0, t, c, t, 1.0, dp, t, t * t, q, c3, sq, dk, c3, sq, batch=n)
            return dqkv
        # dV = P^T dO
        gemm(1, 0, t, c, t, 1.0, p, t, t * t, do, c, t * c, dv, c3, sq, batch=n)
        # dS = scale * P .* (dO V^T - D), D[i] = sum_j P[i][j] dP[i][j] = dO[i] . O[i]: the softmax backward rides in the
        # epilogue of the dP product (no dP round trip through HBM, no separate softmax-backward pass)
        dp = torch.empty_like(p)
        if FUSED_SOFTMAX_BWD:
            drow = torch.empty(n * t, dtype=torch.float32, device=p.device)
            _lib.check(L.odvae_rowdot_f32(do.data_ptr(), o.data_ptr(), n * t, c, drow.data_ptr(), _lib.stream_ptr()), "rowdot")
            tag = KERNEL_EVENTS.begin(secondary=True)
            _lib.check(L.odvae_gemm_softmax_bwd_f32(t, t, c, scale, do.data_ptr(), c, t * c, v.data_ptr(), c3, sq, p.data_ptr(),
                                                    drow.data_ptr(), t, dp.data_ptr(), t, t * t, n, _lib.stream_ptr()),
                       "gemm_softmax_bwd")
            KERNEL_EVENTS.end("gemm_f32", 2.0 * t * t * c * n, tag, 4.0 * n * (2 * t * c + 2 * t * t))
        else:
            gemm(0, 1, t, t, c, 1.0, do, c, t * c, v, c3, sq, dp, t, t * t, batch=n)
            _lib.check(L.odvae_softmax_rows_bwd_f32(p.data_ptr(), dp.data_ptr(), dp.data_ptr(), n * t, t, scale,
                                                    _lib.stream_ptr()), "softmax_rows_bwd")
        # dQ = dS K ; dK = dS^T Q
        gemm(0, 0, t, c, t, 1.0, dp, t, t * t, k, c3, sq, dq, c3, sq, batch=n)
        gemm(1, 0, t, c, t, 1.0, dp, t, t * t, q, c3, sq, dk, c3, sq, batch=n)
        return dqkv


# f32 path at long sequences.  _Attention keeps the probabilities P [N, T, T] for the backward: 64 MiB per image and block at T = 4 096
# (256 x 256 input), but 1 GiB at T = 16 384 (512 x 512) -- five blocks at B = 32 would hold 160 GiB.  Past this budget (bytes of P per
# block; ODVAE_ATTN_SCORE_BUDGET_GB) the block runs in groups of images through ONE score buffer and keeps only q, k, v and o: the
# backward recomputes P group by group (a fifth product, as a fused attention would) -- no T x T tensor outlives its group.
ATTN_SCORE_BUDGET = int(float(os.environ.get("ODVAE_ATTN_SCORE_BUDGET_GB", "4")) * 2 ** 30)


class _AttentionRecompute(Function):
    """softmax(q k^T * C^-0.5) v as _Attention, with the T x T scores living in a scratch buffer of at most ATTN_SCORE_BUDGET bytes."""

    @staticmethod
    def _views(t_, c):
        off = t_.storage_offset()
        return tuple(t_.as_strided((1,), (1,), off + i * c) for i in range(3))

    @staticmethod
    def _probabilities(L, qkv, g0, g, t, c, scale, p):
        c3, sq = 3 * c, t * 3 * c
        q, k, _ = _AttentionRecompute._views(qkv[g0:g0 + g], c)
        gemm(0, 1, t, t, c, 1.0, q, c3, sq, k, c3, sq, p, t, t * t, batch=g)
        _lib.check(L.odvae_softmax_rows_f32(p.data_ptr(), p.data_ptr(), g * t, t, scale, _lib.stream_ptr()), "softmax_rows")

    @staticmethod
    def forward(ctx, qkv, group):
        L = _L()
        qkv = _cl(qkv)
        n, c3, h, w = qkv.shape
        c, t = c3 // 3, h * w
        scale = float(c) ** -0.5
        o = _new_cl(n, c, h, w, qkv)
        p = torch.empty(min(group, n), t, t, dtype=torch.float32, device=qkv.device)
        for g0 in range(0, n, group):
            g = min(group, n - g0)
            _AttentionRecompute._probabilities(L, qkv, g0, g, t, c, scale, p)
            v = _AttentionRecompute._views(qkv[g0:g0 + g], c)[2]
            gemm(0, 0, t, c, t, 1.0, p, t, t * t, v, c3, t * c3, o[g0:g0 + g], c, t * c, batch=g)
        ctx.group = group
        ctx.save_for_backward(qkv, o)
        return o

    @staticmethod
    def backward(ctx, do):
        L = _L()
        qkv, o = ctx.saved_tensors
        do = _cl(do)
        n, c3, h, w = qkv.shape
        c, t = c3 // 3, h * w
        scale = float(c) ** -0.5
        sq = t * c3
        group = max(1, ctx.group // 2)          # two score-sized buffers live here (P and dS)
        dqkv = _new_cl(n, c3, h, w, qkv)
        p = torch.empty(min(group, n), t, t, dtype=torch.float32, device=qkv.device)
        dp = torch.empty_like(p)
        drow = torch.empty(n * t, dtype=torch.float32, device=qkv.device)
        _lib.check(L.odvae_rowdot_f32(do.data_ptr(), o.data_ptr(), n * t, c, drow.data_ptr(), _lib.stream_ptr()), "rowdot")
        for g0 in range(0, n, group):
            g = min(group, n - g0)
            _AttentionRecompute._probabilities(L, qkv, g0, g, t, c, scale, p)
            q, k, v = _AttentionRecompute._views(qkv[g0:g0 + g], c)
            dq, dk, dv = _AttentionRecompute._views(dqkv[g0:g0 + g], c)
            dog = do[g0:g0 + g]
            gemm(1, 0, t, c, t, 1.0, p, t, t * t, dog, c, t * c, dv, c3, sq, batch=g)                       # dV = P^T dO
            tag = KERNEL_EVENTS.begin(secondary=True)
            _lib.check(L.odvae_gemm_softmax_bwd_f32(t, t, c, scale, dog.data_ptr(), c, t * c, v.data_ptr(), c3, sq, p.data_ptr(),
                                                    drow.data_ptr() + 4 * g0 * t, t, dp.data_ptr(), t, t * t, g, _lib.stream_ptr()),
                       "gemm_softmax_bwd")                                                                 # dS = scale P (dO V^T - D)
            KERNEL_EVENTS.end("gemm_f32", 2.0 * t * t * c * g, tag, 4.0 * g * (2 * t * c + 2 * t * t))
            gemm(0, 0, t, c, t, 1.0, dp, t, t * t, k, c3, sq, dq, c3, sq, batch=g)                          # dQ = dS K
            gemm(1, 0, t, c, t, 1.0, dp, t, t * t, q, c3, sq, dk, c3, sq, batch=g)                          # dK = dS^T Q
        return dqkv, None


def attention_qkv(qkv):
    if qkv.dtype == BF16:
        return _FlashAttention.apply(qkv)
    n, c3, h, w = qkv.shape
    per_image = (h * w) ** 2 * 4
    if n * per_image > ATTN_SCORE_BUDGET:
        return _AttentionRecompute.apply(qkv, max(1, ATTN_SCORE_BUDGET // per_image))
    return _Attention.apply(qkv)


# ------------------------------------------------------------------------------------------------------
# GroupNorm (+ swish)
# ------------------------------------------------------------------------------------------------------
class _GroupNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, swish, with_skip=False, partials=None, link=None):
        """partials [N][chunks][groups][2]: the statistics of x as the conv that produced it left them (ops.conv3x3(gn_stats=True)):
        no statistics pass.  link (_GnBwdLink, swish only): filled here for the conv that reads the result (GN_FUSED_BWD)."""
        L = _L()
        x = _cl(x)
        n, c, h, w = x.shape
        g = gamma.detach().contiguous()
        b = beta.detach().contiguous()
        y = _new_cl(n, c, h, w, x)
        mean = torch.empty(n, groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty(n, groups, dtype=torch.float32, device=x.device)
        tag = KERNEL_EVENTS.begin(secondary=True)
        if partials is not None and partials.shape[0] == n and partials.shape[2] == groups:
            _lib.check(L.odvae_groupnorm_fwd_partials_f32(x.data_ptr(), n, h * w, c, groups, g.data_ptr(), b.data_ptr(), float(eps),
                                                          int(swish), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                          partials.data_ptr(), int(partials.shape[1]), _lib.stream_ptr()), "groupnorm_fwd_partials")
        else:
            wp, wn = _ws(L.odvae_groupnorm_workspace_bytes(n, h * w, c, groups), x)
            _lib.check(L.odvae_groupnorm_fwd_f32(x.data_ptr(), n, h * w, c, groups, g.data_ptr(), b.data_ptr(), float(eps),
                                                 int(swish), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), wp, wn,
                                                 _lib.stream_ptr()), "groupnorm_fwd")
        # algorithmic traffic: x read once, y written once
        KERNEL_EVENTS.end("groupnorm", 0.0, tag, 4.0 * 2 * n * h * w * c, issued=0.0)
        ctx.groups, ctx.swish = groups, int(swish)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.set_materialize_grads(False)   # an unused output's gradient stays None instead of a tensor of zeros
        ctx.link = None
        if link is not None and swish:
            link.node, link.groups = weakref.ref(ctx), groups
            ctx.link = link
        if with_skip:
            return y, x.view_as(x)   # the skip connection's handle on x: its gradient comes back into this node
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        L = _L()
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        if dy is None:               # only the skip branch carried a gradient
            return dskip, None, None, None, None, None, None, None, None
        dy = _cl(dy)
        sums = ctx.link.take_sums(dy) if ctx.link is not None else None
        if dskip is not None:
            dskip = _cl(dskip)
        n, c, h, w = x.shape
        dx = _new_cl(n, c, h, w, x)
        dg = torch.empty(c, dtype=torch.float32, device=x.device)
        db = torch.empty(c, dtype=torch.float32, device=x.device)
        g = gamma.detach().contiguous()
        b = beta.detach().contiguous()
        wp, wn = _ws(L.odvae_groupnorm_workspace_bytes(n, h * w, c, ctx.groups), x)
        tag = KERNEL_EVENTS.begin(secondary=True)
        if sums is not None:         # the data-gradient launch that made dy left the first pass's sums: finalize + apply only
            global GN_FUSED_BWD_HITS
            GN_FUSED_BWD_HITS += 1
            _lib.check(L.odvae_groupnorm_bwd_partials_f32(x.data_ptr(), dy.data_ptr(), n, h * w, c, ctx.groups, g.data_ptr(),
                                                          b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ctx.swish, dx.data_ptr(),
                                                          dg.data_ptr(), db.data_ptr(), _lib.ptr(dskip), sums.data_ptr(), int(sums.shape[1]),
                                                          wp, wn, _lib.stream_ptr()), "groupnorm_bwd_partials")
        else:
            _lib.check(L.odvae_groupnorm_bwd_f32(x.data_ptr(), dy.data_ptr(), n, h * w, c, ctx.groups, g.data_ptr(),
                                                 b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ctx.swish, dx.data_ptr(),
                                                 dg.data_ptr(), db.data_ptr(), _lib.ptr(dskip), wp, wn, _lib.stream_ptr()),
                       "groupnorm_bwd")
        # algorithmic traffic: x, dy (and the folded skip gradient) read once, dx written once
        KERNEL_EVENTS.end("groupnorm", 0.0, tag, 4.0 * (3 + (dskip is not None)) * n * h * w * c, issued=0.0)
        return dx, dg, db, None, None, None, None, None, None


def _gn_partials_of(x, groups):
    """Statistics the producing conv attached to this very tensor object (ops.conv3x3(gn_stats=True)), if they fit."""
    tagged = getattr(x, "_gn_partials", None)
    if tagged is None or not GN_FUSED_STATS or groups != GN_GROUPS:
        return None
    p, ptr, version, shape = tagged
    if ptr != x.data_ptr() or version != x._version or shape != tuple(x.shape) or p.device != x.device:
        return None      # the tensor was written to (or is not the conv's output any more): take the statistics pass
    n, _, h, w = x.shape
    chunks = _L().odvae_conv_bf16_stats_chunks(h, w) if x.dtype == BF16 else _L().odvae_conv3x3_wino4_stats_chunks(h, w)
    if p.shape[0] != n or p.shape[2] != groups or p.shape[1] != chunks:
        return None
    return p


def _f32_gn_link(x, groups, swish):
    ok = GN_FUSED_BWD and not GN_FUSED_BWD_SUSPENDED and swish and groups == GN_GROUPS and torch.is_grad_enabled() and x.requires_grad
    return _GnBwdLink() if ok else None


class gn_fused_bwd_suspended:
    """Context manager: no GroupNorm-backward links for the forward calls inside (activation-checkpointed units)."""

    def __enter__(self):
        global GN_FUSED_BWD_SUSPENDED
        GN_FUSED_BWD_SUSPENDED += 1

    def __exit__(self, *exc):
        global GN_FUSED_BWD_SUSPENDED
        GN_FUSED_BWD_SUSPENDED -= 1
        return False


def _tag_gn_output(y, link):
    if link is not None and link.node is not None:
        link.out = (y.data_ptr(), y._version, tuple(y.shape))
        y._gn_bwd_link = link
    return y


def group_norm(x, gamma, beta, groups=32, eps=1e-6, swish=False):
    if x.dtype == BF16:
        return _GroupNormB.apply(x, gamma, beta, groups, eps, swish, False, _gn_partials_of(x, groups))
    link = _f32_gn_link(x, groups, swish)
    return _tag_gn_output(_GroupNorm.apply(x, gamma, beta, groups, eps, swish, False, _gn_partials_of(x, groups), link), link)


def group_norm_skip(x, gamma, beta, groups=32, eps=1e-6, swish=False):
    """(GroupNorm(x), x): the second output is x itself, to be used by the block's skip connection (`x + h`).  Its
    gradient returns into this node and is summed into dx inside the GroupNorm backward pass (one pass instead of
    autograd's separate 3-pass add)."""
    if x.dtype == BF16:
        return _GroupNormB.apply(x, gamma, beta, groups, eps, swish, True, _gn_partials_of(x, groups))
    link = _f32_gn_link(x, groups, swish)
    y, skip = _GroupNorm.apply(x, gamma, beta, groups, eps, swish, True, _gn_partials_of(x, groups), link)
    return _tag_gn_output(y, link), skip


# "norm" activation-checkpoint policy: the tensor a = act(GroupNorm(x)) is NOT kept for the backward of the conv that consumes it; that conv's
# saved input is replaced (saved-tensor hooks) by what it takes to re-make it -- x, mean, rstd, gamma, beta -- and one apply pass rebuilds it when
# the conv's backward asks for its saved tensors.  Against the unit policy (torch.utils.checkpoint around a whole ResnetBlock) nothing but the
# GroupNorm apply is recomputed: no conv, no attention forward runs twice.
def group_norm_apply(x, gamma, beta, mean, rstd, groups, swish):
    """act(GroupNorm(x)) from a forward call's statistics (no autograd): the recompute of the "norm" policy."""
    L = _L()
    x = _cl(x, x.dtype if x.dtype == BF16 else torch.float32)
    n, c, h, w = x.shape
    y = _new_cl(n, c, h, w, x, dtype=x.dtype)
    fn = L.odvae_groupnorm_apply_bf16 if x.dtype == BF16 else L.odvae_groupnorm_apply_f32
    _lib.check(fn(x.data_ptr(), n, h * w, c, int(groups), gamma.detach().contiguous().data_ptr(), beta.detach().contiguous().data_ptr(),
                  mean.data_ptr(), rstd.data_ptr(), int(swish), y.data_ptr(), _lib.stream_ptr()), "groupnorm_apply")
    return y


class _RemakeFromNorm:
    __slots__ = ("x", "gamma", "beta", "mean", "rstd", "groups", "swish")

    def __init__(self, node):
        self.x, self.gamma, self.beta, self.mean, self.rstd = node.saved_tensors
        self.groups, self.swish = node.groups, node.swish

    def make(self):
        with torch.no_grad():
            return group_norm_apply(self.x, self.gamma, self.beta, self.mean, self.rstd, self.groups, self.swish)


class remake_from_norm(torch.autograd.graph.saved_tensors_hooks):
    """with remake_from_norm(a): y = conv(a) -- `a` must be the output of ops.group_norm / group_norm_skip (its grad_fn holds x, mean, rstd)."""

    def __init__(self, a):
        node = a.grad_fn
        ok = node is not None and hasattr(node, "groups") and hasattr(node, "swish") and len(getattr(node, "saved_tensors", ())) == 5
        key = (a.data_ptr(), tuple(a.shape), a.dtype) if ok else None
        recipe = _RemakeFromNorm(node) if ok else None

        def pack(t):
            if key is not None and t.data_ptr() == key[0] and tuple(t.shape) == key[1] and t.dtype == key[2]:
                return recipe
            return t

        def unpack(obj):
            return obj.make() if isinstance(obj, _RemakeFromNorm) else obj

        super().__init__(pack, unpack)


# ------------------------------------------------------------------------------------------------------
# input rescale, posterior, reconstruction term
# ------------------------------------------------------------------------------------------------------
def rescale_minmax(x_nchw):
    """2(x-min)/(max-min)-1 over the whole local batch (src/models/autoencoder.py:434-436); returns NHWC memory."""
    L = _L()
    _lib.require_device(x_nchw)
    x = x_nchw.detach().contiguous()
    n, c, h, w = x.shape
    y = _new_cl(n, c, h, w, x)
    mm = torch.empty(2, dtype=torch.float32, device=x.device)
    wp, wn = _ws(16384, x)
    _lib.check(L.odvae_rescale_minmax_f32(x.data_ptr(), y.data_ptr(), n, c, h * w, mm.data_ptr(), wp, wn,
                                          _lib.stream_ptr()), "rescale_minmax")
    return y


class _GaussianSample(Function):
    @staticmethod
    def forward(ctx, moments, eps):
        L = _L()
        moments = _cl(moments)
        eps = _cl(eps)
        n, c2, h, w = moments.shape
        cz = c2 // 2
        z = _new_cl(n, cz, h, w, moments)
        _lib.check(L.odvae_gaussian_sample_f32(moments.data_ptr(), eps.data_ptr(), z.data_ptr(), n, h * w, cz,
                                               _lib.stream_ptr()), "gaussian_sample")
        ctx.save_for_backward(moments, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        L = _L()
        moments, eps = ctx.saved_tensors
        dz = _cl(dz)
        n, c2, h, w = moments.shape
        dm = _new_cl(n, c2, h, w, moments)
        _lib.check(L.odvae_gaussian_bwd_f32(moments.data_ptr(), eps.data_ptr(), dz.data_ptr(), None, dm.data_ptr(),
                                            n, h * w, c2 // 2, _lib.stream_ptr()), "gaussian_bwd(sample)")
        return dm, None


class _GaussianKL(Function):
    @staticmethod
    def forward(ctx, moments):
        L = _L()
        moments = _cl(moments)
        n, c2, h, w = moments.shape
        kl = torch.empty(n, dtype=torch.float32, device=moments.device)
        _lib.check(L.odvae_gaussian_kl_f32(moments.data_ptr(), kl.data_ptr(), n, h * w, c2 // 2, _lib.stream_ptr()),
                   "gaussian_kl")
        ctx.save_for_backward(moments)
        return kl

    @staticmethod
    def backward(ctx, dkl):
        L = _L()
        (moments,) = ctx.saved_tensors
        n, c2, h, w = moments.shape
        dkl = dkl.contiguous()
        dm = _new_cl(n, c2, h, w, moments)
        _lib.check(L.odvae_gaussian_bwd_f32(moments.data_ptr(), None, None, dkl.data_ptr(), dm.data_ptr(),
                                            n, h * w, c2 // 2, _lib.stream_ptr()), "gaussian_bwd(kl)")
        return dm


def gaussian_sample(moments, eps):
    return _GaussianSample.apply(moments, eps)


def gaussian_kl(moments):
    return _GaussianKL.apply(moments)


class _L1MaskedSum(Function):
    """per-sample sum |x*m - xr*m| (contperceptual.py:137 with the mask_2d_bbox products of :252-255 folded in)."""

    @staticmethod
    def forward(ctx, x, xr, mask):
        L = _L()
        x = _cl(x)
        xr = _cl(xr)
        n, c, h, w = x.shape
        m = mask.detach().reshape(n, h * w).contiguous() if mask is not None else None
        _lib.require_device(m)
        out = torch.empty(n, dtype=torch.float32, device=x.device)
        wp, wn = _ws(n * 1024, x)
        _lib.check(L.odvae_l1_masked_sum_f32(x.data_ptr(), xr.data_ptr(), _lib.ptr(m), out.data_ptr(), n, h * w, c,
                                             wp, wn, _lib.stream_ptr()), "l1_masked_sum")
        ctx.save_for_backward(x, xr, m)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _L()
        x, xr, m = ctx.saved_tensors
        n, c, h, w = x.shape
        g = g.contiguous()
        dxr = _new_cl(n, c, h, w, x)
        _lib.check(L.odvae_l1_masked_bwd_f32(x.data_ptr(), xr.data_ptr(), _lib.ptr(m), g.data_ptr(), dxr.data_ptr(),
                                             n, h * w, c, _lib.stream_ptr()), "l1_masked_bwd")
        return None, dxr, None


def l1_masked_sum(x, xr, mask=None):
    return _L1MaskedSum.apply(x, xr, mask)


def nhwc_to_nchw(x):
    """Contiguous NCHW copy of an NHWC-in-memory tensor (hand-off to NCHW consumers)."""
    L = _L()
    x = _cl(x)
    n, c, h, w = x.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    _lib.check(L.odvae_nhwc_to_nchw_f32(x.data_ptr(), y.data_ptr(), n, c, h * w, _lib.stream_ptr()), "nhwc_to_nchw")
    return y


class _MulMask(Function):
    @staticmethod
    def forward(ctx, x, mask):
        L = _L()
        x = _cl(x)
        n, c, h, w = x.shape
        m = mask.detach().to(x.device).float().reshape(n, h * w).contiguous()
        y = _new_cl(n, c, h, w, x)
        _lib.check(L.odvae_mul_mask_f32(x.data_ptr(), m.data_ptr(), y.data_ptr(), n * h * w, c, _lib.stream_ptr()), "mul_mask")
        ctx.save_for_backward(m)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        (m,) = ctx.saved_tensors
        dy = _cl(dy)
        n, c, h, w = dy.shape
        dx = _new_cl(n, c, h, w, dy)
        _lib.check(L.odvae_mul_mask_f32(dy.data_ptr(), m.data_ptr(), dx.data_ptr(), n * h * w, c, _lib.stream_ptr()), "mul_mask bwd")
        return dx, None


def mul_mask(x, mask):
    """x * mask_2d_bbox ([B,1,H,W]); identity when mask is None."""
    return x if mask is None else _MulMask.apply(x, mask)


class _LatentCombine(Function):
    """z * mask + add on the latent (dropout keep-mask already scaled by 1/(1-p); noise or enc_pose as `add`)."""

    @staticmethod
    def forward(ctx, z, mask, add):
        L = _L()
        z = _cl(z)
        m = _cl(mask) if mask is not None else None
        a = _cl(add) if add is not None else None
        n, c, h, w = z.shape
        out = _new_cl(n, c, h, w, z)
        _lib.check(L.odvae_latent_combine_f32(z.data_ptr(), _lib.ptr(m), _lib.ptr(a), out.data_ptr(), z.numel(),
                                              _lib.stream_ptr()), "latent_combine")
        ctx.save_for_backward(m)
        ctx.has_add = add is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _L()
        (m,) = ctx.saved_tensors
        dout = _cl(dout)
        dz = dout
        if m is not None and ctx.needs_input_grad[0]:
            n, c, h, w = dout.shape
            dz = _new_cl(n, c, h, w, dout)
            _lib.check(L.odvae_latent_combine_f32(dout.data_ptr(), m.data_ptr(), None, dz.data_ptr(), dout.numel(),
                                                  _lib.stream_ptr()), "latent_combine bwd")
        return dz, None, (dout if ctx.has_add and ctx.needs_input_grad[2] else None)


def latent_combine(z, mask=None, add=None):
    return _LatentCombine.apply(z, mask, add)


# ------------------------------------------------------------------------------------------------------
# PatchGAN discriminator pieces (gan_f32.hip)
# ------------------------------------------------------------------------------------------------------
def _conv4x4_out(h, stride):
    return (h + 2 - 4) // stride + 1


class _Conv4x4(Function):
    """Conv2d(k=4, pad=1, stride 1|2) = im2col + MFMA GEMM; weight OIHW [Cout][Cin][4][4]."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        L = _L()
        x = _cl(x)
        n, cin, hi, wi = x.shape
        cout = weight.shape[0]
        ho, wo = _conv4x4_out(hi, stride), _conv4x4_out(wi, stride)
        m, k = n * ho * wo, 16 * cin
        st = _lib.stream_ptr()
        wg = torch.empty(cout, k, dtype=torch.float32, device=x.device)
        _lib.check(L.odvae_weight4x4_reorder_f32(weight.detach().contiguous().data_ptr(), wg.data_ptr(), cout, cin, 1, st), "weight4x4_reorder")
        cols = torch.empty(m, k, dtype=torch.float32, device=x.device)
        _lib.check(L.odvae_im2col4x4_f32(x.data_ptr(), cols.data_ptr(), n, hi, wi, cin, ho, wo, stride, st), "im2col4x4")
        y = _new_cl(n, cout, ho, wo, x)
        b = bias.detach().contiguous() if bias is not None else None
        gemm(0, 1, m, cout, k, 1.0, cols, k, 0, wg, k, 0, y, cout, 0, b)
        ctx.stride, ctx.has_bias, ctx.xshape = stride, bias is not None, (n, cin, hi, wi)
        ctx.save_for_backward(cols, wg)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        cols, wg = ctx.saved_tensors
        dy = _cl(dy)
        n, cin, hi, wi = ctx.xshape
        _, cout, ho, wo = dy.shape
        m, k = n * ho * wo, 16 * cin
        st = _lib.stream_ptr()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dcols = torch.empty(m, k, dtype=torch.float32, device=dy.device)
            gemm(0, 0, m, k, cout, 1.0, dy, cout, 0, wg, k, 0, dcols, k, 0)
            dx = _new_cl(n, cin, hi, wi, dy)
            _lib.check(L.odvae_col2im4x4_f32(dcols.data_ptr(), dx.data_ptr(), n, hi, wi, cin, ho, wo, ctx.stride, st), "col2im4x4")
        if ctx.needs_input_grad[1]:
            dwg = torch.empty(cout, k, dtype=torch.float32, device=dy.device)
            gemm(1, 0, cout, k, m, 1.0, dy, cout, 0, cols, k, 0, dwg, k, 0)
            dw = torch.empty(cout, cin, 4, 4, dtype=torch.float32, device=dy.device)
            _lib.check(L.odvae_weight4x4_reorder_f32(dwg.data_ptr(), dw.data_ptr(), cout, cin, 0, st), "weight4x4_reorder")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy, m, cout)
        return dx, dw, db, None


def conv4x4(x, weight, bias, stride):
    """PatchGAN convolution.  Cout = 1 (the logit head) is zero-padded to 4 output channels for the GEMM's
    float4 staging and sliced back (torch cat/slice on [4,C,4,4] / [B,4,30,30]-sized tensors)."""
    cout = weight.shape[0]
    if cout % 4 == 0:
        return _Conv4x4.apply(x, weight, bias, stride)
    pad = 4 - cout % 4
    wp = torch.cat([weight, weight.new_zeros((pad,) + tuple(weight.shape[1:]))], dim=0)
    bp = torch.cat([bias, bias.new_zeros(pad)]) if bias is not None else None
    return _Conv4x4.apply(x, wp, bp, stride)[:, :cout]


class _BatchNormLReLU(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, slope, train):
        L = _L()
        x = _cl(x)
        n, c, h, w = x.shape
        rows = n * h * w
        y = _new_cl(n, c, h, w, x)
        if train:
            mean = torch.empty(c, dtype=torch.float32, device=x.device)
            rstd = torch.empty(c, dtype=torch.float32, device=x.device)
        else:  # eval: running estimates (tiny [C] torch ops)
            mean = running_mean.detach().clone()
            rstd = torch.rsqrt(running_var.detach() + eps)
        wp, wn = _ws(L.odvae_batchnorm_workspace_bytes(rows, c), x)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        _lib.check(L.odvae_batchnorm_lrelu_fwd_f32(x.data_ptr(), rows, c, g.data_ptr(), b.data_ptr(), float(eps), float(momentum),
                                                   float(slope), int(train), mean.data_ptr(), rstd.data_ptr(),
                                                   _lib.ptr(running_mean) if train else None,
                                                   _lib.ptr(running_var) if train else None, y.data_ptr(), wp, wn,
                                                   _lib.stream_ptr()), "batchnorm_lrelu_fwd")
        ctx.slope, ctx.train = float(slope), int(train)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        dy = _cl(dy)
        n, c, h, w = x.shape
        rows = n * h * w
        dx = _new_cl(n, c, h, w, x)
        dg = torch.empty(c, dtype=torch.float32, device=x.device)
        db = torch.empty(c, dtype=torch.float32, device=x.device)
        wp, wn = _ws(L.odvae_batchnorm_workspace_bytes(rows, c), x)
        _lib.check(L.odvae_batchnorm_lrelu_bwd_f32(x.data_ptr(), dy.data_ptr(), rows, c, gamma.detach().contiguous().data_ptr(),
                                                   beta.detach().contiguous().data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                   ctx.slope, ctx.train, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), wp, wn,
                                                   _lib.stream_ptr()), "batchnorm_lrelu_bwd")
        return dx, dg, db, None, None, None, None, None, None


def batchnorm_lrelu(x, bn, slope):
    """BatchNorm2d `bn` (parameter/buffer holder) + LeakyReLU(slope); updates running stats in training mode."""
    train = bn.training or not bn.track_running_stats
    if train and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    return _BatchNormLReLU.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum, slope, train)


class _LeakyReLU(Function):
    @staticmethod
    def forward(ctx, x, slope):
        L = _L()
        x = _cl(x)
        y = _new_cl(*x.shape, x)
        _lib.check(L.odvae_leaky_relu_f32(x.data_ptr(), y.data_ptr(), float(slope), x.numel(), _lib.stream_ptr()), "leaky_relu")
        ctx.slope = float(slope)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        (x,) = ctx.saved_tensors
        dy = _cl(dy)
        dx = _new_cl(*x.shape, x)
        _lib.check(L.odvae_leaky_relu_bwd_f32(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), ctx.slope, x.numel(), _lib.stream_ptr()),
                   "leaky_relu_bwd")
        return dx, None


def leaky_relu(x, slope):
    return _LeakyReLU.apply(x, slope)


# ------------------------------------------------------------------------------------------------------
# LPIPS-style perceptual network pieces (lpips_f32.hip)
# ------------------------------------------------------------------------------------------------------
class _ScalingLayer(Function):
    @staticmethod
    def forward(ctx, x, shift, scale):
        L = _L()
        x = _cl(x)
        n, c, h, w = x.shape
        sh, sc = shift.detach().reshape(-1).contiguous(), scale.detach().reshape(-1).contiguous()
        y = _new_cl(n, c, h, w, x)
        _lib.check(L.odvae_scaling_layer_f32(x.data_ptr(), sh.data_ptr(), sc.data_ptr(), y.data_ptr(), n * h * w, c, 0,
                                             _lib.stream_ptr()), "scaling_layer")
        ctx.save_for_backward(sh, sc)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        sh, sc = ctx.saved_tensors
        dy = _cl(dy)
        n, c, h, w = dy.shape
        dx = _new_cl(n, c, h, w, dy)
        _lib.check(L.odvae_scaling_layer_f32(dy.data_ptr(), sh.data_ptr(), sc.data_ptr(), dx.data_ptr(), n * h * w, c, 1,
                                             _lib.stream_ptr()), "scaling_layer bwd")
        return dx, None, None


def scale_shift(x, shift, scale):
    return _ScalingLayer.apply(x, shift, scale)


class _MaxPool2x2(Function):
    @staticmethod
    def forward(ctx, x):
        L = _L()
        x = _cl(x)
        n, c, h, w = x.shape
        y = _new_cl(n, c, h // 2, w // 2, x)
        _lib.check(L.odvae_maxpool2x2_f32(x.data_ptr(), y.data_ptr(), n, h // 2, w // 2, c, _lib.stream_ptr()), "maxpool2x2")
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        x, y = ctx.saved_tensors
        dy = _cl(dy)
        n, c, h, w = x.shape
        dx = _new_cl(n, c, h, w, x)
        _lib.check(L.odvae_maxpool2x2_bwd_f32(x.data_ptr(), y.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, h // 2, w // 2, c,
                                              _lib.stream_ptr()), "maxpool2x2_bwd")
        return dx


def maxpool2x2(x):
    return _MaxPool2x2.apply(x)


class _LpipsDistance(Function):
    """[B] spatial mean of lin_w . (normalize(f0) - normalize(f1))^2; differentiable w.r.t. f1 (the reconstruction)."""

    @staticmethod
    def forward(ctx, f0, f1, lin_w):
        L = _L()
        f0, f1 = _cl(f0), _cl(f1)
        n, c, h, w = f0.shape
        wv = lin_w.detach().reshape(-1).contiguous()
        out = torch.empty(n, dtype=torch.float32, device=f0.device)
        wp, wn = _ws(n * 1024, f0)
        _lib.check(L.odvae_lpips_distance_f32(f0.data_ptr(), f1.data_ptr(), wv.data_ptr(), out.data_ptr(), n, h * w, c, wp, wn,
                                              _lib.stream_ptr()), "lpips_distance")
        ctx.save_for_backward(f0, f1, wv)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _L()
        f0, f1, wv = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("lpips_layer_distance: gradient w.r.t. the first (input) branch is not on the OD-VAE path")
        n, c, h, w = f1.shape
        df1 = None
        if ctx.needs_input_grad[1]:
            df1 = _new_cl(n, c, h, w, f1)
            _lib.check(L.odvae_lpips_distance_bwd_f32(f0.data_ptr(), f1.data_ptr(), wv.data_ptr(), g.contiguous().data_ptr(),
                                                      df1.data_ptr(), n, h * w, c, _lib.stream_ptr()), "lpips_distance_bwd")
        return None, df1, None


def lpips_layer_distance(f0, f1, lin_w):
    return _LpipsDistance.apply(f0, f1, lin_w)


# ------------------------------------------------------------------------------------------------------
# bf16 mixed-precision path (BASELINE.json configs[4]): bf16 activations in HBM, f32 master weights / gradients / statistics
# ------------------------------------------------------------------------------------------------------
_I31 = 0x7FFFFFF0


def _conv_b_raw(mode, x, pack, cout, bias, residual, out_f32, cin_alg=None, stats=False):
    """One odvae_conv_bf16 call.  mode 4 (1x1) flattens the pixels to [1][M/wi][wi]; images are processed in groups small enough
    for the 2 GiB buffer descriptors.  cin_alg: the reduction width the algorithm has (3 for conv_in, whose image is zero-padded to
    8 channels): the FLOP / byte figures handed to KERNEL_EVENTS are the direct-form work of SURVEY.md 8(d), not the padded work."""
    L = _L()
    n, cx, hi, wi = x.shape
    if mode in (0, 4):
        ho, wo = hi, wi
    elif mode == 1:
        ho, wo = hi // 2, wi // 2
    else:
        ho, wo = 2 * hi, 2 * wi
    y = _new_cl(n, cout, ho, wo, x, dtype=torch.float32 if out_f32 else BF16)
    esz = 4 if out_f32 else 2
    if mode != 4:
        tag = KERNEL_EVENTS.begin()
        partial = None
        if stats:     # (mode 0, bf16 output) the epilogue also leaves the GroupNorm statistics of y per output tile
            partial = torch.empty(n, L.odvae_conv_bf16_stats_chunks(ho, wo), GN_GROUPS, 2, dtype=torch.float32, device=x.device)
            _lib.check(L.odvae_conv_bf16_stats(x.data_ptr(), n, hi, wi, cx, pack.data_ptr(), cout, _lib.ptr(bias), _lib.ptr(residual),
                                               y.data_ptr(), partial.data_ptr(), GN_GROUPS, _lib.stream_ptr()), "conv_bf16_stats")
        else:
            _lib.check(L.odvae_conv_bf16(mode, x.data_ptr(), n, hi, wi, cx, pack.data_ptr(), cout, _lib.ptr(bias), _lib.ptr(residual),
                                         y.data_ptr(), ho, wo, int(out_f32), _lib.stream_ptr()), "conv_bf16(mode=%d)" % mode)
        ca = cin_alg or cx
        px = hi * wi if mode == 3 else ho * wo     # mode 3 (data gradient of the stride-2 conv): nine taps per LOW-res pixel
        KERNEL_EVENTS.end("conv_bf16", 2.0 * 9 * ca * cout * n * px, tag,
                          2.0 * n * hi * wi * ca + esz * n * ho * wo * cout * (2 if residual is not None else 1) + 2.0 * 9 * ca * cout)
        return (y, partial) if stats else y
    per = hi * wi
    grp = max(1, min(n, _I31 // max(per * cx * 2, per * cout * esz)))
    for a in range(0, n, grp):
        b = min(n, a + grp)
        m = (b - a) * per
        w16 = next(d for d in (16, 8, 4, 2, 1) if m % d == 0)
        tag = KERNEL_EVENTS.begin(secondary=True)
        _lib.check(L.odvae_conv_bf16(4, x.data_ptr() + a * per * cx * 2, 1, m // w16, w16, cx, pack.data_ptr(), cout, _lib.ptr(bias),
                                     None if residual is None else residual.data_ptr() + a * per * cout * 2,
                                     y.data_ptr() + a * per * cout * esz, m // w16, w16, int(out_f32), _lib.stream_ptr()), "conv_bf16(1x1)")
        KERNEL_EVENTS.end("conv1x1_bf16", 2.0 * cx * cout * m, tag, 2.0 * m * cx + esz * m * cout)
    return y


def _pad8(c):
    return (c + 7) // 8 * 8


class _ConvB(Function):
    """3x3 (modes 0 / 1 / 2) or 1x1 (mode 4) convolution on bf16 NHWC activations; weight / bias are the f32 master parameters
    (OIHW), their gradients come back in f32.  x may carry zero channels beyond weight.shape[1] (the 3-channel image padded to 8)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, mode, out_f32, gn_stats=False):
        """gn_stats=True: returns (y, partials) -- partials [N][tiles][32][2], the GroupNorm statistics of y from the conv's own epilogue
        (not differentiable), or None where the kernel does not offer them."""
        L = _L()
        x = _cl(x, BF16)
        res = _cl(residual, BF16) if residual is not None else None
        cout, cin = weight.shape[0], weight.shape[1]
        if x.shape[1] != cin and L.odvae_conv_bf16_reduce_pad(x.shape[1]) != L.odvae_conv_bf16_reduce_pad(cin):
            raise ValueError("conv_bf16: input has %d channels, weight expects %d" % (x.shape[1], cin))
        fwd_pack, dpack = pack_conv3x3(weight, True, bool(ctx.needs_input_grad[0]), "bf16")
        b = bias.detach().contiguous() if bias is not None else None
        partial = None
        if gn_stats and mode == 0 and not out_f32 and GN_FUSED_STATS and L.odvae_conv_bf16_stats_supported(cout, GN_GROUPS):
            y, partial = _conv_b_raw(mode, x, fwd_pack, cout, b, res, out_f32, cin_alg=cin, stats=True)
        else:
            y = _conv_b_raw(mode, x, fwd_pack, cout, b, res, out_f32, cin_alg=cin)
        ctx.mode, ctx.has_bias, ctx.has_res, ctx.dpack = mode, bias is not None, residual is not None, dpack
        ctx.pack_epoch = PACK_CACHE.epoch
        ctx.wshape = tuple(weight.shape)
        ctx.save_for_backward(x)
        if gn_stats:
            if partial is not None:
                ctx.mark_non_differentiable(partial)
            ctx.set_materialize_grads(False)   # (as _Conv3x3: no tensor of zeros for the statistics output)
            return y, partial
        return y

    @staticmethod
    def backward(ctx, dy, _dpartial=None):
        L = _L()
        if dy is None:
            return None, None, None, None, None, None, None
        (x,) = ctx.saved_tensors
        mode = ctx.mode
        cout, cin = ctx.wshape[0], ctx.wshape[1]
        n, cx, hi, wi = x.shape
        _, _, ho, wo = dy.shape
        dyb, cp = dy, cout
        if dy.dtype != BF16:                 # f32 upstream gradient (reconstruction / moments): one cast (+ channel pad) pass
            cp = _pad8(cout)
            dyb = cast_pad_bf16(_cl(dy), cp)
        else:
            dyb = _cl(dy, BF16)
        dx = dw = db = None
        if ctx.needs_input_grad[0] and not WEIGHT_GRADIENT_ONLY:
            if cx != cin:
                raise NotImplementedError("data gradient through a channel-padded input")
            PACK_CACHE.check_epoch(ctx.pack_epoch, "conv_bf16 backward")
            if mode == 0:
                dx = _conv_b_raw(0, dyb, ctx.dpack, cin, None, None, False)
            elif mode == 1:
                dx = _conv_b_raw(3, dyb, ctx.dpack, cin, None, None, False)
            elif mode == 4:
                dx = _conv_b_raw(4, dyb, ctx.dpack, cin, None, None, False)
            else:                            # Upsample: gradient w.r.t. the upsampled image, then its 2x2 sum-pool
                du = _conv_b_raw(0, dyb, ctx.dpack, cin, None, None, False)
                dx = _new_cl(n, cin, hi, wi, x, dtype=BF16)
                _lib.check(L.odvae_upsample2x_bwd_bf16(du.data_ptr(), dx.data_ptr(), n, hi, wi, cin, _lib.stream_ptr()), "upsample2x_bwd_bf16")
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        fold_db = want_db and dy.dtype == BF16 and ctx.needs_input_grad[1]   # bias gradient rides in the weight-gradient pass over dy
        if ctx.needs_input_grad[1]:
            taps = ctx.wshape[2] * ctx.wshape[3]
            dwf = torch.empty((cp, cx) + tuple(ctx.wshape[2:]), dtype=torch.float32, device=x.device)
            dbf = torch.empty(cp, dtype=torch.float32, device=x.device) if fold_db else None
            def _wgrad(xg, dyg, geo, ng, dw_out, db_out):
                need = L.odvae_conv_wgrad_bf16_workspace_bytes(mode, geo[0], geo[4], geo[5], cx, cp)
                wp, wn = _ws(need, x)
                tag = KERNEL_EVENTS.begin(secondary=True)
                _lib.check(L.odvae_conv_wgrad_bf16(mode, xg.data_ptr(), dyg.data_ptr(), *geo, dw_out.data_ptr(), _lib.ptr(db_out), wp, wn,
                                                   _lib.stream_ptr()), "conv_wgrad_bf16(mode=%d)" % mode)
                KERNEL_EVENTS.end("conv_wgrad_bf16", 2.0 * taps * cx * cp * ng * ho * wo, tag, 2.0 * ng * (hi * wi * cx + ho * wo * cp))

            if mode == 4:
                # a 1x1 conv is one [pixels x cx]^T [pixels x cp] product; the kernel addresses with 32-bit offsets, so image groups
                # of under 2 GiB are reduced one after the other (the same split the forward / data-gradient take in _conv_b_raw)
                per = hi * wi
                grp = max(1, min(n, WGRAD_1X1_GROUP or _I31 // (per * max(cx, cp) * 2)))
                for g0 in range(0, n, grp):
                    ng = min(grp, n - g0)
                    m = ng * per
                    w16 = next(d for d in (16, 8, 4, 2, 1) if m % d == 0)
                    geo = (1, m // w16, w16, cx, m // w16, w16, cp)
                    if g0 == 0:
                        _wgrad(x[g0:g0 + ng], dyb[g0:g0 + ng], geo, ng, dwf, dbf)
                    else:
                        dw_part = torch.empty_like(dwf)
                        db_part = torch.empty_like(dbf) if fold_db else None
                        _wgrad(x[g0:g0 + ng], dyb[g0:g0 + ng], geo, ng, dw_part, db_part)
                        dwf += dw_part
                        if fold_db:
                            dbf += db_part
            else:
                _wgrad(x, dyb, (n, hi, wi, cx, ho, wo, cp), n, dwf, dbf)
            dw = dwf if (cp == cout and cx == cin) else dwf[:cout, :cin].contiguous()
            if fold_db:
                db = dbf if cp == cout else dbf[:cout].contiguous()
        if want_db and not fold_db:
            if dy.dtype != BF16:   # f32 upstream gradient (reconstruction / moments): sum the f32 values themselves
                db = _colsum(_cl(dy), n * ho * wo, cout)
            else:
                db = torch.empty(cout, dtype=torch.float32, device=x.device)
                rows = n * ho * wo
                wp, wn = _ws(L.odvae_colsum_bf16_workspace_bytes(rows, cout), x)
                _lib.check(L.odvae_colsum_bf16(dyb.data_ptr(), rows, cout, db.data_ptr(), wp, wn, _lib.stream_ptr()), "colsum_bf16")
        dres = dyb if ctx.has_res and ctx.needs_input_grad[3] else None
        return dx, dw, db, dres, None, None, None


WGRAD_1X1_GROUP = 0   # tests: force the image-group split of the 1x1 weight gradient at small sizes (0 = only past 2 GiB)


def cast_pad_bf16(x, cp=None):
    """bf16 NHWC copy of an f32 NHWC tensor, channels zero-padded to cp (raw, no autograd)."""
    L = _L()
    n, c, h, w = x.shape
    cp = c if cp is None else cp
    y = _new_cl(n, cp, h, w, x, dtype=BF16)
    _lib.check(L.odvae_cast_pad_bf16(x.data_ptr(), n * h * w, c, cp, y.data_ptr(), _lib.stream_ptr()), "cast_pad_bf16")
    return y


class _ToBF16(Function):
    """f32 -> bf16 hand-off into the mixed-precision network (optionally zero-padding the channels to a multiple of 8); the
    gradient comes back as f32 on the original channels."""

    @staticmethod
    def forward(ctx, x, cp):
        x = _cl(x)
        ctx.c = x.shape[1]
        return cast_pad_bf16(x, cp)

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        dy = _cl(dy, BF16)
        n, cp, h, w = dy.shape
        dx = _new_cl(n, cp, h, w, dy)
        _lib.check(L.odvae_cast_f32_from_bf16(dy.data_ptr(), dy.numel(), dx.data_ptr(), _lib.stream_ptr()), "cast_f32_from_bf16")
        return (dx if cp == ctx.c else dx[:, :ctx.c]), None


def to_bf16(x, pad_channels_to=None):
    if x.dtype == BF16:
        return x
    c = x.shape[1]
    cp = c if pad_channels_to is None else max(c, (c + pad_channels_to - 1) // pad_channels_to * pad_channels_to)
    return _ToBF16.apply(x, cp)


class _GroupNormB(Function):
    """GroupNorm(+swish) on bf16 activations: statistics and arithmetic in f32, one rounding on the way out."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, swish, with_skip=False, partials=None):
        L = _L()
        x = _cl(x, BF16)
        n, c, h, w = x.shape
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y = _new_cl(n, c, h, w, x, dtype=BF16)
        mean = torch.empty(n, groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty(n, groups, dtype=torch.float32, device=x.device)
        tag = KERNEL_EVENTS.begin(secondary=True)
        if partials is not None and partials.shape[0] == n and partials.shape[2] == groups:
            _lib.check(L.odvae_groupnorm_fwd_partials_bf16(x.data_ptr(), n, h * w, c, groups, g.data_ptr(), b.data_ptr(), float(eps), int(swish),
                                                           y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), partials.data_ptr(),
                                                           int(partials.shape[1]), _lib.stream_ptr()), "groupnorm_fwd_partials_bf16")
        else:
            wp, wn = _ws(L.odvae_groupnorm_bf16_workspace_bytes(n, h * w, c, groups), x)
            _lib.check(L.odvae_groupnorm_fwd_bf16(x.data_ptr(), n, h * w, c, groups, g.data_ptr(), b.data_ptr(), float(eps), int(swish),
                                                  y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), wp, wn, _lib.stream_ptr()), "groupnorm_fwd_bf16")
        KERNEL_EVENTS.end("groupnorm", 0.0, tag, 2.0 * 2 * n * h * w * c, issued=0.0)
        ctx.groups, ctx.swish = groups, int(swish)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.set_materialize_grads(False)
        if with_skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        L = _L()
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dskip, None, None, None, None, None, None, None
        dy = _cl(dy, BF16)
        if dskip is not None:
            dskip = _cl(dskip, BF16)
        n, c, h, w = x.shape
        dx = _new_cl(n, c, h, w, x, dtype=BF16)
        dg = torch.empty(c, dtype=torch.float32, device=x.device)
        db = torch.empty(c, dtype=torch.float32, device=x.device)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        wp, wn = _ws(L.odvae_groupnorm_bf16_workspace_bytes(n, h * w, c, ctx.groups), x)
        tag = KERNEL_EVENTS.begin(secondary=True)
        _lib.check(L.odvae_groupnorm_bwd_bf16(x.data_ptr(), dy.data_ptr(), n, h * w, c, ctx.groups, g.data_ptr(), b.data_ptr(),
                                              mean.data_ptr(), rstd.data_ptr(), ctx.swish, dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                              _lib.ptr(dskip), wp, wn, _lib.stream_ptr()), "groupnorm_bwd_bf16")
        KERNEL_EVENTS.end("groupnorm", 0.0, tag, 2.0 * (3 + (dskip is not None)) * n * h * w * c, issued=0.0)
        return dx, dg, db, None, None, None, None, None


class _FlashAttention(Function):
    """softmax(q k^T C^-1/2) v from the packed bf16 projection [N, 3C, H, W]; the T x T scores stay on the CU (flash_attn_bf16.hip);
    only o and the per-row log-sum-exp are kept for the backward."""

    @staticmethod
    def forward(ctx, qkv):
        L = _L()
        qkv = _cl(qkv, BF16)
        n, c3, h, w = qkv.shape
        c, t = c3 // 3, h * w
        if not L.odvae_flash_attn_supported(n, t, c):
            raise _lib.HipLibraryError("flash attention: unsupported shape N=%d T=%d C=%d" % (n, t, c))
        o = _new_cl(n, c, h, w, qkv, dtype=BF16)
        lse = torch.empty(n, t, dtype=torch.float32, device=qkv.device)
        tag = KERNEL_EVENTS.begin(secondary=True)
        _lib.check(L.odvae_flash_attn_fwd_bf16(qkv.data_ptr(), n, t, c, float(c) ** -0.5, o.data_ptr(), lse.data_ptr(), _lib.stream_ptr()),
                   "flash_attn_fwd")
        KERNEL_EVENTS.end("flash_attn", 4.0 * t * t * c * n, tag, 2.0 * n * t * 4 * c)
        ctx.save_for_backward(qkv, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        L = _L()
        qkv, o, lse = ctx.saved_tensors
        do = _cl(do, BF16)
        n, c3, h, w = qkv.shape
        c, t = c3 // 3, h * w
        dqkv = _new_cl(n, c3, h, w, qkv, dtype=BF16)
        delta = torch.empty(n * t, dtype=torch.float32, device=qkv.device)
        tag = KERNEL_EVENTS.begin(secondary=True)
        _lib.check(L.odvae_flash_attn_bwd_bf16(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), n, t, c, float(c) ** -0.5,
                                               dqkv.data_ptr(), delta.data_ptr(), _lib.stream_ptr()), "flash_attn_bwd")
        # algorithmic: the five products of the backward (S, dP, dV, dK, dQ); issued: seven (S and dP are formed in both kernels)
        KERNEL_EVENTS.end("flash_attn", 10.0 * t * t * c * n, tag, 2.0 * n * t * 8 * c, issued=14.0 * t * t * c * n)
        return dqkv


# ------------------------------------------------------------------------------------------------------
# pose head: every loss term in one launch (pose_f32.hip)
# ------------------------------------------------------------------------------------------------------
class _PoseLosses(Function):
    """out[9] = pose, class, bbox, fill, kl_bbox losses + the logged per-component means (t1, t2, t3, v3) of
    src/modules/losses/contperceptual.py:111-132,176-212.  Differentiable w.r.t. dec_pose and the box-posterior moments."""

    @staticmethod
    def forward(ctx, dec_pose, moments, pose_gt, bbox_gt, fill_gt, class_gt, prior, prior_idx, bg_idx, l2, yaw, gamma, alpha):
        L = _L()
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        dec_pose, moments, pose_gt, bbox_gt, fill_gt, prior = (f32(t) for t in (dec_pose, moments, pose_gt, bbox_gt, fill_gt, prior))
        _lib.require_device(dec_pose, moments, pose_gt, bbox_gt, fill_gt, prior)
        class_gt = class_gt.detach().to(torch.int64).contiguous()
        prior_idx = prior_idx.detach().to(torch.int32).contiguous()
        b, w = dec_pose.shape
        nc = w - 8
        if moments.shape != (b, 16) or pose_gt.shape != (b, 4) or bbox_gt.shape != (b, 3) or fill_gt.numel() != b or prior.shape[1:] != (3, 8):
            raise ValueError("pose_losses: unexpected shapes %s %s %s %s %s %s" % tuple(tuple(t.shape) for t in (dec_pose, moments, pose_gt, bbox_gt, fill_gt, prior)))
        out = torch.empty(9, dtype=torch.float32, device=dec_pose.device)
        jac_pose = torch.empty(4, b, w, dtype=torch.float32, device=dec_pose.device)
        jac_mom = torch.empty(b, 16, dtype=torch.float32, device=dec_pose.device)
        _lib.check(L.odvae_pose_losses_f32(dec_pose.data_ptr(), pose_gt.data_ptr(), bbox_gt.data_ptr(), fill_gt.data_ptr(), class_gt.data_ptr(),
                                           moments.data_ptr(), prior.data_ptr(), prior_idx.data_ptr(), b, nc, prior.shape[0], int(bg_idx), int(l2),
                                           int(yaw), float(gamma), float(alpha), out.data_ptr(), jac_pose.data_ptr(), jac_mom.data_ptr(),
                                           _lib.stream_ptr()), "pose_losses")
        ctx.save_for_backward(jac_pose, jac_mom)
        ctx.dims = (b, nc)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _L()
        jac_pose, jac_mom = ctx.saved_tensors
        b, nc = ctx.dims
        g = g.contiguous()
        d_pose = torch.empty(b, 8 + nc, dtype=torch.float32, device=g.device)
        d_mom = torch.empty(b, 16, dtype=torch.float32, device=g.device)
        _lib.check(L.odvae_pose_losses_bwd_f32(g.data_ptr(), jac_pose.data_ptr(), jac_mom.data_ptr(), b, nc, d_pose.data_ptr(), d_mom.data_ptr(),
                                               _lib.stream_ptr()), "pose_losses_bwd")
        return (d_pose, d_mom) + (None,) * 11


def pose_losses(dec_pose, moments, pose_gt, bbox_gt, fill_gt, class_gt, prior, prior_idx, bg_idx=1, l2=False, yaw=True, gamma=2.0, alpha=0.25):
    return _PoseLosses.apply(dec_pose, moments, pose_gt, bbox_gt, fill_gt, class_gt, prior, prior_idx, bg_idx, l2, yaw, gamma, alpha)


# ------------------------------------------------------------------------------------------------------
# pose head MLPs: skinny dense layers (linear_f32.hip)
# ------------------------------------------------------------------------------------------------------
LINEAR_ACTS = {None: 0, "none": 0, "tanh": 1, "swish": 2, "silu": 2, "relu": 3}


class _LinearAct(Function):
    """y = act(x . W^T + b) for the batch-sized rows of the pose MLPs (pose_encoder.py:59-131, pose_decoder.py:60-97); the
    epilogue kernel fuses the split-K sum, the bias and the activation and keeps the pre-activation for the backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        L = _L()
        x2 = x.detach().to(torch.float32).reshape(-1, x.shape[-1]).contiguous()
        w = weight.detach().contiguous()
        b = None if bias is None else bias.detach().contiguous()
        _lib.require_device(x2, w, *(() if b is None else (b,)))
        m, k = x2.shape
        n = w.shape[0]
        if w.shape[1] != k or (b is not None and b.numel() != n):
            raise ValueError("linear_act: x %s, weight %s, bias %s" % (tuple(x.shape), tuple(w.shape), None if b is None else tuple(b.shape)))
        y = torch.empty(m, n, dtype=torch.float32, device=x2.device)
        pre = torch.empty_like(y) if act else None
        wp, wn = _ws(L.odvae_linear_workspace_bytes(m, n, k), y)
        _lib.check(L.odvae_linear_fwd_f32(x2.data_ptr(), w.data_ptr(), _lib.ptr(b), m, n, k, act, y.data_ptr(), _lib.ptr(pre), wp, wn,
                                          _lib.stream_ptr()), "linear_fwd")
        ctx.save_for_backward(x2, w, pre)
        ctx.act, ctx.has_bias, ctx.x_shape = act, b is not None, x.shape
        return y.reshape(x.shape[:-1] + (n,))

    @staticmethod
    def backward(ctx, dy):
        L = _L()
        x2, w, pre = ctx.saved_tensors
        m, k = x2.shape
        n = w.shape[0]
        dy = dy.to(torch.float32).reshape(m, n).contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dpre = torch.empty_like(dy)
        dx = torch.empty(m, k, dtype=torch.float32, device=dy.device) if need_x else None
        dw = torch.empty(n, k, dtype=torch.float32, device=dy.device) if need_w else None
        wp, wn = _ws(L.odvae_linear_workspace_bytes(m, n, k), dy)
        _lib.check(L.odvae_linear_bwd_f32(x2.data_ptr(), w.data_ptr(), _lib.ptr(pre), dy.data_ptr(), m, n, k, ctx.act, dpre.data_ptr(),
                                          _lib.ptr(dx), _lib.ptr(dw), wp, wn, _lib.stream_ptr()), "linear_bwd")
        db = _colsum(dpre, m, n) if need_b else None
        return (dx.reshape(ctx.x_shape) if need_x else None), dw, db, None


def linear_act(x, weight, bias=None, act=None):
    """act(x @ weight.T + bias) on the small-batch MFMA kernel; act in LINEAR_ACTS."""
    return _LinearAct.apply(x, weight, bias, LINEAR_ACTS[act])
