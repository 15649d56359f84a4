"""Adam over flat HBM arenas: one kernel launch per optimizer step, one for the global grad norm.

Semantics are torch.optim.Adam(lr, betas=(0.5, 0.9)) as configured in src/models/autoencoder.py:365-377 plus the
trainer's gradient_clip_val (configs/autoencoder/pose/autoencoder_kl_16x16x16.yaml:140, clip_grad_norm_).
Parameters and gradients are re-pointed at views of two contiguous fp32 buffers (71 M floats = 284 MB each for
optimizer 0), so the data-parallel reducer (parallel.py) can all-reduce contiguous buckets in place and the
optimizer touches each byte exactly once.  The step itself needs a HIP device (no CPU fallback).
"""
import torch

from . import lib as _lib

_ALIGN = 64  # floats; keeps every parameter view 256-byte aligned (float4 kernels, GEMM operands)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        # weight_decay / amsgrad are carried (at the values the reference runs with) so that the saved param_groups read
        # like torch.optim.Adam's; nothing else is implemented
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._flat = None
        self._steps = 0
        self._clip = None       # device [norm, coef] written by clip_grad_norm_
        self._clip_armed = False

    # ---- arenas -------------------------------------------------------------------------------------------------
    def _params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def materialize(self):
        """Move parameters and gradients into the flat arenas (idempotent).  Call after model.to(device)."""
        if self._flat is not None:
            return self._flat
        ps = self._params()
        if not ps:
            raise ValueError("FusedAdam got no parameters")
        dev = ps[0].device
        if dev.type != "cuda":
            raise _lib.HipLibraryError("FusedAdam needs parameters on a HIP device (no CPU fallback); got %s" % dev)
        had_grad = [p.grad is not None for p in ps]
        offs, total = [], 0
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FusedAdam: all parameters must be float32 on %s" % dev)
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(ps, offs):
            flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat_p[o:o + p.numel()].view(p.shape)
            g = flat_g[o:o + p.numel()].view(p.shape)
            if p.grad is not None:
                g.copy_(p.grad)
            p.grad = g
        self._flat = dict(p=flat_p, g=flat_g, m=torch.zeros_like(flat_p), v=torch.zeros_like(flat_p),
                          offsets=offs, total=total, params=ps,
                          gviews=[flat_g[o:o + p.numel()].view(p.shape) for p, o in zip(ps, offs)],
                          gptrs=[flat_g.data_ptr() + 4 * o for o in offs])
        self._clip = torch.ones(2, dtype=torch.float32, device=dev)
        # torch.optim.Adam keeps a step count per parameter and skips parameters without a gradient (e.g. the
        # decoder during encoder pre-training); hooks record which parameters a backward actually reached
        self._counts = [0] * len(ps)
        self._touched = {i for i, p in enumerate(ps) if had_grad[i]}
        for i, p in enumerate(ps):
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self._touched.add(i))
        return self._flat

    @property
    def flat_grad(self):
        return self.materialize()["g"]

    def param_slices(self):
        """[(param, offset, numel)] in arena order (used by the gradient bucketing in parallel.py)."""
        f = self.materialize()
        return [(p, o, p.numel()) for p, o in zip(f["params"], f["offsets"])]

    def gather_grads(self, indices=None):
        """Bring the gradients of parameters `indices` (default: all) into the arena and re-point `.grad` at their arena views.
        After `zero_grad(set_to_none=True)` autograd hands each parameter the tensor the weight-gradient kernel produced (no
        accumulate kernel per parameter); those tensors are copied into the arena by ONE multi-tensor copy here, slices of
        parameters the backward did not reach are zeroed (the norm kernel reads the whole arena).  Gradients that already live
        in the arena are left alone."""
        f = self._flat
        views, ptrs, params = f["gviews"], f["gptrs"], f["params"]      # cached: this loop runs once per step on the host
        dst, src, zero = [], [], []
        for i in (range(len(params)) if indices is None else indices):
            p = params[i]
            g = p.grad
            if g is None:
                zero.append(views[i])
                p.grad = views[i]
            elif g.data_ptr() != ptrs[i]:
                dst.append(views[i])
                src.append(g)
                p.grad = views[i]
        with torch.no_grad():
            if dst:
                torch._foreach_copy_(dst, src)
            if zero:
                torch._foreach_zero_(zero)

    def _ensure_grad_views(self):
        self.gather_grads()

    def zero_grad(self, set_to_none=False):
        """set_to_none=False: gradients stay views of the arena, zeroing is one memset.  set_to_none=True (what the trainer
        uses): `.grad` is dropped, the next backward's gradients are gathered into the arena by `gather_grads`."""
        if self._flat is None:
            return super().zero_grad(set_to_none=set_to_none)
        if set_to_none:
            for p in self._flat["params"]:
                p.grad = None
            return
        self._flat["g"].zero_()
        for p, o in zip(self._flat["params"], self._flat["offsets"]):
            if p.grad is None or p.grad.data_ptr() != self._flat["g"][o:o + 1].data_ptr():
                p.grad = self._flat["g"][o:o + p.numel()].view(p.shape)

    # ---- clip + step ----------------------------------------------------------------------------------------------
    def clip_grad_norm_(self, max_norm):
        """Global L2 norm over this optimizer's gradients; the clip coefficient stays on the device and is applied
        inside the Adam kernel (no extra pass over the gradients, no host sync).  Returns the device norm tensor."""
        f = self.materialize()
        self._ensure_grad_views()
        L = _lib.load()
        wp, wn = _lib.workspace.get(8192, f["g"].device)
        _lib.check(L.odvae_grad_norm_f32(f["g"].data_ptr(), f["total"], float(max_norm), self._clip.data_ptr(), wp, wn,
                                         _lib.stream_ptr()), "grad_norm")
        self._clip_armed = True
        return self._clip[0]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:   # PL's automatic optimisation hands training_step + backward in as the closure
            with torch.enable_grad():
                loss = closure()
        f = self.materialize()
        self._ensure_grad_views()
        g0 = self.param_groups[0]
        self._steps += 1
        L = _lib.load()
        # contiguous arena runs of touched parameters that share a step count -> one launch each (1-6 in practice)
        runs, cur = [], None
        n_params = len(f["params"])
        for i in range(n_params):
            if i in self._touched:
                self._counts[i] += 1
                end = f["offsets"][i + 1] if i + 1 < n_params else f["total"]
                if cur is not None and cur[2] == self._counts[i] and cur[1] == f["offsets"][i]:
                    cur[1] = end
                else:
                    cur = [f["offsets"][i], end, self._counts[i]]
                    runs.append(cur)
        clip = self._clip.data_ptr() if self._clip_armed else None
        for beg, end, count in runs:
            _lib.check(L.odvae_adam_step_f32(f["p"].data_ptr() + 4 * beg, f["g"].data_ptr() + 4 * beg,
                                             f["m"].data_ptr() + 4 * beg, f["v"].data_ptr() + 4 * beg, end - beg,
                                             float(g0["lr"]), float(g0["betas"][0]), float(g0["betas"][1]),
                                             float(g0["eps"]), count, clip, _lib.stream_ptr()), "adam_step")
        self._touched.clear()
        self._clip_armed = False
        from . import ops
        ops.PACK_CACHE.bump()   # parameters changed under torch's version counters: conv weight packs are stale
        return loss


    # ---- checkpoint interchange: torch.optim.Adam's state layout ------------------------------------------------------------
    def state_dict(self):
        """{"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]} exactly as torch.optim.Adam writes it
        (Lightning stores this under `optimizer_states`), so a resume -- here or under the reference's Adam -- keeps the
        moments and the bias-correction step.  Parameters that never received a gradient carry no state, as in torch."""
        if self._flat is None:
            return super().state_dict()
        f = self._flat
        self.state.clear()
        for i, (p, o) in enumerate(zip(f["params"], f["offsets"])):
            if self._counts[i] > 0:
                n = p.numel()
                self.state[p] = {"step": torch.tensor(float(self._counts[i])),
                                 "exp_avg": f["m"][o:o + n].view(p.shape).clone(),
                                 "exp_avg_sq": f["v"][o:o + n].view(p.shape).clone()}
        out = super().state_dict()
        self.state.clear()   # the arenas stay the only live copy
        return out

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)   # torch maps indices -> parameters and casts the moments to their device
        f = self.materialize()
        f["m"].zero_()
        f["v"].zero_()
        for i, (p, o) in enumerate(zip(f["params"], f["offsets"])):
            st = self.state.get(p)
            self._counts[i] = 0
            if st:
                n = p.numel()
                f["m"][o:o + n].copy_(st["exp_avg"].reshape(-1))
                f["v"][o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                self._counts[i] = int(round(float(st["step"])))
        self._steps = max(self._counts) if self._counts else 0
        self.state.clear()


def make_adam(params, lr, betas):
    return FusedAdam(params, lr=lr, betas=betas)
