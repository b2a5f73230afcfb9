"""AdamW over a flat parameter arena (one fused HIP kernel per step).

Replaces torch.optim.AdamW(params, lr, fused=True) as built at /root/reference/trainer.py:33-38 with the
same defaults (betas (0.9, 0.999), eps 1e-8, decoupled weight_decay 0.01 on EVERY parameter, PReLU slopes
and biases included).  MI355X-first layout: all parameters of a model live in ONE contiguous float32
buffer and all their gradients in another, so
  * zero_grad is one memset, the optimizer step is one kernel (fsr_adamw_step),
  * the data-parallel gradient exchange is ONE RCCL all-reduce of the gradient arena (no bucketing copies),
  * conv weight-gradient kernels accumulate straight into their slice of the gradient arena.
Parameters keep their identity (nn.Parameter objects, names, shapes); only their storage is re-pointed.
"""
import torch

from . import _lib as L
from .ops import _p, _stream


class ArenaAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, fused=True):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._params = params
        total = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.float32, device=dev)   # device-side step counter (graph-replay safe)
        self._epoch = 0
        self.grad_scale = 1.0  # set to 1/world_size when gradients are SUM all-reduced
        # Dynamic loss scaling (fp16 mode, Trainer): device float[4] {scale, clean iterations, non-finite flag, skipped
        # iterations}.  When set, step() first checks the gradient arena for inf / NaN and the update divides by the scale on
        # the device -- or does nothing at all while the flag is up (a single overflow would otherwise poison exp_avg,
        # exp_avg_sq and the parameters for good, silently, under hipGraph replay).
        self.scale_state = None
        off = 0
        self._slices = []
        for p in params:
            n = p.numel()
            if p.dtype != torch.float32:
                raise L.FsrError("ArenaAdamW: parameters must be float32")
            self.flat_param[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat_param[off:off + n].view(p.shape)
            p.grad = self.flat_grad[off:off + n].view(p.shape)
            if (p.dim() == 4 and p.shape[2:] == (3, 3)) or p.dim() == 1:
                p._fsr_grad = p.grad  # conv3x3 weight-gradient kernels accumulate here directly (weights; first-layer biases)
            self._slices.append((off, n))
            off += n

    def zero_grad(self, set_to_none=False):
        """One memset.  Gradients are never set to None: they are permanent views of the arena."""
        self.flat_grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise L.FsrError("ArenaAdamW does not take a closure")
        g = self.param_groups[0]
        self._epoch += 1
        if self.scale_state is not None:
            L.check(L.lib().fsr_grad_nonfinite(_p(self.flat_grad), self.flat_grad.numel(), _p(self.scale_state), _stream()),
                    "fsr_grad_nonfinite")
            L.check(L.lib().fsr_adamw_step_scaled(_p(self.flat_param), _p(self.flat_grad), _p(self.exp_avg), _p(self.exp_avg_sq),
                                                  self.flat_param.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                                  float(g["eps"]), float(g["weight_decay"]), _p(self.step_dev),
                                                  float(self.grad_scale), _p(self.scale_state), _stream()), "fsr_adamw_step_scaled")
        else:
            L.check(L.lib().fsr_adamw_step(_p(self.flat_param), _p(self.flat_grad), _p(self.exp_avg), _p(self.exp_avg_sq),
                                           self.flat_param.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                           float(g["eps"]), float(g["weight_decay"]), _p(self.step_dev), float(self.grad_scale),
                                           _stream()), "fsr_adamw_step")
        # the kernel writes through raw pointers, so torch's version counters do not move: publish an
        # epoch the packed-filter cache (ops.packed_filter) keys on instead
        for p in self._params:
            p._fsr_epoch = self._epoch

    def mark_updated(self):
        """Parameters were modified behind torch's back (a replayed hipGraph ran fsr_adamw_step): invalidate caches
        keyed on them (ops.packed_filter)."""
        self._epoch += 1
        for p in self._params:
            p._fsr_epoch = self._epoch

    def state_dict(self):
        """torch.optim.AdamW-shaped state (per-parameter step / exp_avg / exp_avg_sq), as trainer.py:149-156 saves."""
        state = {}
        for i, (off, n) in enumerate(self._slices):
            shape = self._params[i].shape
            state[i] = {"step": self.step_dev.detach().cpu().reshape(()).clone(),
                        "exp_avg": self.exp_avg[off:off + n].view(shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view(shape).clone()}
        g = dict(self.param_groups[0])
        g["params"] = list(range(len(self._params)))
        sd = {"state": state, "param_groups": [g]}
        if self.scale_state is not None:
            # extra top-level key (torch.optim.Optimizer.load_state_dict reads only "state" / "param_groups", so reference-format
            # loaders ignore it): {scale, clean iterations, non-finite flag, skipped iterations} of the dynamic fp16 loss scale --
            # a resumed run continues at the scale it had backed off to instead of overflowing its way down again
            sd["fsr_loss_scale_state"] = self.scale_state.detach().cpu().clone()
        return sd

    def load_state_dict(self, sd):
        for i, (off, n) in enumerate(self._slices):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_dev.fill_(float(st["step"]))
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in sd["param_groups"][0]:
                self.param_groups[0][k] = sd["param_groups"][0][k]
        saved = sd.get("fsr_loss_scale_state")
        if saved is not None and self.scale_state is not None:
            v = saved.to(torch.float32).reshape(-1).clone()
            v[2] = 0.0                                   # never resume with the non-finite flag up
            self.scale_state.copy_(v)
