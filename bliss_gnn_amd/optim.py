"""``th.optim.Adam(self.parameters(), lr)`` (train_lightning.py:205-206) as ONE gfx950 launch per step (csrc/optim.hip) for the
bf16 module the reference trains (:596, :607: parameters, gradients, moments all bf16).  Step count and learning rate live
on the device: the step can be recorded into a HIP graph and ``StepLR`` (:208) still takes effect on replay."""
import ctypes as C

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise NotImplementedError("one parameter group, like the reference's optimiser (train_lightning.py:206)")
        ps = self.param_groups[0]["params"]
        if not ps or any(p.dtype != torch.bfloat16 or not p.is_cuda or not p.is_contiguous() for p in ps):
            raise TypeError("bliss_gnn_amd.optim.Adam updates contiguous bf16 parameters on the GPU (the reference's precision)")
        if len(ps) > _lib.ADAM_MAX_TENSORS:
            raise NotImplementedError("more than %d parameter tensors" % _lib.ADAM_MAX_TENSORS)
        dev = ps[0].device
        self._state = torch.zeros(4, dtype=torch.float32, device=dev)
        self._state[1] = lr
        self._lr_on_device = lr
        for p in ps:
            self.state[p] = dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))

    @property
    def step_count(self):
        return int(self._state[0].item())

    def state_dict(self):
        """torch's layout plus the device-resident step count (``'bliss_step'``): without it a resumed run would restart the
        bias correction at step 0."""
        sd = super().state_dict()
        sd["bliss_step"] = self.step_count
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        step = state_dict.pop("bliss_step", None)
        super().load_state_dict(state_dict)
        if step is not None:
            self._state[0] = float(step)
        self._state[1] = float(self.param_groups[0]["lr"])
        self._lr_on_device = float(self.param_groups[0]["lr"])

    def sync_lr(self):
        """Push ``param_groups[0]['lr']`` (what a torch lr_scheduler rewrites) to the device; call outside graph capture."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            self._state[1] = lr
            self._lr_on_device = lr

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        group = self.param_groups[0]
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        t = _lib.AdamTensors()
        n = 0
        for p in group["params"]:
            if p.grad is None:
                continue
            g = p.grad
            if g.dtype != torch.bfloat16 or not g.is_contiguous():
                g = g.to(torch.bfloat16).contiguous()
                p.grad = g
            st = self.state[p]
            t.param[n], t.grad[n], t.exp_avg[n], t.exp_avg_sq[n], t.numel[n] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                                            st["exp_avg_sq"].data_ptr(), p.numel())
            n += 1
        if n == 0:
            return
        t.count = n
        b1, b2 = group["betas"]
        _lib.check(_lib.lib.bliss_adam_step(C.byref(t), self._state.data_ptr(), float(b1), float(b2), float(group["eps"]),
                                            float(group["weight_decay"]), torch.cuda.current_stream().cuda_stream), "bliss_adam_step")
