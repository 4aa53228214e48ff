"""
Adam over one flat parameter buffer, the optimiser of the reference's training recipe
(torch.optim.Adam, lr 1e-2, weight_decay 1e-5: scripts/configs/materials_tensor.yaml:103-107;
pretrained/20230627/config_final.yaml:43-47).

``FlatAdam(model.parameters(), ...)`` re-homes every parameter as a view into ONE contiguous fp32 buffer and gives each
a gradient view into a second one (autograd accumulates into an existing ``.grad`` in place), so a step is a single
elementwise launch of ``matten_adam_step`` and ``zero_grad`` a single memset -- instead of the multi-tensor machinery
over ~60 small tensors.  State (step count included) lives on the device: a step captures into a hipGraph as it is
(``matten_amd.graphs.GraphedTrainStep``).  Same update rule and state names as ``torch.optim.Adam`` (amsgrad off).
"""
from typing import Iterable

import torch

from . import _lib, ops
from .nn._tables import bump_weights_epoch


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FlatAdam: no parameter requires a gradient")
        if any(p.dtype != torch.float32 or not p.is_cuda for p in params):
            raise _lib.MattenHipError("FlatAdam: parameters must be fp32 tensors on the MI355X (no CPU fallback)")
        if len({p.device for p in params}) != 1:
            raise ValueError("FlatAdam: all parameters on one device")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev = params[0].device
        # 16-byte aligned slots: the kernel moves four floats per lane
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self._n = n
        self.flat_params = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.flat_params[o:o + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                p.grad = self.flat_grads[o:o + p.numel()].view_as(p)
        self._params, self._offs = params, offs
        # the optimiser state under torch's names (state_dict / snapshotting tools look here); one entry for the lot
        self.state[params[0]] = {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    def zero_grad(self, set_to_none: bool = False) -> None:
        """one memset; the gradient views stay in place (set_to_none would detach them from the flat buffer)"""
        self.flat_grads.zero_()
        for p, o in zip(self._params, self._offs):
            if p.grad is None or p.grad.data_ptr() != self.flat_grads.data_ptr() + 4 * o:
                p.grad = self.flat_grads[o:o + p.numel()].view_as(p)

    def load_state_dict(self, state_dict) -> None:
        """torch's loader would REPLACE the state entry with fresh tensors that step() never looks at (a resumed run would
        restart with zero moments and step 0): copy the loaded moments and step count into the live flat buffers and
        point ``self.state`` back at them.  Accepts what ``state_dict()`` of this class writes (one state entry holding
        the flat tensors) and what ``torch.optim.Adam`` over the same parameter list writes (one entry per parameter)."""
        groups, state = state_dict["param_groups"], state_dict["state"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self._params):
            raise ValueError("FlatAdam.load_state_dict: the saved optimiser covers a different parameter list")
        for k, v in groups[0].items():
            if k != "params":
                self.param_groups[0][k] = v
        ids = list(groups[0]["params"])
        with torch.no_grad():
            first = state.get(ids[0])
            if first is not None and first["exp_avg"].numel() == self._n and len(state) == 1:   # this class's own layout
                self.exp_avg.copy_(first["exp_avg"].reshape(-1))
                self.exp_avg_sq.copy_(first["exp_avg_sq"].reshape(-1))
                self.step_count.fill_(float(first["step"]))
            elif state:                                                                         # torch.optim.Adam's layout
                steps = set()
                for pid, p, o in zip(ids, self._params, self._offs):
                    st = state.get(pid)
                    if st is None:
                        raise ValueError(f"FlatAdam.load_state_dict: no state for parameter {pid}")
                    if st["exp_avg"].numel() != p.numel():
                        raise ValueError(f"FlatAdam.load_state_dict: state of parameter {pid} has the wrong size")
                    self.exp_avg[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(float(st["step"]))
                if len(steps) != 1:
                    raise ValueError("FlatAdam.load_state_dict: parameters with different step counts")
                self.step_count.fill_(steps.pop())
            else:
                self.exp_avg.zero_(), self.exp_avg_sq.zero_(), self.step_count.zero_()
        self.state.clear()
        self.state[self._params[0]] = {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        base = self.flat_params.data_ptr()
        for p, o in zip(self._params, self._offs):   # model.to() / .double() / p.data = ... after construction detach a parameter
            if p.data_ptr() != base + 4 * o:
                raise _lib.MattenHipError("FlatAdam: a parameter no longer lives in the flat buffer (model.to() / .double() / "
                                          "p.data = ... after the optimiser was built); build the optimiser last")
        for p, o in zip(self._params, self._offs):   # a gradient that was replaced (set_to_none, clipping into new tensors)
            if p.grad is not None and p.grad.data_ptr() != self.flat_grads.data_ptr() + 4 * o:
                self.flat_grads[o:o + p.numel()].view_as(p).copy_(p.grad)
                p.grad = self.flat_grads[o:o + p.numel()].view_as(p)
        g = self.param_groups[0]
        self.step_count += 1.0
        lib = _lib.load()
        _lib.check(lib.matten_adam_step(self.flat_params.data_ptr(), self.flat_grads.data_ptr(), self.exp_avg.data_ptr(),
                                        self.exp_avg_sq.data_ptr(), self._n, self.step_count.data_ptr(), float(g["lr"]),
                                        float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                        float(g["weight_decay"]), ops._stream()), "matten_adam_step")
        bump_weights_epoch()   # the kernel wrote through raw pointers: no parameter's _version moved
        return loss
