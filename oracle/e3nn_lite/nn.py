"""
Restatement of ``e3nn.nn`` (v0.5.1) pieces used by the MatTen hot path [e3nn-recalled].
ORACLE / TEST INFRASTRUCTURE (see oracle/__init__.py).

  FullyConnectedNet  <- reference nn/utils.py:246-251 (radial MLP)
  Gate               <- reference nn/utils.py:134-140
  BatchNorm          <- reference nn/utils.py:418
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import torch

from . import o3


def moment(f: Callable, n: int, dtype=None, device=None) -> torch.Tensor:
    """E_{z~N(0,1)} f(z)^n estimated like e3nn.math._normalize_activation.moment (1e6 fp64 samples, seed 0)."""
    gen = torch.Generator(device="cpu").manual_seed(0)
    z = torch.randn(1_000_000, generator=gen, dtype=torch.float64).to(dtype=dtype, device=device)
    return f(z).pow(n).mean()


class normalize2mom(torch.nn.Module):
    _is_id: bool
    cst: float

    def __init__(self, f: Callable):
        super().__init__()
        with torch.no_grad():
            cst = moment(f, 2, dtype=torch.float64, device="cpu").pow(-0.5).item()
        self._is_id = abs(cst - 1) < 1e-4
        self.f = f
        self.cst = cst

    def forward(self, x):
        if self._is_id:
            return self.f(x)
        return self.f(x).mul(self.cst)


class _Layer(torch.nn.Module):
    def __init__(self, h_in: int, h_out: int, act, var_in: float, var_out: float):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(h_in, h_out))
        self.act = act
        self.h_in = h_in
        self.h_out = h_out
        self.var_in = var_in
        self.var_out = var_out

    def forward(self, x: torch.Tensor):
        if self.act is not None:
            w = self.weight / (self.h_in * self.var_in) ** 0.5
            x = x @ w
            x = self.act(x)
            x = x * self.var_out**0.5
        else:
            w = self.weight / (self.h_in * self.var_in / self.var_out) ** 0.5
            x = x @ w
        return x


class FullyConnectedNet(torch.nn.Sequential):
    """e3nn.nn.FullyConnectedNet: bias-free, weights ~N(0,1), x@W/sqrt(h_in), normalize2mom(act)."""

    def __init__(self, hs: List[int], act=None, variance_in: float = 1, variance_out: float = 1, out_act: bool = False):
        super().__init__()
        self.hs = list(hs)
        if act is not None:
            act = normalize2mom(act)
        var_in = variance_in
        for i, (h1, h2) in enumerate(zip(self.hs, self.hs[1:])):
            if i == len(self.hs) - 2:
                var_out = variance_out
                a = act if out_act else None
            else:
                var_out = 1
                a = act
            layer = _Layer(h1, h2, a, var_in, var_out)
            setattr(self, f"layer{i}", layer)
            var_in = var_out


def _act_parity(act) -> int:
    x = torch.linspace(0, 10, 256)
    a1, a2 = act(x), act(-x)
    if (a1 - a2).abs().max() < 1e-5:
        return 1
    if (a1 + a2).abs().max() < 1e-5:
        return -1
    return 0


class Activation(torch.nn.Module):
    """e3nn.nn.Activation: scalar activation, each act wrapped in normalize2mom."""

    def __init__(self, irreps_in, acts):
        super().__init__()
        irreps_in = o3.Irreps(irreps_in)
        assert len(irreps_in) == len(acts), (irreps_in, acts)
        acts = [normalize2mom(act) if act is not None else None for act in acts]
        irreps_out = []
        for (mul, (l_in, p_in)), act in zip(irreps_in, acts):
            if act is not None:
                assert l_in == 0
                p_act = _act_parity(act)
                p_out = p_act if p_in == -1 else p_in
                irreps_out.append((mul, (0, p_out)))
                if p_out == 0:
                    raise ValueError("activation: the parity is violated")
            else:
                irreps_out.append((mul, (l_in, p_in)))
        self.irreps_in = irreps_in
        self.irreps_out = o3.Irreps(irreps_out)
        self.acts = torch.nn.ModuleList(acts)

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        output = []
        index = 0
        for (mul, ir), act in zip(self.irreps_in, self.acts):
            if act is not None:
                output.append(act(features.narrow(-1, index, mul)))
            else:
                output.append(features.narrow(-1, index, mul * ir.dim))
            index += mul * ir.dim
        if len(output) > 1:
            return torch.cat(output, dim=-1)
        elif len(output) == 1:
            return output[0]
        return torch.zeros_like(features)


class _Sortcut(torch.nn.Module):
    def __init__(self, *irreps_outs):
        super().__init__()
        self.irreps_outs = tuple(o3.Irreps(irreps).simplify() for irreps in irreps_outs)
        irreps_in = sum(self.irreps_outs, o3.Irreps([]))

        i = 0
        instructions = []
        for irreps_out in self.irreps_outs:
            instructions += [tuple(range(i, i + len(irreps_out)))]
            i += len(irreps_out)
        assert len(irreps_in) == i, (len(irreps_in), i)

        irreps_in_sorted, p, _ = irreps_in.sort()
        instructions = [tuple(p[i] for i in x) for x in instructions]
        self._sorted = irreps_in_sorted
        self._instructions = instructions
        self.irreps_in = irreps_in_sorted.simplify()

    def forward(self, x):
        sl = self._sorted.slices()
        outs = []
        for ins in self._instructions:
            if len(ins) == 0:
                outs.append(x[..., :0])
            else:
                outs.append(torch.cat([x[..., sl[i]] for i in ins], dim=-1))
        return tuple(outs)


class Gate(torch.nn.Module):
    """e3nn.nn.Gate."""

    def __init__(self, irreps_scalars, act_scalars, irreps_gates, act_gates, irreps_gated):
        super().__init__()
        irreps_scalars = o3.Irreps(irreps_scalars)
        irreps_gates = o3.Irreps(irreps_gates)
        irreps_gated = o3.Irreps(irreps_gated)

        if len(irreps_gates) > 0 and irreps_gates.lmax > 0:
            raise ValueError(f"Gate scalars must be scalars, instead got irreps_gates = {irreps_gates}")
        if len(irreps_scalars) > 0 and irreps_scalars.lmax > 0:
            raise ValueError(f"Scalars must be scalars, instead got irreps_scalars = {irreps_scalars}")
        if irreps_gates.num_irreps != irreps_gated.num_irreps:
            raise ValueError(
                f"There are {irreps_gated.num_irreps} irreps in irreps_gated, "
                f"but a different number ({irreps_gates.num_irreps}) of gate scalars in irreps_gates"
            )

        self.sc = _Sortcut(irreps_scalars, irreps_gates, irreps_gated)
        self.irreps_scalars, self.irreps_gates, self.irreps_gated = self.sc.irreps_outs
        self._irreps_in = self.sc.irreps_in

        self.act_scalars = Activation(irreps_scalars, act_scalars)
        irreps_scalars = self.act_scalars.irreps_out

        self.act_gates = Activation(irreps_gates, act_gates)
        irreps_gates = self.act_gates.irreps_out

        self.mul = o3.ElementwiseTensorProduct(irreps_gated, irreps_gates)
        irreps_gated = self.mul.irreps_out

        self._irreps_out = irreps_scalars + irreps_gated

    def forward(self, features):
        scalars, gates, gated = self.sc(features)
        scalars = self.act_scalars(scalars)
        if gates.shape[-1]:
            gates = self.act_gates(gates)
            gated = self.mul(gated, gates)
            features = torch.cat([scalars, gated], dim=-1)
        else:
            features = scalars
        return features

    @property
    def irreps_in(self):
        return self._irreps_in

    @property
    def irreps_out(self):
        return self._irreps_out


class NormActivation(torch.nn.Module):
    """e3nn.nn.NormActivation (e3nn 0.5.1 nn/_normact.py), restated: every irrep channel is scaled by
    ``f(|x|) / |x|`` (``normalize=True``), |x| the Euclidean norm over the channel's 2l+1 components
    (o3.Norm: the 'uuu' product l x l -> 0e with path weight 2l+1 in component normalisation = sum_m x_m^2),
    squared norms below epsilon^2 clamped to epsilon^2 before the square root (so no gradient flows through the
    clamped norm), the scaling applied by o3.ElementwiseTensorProduct (0e x l -> l: plain multiplication).
    ``scalar_nonlinearity`` is used as given (no normalize2mom).  bias=False only (the reference's call,
    matten nn/utils.py:142-150)."""

    def __init__(self, irreps_in, scalar_nonlinearity: Callable, normalize: bool = True, epsilon: float = None,
                 bias: bool = False):
        super().__init__()
        from .o3 import Irreps

        if bias:
            raise NotImplementedError("NormActivation(bias=True) is not used by the reference")
        self.irreps_in = Irreps(irreps_in)
        self.irreps_out = Irreps(irreps_in)
        if epsilon is None and normalize:
            epsilon = 1e-8
        elif epsilon is not None and not normalize:
            raise ValueError("epsilon and normalize = False don't make sense together")
        elif not normalize:
            epsilon = 0.0
        self._eps_squared = epsilon * epsilon
        self.scalar_nonlinearity = scalar_nonlinearity
        self.normalize = normalize

    def forward(self, features):
        out, ix = [], 0
        for mul, ir in self.irreps_in:
            d = ir.dim
            field = features[..., ix: ix + mul * d].reshape(features.shape[:-1] + (mul, d))
            ix += mul * d
            norms = field.pow(2).sum(-1)
            if self._eps_squared > 0:
                norms = torch.where(norms < self._eps_squared, torch.full_like(norms, self._eps_squared), norms)
                norms = norms.sqrt()
            scalings = self.scalar_nonlinearity(norms)
            if self.normalize:
                scalings = scalings / norms
            out.append((field * scalings[..., None]).reshape(features.shape[:-1] + (mul * d,)))
        return torch.cat(out, dim=-1)


class BatchNorm(torch.nn.Module):
    """e3nn.nn.BatchNorm (defaults: eps 1e-5, momentum 0.1, affine, reduce mean, component)."""

    def __init__(self, irreps, eps=1e-5, momentum=0.1, affine=True, reduce="mean", instance=False, normalization="component"):
        super().__init__()
        self.irreps = o3.Irreps(irreps)
        self.eps = eps
        self.momentum = momentum
        self.affine = affine
        self.instance = instance
        self.reduce = reduce
        self.normalization = normalization

        num_scalar = sum(mul for mul, ir in self.irreps if ir.is_scalar())
        num_features = self.irreps.num_irreps

        if self.instance:
            self.register_buffer("running_mean", None)
            self.register_buffer("running_var", None)
        else:
            self.register_buffer("running_mean", torch.zeros(num_scalar))
            self.register_buffer("running_var", torch.ones(num_features))

        if affine:
            self.weight = torch.nn.Parameter(torch.ones(num_features))
            self.bias = torch.nn.Parameter(torch.zeros(num_scalar))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def _roll_avg(self, curr, update):
        return (1 - self.momentum) * curr + self.momentum * update.detach()

    def forward(self, input):
        batch, *size, dim = input.shape
        input = input.reshape(batch, -1, dim)  # [batch, sample, stacked features]

        if self.training and not self.instance:
            new_means = []
            new_vars = []

        fields = []
        ix = 0
        irm = 0
        irv = 0
        iw = 0
        ib = 0

        for mul, ir in self.irreps:
            d = ir.dim
            field = input[:, :, ix : ix + mul * d]
            ix += mul * d
            field = field.reshape(batch, -1, mul, d)

            if ir.is_scalar():
                if self.training or self.instance:
                    if self.instance:
                        field_mean = field.mean(1).reshape(batch, mul)
                    else:
                        field_mean = field.mean([0, 1]).reshape(mul)
                        new_means.append(self._roll_avg(self.running_mean[irm : irm + mul], field_mean))
                else:
                    field_mean = self.running_mean[irm : irm + mul]
                irm += mul
                field = field - field_mean.reshape(-1, 1, mul, 1)

            if self.training or self.instance:
                if self.normalization == "norm":
                    field_norm = field.pow(2).sum(3)
                elif self.normalization == "component":
                    field_norm = field.pow(2).mean(3)
                else:
                    raise ValueError(self.normalization)
                if self.reduce == "mean":
                    field_norm = field_norm.mean(1)
                elif self.reduce == "max":
                    field_norm = field_norm.max(1).values
                else:
                    raise ValueError(self.reduce)
                if not self.instance:
                    field_norm = field_norm.mean(0)
                    new_vars.append(self._roll_avg(self.running_var[irv : irv + mul], field_norm))
            else:
                field_norm = self.running_var[irv : irv + mul]
            irv += mul

            field_norm = (field_norm + self.eps).pow(-0.5)

            if self.affine:
                weight = self.weight[iw : iw + mul]
                iw += mul
                field_norm = field_norm * weight

            field = field * field_norm.reshape(-1, 1, mul, 1)

            if self.affine and ir.is_scalar():
                bias = self.bias[ib : ib + mul]
                ib += mul
                field = field + bias.reshape(mul, 1)

            fields.append(field.reshape(batch, -1, mul * d))

        assert ix == dim

        if self.training and not self.instance:
            assert irm == self.running_mean.numel()
            assert irv == self.running_var.size(0)
        if self.affine:
            assert iw == self.weight.size(0)
            assert ib == self.bias.numel()

        if self.training and not self.instance:
            if len(new_means) > 0:
                torch.cat(new_means, out=self.running_mean)
            if len(new_vars) > 0:
                torch.cat(new_vars, out=self.running_var)

        output = torch.cat(fields, dim=2)
        return output.reshape(batch, *size, dim)
