"""
Restatement of the parts of ``e3nn.o3`` (v0.5.1) the MatTen hot path calls.

ORACLE / TEST INFRASTRUCTURE (see oracle/__init__.py).  e3nn is not vendored in the
reference and not installable here, so everything below is written from e3nn 0.5.1's
published algorithm [e3nn-recalled]; reference call sites are cited per symbol.

Conventions restated:
  * Irrep ordering is the tuple order (l, p) with p in {-1, +1}  ->  0o < 0e < 1o < 1e ...
  * data layout of an irreps block is [mul, 2l+1] row-major ("mul_ir")
  * real spherical harmonics with polar axis y, m = -l..l
  * real Wigner-3j = complex CG pushed through the real<->complex change of basis,
    Frobenius-normalised
"""
from __future__ import annotations

import collections
import functools
import itertools
import math
from fractions import Fraction
from typing import Iterator, List, NamedTuple, Optional, Sequence, Tuple, Union

import torch


# --------------------------------------------------------------------------------------
# Irrep / Irreps
# --------------------------------------------------------------------------------------
class Irrep(tuple):
    """(l, p) -- e3nn.o3.Irrep.  Used all over the reference, e.g. nn/utils.py:358-367."""

    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                name = l.strip()
                l = int(name[:-1])
                p = {"e": 1, "o": -1, "y": (-1) ** l}[name[-1]]
            elif isinstance(l, tuple):
                l, p = l
        assert isinstance(l, int) and l >= 0, l
        assert p in (-1, 1), p
        return super().__new__(cls, (l, p))

    @property
    def l(self) -> int:  # noqa: E743
        return self[0]

    @property
    def p(self) -> int:
        return self[1]

    @property
    def dim(self) -> int:
        return 2 * self.l + 1

    def is_scalar(self) -> bool:
        return self.l == 0 and self.p == 1

    def __repr__(self):
        return f"{self.l}{'e' if self.p == 1 else 'o'}"

    def __mul__(self, other) -> Iterator["Irrep"]:
        other = Irrep(other)
        p = self.p * other.p
        lmin = abs(self.l - other.l)
        lmax = self.l + other.l
        for l in range(lmin, lmax + 1):
            yield Irrep(l, p)

    def __rmul__(self, mul):
        assert isinstance(mul, int)
        return Irreps([(mul, self)])

    def __add__(self, other):
        return Irreps(self) + Irreps(other)


class _MulIr(tuple):
    def __new__(cls, mul, ir=None):
        if ir is None:
            mul, ir = mul
        assert isinstance(mul, int)
        return super().__new__(cls, (mul, Irrep(ir)))

    @property
    def mul(self) -> int:
        return self[0]

    @property
    def ir(self) -> Irrep:
        return self[1]

    @property
    def dim(self) -> int:
        return self.mul * self.ir.dim

    def __repr__(self):
        return f"{self.mul}x{self.ir}"


class Irreps(tuple):
    """e3nn.o3.Irreps: tuple of (mul, Irrep)."""

    def __new__(cls, irreps=None):
        if isinstance(irreps, Irreps):
            return super().__new__(cls, irreps)
        out = []
        if isinstance(irreps, Irrep):
            out.append(_MulIr(1, irreps))
        elif isinstance(irreps, str):
            if irreps.strip() != "":
                for mul_ir in irreps.split("+"):
                    if "x" in mul_ir:
                        mul, ir = mul_ir.split("x")
                        out.append(_MulIr(int(mul), Irrep(ir)))
                    else:
                        out.append(_MulIr(1, Irrep(mul_ir)))
        elif irreps is None:
            pass
        else:
            for mul_ir in irreps:
                if isinstance(mul_ir, str):
                    out.append(_MulIr(1, Irrep(mul_ir)))
                elif isinstance(mul_ir, Irrep):
                    out.append(_MulIr(1, mul_ir))
                elif isinstance(mul_ir, _MulIr):
                    out.append(mul_ir)
                else:
                    mul, ir = mul_ir
                    out.append(_MulIr(int(mul), Irrep(ir)))
        return super().__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax: int, p: int = -1) -> "Irreps":
        return Irreps([(1, (l, p**l)) for l in range(lmax + 1)])

    def slices(self) -> List[slice]:
        s = []
        i = 0
        for mul_ir in self:
            s.append(slice(i, i + mul_ir.dim))
            i += mul_ir.dim
        return s

    def __getitem__(self, i):
        x = super().__getitem__(i)
        if isinstance(i, slice):
            return Irreps(x)
        return x

    def __contains__(self, ir) -> bool:
        ir = Irrep(ir)
        return ir in (irrep for _, irrep in self)

    def count(self, ir) -> int:
        ir = Irrep(ir)
        return sum(mul for mul, irrep in self if ir == irrep)

    def __add__(self, irreps):
        irreps = Irreps(irreps)
        return Irreps(super().__add__(irreps))

    def __mul__(self, other):
        if isinstance(other, int):
            return Irreps(super().__mul__(other))
        raise NotImplementedError

    def simplify(self) -> "Irreps":
        out = []
        for mul, ir in self:
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + mul, ir)
            elif mul > 0:
                out.append((mul, ir))
        return Irreps(out)

    def remove_zero_multiplicities(self) -> "Irreps":
        return Irreps([(mul, ir) for mul, ir in self if mul > 0])

    def sort(self):
        Ret = collections.namedtuple("sort", ["irreps", "p", "inv"])
        out = [(ir, i, mul) for i, (mul, ir) in enumerate(self)]
        out = sorted(out)
        inv = tuple(i for _, i, _ in out)
        p = _perm_inverse(inv)
        irreps = Irreps([(mul, ir) for ir, _, mul in out])
        return Ret(irreps, p, inv)

    @property
    def dim(self) -> int:
        return sum(mul * ir.dim for mul, ir in self)

    @property
    def num_irreps(self) -> int:
        return sum(mul for mul, _ in self)

    @property
    def ls(self) -> List[int]:
        return [l for mul, (l, p) in self for _ in range(mul)]

    @property
    def lmax(self) -> int:
        if len(self) == 0:
            raise ValueError("Cannot get lmax of empty Irreps")
        return max(self.ls)

    def __repr__(self):
        return "+".join(f"{mul_ir}" for mul_ir in self)


def _perm_inverse(p):
    return tuple(p.index(i) for i in range(len(p)))


# --------------------------------------------------------------------------------------
# Wigner 3j
# --------------------------------------------------------------------------------------
def _su2_clebsch_gordan_coeff(idx1, idx2, idx3) -> float:
    j1, m1 = idx1
    j2, m2 = idx2
    j3, m3 = idx3
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max([-j1 + j2 + m3, -j1 + m1, 0]))
    vmax = int(min([j2 + j3 + m1, j3 - j1 + j2, j3 + m3]))

    def f(n):
        assert n == round(n)
        return math.factorial(round(n))

    C = (
        (2.0 * j3 + 1.0)
        * Fraction(
            f(j3 + j1 - j2) * f(j3 - j1 + j2) * f(j1 + j2 - j3) * f(j3 + m3) * f(j3 - m3),
            f(j1 + j2 + j3 + 1) * f(j1 - m1) * f(j1 + m1) * f(j2 - m2) * f(j2 + m2),
        )
    ) ** 0.5

    S = 0
    for v in range(vmin, vmax + 1):
        S += (-1) ** int(v + j2 + m2) * Fraction(
            f(j2 + j3 + m1 - v) * f(j1 - m1 + v),
            f(v) * f(j3 - j1 + j2 - v) * f(j3 + m3 - v) * f(v + j1 - j2 - m3),
        )
    return float(C * S)


def _su2_clebsch_gordan(j1: int, j2: int, j3: int) -> torch.Tensor:
    mat = torch.zeros((2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1), dtype=torch.float64)
    if abs(j1 - j2) <= j3 <= j1 + j2:
        for m1 in range(-j1, j1 + 1):
            for m2 in range(-j2, j2 + 1):
                if abs(m1 + m2) <= j3:
                    mat[j1 + m1, j2 + m2, j3 + m1 + m2] = _su2_clebsch_gordan_coeff(
                        (j1, m1), (j2, m2), (j3, m1 + m2)
                    )
    return mat


def change_basis_real_to_complex(l: int) -> torch.Tensor:
    # https://en.wikipedia.org/wiki/Spherical_harmonics#Real_form
    q = torch.zeros((2 * l + 1, 2 * l + 1), dtype=torch.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / 2**0.5
        q[l + m, l - abs(m)] = -1j / 2**0.5
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / 2**0.5
        q[l + m, l - abs(m)] = 1j * (-1) ** m / 2**0.5
    q = (-1j) ** l * q  # makes the Clebsch-Gordan coefficients real
    return q


@functools.lru_cache(maxsize=None)
def _so3_clebsch_gordan(l1: int, l2: int, l3: int) -> torch.Tensor:
    Q1 = change_basis_real_to_complex(l1)
    Q2 = change_basis_real_to_complex(l2)
    Q3 = change_basis_real_to_complex(l3)
    C = _su2_clebsch_gordan(l1, l2, l3).to(dtype=torch.complex128)
    C = torch.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, torch.conj(Q3.T), C)
    assert torch.all(torch.abs(torch.imag(C)) < 1e-9)
    C = torch.real(C)
    C = C / torch.norm(C)
    return C


def wigner_3j(l1: int, l2: int, l3: int, dtype=None) -> torch.Tensor:
    """e3nn.o3.wigner_3j: real, Frobenius-normalised, shape [2l1+1, 2l2+1, 2l3+1] (fp64 master)."""
    assert abs(l2 - l3) <= l1 <= l2 + l3
    C = _so3_clebsch_gordan(l1, l2, l3)
    if dtype is None:
        dtype = torch.get_default_dtype()
    return C.to(dtype=dtype).clone()


# --------------------------------------------------------------------------------------
# Spherical harmonics
# --------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def _sh_recursion_constants(lmax: int) -> Tuple[float, ...]:
    n = torch.tensor([[0.3, -0.5, 0.8124038404635961]], dtype=torch.float64)
    n = n / n.norm()
    ys = [torch.ones_like(n[..., :1]), n]
    consts = [1.0, 1.0]
    for l in range(2, lmax + 1):
        C = wigner_3j(l, l - 1, 1, dtype=torch.float64)
        y = torch.einsum("kij,...i,...j->...k", C, ys[l - 1], ys[1])
        c = 1.0 / y.norm().item()
        consts.append(c)
        ys.append(c * y)
    return tuple(consts)


def spherical_harmonics(
    ls: Sequence[int], x: torch.Tensor, normalize: bool, normalization: str = "integral"
) -> torch.Tensor:
    """
    e3nn.o3.spherical_harmonics / o3.SphericalHarmonics.forward.
    Reference call site: nn/_nequip.py:167-174 (normalize=True, normalization="component").
    """
    assert normalization in ("integral", "component", "norm")
    if normalize:
        x = torch.nn.functional.normalize(x, dim=-1)
    lmax = max(ls)
    consts = _sh_recursion_constants(lmax)
    ys = [torch.ones_like(x[..., :1])]
    if lmax >= 1:
        ys.append(x)
    for l in range(2, lmax + 1):
        C = wigner_3j(l, l - 1, 1, dtype=x.dtype)
        ys.append(consts[l] * torch.einsum("kij,...i,...j->...k", C, ys[l - 1], ys[1]))
    out = []
    for l in ls:
        y = ys[l]
        if normalization == "integral":
            y = y * (math.sqrt(2 * l + 1) / math.sqrt(4 * math.pi))
        elif normalization == "component":
            y = y * math.sqrt(2 * l + 1)
        out.append(y)
    return torch.cat(out, dim=-1)


class SphericalHarmonics(torch.nn.Module):
    def __init__(self, irreps_out, normalize: bool, normalization: str = "integral"):
        super().__init__()
        if isinstance(irreps_out, int):
            irreps_out = Irreps.spherical_harmonics(irreps_out)
        self.irreps_out = Irreps(irreps_out)
        for mul, (l, p) in self.irreps_out:
            assert p == (-1) ** l, "spherical harmonics have parity (-1)^l"
        self._ls = [l for mul, (l, p) in self.irreps_out for _ in range(mul)]
        self.normalize = normalize
        self.normalization = normalization

    def forward(self, x):
        return spherical_harmonics(self._ls, x, self.normalize, self.normalization)


def rand_matrix(generator=None, dtype=torch.float64) -> torch.Tensor:
    """Random proper rotation (QR of a Gaussian matrix) -- stand-in for e3nn.o3.rand_matrix."""
    a = torch.randn(3, 3, dtype=dtype, generator=generator)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def wigner_D_from_sh(l: int, R: torch.Tensor) -> torch.Tensor:
    """
    Real Wigner-D of the rotation R in the basis of the real SH above:  Y_l(R x) = D_l(R) Y_l(x).
    Obtained by least squares over random directions (test helper, fp64).
    """
    g = torch.Generator().manual_seed(1234 + l)
    x = torch.randn(8 * (2 * l + 1) + 8, 3, dtype=torch.float64, generator=g)
    Y = spherical_harmonics([l], x, True, "component")
    YR = spherical_harmonics([l], x @ R.to(torch.float64).T, True, "component")
    # YR = Y @ D^T
    Dt = torch.linalg.lstsq(Y, YR).solution
    return Dt.T


# --------------------------------------------------------------------------------------
# TensorProduct
# --------------------------------------------------------------------------------------
class Instruction(NamedTuple):
    i_in1: int
    i_in2: int
    i_out: int
    connection_mode: str
    has_weight: bool
    path_weight: float
    path_shape: tuple


class TensorProduct(torch.nn.Module):
    """
    e3nn.o3.TensorProduct restricted to the connection modes the hot path uses ('uvu', 'uvw').
    Reference call sites: nn/utils.py:230-237 (uvu, external weights), nn/conv.py:59-86 (via
    FullyConnectedTensorProduct, uvw, internal shared weights).
    """

    def __init__(
        self,
        irreps_in1,
        irreps_in2,
        irreps_out,
        instructions,
        in1_var=None,
        in2_var=None,
        out_var=None,
        irrep_normalization: str = "component",
        path_normalization: str = "element",
        internal_weights: Optional[bool] = None,
        shared_weights: Optional[bool] = None,
    ):
        super().__init__()
        self.irreps_in1 = Irreps(irreps_in1)
        self.irreps_in2 = Irreps(irreps_in2)
        self.irreps_out = Irreps(irreps_out)

        instructions = [x if len(x) == 6 else x + (1.0,) for x in instructions]
        ins_list = []
        for i_in1, i_in2, i_out, mode, has_weight, path_weight in instructions:
            shape = {
                "uvw": (self.irreps_in1[i_in1].mul, self.irreps_in2[i_in2].mul, self.irreps_out[i_out].mul),
                "uvu": (self.irreps_in1[i_in1].mul, self.irreps_in2[i_in2].mul),
            }[mode]
            ins_list.append(Instruction(i_in1, i_in2, i_out, mode, has_weight, path_weight, shape))
        instructions = ins_list

        if in1_var is None:
            in1_var = [1.0 for _ in self.irreps_in1]
        if in2_var is None:
            in2_var = [1.0 for _ in self.irreps_in2]
        if out_var is None:
            out_var = [1.0 for _ in self.irreps_out]

        def num_elements(ins):
            return {
                "uvw": (self.irreps_in1[ins.i_in1].mul * self.irreps_in2[ins.i_in2].mul),
                "uvu": self.irreps_in2[ins.i_in2].mul,
            }[ins.connection_mode]

        normalization_coefficients = []
        for ins in instructions:
            mul_ir_in1 = self.irreps_in1[ins.i_in1]
            mul_ir_in2 = self.irreps_in2[ins.i_in2]
            mul_ir_out = self.irreps_out[ins.i_out]
            assert mul_ir_in1.ir.p * mul_ir_in2.ir.p == mul_ir_out.ir.p
            assert abs(mul_ir_in1.ir.l - mul_ir_in2.ir.l) <= mul_ir_out.ir.l <= mul_ir_in1.ir.l + mul_ir_in2.ir.l

            if irrep_normalization == "component":
                alpha = mul_ir_out.ir.dim
            elif irrep_normalization == "norm":
                alpha = mul_ir_in1.ir.dim * mul_ir_in2.ir.dim
            else:
                alpha = 1

            if path_normalization == "element":
                x = sum(
                    in1_var[i.i_in1] * in2_var[i.i_in2] * num_elements(i)
                    for i in instructions
                    if i.i_out == ins.i_out
                )
            elif path_normalization == "path":
                x = in1_var[ins.i_in1] * in2_var[ins.i_in2] * num_elements(ins)
                x *= len([i for i in instructions if i.i_out == ins.i_out])
            else:
                x = 1
            if x > 0.0:
                alpha /= x
            alpha *= out_var[ins.i_out]
            alpha *= ins.path_weight
            normalization_coefficients.append(math.sqrt(alpha))

        self.instructions = [
            Instruction(i.i_in1, i.i_in2, i.i_out, i.connection_mode, i.has_weight, c, i.path_shape)
            for i, c in zip(instructions, normalization_coefficients)
        ]

        if shared_weights is False and internal_weights is None:
            internal_weights = False
        if shared_weights is None:
            shared_weights = True
        if internal_weights is None:
            internal_weights = shared_weights and any(i.has_weight for i in self.instructions)
        assert shared_weights or not internal_weights
        self.internal_weights = internal_weights
        self.shared_weights = shared_weights

        self.weight_numel = sum(math.prod(i.path_shape) for i in self.instructions if i.has_weight)
        if internal_weights and self.weight_numel > 0:
            self.weight = torch.nn.Parameter(torch.randn(self.weight_numel))
        else:
            self.register_buffer("weight", torch.Tensor())

        if self.irreps_out.dim > 0:
            output_mask = torch.cat(
                [
                    torch.ones(mul * ir.dim)
                    if any(
                        (ins.i_out == i_out) and (ins.path_weight != 0) and (0 not in ins.path_shape)
                        for ins in self.instructions
                    )
                    else torch.zeros(mul * ir.dim)
                    for i_out, (mul, ir) in enumerate(self.irreps_out)
                ]
            )
        else:
            output_mask = torch.ones(0)
        self.register_buffer("output_mask", output_mask)

    def forward(self, x1: torch.Tensor, x2: torch.Tensor, weight: Optional[torch.Tensor] = None) -> torch.Tensor:
        if weight is None:
            assert self.internal_weights or self.weight_numel == 0
            weight = self.weight
        else:
            if self.shared_weights:
                assert weight.shape == (self.weight_numel,)
            else:
                assert weight.shape[-1] == self.weight_numel

        batch = x1.shape[0]
        assert x1.shape == (batch, self.irreps_in1.dim), (x1.shape, self.irreps_in1.dim)
        assert x2.shape == (batch, self.irreps_in2.dim), (x2.shape, self.irreps_in2.dim)

        x1s = [
            x1[:, s].reshape(batch, mul, ir.dim) for s, (mul, ir) in zip(self.irreps_in1.slices(), self.irreps_in1)
        ]
        x2s = [
            x2[:, s].reshape(batch, mul, ir.dim) for s, (mul, ir) in zip(self.irreps_in2.slices(), self.irreps_in2)
        ]

        outs: List[List[torch.Tensor]] = [[] for _ in self.irreps_out]
        flat = 0
        z = "" if self.shared_weights else "z"
        for ins in self.instructions:
            mul_ir_in1 = self.irreps_in1[ins.i_in1]
            mul_ir_in2 = self.irreps_in2[ins.i_in2]
            mul_ir_out = self.irreps_out[ins.i_out]
            if mul_ir_in1.dim == 0 or mul_ir_in2.dim == 0 or mul_ir_out.dim == 0:
                continue
            a = x1s[ins.i_in1]
            b = x2s[ins.i_in2]
            if ins.has_weight:
                n = math.prod(ins.path_shape)
                w = weight[..., flat : flat + n].reshape(((-1,) if not self.shared_weights else ()) + ins.path_shape)
                flat += n
            w3j = wigner_3j(mul_ir_in1.ir.l, mul_ir_in2.ir.l, mul_ir_out.ir.l, dtype=x1.dtype).to(x1.device)
            # Contraction order: e3nn's generated code contracts each instruction with an optimised
            # einsum path (opt_einsum_fx); the orders below are such paths, written out.
            if ins.connection_mode == "uvw":
                assert ins.has_weight
                if mul_ir_in2.ir.l == 0 and self.shared_weights:
                    # scalar second operand (the one-hot species vector of reference nn/conv.py:59-86)
                    wz = torch.einsum("uvw,zv->zuw", w, b[:, :, 0])
                    r = torch.einsum("zuw,zui,ik->zwk", wz, a, w3j[:, 0, :])
                else:
                    r = torch.einsum(f"{z}uvw,ijk,zui,zvj->zwk", w, w3j, a, b)
            elif ins.connection_mode == "uvu":
                assert mul_ir_in1.mul == mul_ir_out.mul
                m = torch.einsum("ijk,zvj->zvik", w3j, b)  # per-sample coupling matrices
                if mul_ir_in2.mul == 1:
                    r = torch.bmm(a, m[:, 0])  # [z,u,k]
                    if ins.has_weight:
                        r = r * (w[..., 0, None] if not self.shared_weights else w[None, :, 0, None])
                else:
                    if ins.has_weight:
                        r = torch.einsum(f"{z}uv,zui,zvik->zuk", w, a, m)
                    else:
                        r = torch.einsum("zui,zvik->zuk", a, m)
            else:
                raise NotImplementedError(ins.connection_mode)
            r = ins.path_weight * r
            outs[ins.i_out].append(r.reshape(batch, mul_ir_out.dim))

        cols = []
        for i_out, mul_ir_out in enumerate(self.irreps_out):
            if mul_ir_out.dim == 0:
                continue
            if outs[i_out]:
                cols.append(functools.reduce(torch.add, outs[i_out]))
            else:
                cols.append(x1.new_zeros(batch, mul_ir_out.dim))
        if cols:
            return torch.cat(cols, dim=1)
        return x1.new_zeros(batch, 0)


class FullyConnectedTensorProduct(TensorProduct):
    """e3nn.o3.FullyConnectedTensorProduct -- reference call sites nn/conv.py:59-61,77-79,84-86."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, irrep_normalization=None, path_normalization=None, **kwargs):
        irreps_in1 = Irreps(irreps_in1).simplify()
        irreps_in2 = Irreps(irreps_in2).simplify()
        irreps_out = Irreps(irreps_out).simplify()
        instr = [
            (i_1, i_2, i_out, "uvw", True, 1.0)
            for i_1, (_, ir_1) in enumerate(irreps_in1)
            for i_2, (_, ir_2) in enumerate(irreps_in2)
            for i_out, (_, ir_out) in enumerate(irreps_out)
            if ir_out in ir_1 * ir_2
        ]
        super().__init__(
            irreps_in1,
            irreps_in2,
            irreps_out,
            instr,
            irrep_normalization=irrep_normalization or "component",
            path_normalization=path_normalization or "element",
            **kwargs,
        )


class Linear(torch.nn.Module):
    """
    e3nn.o3.Linear (no bias, internal shared weights).
    Reference call sites: nn/nodewise.py:111-116, model_factory/tfn_scalar_tensor.py:49-51.
    """

    def __init__(self, irreps_in, irreps_out, path_normalization: str = "element"):
        super().__init__()
        self.irreps_in = Irreps(irreps_in)
        self.irreps_out = Irreps(irreps_out)
        ins = [
            (i_in, i_out)
            for i_in, (_, ir_in) in enumerate(self.irreps_in)
            for i_out, (_, ir_out) in enumerate(self.irreps_out)
            if ir_in == ir_out
        ]

        def alpha(i_in, i_out):
            x = sum(
                self.irreps_in[j_in if path_normalization == "element" else i_in].mul
                for j_in, j_out in ins
                if j_out == i_out
            )
            return 1.0 if x == 0 else x

        self.instructions = [(i_in, i_out, alpha(i_in, i_out) ** (-0.5)) for i_in, i_out in ins]
        self.weight_numel = sum(self.irreps_in[i].mul * self.irreps_out[o].mul for i, o, _ in self.instructions)
        self.weight = torch.nn.Parameter(torch.randn(self.weight_numel))
        output_mask = torch.cat(
            [
                torch.ones(mul * ir.dim)
                if any(o == i_out for _, o, _ in self.instructions)
                else torch.zeros(mul * ir.dim)
                for i_out, (mul, ir) in enumerate(self.irreps_out)
            ]
        ) if self.irreps_out.dim > 0 else torch.ones(0)
        self.register_buffer("output_mask", output_mask)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        batch = x.shape[0]
        xs = [x[:, s].reshape(batch, mul, ir.dim) for s, (mul, ir) in zip(self.irreps_in.slices(), self.irreps_in)]
        outs: List[List[torch.Tensor]] = [[] for _ in self.irreps_out]
        flat = 0
        for i_in, i_out, pw in self.instructions:
            mi, mo = self.irreps_in[i_in].mul, self.irreps_out[i_out].mul
            w = self.weight[flat : flat + mi * mo].reshape(mi, mo)
            flat += mi * mo
            r = pw * torch.einsum("uw,zui->zwi", w, xs[i_in])
            outs[i_out].append(r.reshape(batch, -1))
        cols = []
        for i_out, mul_ir in enumerate(self.irreps_out):
            if mul_ir.dim == 0:
                continue
            if outs[i_out]:
                cols.append(functools.reduce(torch.add, outs[i_out]))
            else:
                cols.append(x.new_zeros(batch, mul_ir.dim))
        return torch.cat(cols, dim=1) if cols else x.new_zeros(batch, 0)


class ElementwiseTensorProduct(torch.nn.Module):
    """e3nn.o3.ElementwiseTensorProduct restricted to (irreps) x (scalars): used inside nn.Gate."""

    def __init__(self, irreps_in1, irreps_in2):
        super().__init__()
        irreps_in1 = Irreps(irreps_in1).simplify()
        irreps_in2 = Irreps(irreps_in2).simplify()
        assert irreps_in1.num_irreps == irreps_in2.num_irreps
        irreps_in1 = list(irreps_in1)
        irreps_in2 = list(irreps_in2)
        i = 0
        while i < len(irreps_in1):
            mul_1, ir_1 = irreps_in1[i]
            mul_2, ir_2 = irreps_in2[i]
            if mul_1 < mul_2:
                irreps_in2[i] = (mul_1, ir_2)
                irreps_in2.insert(i + 1, (mul_2 - mul_1, ir_2))
            if mul_2 < mul_1:
                irreps_in1[i] = (mul_2, ir_1)
                irreps_in1.insert(i + 1, (mul_1 - mul_2, ir_1))
            i += 1
        out = []
        for (mul, ir_1), (mul_2, ir_2) in zip(irreps_in1, irreps_in2):
            assert mul == mul_2
            assert Irrep(ir_2).l == 0, "only scalar second operand restated"
            out.append((mul, (Irrep(ir_1).l, Irrep(ir_1).p * Irrep(ir_2).p)))
        self.irreps_in1 = Irreps(irreps_in1)
        self.irreps_in2 = Irreps(irreps_in2)
        self.irreps_out = Irreps(out)

    def forward(self, x1, x2):
        batch = x1.shape[0]
        cols = []
        for s1, s2, (mul, ir) in zip(self.irreps_in1.slices(), self.irreps_in2.slices(), self.irreps_in1):
            a = x1[:, s1].reshape(batch, mul, ir.dim)
            b = x2[:, s2].reshape(batch, mul, 1)
            # 'uuu' path with component normalisation: sqrt(2l+1) * w3j(l,0,l) = identity
            cols.append((a * b).reshape(batch, -1))
        return torch.cat(cols, dim=1) if cols else x1.new_zeros(batch, 0)
