"""
``e3nn.io.CartesianTensor`` / ``e3nn.o3.ReducedTensorProducts`` (v0.5.1) restated [e3nn-recalled].
ORACLE / TEST INFRASTRUCTURE.

Reference call sites: utils.py:110-133 (CartesianTensorWrapper / ToCartesian),
model_factory/tfn_scalar_tensor.py:44-57, dataset/structure_scalar_tensor.py:262-267.

The change of basis Q [n_irreps_dim, 3, ..., 3] is built exactly the way e3nn builds it:
  1. germinate the permutation group of the index formula, take the orbit basis P of the
     symmetric subspace (``reduce_permutation``),
  2. enumerate every coupling path of (1o)^(x)n into irreps with successive real wigner-3j
     (``_wigner_nj``, 'component' normalisation, paths stably sorted by irrep),
  3. per irrep, find the combinations of paths that lie in the symmetric subspace (null space
     of the [[RR,-RP],[-RP^T,PP]] problem on component 0), Gram-Schmidt them in path order
     (``orthonormalize``) and rescale every basis tensor to squared norm 2l+1.
Whether this reproduces e3nn's Q entry by entry cannot be checked here (parity unpinned);
what is checked in tests: orthonormality, the index symmetries, and block-wise equivariance.
"""
from __future__ import annotations

import collections
import itertools
from typing import Dict, List, Tuple

import torch

from . import o3


def _perm_inverse(p):
    return tuple(p.index(i) for i in range(len(p)))


def _perm_compose(p1, p2):
    # p: i |-> p[i];  (p1 o p2)(i) = p1[p2[i]]
    return tuple(p1[p2[i]] for i in range(len(p1)))


def germinate_formulas(formula: str):
    formulas = [(-1 if f.startswith("-") else 1, f.replace("-", "")) for f in formula.split("=")]
    s0, f0 = formulas[0]
    assert s0 == 1
    for _s, f in formulas:
        if len(set(f)) != len(f) or set(f) != set(f0):
            raise RuntimeError(f"{f} is not a permutation of {f0}")
        if len(f0) != len(f):
            raise RuntimeError(f"{f0} and {f} don't have the same number of indices")
    formulas = {(s, tuple(f.index(i) for i in f0)) for s, f in formulas}
    while True:
        n = len(formulas)
        formulas = formulas.union([(s, _perm_inverse(p)) for s, p in formulas])
        formulas = formulas.union(
            [(s1 * s2, _perm_compose(p1, p2)) for s1, p1 in formulas for s2, p2 in formulas]
        )
        if len(formulas) == n:
            break
    return f0, formulas


def reduce_permutation(f0, formulas, dims: List[int]) -> torch.Tensor:
    full_base = list(itertools.product(*(range(d) for d in dims)))
    base = set()
    for x in full_base:
        xs = {(s, tuple(x[i] for i in p)) for s, p in formulas}
        if (-1, x) not in xs:
            base.add(frozenset({frozenset(xs), frozenset({(-s, x) for s, x in xs})}))
    base = sorted([sorted([sorted(xs) for xs in x]) for x in base])
    d_sym = len(base)
    Q = torch.zeros(d_sym, len(full_base), dtype=torch.float64)
    for i, x in enumerate(base):
        x = max(x, key=lambda xs: sum(s for s, x in xs))
        for s, e in x:
            j = 0
            for k, d in zip(e, dims):
                j *= d
                j += k
            Q[i, j] = s / len(x) ** 0.5
    return Q.reshape(d_sym, *dims)


def _wigner_nj(irrepss: List[o3.Irreps]):
    irrepss = [o3.Irreps(irreps) for irreps in irrepss]
    if len(irrepss) == 1:
        (irreps,) = irrepss
        ret = []
        e = torch.eye(irreps.dim, dtype=torch.float64)
        i = 0
        for mul, ir in irreps:
            for _ in range(mul):
                sl = slice(i, i + ir.dim)
                ret += [(ir, ("in", 0, sl.start, sl.stop), e[sl])]
                i += ir.dim
        return ret

    *irrepss_left, irreps_right = irrepss
    ret = []
    for ir_left, path_left, C_left in _wigner_nj(irrepss_left):
        i = 0
        for mul, ir in irreps_right:
            for ir_out in ir_left * ir:
                C = o3.wigner_3j(ir_out.l, ir_left.l, ir.l, dtype=torch.float64)
                C = C * ir_out.dim**0.5  # normalization == "component"
                C = torch.einsum("jk,ijl->ikl", C_left.flatten(1), C)
                C = C.reshape(ir_out.dim, *(irreps.dim for irreps in irrepss_left), ir.dim)
                for u in range(mul):
                    E = torch.zeros(
                        ir_out.dim, *(irreps.dim for irreps in irrepss_left), irreps_right.dim, dtype=torch.float64
                    )
                    sl = slice(i + u * ir.dim, i + (u + 1) * ir.dim)
                    E[..., sl] = C
                    ret += [(ir_out, ("tp", (ir_left, ir, ir_out), path_left, len(irrepss_left), sl.start, sl.stop), E)]
            i += mul * ir.dim
    return sorted(ret, key=lambda x: x[0])


def orthonormalize(original: torch.Tensor, eps: float = 1e-9):
    assert original.dim() == 2
    dim = original.shape[1]
    final = []
    matrix = []
    for i, x in enumerate(original):
        cx = x.new_zeros(len(original))
        cx[i] = 1
        for j, y in enumerate(final):
            c = torch.dot(x, y)
            x = x - c * y
            cx = cx - c * matrix[j]
        if x.norm() > 2 * eps:
            c = 1 / x.norm()
            x = c * x
            cx = c * cx
            x[x.abs() < eps] = 0
            cx[cx.abs() < eps] = 0
            c = x[x.nonzero()[0, 0]].sign()
            x = c * x
            cx = c * cx
            final += [x]
            matrix += [cx]
    final = torch.stack(final) if len(final) > 0 else original.new_zeros((0, dim))
    matrix = torch.stack(matrix) if len(matrix) > 0 else original.new_zeros((0, len(original)))
    return final, matrix


class ReducedTensorProducts:
    """Only what CartesianTensor needs: ``irreps_out`` and ``change_of_basis`` (fp64 master copy)."""

    def __init__(self, formula: str, eps: float = 1e-9, **irreps):
        f0, formulas = germinate_formulas(formula)
        irreps = {i: o3.Irreps(irs) for i, irs in irreps.items()}
        for _sign, p in formulas:
            f = "".join(f0[i] for i in p)
            for i, j in zip(f0, f):
                if i in irreps and j in irreps and irreps[i] != irreps[j]:
                    raise RuntimeError(f"irreps of {i} and {j} should be the same")
                if i in irreps:
                    irreps[j] = irreps[i]
                if j in irreps:
                    irreps[i] = irreps[j]
        for i in f0:
            if i not in irreps:
                raise RuntimeError(f"index {i} has no irreps associated to it")

        base_perm = reduce_permutation(f0, formulas, [irreps[i].dim for i in f0])

        Ps = collections.defaultdict(list)
        for ir, path, base_o3 in _wigner_nj([irreps[i] for i in f0]):
            Ps[ir].append((path, base_o3))

        change_of_basis = []
        irreps_out = []

        P = base_perm.flatten(1)  # [permutation basis, input basis]
        PP = P @ P.T

        for ir in Ps:
            mul = len(Ps[ir])
            base_o3 = torch.stack([R for _, R in Ps[ir]])
            R = base_o3.flatten(2)  # [multiplicity, ir, input basis]

            # component j = 0 only (as e3nn does)
            RR = R[:, 0] @ R[:, 0].T
            RP = R[:, 0] @ P.T
            prob = torch.cat([torch.cat([RR, -RP], dim=1), torch.cat([-RP.T, PP], dim=1)], dim=0)
            eigenvalues, eigenvectors = torch.linalg.eigh(prob)
            X = eigenvectors[:, eigenvalues < eps][:mul].T  # [solutions, multiplicity]
            proj = X.T @ X  # top-left block of the projector onto the null space (basis independent)

            X, _ = orthonormalize(proj, eps)

            for x in X:
                C = torch.einsum("u,ui...->i...", x, base_o3)
                correction = (ir.dim / C.pow(2).sum()) ** 0.5
                C = correction * C
                change_of_basis.append(C)
                irreps_out.append((1, ir))

        self.irreps_in = [irreps[i] for i in f0]
        self.irreps_out = o3.Irreps(irreps_out).simplify()
        self.change_of_basis = torch.cat(change_of_basis)  # fp64


class CartesianTensor(o3.Irreps):
    """e3nn.io.CartesianTensor: an Irreps that knows its Cartesian index formula."""

    def __new__(cls, formula: str):
        indices = formula.split("=")[0].replace("-", "")
        rtp = ReducedTensorProducts(formula, **{i: "1o" for i in indices})
        ret = super().__new__(cls, rtp.irreps_out)
        ret.formula = formula
        ret.indices = indices
        ret._rtp = rtp
        return ret

    def reduced_tensor_products(self) -> ReducedTensorProducts:
        return self._rtp

    def change_of_basis(self, dtype=None) -> torch.Tensor:
        Q = self._rtp.change_of_basis
        return Q.to(dtype or torch.get_default_dtype())

    def from_cartesian(self, data: torch.Tensor) -> torch.Tensor:
        Q = self.change_of_basis(data.dtype).to(data.device).flatten(-len(self.indices))
        return data.flatten(-len(self.indices)) @ Q.T

    def to_cartesian(self, data: torch.Tensor) -> torch.Tensor:
        Q = self.change_of_basis(data.dtype).to(data.device)
        cartesian_tensor = data @ Q.flatten(-len(self.indices))
        shape = list(data.shape[:-1]) + list(Q.shape[1:])
        return cartesian_tensor.view(shape)
