"""
Irreps algebra and constant tables for the MI355X kernels (host side, numpy fp64).

This is the product's own, minimal irreps layer: just enough to turn the reference's hyper-
parameters (irreps strings such as ``32x0o+32x0e+16x1o``) into the integer/float tables the HIP
kernels consume.  Conventions follow e3nn 0.5.1 as used by the reference:
irreps sorted by (l, p) with odd before even, blocks laid out [mul, 2l+1], real spherical
harmonics with polar axis y.  Nothing here runs per batch.
"""
from __future__ import annotations

import functools
import itertools
import math
from typing import Iterable, Iterator, List, Sequence, Tuple, Union

import numpy as np


class Irrep(tuple):
    """(l, p): degree and parity (+1 even / -1 odd)."""

    __slots__ = ()

    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                s = l.strip()
                p = {"e": 1, "o": -1}[s[-1]]
                l = int(s[:-1])
            else:
                l, p = l
        if not (isinstance(l, int) and l >= 0 and p in (1, -1)):
            raise ValueError(f"bad irrep ({l!r}, {p!r})")
        return tuple.__new__(cls, (l, p))

    l = property(lambda self: self[0])  # noqa: E741
    p = property(lambda self: self[1])
    dim = property(lambda self: 2 * self[0] + 1)

    def is_scalar(self) -> bool:
        return self == (0, 1)

    def __mul__(self, other) -> Iterator["Irrep"]:
        other = Irrep(other)
        for l in range(abs(self.l - other.l), self.l + other.l + 1):
            yield Irrep(l, self.p * other.p)

    def __repr__(self):
        return f"{self.l}{'e' if self.p == 1 else 'o'}"


class MulIr(tuple):
    __slots__ = ()

    def __new__(cls, mul, ir=None):
        if ir is None:  # MulIr((mul, ir)): the form pickle / copy re-create a tuple subclass with
            mul, ir = mul
        return tuple.__new__(cls, (int(mul), Irrep(ir)))

    mul = property(lambda self: self[0])
    ir = property(lambda self: self[1])
    dim = property(lambda self: self[0] * self[1].dim)

    def __repr__(self):
        return f"{self.mul}x{self.ir}"


class Irreps(tuple):
    """Ordered direct sum of mul x irrep blocks."""

    __slots__ = ()

    def __new__(cls, spec=None):
        if isinstance(spec, Irreps):
            return spec
        items: List[MulIr] = []
        if spec is None:
            pass
        elif isinstance(spec, Irrep):
            items.append(MulIr(1, spec))
        elif isinstance(spec, str):
            for tok in spec.split("+"):
                tok = tok.strip()
                if not tok:
                    continue
                if "x" in tok:
                    m, ir = tok.split("x")
                    items.append(MulIr(int(m), ir))
                else:
                    items.append(MulIr(1, tok))
        elif isinstance(spec, int):
            raise TypeError("use Irreps.spherical_harmonics(lmax)")
        else:
            for it in spec:
                if isinstance(it, (str, Irrep)):
                    items.append(MulIr(1, it))
                else:
                    items.append(MulIr(it[0], it[1]))
        return tuple.__new__(cls, items)

    @classmethod
    def spherical_harmonics(cls, lmax: int) -> "Irreps":
        return cls([(1, (l, (-1) ** l)) for l in range(lmax + 1)])

    # --- sizes -------------------------------------------------------------------------
    @property
    def dim(self) -> int:
        return sum(b.dim for b in self)

    @property
    def num_irreps(self) -> int:
        return sum(b.mul for b in self)

    @property
    def lmax(self) -> int:
        return max(b.ir.l for b in self)

    @property
    def ls(self) -> List[int]:
        return [b.ir.l for b in self for _ in range(b.mul)]

    def offsets(self) -> List[int]:
        out, o = [], 0
        for b in self:
            out.append(o)
            o += b.dim
        return out

    # --- algebra -----------------------------------------------------------------------
    def __contains__(self, ir) -> bool:
        ir = Irrep(ir)
        return any(b.ir == ir for b in self)

    def __add__(self, other) -> "Irreps":
        return Irreps(tuple(self) + tuple(Irreps(other)))

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return Irreps(r) if isinstance(i, slice) else r

    def simplify(self) -> "Irreps":
        out: List[Tuple[int, Irrep]] = []
        for mul, ir in self:
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + mul, ir)
            elif mul > 0:
                out.append((mul, ir))
        return Irreps(out)

    def sort(self):
        """Stable sort by (l, p).  Returns (sorted irreps, p, inv) with sorted[p[i]] = self[i]."""
        order = sorted(range(len(self)), key=lambda i: (self[i].ir, i))
        p = [0] * len(self)
        for new, old in enumerate(order):
            p[old] = new
        return Irreps([self[i] for i in order]), tuple(p), tuple(order)

    def __repr__(self):
        return "+".join(repr(b) for b in self)

    def __eq__(self, other):
        try:
            other = Irreps(other)
        except Exception:
            return NotImplemented
        return tuple.__eq__(self, other)

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = tuple.__hash__


# ------------------------------------------------------------------------------------------
# real Clebsch-Gordan tensors
# ------------------------------------------------------------------------------------------
def _cg_complex(j1: int, m1: int, j2: int, m2: int, j3: int, m3: int) -> float:
    """<j1 m1 j2 m2 | j3 m3> (Condon-Shortley), Racah's formula with exact integer factorials."""
    if m3 != m1 + m2 or not (abs(j1 - j2) <= j3 <= j1 + j2):
        return 0.0
    f = math.factorial
    pref = (2 * j3 + 1) * f(j3 + j1 - j2) * f(j3 - j1 + j2) * f(j1 + j2 - j3) / f(j1 + j2 + j3 + 1)
    pref *= f(j3 + m3) * f(j3 - m3) / (f(j1 - m1) * f(j1 + m1) * f(j2 - m2) * f(j2 + m2))
    s = 0.0
    vmin = max(-j1 + j2 + m3, -j1 + m1, 0)
    vmax = min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3)
    for v in range(vmin, vmax + 1):
        s += (-1) ** (v + j2 + m2) * (
            f(j2 + j3 + m1 - v) * f(j1 - m1 + v) / (f(v) * f(j3 - j1 + j2 - v) * f(j3 + m3 - v) * f(v + j1 - j2 - m3))
        )
    return math.sqrt(pref) * s


def _real_to_complex(l: int) -> np.ndarray:
    """Unitary taking the real SH basis used here (m=-l..l, y polar) to the complex one, times (-i)^l."""
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    r = 1 / math.sqrt(2)
    for m in range(-l, 0):
        q[l + m, l - m] = r
        q[l + m, l + m] = -1j * r
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + m] = (-1) ** m * r
        q[l + m, l - m] = 1j * (-1) ** m * r
    return (-1j) ** l * q


@functools.lru_cache(maxsize=None)
def wigner_3j(l1: int, l2: int, l3: int) -> np.ndarray:
    """Real, Frobenius-normalised coupling tensor C[i,j,k] of (l1 x l2 -> l3); fp64, read-only."""
    if not abs(l1 - l2) <= l3 <= l1 + l2:
        raise ValueError((l1, l2, l3))
    c = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            if abs(m1 + m2) <= l3:
                c[l1 + m1, l2 + m2, l3 + m1 + m2] = _cg_complex(l1, m1, l2, m2, l3, m1 + m2)
    q1, q2, q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    r = np.einsum("ij,kl,mn,ikn->jlm", q1, q2, np.conj(q3.T), c.astype(np.complex128))
    if np.abs(r.imag).max() > 1e-9:
        raise AssertionError("real CG has an imaginary part")
    r = r.real
    r = r / np.linalg.norm(r)
    r.setflags(write=False)
    return r


# ------------------------------------------------------------------------------------------
# Cartesian <-> irreps basis for symmetric Cartesian tensors (e3nn CartesianTensor semantics)
# ------------------------------------------------------------------------------------------
def _index_group(formula: str):
    parts = [(-1 if f.startswith("-") else 1, f.replace("-", "")) for f in formula.split("=")]
    f0 = parts[0][1]
    gens = {(s, tuple(f.index(c) for c in f0)) for s, f in parts}
    group = set(gens)
    while True:
        new = set(group)
        for s, p in group:
            inv = tuple(p.index(i) for i in range(len(p)))
            new.add((s, inv))
        for (s1, p1), (s2, p2) in itertools.product(group, repeat=2):
            new.add((s1 * s2, tuple(p1[p2[i]] for i in range(len(p1)))))
        if len(new) == len(group):
            return f0, group
        group = new


def _symmetric_basis(f0: str, group, d: int = 3) -> np.ndarray:
    """Orthonormal orbit basis of the tensors invariant under the signed index permutations."""
    n = len(f0)
    classes = set()
    for x in itertools.product(range(d), repeat=n):
        xs = {(s, tuple(x[i] for i in p)) for s, p in group}
        if (-1, x) not in xs:
            classes.add(frozenset({frozenset(xs), frozenset({(-s, y) for s, y in xs})}))
    base = sorted([sorted([sorted(xs) for xs in cl]) for cl in classes])
    P = np.zeros((len(base), d**n))
    for r, cl in enumerate(base):
        xs = max(cl, key=lambda xs: sum(s for s, _ in xs))
        for s, e in xs:
            j = 0
            for k in e:
                j = j * d + k
            P[r, j] = s / math.sqrt(len(xs))
    return P


def _coupling_paths(n: int):
    """All ways of coupling n vectors (1o) into irreps: list of (Irrep, basis[2l+1, 3^n]), stably sorted."""
    paths = [(Irrep(1, -1), np.eye(3))]
    for depth in range(1, n):
        nxt = []
        for ir_left, c_left in paths:
            for ir_out in ir_left * Irrep(1, -1):
                c = wigner_3j(ir_out.l, ir_left.l, 1) * math.sqrt(ir_out.dim)
                b = np.einsum("jk,ijl->ikl", c_left.reshape(ir_left.dim, -1), c)
                nxt.append((ir_out, b.reshape(ir_out.dim, -1)))
        paths = sorted(nxt, key=lambda t: t[0])
        _ = depth
    return paths


def _gram_schmidt_rows(a: np.ndarray, eps: float = 1e-9) -> np.ndarray:
    out: List[np.ndarray] = []
    for x in a:
        x = x.copy()
        for y in out:
            x -= np.dot(x, y) * y
        nrm = np.linalg.norm(x)
        if nrm > 2 * eps:
            x /= nrm
            x[np.abs(x) < eps] = 0
            x *= np.sign(x[np.nonzero(x)[0][0]])
            out.append(x)
    return np.array(out).reshape(len(out), a.shape[1])


@functools.lru_cache(maxsize=None)
def cartesian_tensor_basis(formula: str) -> Tuple[Irreps, np.ndarray]:
    """
    (irreps, Q[irreps.dim, 3, ..., 3]) with  cart = einsum('q...,bq->b...', Q, x)  and
    x = cart.flatten() @ Q.flatten(1).T : the change of basis e3nn's
    CartesianTensor(formula) / ReducedTensorProducts builds (reference utils.py:110-133).
    """
    f0, group = _index_group(formula)
    n = len(f0)
    P = _symmetric_basis(f0, group)
    by_ir = {}
    for ir, b in _coupling_paths(n):
        by_ir.setdefault(ir, []).append(b)
    blocks, irreps = [], []
    for ir, bases in by_ir.items():
        B = np.stack(bases)  # [mul, 2l+1, 3^n]
        R0 = B[:, 0]  # component 0 of every path
        A = np.block([[R0 @ R0.T, -R0 @ P.T], [-(R0 @ P.T).T, P @ P.T]])
        w, v = np.linalg.eigh(A)
        null = v[:, w < 1e-9]
        X = null[: len(bases)]  # [mul, n_solutions]
        proj = X @ X.T
        for coeff in _gram_schmidt_rows(proj):
            C = np.einsum("u,uik->ik", coeff, B)
            C *= math.sqrt(ir.dim / (C**2).sum())
            blocks.append(C)
            irreps.append((1, ir))
    Q = np.concatenate(blocks).reshape((-1,) + (3,) * n)
    Q.setflags(write=False)
    return Irreps(irreps).simplify(), Q
