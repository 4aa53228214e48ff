#!/usr/bin/env python3
"""
Generates cg_gen.h: straight-line CDNA-friendly code for every real Clebsch-Gordan contraction
(l1 x l2 -> l3, all l <= LMAX) used by the per-path tensor-product kernel.

For one edge and one channel the kernel needs
    acc[k] += sqrt(2 l3+1) * sum_ij C^{l1 l2 l3}_{ijk} * xw[i] * y[j]        (xw = w * x)
The coupling tensors are 84-97 % zeros, so each one is emitted as fully unrolled scalar code with
the non-zero coefficients as literals (they end up as SGPR/inline constants, nothing is fetched).
Two algebraically equal schedules are costed per triple and the cheaper one is emitted:
    pair form :  p = xw[i]*y[j] once per distinct (i,j);   acc[k] += c * p
    M    form :  M[i][k] = sum_j c*y[j];                    acc[k] += xw[i] * M[i][k]

Run:  python gen_cg.py > cg_gen.h      (deterministic; tests/test_host.py checks the committed file)
"""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from matten_amd.o3 import wigner_3j  # noqa: E402

LMAX = 4


def lit(c: float) -> str:
    return f"{c:.9e}f"


def triples():
    for l1 in range(LMAX + 1):
        for l2 in range(LMAX + 1):
            for l3 in range(abs(l1 - l2), min(LMAX, l1 + l2) + 1):
                yield l1, l2, l3


def emit_triple(l1, l2, l3, out):
    C = wigner_3j(l1, l2, l3) * math.sqrt(2 * l3 + 1)
    d1, d2, d3 = 2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1
    nz = [(i, j, k, float(C[i, j, k])) for i in range(d1) for j in range(d2) for k in range(d3) if abs(C[i, j, k]) > 1e-12]
    pairs = sorted({(i, j) for i, j, _, _ in nz})
    mik = sorted({(i, k) for i, _, k, _ in nz})
    cost_pair = len(pairs) + len(nz)
    cost_m = len(nz) + len(mik)
    out.append(f"// ({l1},{l2},{l3}): nnz={len(nz)} pairs={len(pairs)} nzM={len(mik)} -> "
               f"{'pair' if cost_pair <= cost_m else 'M'} form, {min(cost_pair, cost_m)} VALU ops")
    out.append(f"template <> struct CG<{l1}, {l2}, {l3}> {{")
    out.append(f"    static constexpr int NNZ = {len(nz)};")
    out.append("    static __device__ __forceinline__ void apply(const float* __restrict__ xw, "
               "const float* __restrict__ y, float* __restrict__ acc) {")
    if cost_pair <= cost_m:
        for (i, j) in pairs:
            terms = [(k, c) for (ii, jj, k, c) in nz if ii == i and jj == j]
            if len(terms) == 1 and abs(abs(terms[0][1]) - 1.0) < 1e-12:
                k, c = terms[0]
                sign = "" if c > 0 else "-"
                out.append(f"        acc[{k}] = fmaf({sign}xw[{i}], y[{j}], acc[{k}]);")
                continue
            out.append(f"        {{ const float p = xw[{i}] * y[{j}];")
            for k, c in terms:
                out.append(f"          acc[{k}] = fmaf({lit(c)}, p, acc[{k}]);")
            out.append("        }")
    else:
        for (i, k) in mik:
            terms = [(j, c) for (ii, j, kk, c) in nz if ii == i and kk == k]
            expr = f"{lit(terms[0][1])} * y[{terms[0][0]}]"
            for j, c in terms[1:]:
                expr = f"fmaf({lit(c)}, y[{j}], {expr})"
            out.append(f"        acc[{k}] = fmaf(xw[{i}], {expr}, acc[{k}]);")
    out.append("    }")
    # adjoint w.r.t. the first operand (training): t[i] += sum_jk C_ijk y[j] g[k].  With it
    #   d/dw = sum_i x[i] t[i]   and   d/dx[i] = w t[i]   for out[k] = w sum_ij C_ijk x[i] y[j]
    out.append("    static __device__ __forceinline__ void adjoint(const float* __restrict__ y, "
               "const float* __restrict__ g, float* __restrict__ t) {")
    jk = sorted({(j, k) for _, j, k, _ in nz})
    for (j, k) in jk:
        terms = [(i, c) for (i, jj, kk, c) in nz if jj == j and kk == k]
        if len(terms) == 1 and abs(abs(terms[0][1]) - 1.0) < 1e-12:
            i, c = terms[0]
            sign = "" if c > 0 else "-"
            out.append(f"        t[{i}] = fmaf({sign}y[{j}], g[{k}], t[{i}]);")
            continue
        out.append(f"        {{ const float p = y[{j}] * g[{k}];")
        for i, c in terms:
            out.append(f"          t[{i}] = fmaf({lit(c)}, p, t[{i}]);")
        out.append("        }")
    out.append("    }")
    out.append("};")
    return min(cost_pair, cost_m), len(nz)


# ---- two edges at a time (vector input blocks, l1 = 1) ----------------------------------------------------------------
# The coupling coefficients do not depend on the edge, so the pair products of TWO edges of the same (node, channel) can be
# added before they meet them:  p = xw0[i] y0[j] + xw1[i] y1[j];  acc[k] += C p  -- 2 pairs + nnz operations for two edges
# instead of 2 (pairs + nnz): -22 % / -25 % vector instructions for the two l1 = 1 kinds, whose edge steps run at the issue
# rate of a single wave (docs/LAB_NOTES.md round 5).  Emitted for l1 = 1 only (the kinds that walk two edge slots per chunk).
PAIR_SUM_L1 = (1,)


def emit_triple2(l1, l2, l3, out):
    C = wigner_3j(l1, l2, l3) * math.sqrt(2 * l3 + 1)
    d1, d2, d3 = 2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1
    nz = [(i, j, k, float(C[i, j, k])) for i in range(d1) for j in range(d2) for k in range(d3) if abs(C[i, j, k]) > 1e-12]
    pairs = sorted({(i, j) for i, j, _, _ in nz})
    out.append(f"// ({l1},{l2},{l3}) for two edges: 2 x {len(pairs)} pair operations + {len(nz)} coefficient operations")
    out.append(f"template <> struct CG2<{l1}, {l2}, {l3}> {{")
    out.append("    static __device__ __forceinline__ void apply(const float* __restrict__ xa, const float* __restrict__ ya, "
               "const float* __restrict__ xb, const float* __restrict__ yb, float* __restrict__ acc) {")
    for (i, j) in pairs:
        terms = [(k, c) for (ii, jj, k, c) in nz if ii == i and jj == j]
        if len(terms) == 1 and abs(abs(terms[0][1]) - 1.0) < 1e-12:
            k, c = terms[0]
            sign = "" if c > 0 else "-"
            out.append(f"        acc[{k}] = fmaf({sign}xa[{i}], ya[{j}], acc[{k}]);")
            out.append(f"        acc[{k}] = fmaf({sign}xb[{i}], yb[{j}], acc[{k}]);")
            continue
        out.append(f"        {{ const float p = fmaf(xb[{i}], yb[{j}], xa[{i}] * ya[{j}]);")
        for k, c in terms:
            out.append(f"          acc[{k}] = fmaf({lit(c)}, p, acc[{k}]);")
        out.append("        }")
    out.append("    }")
    out.append("};")


# l2 ranges fused into one "group" per input block: single source of truth is matten_amd/plan.py
from matten_amd.plan import TP_GROUPS as GROUPS, TP_MAX_COMBOS as MAX_COMBOS, tp_groups_hash  # noqa: E402


def emit_group(l1, gi, combos, out):
    combos = list(combos)
    lo, hi = min(l2 for l2, _ in combos), max(l2 for l2, _ in combos)   # the harmonics a lane keeps: the hull of its l2
    assert len(combos) <= MAX_COMBOS, (l1, gi, len(combos))
    d1 = 2 * l1 + 1
    offs, o = [], 0
    for (_, l3) in combos:
        offs.append(o)
        o += 2 * l3 + 1
    y0 = lo * lo
    ny = (hi + 1) ** 2 - y0
    out.append(f"// in1 l={l1}, edge harmonics l2={lo}..{hi}: {len(combos)} couplings "
               f"{' '.join('(%d,%d)' % c for c in combos)}, {o} accumulators")
    out.append(f"template <> struct Group<{l1}, {gi}> {{")
    out.append(f"    static constexpr int NC = {len(combos)}, NACC = {o}, D1 = {d1}, Y0 = {y0}, NY = {ny};")
    out.append("    static constexpr int L2[NC] = {" + ", ".join(str(c[0]) for c in combos) + "};")
    out.append("    static constexpr int L3[NC] = {" + ", ".join(str(c[1]) for c in combos) + "};")
    out.append("    static constexpr int OFF[NC] = {" + ", ".join(str(v) for v in offs) + "};")
    out.append("    // x[D1]: features of this channel; y[NY]: harmonics l2=lo..hi; w[NC]: per-coupling edge weight")
    out.append("    static __device__ __forceinline__ void apply(unsigned mask, const float* __restrict__ x, "
               "const float* __restrict__ y, const float* __restrict__ w, float* __restrict__ acc) {")
    for l2 in range(lo, hi + 1):
        members = [(c, l3, off) for c, ((l2_, l3), off) in enumerate(zip(combos, offs)) if l2_ == l2]
        if not members:
            continue
        rows = shared_rows(l1, l2) if len(members) == len(range(abs(l1 - l2), min(LMAX, l1 + l2) + 1)) else 0
        if rows:
            emit_shared_products(l1, l2, members, rows, l2 * l2 - y0, out)
            continue
        for c, l3, off in members:
            out.append(f"        if (mask & {1 << c}u) {{")
            out.append(f"            float xw[{d1}];")
            out.append(f"            _Pragma(\"unroll\") for (int i = 0; i < {d1}; ++i) xw[i] = w[{c}] * x[i];")
            out.append(f"            CG<{l1}, {l2}, {l3}>::apply(xw, y + {l2 * l2 - y0}, acc + {off});")
            out.append("        }")
    out.append("    }")
    if l1 in PAIR_SUM_L1:
        out.append("    // two edges (a, b) of the same (node, channel) in one pass: see CG2")
        out.append("    static constexpr bool HAS_APPLY2 = true;")
        out.append("    static __device__ __forceinline__ void apply2(unsigned mask, const float* __restrict__ xa, "
                   "const float* __restrict__ ya, const float* __restrict__ wa, const float* __restrict__ xb, "
                   "const float* __restrict__ yb, const float* __restrict__ wb, float* __restrict__ acc) {")
        for c, ((l2, l3), off) in enumerate(zip(combos, offs)):
            out.append(f"        if (mask & {1 << c}u) {{")
            out.append(f"            float xwa[{d1}], xwb[{d1}];")
            out.append(f"            _Pragma(\"unroll\") for (int i = 0; i < {d1}; ++i) xwa[i] = wa[{c}] * xa[i], xwb[i] = wb[{c}] * xb[i];")
            out.append(f"            CG2<{l1}, {l2}, {l3}>::apply(xwa, ya + {l2 * l2 - y0}, xwb, yb + {l2 * l2 - y0}, acc + {off});")
            out.append("        }")
        out.append("    }")
    else:
        out.append("    static constexpr bool HAS_APPLY2 = false;")
    out.append("};")


# ---- shared products (experiment: off unless MATTEN_CG_SHARED_ROWS is set) ---------------------
# Per coupling the code above costs (2 l1 + 1) + pairs + nnz operations: xw = w x, p = xw_i y_j, acc_k += C p.  The
# products x_i y_j do not depend on the coupling, only the weight does: all l3 of one (l1, l2) can share them,
#     P_ij = x_i y_j (once);    t_k = sum_ij C_ijk P_ij;    acc_k += w t_k,
# which for l1 >= 2 is 15-29 % fewer operations.  The input components are taken SHARED_ROWS at a time: P of those rows,
# then per coupling (uniform branch on its mask bit) the partial t of the rows and one fma per touched component.
SHARED_ROWS = int(os.environ.get("MATTEN_CG_SHARED_ROWS", "0"))        # 0: off
SHARED_MIN_L1 = int(os.environ.get("MATTEN_CG_SHARED_MIN_L1", "2"))
SHARED_MAX_P = int(os.environ.get("MATTEN_CG_SHARED_MAX_P", "27"))     # registers for P


def coupling_nonzeros(l1, l2, l3):
    C = wigner_3j(l1, l2, l3) * math.sqrt(2 * l3 + 1)
    return [(i, j, k, float(C[i, j, k])) for i in range(2 * l1 + 1) for j in range(2 * l2 + 1)
            for k in range(2 * l3 + 1) if abs(C[i, j, k]) > 1e-12]


def shared_rows(l1, l2):
    """rows of x per chunk for the shared-product form of (l1, l2), or 0 for the per-coupling form"""
    if not SHARED_ROWS or l1 < SHARED_MIN_L1:
        return 0
    d1, d2 = 2 * l1 + 1, 2 * l2 + 1
    l3s = range(abs(l1 - l2), min(LMAX, l1 + l2) + 1)
    if len(l3s) < 2:
        return 0
    rows = min(SHARED_ROWS, d1)
    while rows > 1 and rows * d2 > SHARED_MAX_P:
        rows -= 1
    now = 0
    for l3 in l3s:
        nz = coupling_nonzeros(l1, l2, l3)
        pairs, mik = {(i, j) for i, j, _, _ in nz}, {(i, k) for i, _, k, _ in nz}
        now += min(len(pairs) + len(nz), len(nz) + len(mik)) + d1
    new = 0
    for i0 in range(0, d1, rows):
        r = range(i0, min(d1, i0 + rows))
        used = set()
        for l3 in l3s:
            sub = [(i, j, k) for i, j, k, _ in coupling_nonzeros(l1, l2, l3) if i in r]
            used |= {(i, j) for i, j, _ in sub}
            new += len(sub) + len({k for _, _, k in sub})
        new += len(used)
    return rows if new < 0.95 * now else 0


def emit_shared_products(l1, l2, members, rows, y_off, out):
    d1 = 2 * l1 + 1
    nzs = {l3: coupling_nonzeros(l1, l2, l3) for _, l3, _ in members}
    out.append(f"        // l2 = {l2}: products x_i y_j shared by the couplings l3 = "
               f"{', '.join(str(l3) for _, l3, _ in members)}, {rows} rows of x at a time")
    for i0 in range(0, d1, rows):
        r = range(i0, min(d1, i0 + rows))
        used = sorted({(i, j) for nz in nzs.values() for i, j, _, _ in nz if i in r})
        out.append("        {")
        for (i, j) in used:
            out.append(f"            const float p{i}_{j} = x[{i}] * y[{y_off + j}];")
        for c, l3, off in members:
            sub = [(i, j, k, v) for i, j, k, v in nzs[l3] if i in r]
            if not sub:
                continue
            out.append(f"            if (mask & {1 << c}u) {{")
            for k in sorted({k for _, _, k, _ in sub}):
                terms = [(i, j, v) for i, j, kk, v in sub if kk == k]
                i, j, v = terms[0]
                expr = f"p{i}_{j} * {lit(v)}"
                for i, j, v in terms[1:]:
                    expr = f"fmaf({lit(v)}, p{i}_{j}, {expr})"
                out.append(f"                acc[{off + k}] = fmaf(w[{c}], {expr}, acc[{off + k}]);")
            out.append("            }")
        out.append("        }")


def main():
    out = [
        "// GENERATED by gen_cg.py -- do not edit.  Real Clebsch-Gordan contractions (e3nn convention,",
        "// 'component' path normalisation sqrt(2 l3+1) folded in) as unrolled literal-coefficient code.",
        "#pragma once",
        "#include <hip/hip_runtime.h>",
        "",
        "namespace matten {",
        "",
        f"constexpr int CG_LMAX = {LMAX};",
        "template <int L1, int L2, int L3> struct CG;",
        "template <int L1, int L2, int L3> struct CG2;   // the same contraction for two edges at once (l1 = 1 only)",
        "",
    ]
    tot_ops = tot_nnz = 0
    for t in triples():
        ops, nnz = emit_triple(*t, out)
        tot_ops += ops
        tot_nnz += nnz
        out.append("")
        if t[0] in PAIR_SUM_L1:
            emit_triple2(*t, out)
            out.append("")
    out.append(f"// total: {tot_nnz} non-zeros, {tot_ops} VALU ops over all triples")
    out.append("")
    out.append(f"constexpr int GROUP_MAX_COMBOS = {MAX_COMBOS};")
    out.append("constexpr int GROUP_KIND_STRIDE = 8;  // kind = l1 * GROUP_KIND_STRIDE + group index")
    out.append(f"constexpr int GROUPS_HASH = {tp_groups_hash()};  // plan.tp_groups_hash() of the lists below (matten_tp_groups_hash)")
    out.append("template <int L1, int G> struct Group;")
    out.append("#define MATTEN_FOR_EACH_GROUP(X) " + " ".join(
        f"X({l1}, {gi})" for l1, ranges in GROUPS.items() for gi in range(len(ranges))))
    out.append("")
    for l1, groups in GROUPS.items():
        for gi, combos in enumerate(groups):
            emit_group(l1, gi, combos, out)
            out.append("")
    out.append("}  // namespace matten")
    print("\n".join(out))


if __name__ == "__main__":
    main()
