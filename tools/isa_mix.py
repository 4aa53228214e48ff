#!/usr/bin/env python3
"""ISA instruction mix of matten_tp_fused's chunk loop, per group kind (round-6 verdict item 1: "62 % of VALU issue is not CG
FMA and nobody has said what it is").

tp_fused_kernel is ONE function with ~40 inlined walks; its loops carry no names.  So every (kind, walk variant, coupling mask)
the bench model launches is compiled ALONE (-DMATTEN_LAB -DTPF_ONLY_L1/_GI/_VARIANT/_MASK, a few seconds each, no GPU needed),
disassembled, and the chunk loop (the innermost loop that holds the matrix instructions) is counted by instruction class:

    cg      fp32 arithmetic: v_fma / v_fmac / v_mul / v_add / v_sub / v_pk_*          (the contraction itself + x*w + D combine)
    mov     v_mov / v_cndmask / v_accvgpr / v_swap / v_perm / v_bfi ...                (copies, selects)
    int     integer / address / compare VALU: v_add_u32, v_lshl*, v_mad_u64_u32, v_mul_lo, v_and, v_cmp*, v_min/max_i ...
    xlane   v_readlane / v_readfirstlane / v_writelane / ds_bpermute / dpp moves
    cvt     v_cvt*
    mfma    v_mfma*
    lds     ds_read* / ds_write*
    vmem    global_* / buffer_* / flat_* / scratch_*
    salu    s_* arithmetic / moves / compares, smem = s_load*
    ctrl    s_waitcnt / s_nop / s_barrier / s_setprio / branches

beside what the ALGORITHM asks of a lane per chunk: nnz = Clebsch-Gordan non-zeros (one FMA each: the "algorithmic CG flops" of
bench.py), gen = the operations gen_cg.py's code needs for them (pair products or M rows + the FMAs; the two-slot form of the
vector blocks counted as generated) and xw = the x*w products.  Inner loops (edge slots of a chunk) are weighted by their trip
count at the bench workload.  Layer weights: units of the kind per node tile x chunks per unit (fcc-64: 18 edges per node).

    python3 tools/isa_mix.py > profiles/r06_tp_fused_isa_mix.txt        [KEEP=/tmp/isa_mix keeps objects and listings]
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter, OrderedDict
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

from isa_mfma_hazard import device_disassembly, parse_functions  # noqa: E402
from matten_amd import plan as mplan  # noqa: E402
from matten_amd.o3 import Irreps, wigner_3j  # noqa: E402

CSRC = os.path.join(ROOT, "matten_amd", "csrc")


def bench_layers():
    """[(name, UVUPlan)] of the four tensor-product launches of the bench forward: the paper model built on the CPU (plans are host
    tables), the last conv layer as inference runs it (dead-output elimination: the read-out's irreps only)"""
    from __graft_entry__ import PAPER_HPARAMS
    from matten_amd.data import synthetic
    from matten_amd.model_factory.tfn_scalar_tensor import ScalarTensorModel

    ds = {"allowed_species": list(synthetic.FCC_METALS), "average_num_neighbors": 18.0}
    model = ScalarTensorModel(backbone_hparams=dict(PAPER_HPARAMS), dataset_hparams=ds).eval()
    convs = [m for m in model.backbone.modules() if type(m).__name__ == "PointConv"]
    out = []
    for i, c in enumerate(convs):
        run = c._view if c._view is not None else c
        out.append((f"layer {i + 1}" + (" (read-out view)" if run is not c else "") + f": W {run.tp.plan.weight_numel}", run.tp.plan))
    return out


DEG = 18   # fcc-64 at 5 A


def classify(mn: str) -> str:
    if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
        return "mfma"
    if mn.startswith("ds_bpermute") or mn.startswith("ds_permute") or mn.startswith("ds_swizzle"):
        return "xlane"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith("s_load") or mn.startswith("s_buffer_load"):
        return "smem"
    if mn.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_branch", "s_cbranch", "s_endpgm", "s_sleep", "s_memtime")):
        return "ctrl"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith(("v_readlane", "v_readfirstlane", "v_writelane")) or "dpp" in mn:
        return "xlane"
    if mn.startswith("v_cvt"):
        return "cvt"
    if re.match(r"v_(fma|fmac|mul|add|sub|mac|mad|pk_fma|pk_mul|pk_add|max|min|rcp|rsq|sqrt|exp|log|sin|cos|fract|floor|trunc|rndne|ldexp)_(f32|f16|legacy_f32)", mn) \
            or mn.startswith(("v_fmaak_f32", "v_fmamk_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_dot")):
        return "cg"
    if mn.startswith(("v_mov", "v_cndmask", "v_accvgpr", "v_swap", "v_perm", "v_bfi", "v_alignbit", "v_pack", "v_pk_mov")):
        return "mov"
    if mn.startswith("v_"):
        return "int"
    return "other"


CLASSES = ["cg", "mov", "int", "xlane", "cvt", "mfma", "lds", "vmem", "salu", "smem", "ctrl"]


def natural_loops(ins):
    """basic blocks + dominators + natural loops of one function.  -> (blocks [(first, last)], loops [frozenset of block ids])
    (a backward BRANCH is not a loop: the compiler places uniform-branch targets out of order)"""
    index = {a: i for i, (a, _, _) in enumerate(ins)}
    tgt_of = {}
    leaders = {0}
    for i, (addr, mn, ops) in enumerate(ins):
        if mn.startswith("s_branch") or mn.startswith("s_cbranch"):
            off = int(ops.split()[0])
            off = off - 65536 if off >= 32768 else off
            t = index.get(addr + 4 + 4 * off)
            tgt_of[i] = t
            if t is not None:
                leaders.add(t)
            if i + 1 < len(ins):
                leaders.add(i + 1)
        elif mn in ("s_endpgm", "s_setpc_b64") and i + 1 < len(ins):
            leaders.add(i + 1)
    starts = sorted(leaders)
    blocks = [(b, (starts[k + 1] - 1) if k + 1 < len(starts) else len(ins) - 1) for k, b in enumerate(starts)]
    bid = {b: k for k, (b, _) in enumerate(blocks)}
    succ = [[] for _ in blocks]
    for k, (b, e) in enumerate(blocks):
        mn = ins[e][1]
        if mn in ("s_endpgm", "s_setpc_b64"):
            continue
        if e in tgt_of:
            if tgt_of[e] is not None:
                succ[k].append(bid[tgt_of[e]])
            if not mn.startswith("s_branch") and e + 1 < len(ins):
                succ[k].append(bid[e + 1])
        elif e + 1 < len(ins):
            succ[k].append(bid[e + 1])
    pred = [[] for _ in blocks]
    for k, ss in enumerate(succ):
        for j in ss:
            pred[j].append(k)
    n = len(blocks)
    full = (1 << n) - 1
    dom = [full] * n
    dom[0] = 1
    changed = True
    order = list(range(n))
    while changed:
        changed = False
        for k in order[1:]:
            d = full
            for p in pred[k]:
                d &= dom[p]
            d |= 1 << k
            if d != dom[k]:
                dom[k] = d
                changed = True
    by_head = {}
    for t in range(n):
        for h in succ[t]:
            if (dom[t] >> h) & 1:    # back edge t -> h
                body = {h, t}
                stack = [t]
                while stack:
                    x = stack.pop()
                    if x == h:
                        continue
                    for p in pred[x]:
                        if p not in body:
                            body.add(p)
                            stack.append(p)
                by_head.setdefault(h, set()).update(body)
    return blocks, [frozenset(b) for b in by_head.values()], dom, {frozenset(b): h for h, b in by_head.items()}


def chunk_loop_mix(ins, inner_trips):
    """instruction classes of ONE iteration of the chunk loop = the smallest natural loop that holds matrix instructions AND the
    workgroup barrier; loops inside it (the edge slots of a chunk) weighted by inner_trips.  Around it: the node-group loop of a
    persistent unit (everything of it outside the chunk loop = per node group: segment bounds, first loads, the agg stores) and the
    blocks that dominate its header (per unit: entry decode, A fragments).
    -> (Counter per class per chunk, instructions in the body, [inner loop sizes], Counter per node group, Counter per unit)"""
    blocks, loops, dom, head_of = natural_loops(ins)
    size = lambda body: sum(blocks[b][1] - blocks[b][0] + 1 for b in body)
    has = lambda body, pred: any(pred(ins[k][1]) for b in body for k in range(blocks[b][0], blocks[b][1] + 1))
    cand = [L for L in loops if has(L, lambda m: m.startswith("v_mfma")) and has(L, lambda m: m == "s_barrier")]
    if not cand:
        raise RuntimeError("no loop with matrix instructions and a barrier")
    chunk = min(cand, key=size)
    inner = [L for L in loops if L < chunk]
    inner = [L for L in inner if not any(L < L2 for L2 in inner)]
    mix = Counter()
    for b in chunk:
        w = inner_trips if any(b in L for L in inner) else 1
        for k in range(blocks[b][0], blocks[b][1] + 1):
            mix[classify(ins[k][1])] += w
    outer = [L for L in cand if chunk < L]
    group, unit = Counter(), Counter()
    if outer:
        rep = min(outer, key=size)
        for b in rep - chunk:
            for k in range(blocks[b][0], blocks[b][1] + 1):
                group[classify(ins[k][1])] += 1
        h = head_of[rep]
        for b in range(len(blocks)):
            if b != h and (dom[h] >> b) & 1:
                for k in range(blocks[b][0], blocks[b][1] + 1):
                    unit[classify(ins[k][1])] += 1
    return mix, size(chunk), [size(L) for L in inner], group, unit


_NNZ = {}


def cg_costs(l1, l2, l3):
    """(nnz, generated ops of the single-slot form) exactly as gen_cg.emit_triple prices them"""
    key = (l1, l2, l3)
    if key not in _NNZ:
        C = wigner_3j(l1, l2, l3)
        nz = [(i, j, k) for i in range(2 * l1 + 1) for j in range(2 * l2 + 1) for k in range(2 * l3 + 1) if abs(C[i, j, k]) > 1e-12]
        pairs = len({(i, j) for i, j, _ in nz})
        mik = len({(i, k) for i, _, k in nz})
        _NNZ[key] = (len(nz), min(pairs + len(nz), len(nz) + mik))
    return _NNZ[key]


def variant_of(cu, paired, l1, gi):
    npw = 64 // cu
    if paired:
        return 0
    if npw > 16:
        return 1
    if (l1 == 0 or l1 == 1) and npw <= 8:
        return 2
    return 3


VARIANT_NAME = {0: "paired", 1: "2 tiles", 2: "two-deep", 3: "plain"}
HOT = {(0, 0): (0x1f, 0xf), (0, 1): (0x7,), (1, 4): (0x3f,), (2, 4): (0x3f,), (1, 2): (0xf,), (1, 3): (0x3,), (2, 2): (0x7,), (2, 3): (0x3f,),
       (3, 2): (0x1f,), (3, 3): (0xf,), (4, 2): (0x7,), (4, 3): (0x3f,)}   # == tp_walk.h HotMask


def main():
    keep = os.environ.get("KEEP")
    work = keep or tempfile.mkdtemp(prefix="isa_mix_")
    os.makedirs(work, exist_ok=True)
    # ---- what the bench model launches ----
    rows = OrderedDict()   # (l1, gi, variant, cmask) -> {"cu", "layers": {name: (units per tile, chunks per unit)}, "masks": set}
    layer_units = {}
    LAYERS = bench_layers()
    for lname, p in LAYERS:
        ge = p.group_entries
        um = mplan.fused_unit_map(ge)
        layer_units[lname] = len(um)
        for u in um:
            u = int(u)
            if (u >> 25) & 1:
                continue     # loader-only padding unit
            e = (u >> 8) & 0xffff
            r = ge[e]
            kind = int(r[0]) & 255
            l1, gi = kind // 8, kind % 8
            cu, mask = 1 << int(r[3]), int(r[4]) & 0xffffffff
            paired = (u >> 26) & 1
            reps = 1 << ((u >> 27) & 3)
            var = variant_of(cu, paired, l1, gi)
            hot = mask in HOT.get((l1, gi), ())
            cmask = mask      # counted with the mask at compile time in every case (see the header of the table); `hot` = production does too
            npw = 64 // cu
            ch = 1 if npw >= 16 else 16 // npw
            chunks = reps * -(-DEG // ch)
            key = (l1, gi, var, cmask, int(r[7]))
            rec = rows.setdefault(key, {"cu": cu, "ch": ch, "layers": Counter(), "masks": set(), "mul": int(r[2]), "merged": bool(int(r[0]) & 256),
                                        "hot": hot, "n_mt": int(r[7])})
            rec["layers"][lname] += chunks
            rec.setdefault("units", Counter())[lname] += 1
            rec.setdefault("groups", Counter())[lname] += reps
            rec["masks"].add(mask)
    # ---- one object per row ----
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
             "-DMATTEN_LAB"]

    def build(key):
        l1, gi, var, cmask, n_mt = key
        obj = os.path.join(work, f"tpf_{l1}_{gi}_{var}_{cmask:x}_{n_mt}.o")
        subprocess.run(["hipcc"] + flags + [f"-DTPF_ONLY_L1={l1}", f"-DTPF_ONLY_GI={gi}", f"-DTPF_ONLY_VARIANT={var}", f"-DTPF_ONLY_MASK={cmask}",
                                             f"-DTPF_ONLY_MT={n_mt}", "-c", os.path.join(CSRC, "tp_fused.hip"), "-o", obj], check=True, capture_output=True)
        text = device_disassembly(obj)
        if keep:
            open(obj[:-2] + ".s", "w").write(text)
        f = parse_functions(text)
        name = next(n for n in f if "tp_fused_kernel" in n)
        return key, f[name]

    with ThreadPoolExecutor(max_workers=int(os.environ.get("JOBS", "6"))) as ex:
        listings = dict(ex.map(build, rows))
    # ---- table ----
    G = mplan.TP_GROUPS
    print("tp_fused chunk loop: VALU / LDS / memory instructions per wave and CHUNK by group kind, against the algorithm's per-lane counts")
    print("(every kind compiled alone with its coupling mask and tile count at compile time: tools/isa_mix.py; mask* = production reads that")
    print(" mask at run time -- one uniform branch per coupling, same vector instructions; `cg` = fp32 arithmetic; nnz / gen / xw = Clebsch-Gordan non-zeros,")
    print(" operations of the generated coupling code, x*w products per lane and chunk; D = the 4 FMAs per matrix tile that add the split products)")
    hdr = f"{'kind':8s} {'walk':9s} {'mask':>7s} {'lanes':>5s} {'slots':>5s} | " + " ".join(f"{c:>5s}" for c in CLASSES) + \
          f" | {'VALU':>5s} {'nnz':>5s} {'gen':>5s} {'xw':>4s} {'D':>3s} | {'nnz/VALU':>8s} {'(gen+xw+D)/VALU':>15s} {'loop':>5s} inner"
    print(hdr)
    agg = {ln: Counter() for ln, _ in LAYERS}
    for key, rec in rows.items():
        l1, gi, var, cmask, n_mt = key
        ch = rec["ch"]
        trips = ch // 2 if var == 2 else ch
        mix, n_body, inner, grp, unit = chunk_loop_mix(listings[key], trips)
        combos = G[l1][gi]
        masks = sorted(rec["masks"])
        # per lane and chunk, for the (first) mask of this row
        m0 = cmask or masks[0]
        live = [c for b, c in enumerate(combos) if (m0 >> b) & 1]
        nnz = sum(cg_costs(l1, l2, l3)[0] for l2, l3 in live) * ch
        gen = sum(cg_costs(l1, l2, l3)[1] for l2, l3 in live) * ch
        xw = (2 * l1 + 1) * len(live) * ch
        dcomb = 4 * n_mt * (2 if var == 1 else 1)
        valu = sum(mix[c] for c in ("cg", "mov", "int", "xlane", "cvt"))
        VAL = ("cg", "mov", "int", "xlane", "cvt")
        g_valu, u_valu = sum(grp[c] for c in VAL), sum(unit[c] for c in VAL)
        print(f"({l1},{gi}){'m' if rec['merged'] else ' '}   {VARIANT_NAME[var]:9s} {('%#x' % cmask) + ('' if rec['hot'] else '*'):>7s} {rec['cu']:5d} {ch:5d} | "
              + " ".join(f"{mix[c]:5.0f}" for c in CLASSES)
              + f" | {valu:5.0f} {nnz:5d} {gen:5d} {xw:4d} {dcomb:3d} | {nnz / valu:8.2f} {(gen + xw + dcomb) / valu:15.2f} {n_body:5d} {inner}"
)
        for ln, chunks in rec["layers"].items():
            for c in CLASSES:
                agg[ln][c] += mix[c] * chunks
            agg[ln]["nnz"] += nnz * chunks
            agg[ln]["gen"] += (gen + xw + dcomb) * chunks
            agg[ln]["valu"] += valu * chunks
            agg[ln]["valu_group"] += g_valu * rec["groups"][ln]
            agg[ln]["valu_unit"] += u_valu * rec["units"][ln]
            agg[ln]["vmem_group"] += grp["vmem"] * rec["groups"][ln]
    print()
    print("per node tile (64 nodes) and layer, chunk loops only (prologue / epilogue / loader-only units not included): wave instructions")
    print(f"{'layer':26s} " + " ".join(f"{c:>8s}" for c in CLASSES) + f" | {'VALU':>8s} {'nnz/VALU':>8s} {'gen/VALU':>8s}  units/tile")
    for ln, _ in LAYERS:
        a = agg[ln]
        print(f"{ln:26s} " + " ".join(f"{a[c]:8.0f}" for c in CLASSES) + f" | {a['valu']:8.0f} {a['nnz'] / a['valu']:8.2f} {a['gen'] / a['valu']:8.2f}  {layer_units[ln]:3d}   outside the chunk loops: {a['valu_group']:7.0f} VALU per node group x groups + {a['valu_unit']:6.0f} per unit x units = {(a['valu_group'] + a['valu_unit']) / a['valu']:.2f} of the loops'; {a['vmem_group']:6.0f} vmem")
    tot = Counter()
    for a in agg.values():
        tot.update(a)
    print(f"{'all four':26s} " + " ".join(f"{tot[c]:8.0f}" for c in CLASSES) + f" | {tot['valu']:8.0f} {tot['nnz'] / tot['valu']:8.2f} {tot['gen'] / tot['valu']:8.2f}        outside the chunk loops: {tot['valu_group'] + tot['valu_unit']:8.0f} VALU = {(tot['valu_group'] + tot['valu_unit']) / tot['valu']:.2f} of the loops'")
    n_tiles = 1000
    print(f"\nx {n_tiles} node tiles (1000 fcc-64 crystals): {tot['valu'] * n_tiles / 4:.3e} VALU wave instructions per launch (mean of four) in the chunk loops "
          f"+ {(tot['valu_group'] + tot['valu_unit']) * n_tiles / 4:.3e} around them (static count: every block of the node-group loop once per group); "
          f"the SQ counter of the whole kernel is in profiles/*tp_fused_valu.json")
    if not keep:
        import shutil

        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
