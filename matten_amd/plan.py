"""
Host-side planners: turn irreps bookkeeping into the flat integer / float tables the gfx950
kernels index with (see include/matten_hip.h).  Built once per module at construction.

Each planner mirrors the instruction enumeration of the e3nn class the reference instantiates,
so parameters keep the reference's flat layout:

  plan_uvu      <- UVUTensorProduct.__init__          reference nn/utils.py:205-237
  plan_fctp     <- o3.FullyConnectedTensorProduct     reference nn/conv.py:59-61,77-79,84-86
  plan_linear   <- o3.Linear                          reference nn/nodewise.py:111
  plan_gate     <- ActivationLayer / nn.Gate          reference nn/utils.py:96-140
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .o3 import Irrep, Irreps, wigner_3j

TP_TILE_NODES = 64  # must equal matten_tp_tile_nodes() of the library (checked at first use)

# Couplings (l2, l3) of one input-block degree l1 contracted by one wave = a "group".  Schemes A-C cut by l2 RANGES; scheme D
# moves two or three couplings of l2 = 3 from the heavy second group of the l1 >= 2 blocks into the light first one: the two
# partners of a block sit in ONE workgroup and meet at a barrier every chunk, and with the range cut they are 296 / 492,
# 402 / 716 and 404 / 902 coupling operations per edge step (the lighter wave waits 12-41 % of its life, tools/tp_trace.py);
# with D 380 / 408, 462 / 656, 550 / 756: -0.75 % per forward (same-box A/B, round 5; balancing further -- 534 / 584 and 674 / 632
# at 61 accumulators in the first groups of l1 = 3, 4 -- is 1.3 % SLOWER than D).  gen_cg.py emits Group<l1, g> from the same lists (cg_gen.h is generated for ONE
# scheme; tests/test_host.py checks the committed header against the default).
def _range_groups(ranges, l1, lmax=4):
    return [[(l2, l3) for l2 in range(lo, hi + 1) for l3 in range(abs(l1 - l2), min(lmax, l1 + l2) + 1)] for lo, hi in ranges]


def _moved(l1, to_first, lmax=4):
    """scheme A's two groups of l1 with the couplings `to_first` (of l2 = 3) moved from the second into the first"""
    g0, g1 = _range_groups([(0, 2), (3, 4)], l1, lmax)
    assert all(c in g1 for c in to_first)
    return [g0 + list(to_first), [c for c in g1 if c not in to_first]]


_GROUP_SCHEMES = {
    "A": {l1: _range_groups([(0, 4)] if l1 == 0 else [(0, 2), (3, 4)], l1) for l1 in range(5)},
    "B": {l1: _range_groups([(0, 4)] if l1 == 0 else [(0, 2), (3, 4)] if l1 == 1 else [(0, 1), (2, 2), (3, 3), (4, 4)], l1)
          for l1 in range(5)},
    "C": {l1: _range_groups([(l2, l2) for l2 in range(5)], l1) for l1 in range(5)},
    "D": {0: _range_groups([(0, 4)], 0), 1: _range_groups([(0, 2), (3, 4)], 1),
          2: _moved(2, [(3, 1), (3, 2)]), 3: _moved(3, [(3, 0), (3, 1)]), 4: _moved(4, [(3, 1), (3, 2)])},
}
TP_GROUPS_SCHEME = os.environ.get("MATTEN_TP_GROUPS", "D")
# ALTERNATIVE groups (generated code like any other; kinds after the regular ones of their l1): taken INSTEAD of the regular
# groups by an input block all of whose couplings one of them holds.  They are the couplings into EVEN l3 of an odd- / even-
# parity input block -- what is left of a block when a layer's output is cut down to 0e + 2e + 4e (the last conv layer behind
# the elasticity read-out): one entry per block chunk instead of two, i.e. half the units (gathers, stage rows, matrix phase,
# barriers), and no dead weight column in the entry (the regular groups keep zero columns for absent couplings).  Measured
# and left out: ONE list for both parities of l1 = 3 so that its 2x3o + 2x3e blocks keep their merged four-lane entry (nine
# couplings, 57 accumulators: last layer 0.60 -> 0.635 ms; the pair runs unmerged on the per-parity lists instead, plan_uvu)
# and l1 = 0 (three couplings instead of five: 0.604 vs 0.610 ms).
# The lists of l1 <= 2 end with ALL couplings of degrees <= 2: the groups of an lmax-2 model (configs[3]), whose blocks otherwise
# sit in the lmax-4 lists with a third to a half of their weight columns dead and, for l1 = 2, in two entries instead of one.
_ALT_GROUPS = {
    0: [[(0, 0), (1, 1), (2, 2)]],
    1: [[(1, 0), (1, 2), (3, 2), (3, 4)], [(2, 2), (4, 4)], [(0, 1), (1, 0), (1, 1), (1, 2), (2, 1), (2, 2)]],
    2: [[(1, 2), (3, 2), (3, 4)], [(0, 2), (2, 0), (2, 2), (2, 4), (4, 2), (4, 4)],
        [(0, 2), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2)]],
    3: [[(1, 2), (1, 4), (3, 0), (3, 2), (3, 4)], [(2, 2), (2, 4), (4, 2), (4, 4)]],
    4: [[(1, 4), (3, 2), (3, 4)], [(0, 4), (2, 2), (2, 4), (4, 0), (4, 2), (4, 4)]],
}
TP_ALT_GROUPS = os.environ.get("MATTEN_TP_ALT_GROUPS", "1") != "0"
TP_GROUPS_REGULAR = {l1: len(g) for l1, g in _GROUP_SCHEMES[TP_GROUPS_SCHEME].items()}   # l1 -> number of regular groups
# l1 -> [group: [(l2, l3), ...]]: the regular groups, then the alternatives
TP_GROUPS = {l1: list(g) + (_ALT_GROUPS.get(l1, []) if TP_GROUPS_SCHEME == "D" else [])
             for l1, g in _GROUP_SCHEMES[TP_GROUPS_SCHEME].items()}


def tp_groups_hash() -> int:
    """31-bit FNV-1a of the coupling-group lists: cg_gen.h carries the value it was generated for (matten::GROUPS_HASH), the
    library exports it (matten_tp_groups_hash) and ops.tp_fused refuses a library whose generated code is for other lists --
    entries planned on one scheme and contracted by the code of another give wrong sums, silently."""
    global _TP_GROUPS_HASH
    if _TP_GROUPS_HASH is None:   # (TP_GROUPS is fixed at import; ops.tp_fused asks on every call)
        h = 0x811C9DC5
        for b in repr(sorted((l1, [list(map(tuple, g)) for g in gs]) for l1, gs in TP_GROUPS.items())).encode():
            h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
        _TP_GROUPS_HASH = h & 0x7FFFFFFF
    return _TP_GROUPS_HASH


_TP_GROUPS_HASH = None


def groups_for_block(l1: int, couplings) -> list:
    """[(group index, couplings of the group)] an input block with these (l2, l3) couplings is contracted by"""
    n_reg = TP_GROUPS_REGULAR[l1]
    couplings = set(couplings)
    if TP_ALT_GROUPS and couplings:
        cover = [gi for gi in range(n_reg, len(TP_GROUPS[l1])) if couplings <= set(TP_GROUPS[l1][gi])]
        if cover:
            gi = min(cover, key=lambda g: (len(TP_GROUPS[l1][g]), g))   # the shortest list that holds them all
            return [(gi, TP_GROUPS[l1][gi])]
    return list(enumerate(TP_GROUPS[l1][:n_reg]))
TP_MAX_COMBOS = 12
TP_MAX_COLS = int(os.environ.get("MATTEN_TP_MAX_COLS", "64"))  # == matten_tp_max_cols() of the library (-DTPF_MAX_COLS)
TP_MAX_COLS_L0 = int(os.environ.get("MATTEN_TP_MAX_COLS_L0", "96"))  # scalar input blocks (see plan_uvu); -DTPF_MAX_COLS_L0
TP_MAX_COLS_L1 = int(os.environ.get("MATTEN_TP_MAX_COLS_L1", str(TP_MAX_COLS)))  # vector input blocks; -DTPF_MAX_COLS_L1
TP_KIND_STRIDE = 8

ACT_CODE = {None: 0, "silu": 1, "tanh": 2, "sigmoid": 3, "ssp": 4, "abs": 5}


LIGHT_LMAX = 1   # scalar and vector input blocks: every chunk of a block cut at 8 channels keeps 8 lanes per node

# matten_tp_fused: two ADJACENT input blocks of equal degree and multiplicity 2 (the model's 2x3o + 2x3e) share one
# four-lane entry (16 nodes per wave, one 16-row stage, half the matrix tiles of two two-lane entries walking 32 nodes
# each).  The entry's record carries the kind with TP_KIND_MERGED set, the union of the two halves' coupling masks and
# the first half's output offsets; the record that FOLLOWS it (kind -1, never scheduled) carries the second half's mask
# and offsets and, in its w_base word, the first half's own mask.  MATTEN_TP_MERGE=0 keeps one entry per block.
TP_KIND_MERGED = 256
TP_MERGE = os.environ.get("MATTEN_TP_MERGE", "1") != "0"

FUSED_UNIT_SHARED = 1 << 24       # the unit's workgroup stages hidden features / harmonics once for its four waves
FUSED_UNIT_LOADER_ONLY = 1 << 25  # padding unit of a shared workgroup: feeds the stage, contracts nothing
FUSED_UNIT_PAIRED = 1 << 26       # shared workgroup of TWO entries on TWO consecutive node groups (waves 0,1 | 2,3)
FUSED_UNIT_REPS_SHIFT = 27        # bits 27-28: log2 of the node groups a PERSISTENT unit walks one after the other


def fused_persist(cu_log2: int, groups: int, mode: str) -> int:
    """node groups a unit of this lanes-per-node class walks in turn (a power of two dividing its groups per tile; paired
    workgroups advance two groups at a time).  MATTEN_TP_PERSIST = "lanes:reps,..." (lanes per node: 2, 4, 8, 16, 32)
    overrides the defaults."""
    spec = os.environ.get("MATTEN_TP_PERSIST", TP_PERSIST_DEFAULT)
    want = 1
    for item in filter(None, spec.split(",")):
        lanes, reps = item.split(":")
        if int(lanes) == (1 << cu_log2):
            want = int(reps)
    avail = groups // 2 if mode == "paired" else groups
    reps = 1
    while reps * 2 <= min(want, 8) and avail % (reps * 2) == 0:
        reps *= 2
    return reps


TP_PERSIST_DEFAULT = "16:4,8:2"   # measured (tools/ab_env.sh, round 5): -2.3 % per forward; 4- and 2-lane classes: no gain


def fused_workgroups(group_entries, order: Optional[str] = None):
    """How matten_tp_fused's entries are cut into workgroups: -> list of (cu_log2, entry ids, mode).
    Entries with equal lanes per node, sorted by kind (stable: kind-homogeneous workgroups where possible), are taken
    four at a time: mode "shared" (the four waves = four entries on ONE node group, hidden features / harmonics staged
    once per workgroup; fewer than four entries are padded with loader-only waves).  What is left over when the class
    has at most 16 nodes per wave goes, two entries at a time, into mode "paired": waves 0, 1 = the two entries on node
    group r, waves 2, 3 = the same entries on node group r + 1 (a lone entry leaves one loader-only wave per node
    group instead of three).  "plain": no sharing (one lane per node, or the 'entry' order of the A/B harness)."""
    order = order or os.environ.get("MATTEN_FUSED_UNIT_ORDER", "node")
    ent = np.asarray(group_entries).reshape(-1, 32)
    out = []
    classes: Dict[int, List[int]] = {}
    for e in range(len(ent)):                      # classes by lanes per node, in order of first appearance
        if int(ent[e][0]) < 0:
            continue                               # second-half record of a merged entry: not a unit
        classes.setdefault(int(ent[e][3]), []).append(e)
    for cu_log2, members in classes.items():
        npw = max(1, 64 >> cu_log2)
        run = sorted(members, key=lambda e: int(ent[e][0]) & (TP_KIND_MERGED - 1))
        if order != "node" or cu_log2 < 1:
            out.append((cu_log2, run, "plain"))
        elif npw <= 16 and os.environ.get("MATTEN_FUSED_PAIRED", "1") != "0":
            n4 = len(run) // 4 * 4
            for k in range(0, n4, 4):
                out.append((cu_log2, run[k:k + 4], "shared"))
            rest = run[n4:]
            if len(rest) == 3:
                out.append((cu_log2, rest, "shared"))
            elif rest:
                out.append((cu_log2, rest, "paired"))
        else:
            nblk = -(-len(run) // 4)
            cuts = [round(k * len(run) / nblk) for k in range(nblk + 1)]
            for k in range(nblk):
                out.append((cu_log2, run[cuts[k]:cuts[k + 1]], "shared"))
    return out


def fused_unit_map(group_entries, order: Optional[str] = None) -> np.ndarray:
    """Wave (unit) index inside a node tile -> packed (flags | entry << 8 | node group) for matten_tp_fused; four
    consecutive units are one workgroup (fused_workgroups).  Shared / paired workgroups come first, node group major."""
    ent = np.asarray(group_entries).reshape(-1, 32)
    assert len(ent) < 65536
    shared_units: List[int] = []
    plain_units: List[int] = []
    for cu_log2, blk, mode in fused_workgroups(ent, order):
        groups = -(-TP_TILE_NODES // max(1, 64 >> cu_log2))
        assert groups < 256
        reps = fused_persist(cu_log2, groups, mode) if mode in ("shared", "paired") else 1
        rbits = (reps.bit_length() - 1) << FUSED_UNIT_REPS_SHIFT
        if mode == "shared":
            for r in range(0, groups, reps):
                shared_units += [FUSED_UNIT_SHARED | rbits | (e << 8) | r for e in blk]
                shared_units += [FUSED_UNIT_SHARED | FUSED_UNIT_LOADER_ONLY | rbits | (blk[-1] << 8) | r] * (4 - len(blk))
        elif mode == "paired":
            assert groups % 2 == 0 and 1 <= len(blk) <= 2
            flags = FUSED_UNIT_SHARED | FUSED_UNIT_PAIRED | rbits
            for r in range(0, groups, 2 * reps):
                for rr in (r, r + 1):
                    shared_units += [flags | (e << 8) | rr for e in blk]
                    shared_units += [flags | FUSED_UNIT_LOADER_ONLY | (blk[-1] << 8) | rr] * (2 - len(blk))
        else:
            plain_units += [(e << 8) | r for e in blk for r in range(groups)]
    return np.array(shared_units + plain_units, dtype=np.int64).astype(np.int32)


def tp_path_exists(irreps_in1, irreps_in2, ir_out) -> bool:
    """reference nn/utils.py:358-367"""
    ir_out = Irrep(ir_out)
    for _, ir1 in Irreps(irreps_in1).simplify():
        for _, ir2 in Irreps(irreps_in2).simplify():
            if ir_out in ir1 * ir2:
                return True
    return False


# ------------------------------------------------------------------------------------------
# 'uvu' tensor product with per-edge weights
# ------------------------------------------------------------------------------------------
@dataclass
class UVUPath:
    i_in1: int
    i_sh: int
    l1: int
    l2: int
    l3: int
    p3: int
    mul: int
    x_off: int       # offset of the in1 block in the node feature row
    w_off: int       # offset of this path's weights in the per-edge weight row (instruction order)
    slot: int        # index of the output slot in the sorted irreps_mid
    out_off: int     # offset of the output slot in the message row
    m_off: int       # offset of the (l1,l2,l3) coupling matrix in the per-edge M table


@dataclass
class UVUPlan:
    irreps_in1: Irreps
    irreps_sh: Irreps
    irreps_mid: Irreps           # sorted, NOT simplified: one slot per path
    irreps_out: Irreps           # irreps_mid.simplify(): what lin2 sees
    paths: List[UVUPath]
    weight_numel: int
    d_in: int
    d_mid: int
    sh_dim: int
    m_total: int
    m_nterms: int
    m_terms_idx: np.ndarray      # uint8 [m_total, m_nterms]
    m_terms_coef: np.ndarray     # f32   [m_total, m_nterms]
    out_meta: np.ndarray         # int32 [d_mid, 4]
    cg_nnz: int = 0
    # per-path kernel (matten_tp_paths): one entry per <=64-channel chunk of a path
    path_entries: np.ndarray = None   # int32 [n_entries, 8]
    unit_start: np.ndarray = None     # int32 [n_entries + 1] waves per node tile, prefix sum
    units_per_tile: int = 0
    # block-fused kernel (matten_tp_blocks): one entry per (input block chunk, l2 group)
    group_entries: np.ndarray = None  # int32 [n_groups, 32]
    group_unit_start: np.ndarray = None
    group_units_per_tile: int = 0
    fused_cols: np.ndarray = None     # int64 [W_fused]: fused weight column -> reference weight column, -1 = zero
    fused_lds_floats_per_wave: int = 0  # LDS tile of matten_tp_fused
    fused_a_tiles: int = 0              # 16-column tiles of all entries (pre-split A operand of matten_tp_fused)
    fused_unit_map: np.ndarray = None   # int32 [fused units per tile]: unit -> flags | entry << 8 | node group
    group_entry_paths: List[Dict[int, int]] = None  # per group entry: coupling c -> index into paths
    group_entry_u0: List[int] = None                # per group entry: first channel within its input block
    group_entry_mul: List[int] = None               # per group entry record: channels of ITS input block it covers (a merged
                                                    # entry's two records: 2 and 2; otherwise the entry's mul)
    # adjoint tables (matten_tp_backward)
    bw_col_meta: np.ndarray = None    # int32 [W, 4] {x_base, out_base, nnz_begin, nnz_count | y_off << 16}
    bw_nnz_ijk: np.ndarray = None     # uint8 [nnz, 4]
    bw_nnz_c: np.ndarray = None       # f32 [nnz]
    bw_in_ptr: np.ndarray = None      # int32 [n_in+1] / bw_in_cols int32 [W]: weight columns grouped by input channel
    bw_in_cols: np.ndarray = None
    # literal-coefficient adjoint (matten_tp_backward_lit): input blocks and their path lists
    bw_blocks: np.ndarray = None      # int32 [n_blocks, 4] {x_off, mul, l1, first path | n_paths << 16}
    bw_sum_lanes: int = 0             # sum of the blocks' mul rounded up to a power of two (matten_tp_backward_lit's launch)
    bw_paths: np.ndarray = None       # int32 [n_paths, 4] {l1*25 + l2*5 + l3, w_off, out_off, first A tile (bw_w_entries)}
    bw_w_entries: np.ndarray = None   # int32 [n_paths, 32]: one pseudo group entry per path for matten_split_a_tiles (words 5, 6, 7 =
                                      # w_off, first tile, ceil(mul / 16)): the w-free adjoint's A fragments in reference column order
    bw_a_tiles: int = 0               # tiles in all of them
    bw_max_l: int = 4                 # largest degree among the paths' (l1, l2, l3): <= 2 selects the adjoint's small instantiation
    bw_wfree_lds_floats: int = 4096   # LDS floats per workgroup of matten_tp_backward_lit_wfree (its widest block in one round, capped)
    bw_max_mul: int = 0


def plan_uvu(irreps_in1, irreps_sh, irreps_target) -> UVUPlan:
    irreps_in1, irreps_sh, irreps_target = Irreps(irreps_in1), Irreps(irreps_sh), Irreps(irreps_target)
    lmax = len(irreps_sh) - 1
    if irreps_sh != Irreps.spherical_harmonics(lmax):
        raise NotImplementedError(
            f"edge attributes must be the spherical harmonics 0..lmax in order, got {irreps_sh}"
        )

    # instruction enumeration, reference nn/utils.py:205-213.  (The `or ir_out == Irreps("0e")`
    # clause there compares an Irrep with an Irreps and never fires.)
    mid: List[Tuple[int, Irrep]] = []
    raw = []
    for i, (mul, ir1) in enumerate(irreps_in1):
        for j, (_, ir2) in enumerate(irreps_sh):
            for ir_out in ir1 * ir2:
                if ir_out in irreps_target:
                    raw.append((i, j, len(mid)))
                    mid.append((mul, ir_out))
    if not mid:
        raise ValueError(f"{irreps_in1} x {irreps_sh} produces no path into {irreps_target}")
    irreps_mid, perm, _ = Irreps(mid).sort()  # reference nn/utils.py:222-228
    mid_off = irreps_mid.offsets()
    x_offs = irreps_in1.offsets()
    sh_offs = irreps_sh.offsets()

    triples: Dict[Tuple[int, int, int], int] = {}
    m_rows_idx: List[List[int]] = []
    m_rows_coef: List[List[float]] = []
    cg_nnz = 0

    def m_offset(l1, l2, l3) -> int:
        nonlocal cg_nnz
        key = (l1, l2, l3)
        if key in triples:
            return triples[key]
        off = len(m_rows_idx)
        triples[key] = off
        C = wigner_3j(l1, l2, l3) * math.sqrt(2 * l3 + 1)  # path coefficient sqrt(2 l3+1), App. A.3
        for i in range(2 * l1 + 1):
            for k in range(2 * l3 + 1):
                js = [j for j in range(2 * l2 + 1) if abs(C[i, j, k]) > 1e-12]
                cg_nnz += len(js)
                m_rows_idx.append([sh_offs[l2] + j for j in js])
                m_rows_coef.append([float(C[i, j, k]) for j in js])
        return off

    paths: List[UVUPath] = []
    w_off = 0
    for i, j, k in raw:
        mul, ir1 = irreps_in1[i]
        ir2 = irreps_sh[j].ir
        slot = perm[k]
        ir3 = irreps_mid[slot].ir
        paths.append(
            UVUPath(i, j, ir1.l, ir2.l, ir3.l, ir3.p, mul, x_offs[i], w_off, slot, mid_off[slot],
                    m_offset(ir1.l, ir2.l, ir3.l))
        )
        w_off += mul

    m_total = len(m_rows_idx)
    m_nterms = max(1, max(len(r) for r in m_rows_idx))
    idx = np.zeros((m_total, m_nterms), dtype=np.uint8)
    coef = np.zeros((m_total, m_nterms), dtype=np.float32)
    for m, (ri, rc) in enumerate(zip(m_rows_idx, m_rows_coef)):
        idx[m, : len(ri)] = ri
        coef[m, : len(rc)] = rc

    d_mid = irreps_mid.dim
    meta = np.zeros((d_mid, 4), dtype=np.int32)
    for p in paths:
        d1, d3 = 2 * p.l1 + 1, 2 * p.l3 + 1
        for u in range(p.mul):
            for k in range(d3):
                o = p.out_off + u * d3 + k
                meta[o] = (p.x_off + u * d1, p.w_off + u, p.m_off + k, d1 | (d3 << 8))
    entries, ustart = [], [0]
    for p in paths:
        d1, d3 = 2 * p.l1 + 1, 2 * p.l3 + 1
        for u0 in range(0, p.mul, 64):
            mul_c = min(64, p.mul - u0)
            cu_log2 = max(0, (mul_c - 1).bit_length())
            cu = 1 << cu_log2
            nodes_per_wave = max(1, 64 // cu)
            waves = -(-TP_TILE_NODES // nodes_per_wave)
            entries.append((p.l1 * 25 + p.l2 * 5 + p.l3, p.x_off + u0 * d1, p.w_off + u0, p.out_off + u0 * d3, mul_c,
                            cu_log2, 0, 0))
            ustart.append(ustart[-1] + waves)
    # ---- block-fused groups: all couplings of one input block and one l2 range (cg_gen.h Group<l1,g>) ----
    gentries, gstart = [], [0]
    gentry_paths: List[Dict[int, int]] = []   # per group entry: coupling index c -> index into `paths`
    gentry_u0: List[int] = []                 # per group entry: first channel of its input block
    fused_a_tiles = 0
    fused_cols: List[int] = []
    lds_need = 0
    by_block: Dict[int, List[UVUPath]] = {}
    for p in paths:
        by_block.setdefault(p.i_in1, []).append(p)
    # Scalar and vector (l1 <= 1) input blocks are the light kinds and gain from wider entries -- 16 instead of 8
    # channels share one matrix product, one staged row and one chunk's fixed costs (-6 % on the l1 = 0 kind) -- but
    # only when the entries of every lanes-per-node class fill whole workgroups of four: a workgroup padded with
    # loader-only waves costs more than the width gains (measured, tools/tp_cols_ab.sh).  Choose per layer.
    def _cap(l1, n_combos, cols0, cols1):
        limit = cols0 if l1 == 0 else cols1 if l1 == 1 else TP_MAX_COLS
        cap = 64
        while cap > 1 and cap * n_combos > limit:
            cap //= 2
        return min(cap, 16)   # (16 lanes per node at most: short alternative coupling lists would take 32)

    def _padding(cols0, cols1):
        """loader-only waves per node tile for this choice"""
        per_class: Dict[int, int] = {}
        for plist in by_block.values():
            l1_, mul_ = plist[0].l1, plist[0].mul
            for _gi, combos_ in groups_for_block(l1_, [(p_.l2, p_.l3) for p_ in plist]):
                if not any((p_.l2, p_.l3) in combos_ for p_ in plist):
                    continue
                cap_ = _cap(l1_, len(combos_), cols0, cols1)
                for u0 in range(0, mul_, cap_):
                    cu = max(1, (min(cap_, mul_ - u0) - 1).bit_length())
                    per_class[cu] = per_class.get(cu, 0) + 1
        idle = 0
        for cu, n in per_class.items():
            npw_ = max(1, 64 >> cu)
            groups_ = -(-TP_TILE_NODES // npw_)
            if npw_ <= 16:      # leftovers of two pair up over two node groups (fused_workgroups)
                idle += {0: 0, 1: 1, 2: 0, 3: 1}[n % 4] * groups_
            else:
                idle += (-n) % 4 * groups_
        return idle

    # fewest loader-only waves first, then the widest entries.  (Going NARROWER than the general width to fill the
    # workgroups of a lone 16-channel scalar block -- four 4-channel entries in the first conv layer -- was measured
    # too: 0.173 -> 0.185 ms; half-idle workgroups of 8-channel entries are still faster.)
    cands0 = sorted({TP_MAX_COLS_L0, TP_MAX_COLS}, reverse=True)
    cands1 = sorted({TP_MAX_COLS_L1, TP_MAX_COLS}, reverse=True)
    cols_l0, cols_l1 = min(((c0, c1) for c0 in cands0 for c1 in cands1),
                           key=lambda c: (_padding(*c), -c[0] - c[1]))
    gentry_mul: List[int] = []
    # adjacent blocks of equal degree and multiplicity 2 share an entry (see TP_KIND_MERGED)
    block_ids = list(by_block.keys())
    partner: Dict[int, int] = {}
    if TP_MERGE:
        k = 0
        while k + 1 < len(block_ids):
            pa, pb = by_block[block_ids[k]][0], by_block[block_ids[k + 1]][0]
            if (pa.mul == 2 and pb.mul == 2 and pa.l1 == pb.l1 and pa.l1 >= 2
                    and pb.x_off == pa.x_off + pa.mul * (2 * pa.l1 + 1)):
                partner[block_ids[k]] = block_ids[k + 1]
                k += 2
            else:
                k += 1
    if TP_ALT_GROUPS:
        # a pair whose halves are EACH covered by an alternative group runs as two plain two-lane entries on those (a merged
        # entry needs one coupling list for both parities: the regular groups, or a nine-coupling union that measured slower):
        # last conv layer 0.585 -> 0.521 ms
        for a_, b_ in list(partner.items()):
            la = by_block[a_][0].l1
            own = lambda blk: groups_for_block(la, [(p.l2, p.l3) for p in by_block[blk]])
            if all(len(own(blk)) == 1 and own(blk)[0][0] >= TP_GROUPS_REGULAR[la] for blk in (a_, b_)):
                del partner[a_]
    second = set(partner.values())

    def add_entry(l1, gi, combos, x_off, u0, mul_c, cu_log2, halves):
        """halves: [(channels of the half, {(l2, l3): path})]; one half = a plain entry"""
        nonlocal fused_a_tiles, lds_need
        d1_ = 2 * l1 + 1
        nodes_per_wave = max(1, 64 // (1 << cu_log2))
        n_tiles16 = max(1, nodes_per_wave // 16)
        # wave tile: [16 * n_tiles16 edge rows][weight columns + 4 (+ 32: units off the shared path park the
        # edge's harmonics behind the weights; sized for them so that any unit order is valid)].  Without the 32
        # a block needs 35.8 instead of 52 KB of LDS, but four blocks per CU do not pay (docs/LAB_NOTES.md round 2).
        # weight block of the entry: [u][c] over ALL couplings of the group (absent ones: zero columns; a block packed to the
        # live couplings was measured 7-9 % slower, docs/LAB_NOTES.md round 4)
        live = list(combos)
        lds_need = max(lds_need, 16 * n_tiles16 * (16 * (-(-(mul_c * len(live)) // 16)) + 32 + 4))
        n_mt = -(-(mul_c * len(live)) // 16)  # 16-column MFMA tiles of the entry's weight block
        merged = len(halves) == 2
        rows = []
        for h, (mul_h, present) in enumerate(halves):
            row = [-1, 0, mul_h, cu_log2, 0, 0, 0, 0] + [0] * 24
            mask = 0
            for c, key in enumerate(combos):
                if key in present:
                    pth = present[key]
                    mask |= 1 << c
                    row[8 + 12 + c] = pth.out_off + u0 * (2 * pth.l3 + 1)
            row[4] = mask
            rows.append(row)
        head = rows[0]
        head[0] = l1 * TP_KIND_STRIDE + gi + (TP_KIND_MERGED if merged else 0)
        head[1], head[2], head[5], head[6], head[7] = x_off + u0 * d1_, mul_c, len(fused_cols), fused_a_tiles, n_mt
        if merged:
            rows[1][5] = head[4]                 # the first half's own mask travels in the continuation record
            head[4] |= rows[1][4]                # the walk contracts the union (absent couplings: zero weight columns)
        fused_a_tiles += n_mt
        # fused weight layout of this entry: [u][live c] -> column of the reference's weight row (-1: absent in this half)
        for (mul_h, present) in halves:
            for uu in range(mul_h):
                for key in live:
                    fused_cols.append(present[key].w_off + u0 + uu if key in present else -1)
        for r_, (mul_h, present) in zip(rows, halves):
            gentries.append(r_)
            gentry_paths.append({c: paths.index(present[key]) for c, key in enumerate(combos) if key in present})
            gentry_u0.append(u0)
            gentry_mul.append(mul_h)
        gstart.append(gstart[-1] + -(-TP_TILE_NODES // nodes_per_wave))

    for i_in1 in block_ids:
        if i_in1 in second:
            continue
        plist = by_block[i_in1]
        l1, mul = plist[0].l1, plist[0].mul
        both = list(plist) + (list(by_block[partner[i_in1]]) if i_in1 in partner else [])
        for gi, combos in groups_for_block(l1, [(p.l2, p.l3) for p in both]):
            present = {(p.l2, p.l3): p for p in plist if (p.l2, p.l3) in combos}
            todo = [(plist, present)]
            if i_in1 in partner:
                plist_b = by_block[partner[i_in1]]
                present_b = {(p.l2, p.l3): p for p in plist_b if (p.l2, p.l3) in combos}
                if present and present_b and _cap(l1, len(combos), cols_l0, cols_l1) >= 4:
                    add_entry(l1, gi, combos, plist[0].x_off, 0, 4, 2, [(2, present), (2, present_b)])
                    continue
                todo.append((plist_b, present_b))
            for pl, pres in todo:
                if not pres:
                    continue
                mul_b = pl[0].mul
                # channels per entry: a power of two <= 64 whose [u][c] weight block stays <= 64 columns, so the
                # fused kernel keeps the whole A operand (4 MFMA tiles x 8 k-steps) in registers
                cap = _cap(l1, len(combos), cols_l0, cols_l1)
                for u0 in range(0, mul_b, cap):
                    mul_c = min(cap, mul_b - u0)
                    # at least two lanes per node (a multiplicity-1 entry idles one of them): at most 32 nodes, i.e. two
                    # 16-edge MFMA tiles, per wave and chunk, which is what the fused kernel's shared LDS stage holds
                    cu_log2 = max(1, (mul_c - 1).bit_length())
                    if l1 <= LIGHT_LMAX and cap == 8 and mul_b >= 8:
                        cu_log2 = 3  # (a ragged last chunk joins its block's lanes-per-node class: one workgroup shape per block)
                    add_entry(l1, gi, combos, pl[0].x_off, u0, mul_c, cu_log2, [(mul_c, pres)])
    # ---- adjoint tables: per weight column its path geometry and the non-zeros of its coupling tensor ----
    nnz_begin: Dict[Tuple[int, int, int], Tuple[int, int]] = {}
    nnz_ijk, nnz_c = [], []
    col_meta = np.zeros((w_off, 4), dtype=np.int64)
    for p in paths:
        key = (p.l1, p.l2, p.l3)
        if key not in nnz_begin:
            C = wigner_3j(*key) * math.sqrt(2 * p.l3 + 1)
            b0 = len(nnz_c)
            for i, j, k in zip(*np.nonzero(np.abs(C) > 1e-12)):
                nnz_ijk.append((i, j, k, 0))
                nnz_c.append(C[i, j, k])
            nnz_begin[key] = (b0, len(nnz_c) - b0)
        b0, cnt = nnz_begin[key]
        d1, d3 = 2 * p.l1 + 1, 2 * p.l3 + 1
        for u in range(p.mul):
            col_meta[p.w_off + u] = (p.x_off + u * d1, p.out_off + u * d3, b0, cnt | (sh_offs[p.l2] << 16))
    # literal adjoint: per input block its paths (the path list is generated block by block already)
    bw_blocks, bw_paths, bw_w_entries, bw_a_tiles = [], [], [], 0
    for i_in1, plist in by_block.items():
        bw_blocks.append((plist[0].x_off, plist[0].mul, plist[0].l1, len(bw_paths) | (len(plist) << 16)))
        assert len(plist) < 32768 and len(bw_paths) < 65536
        for pth in plist:
            n_mt = -(-pth.mul // 16)
            bw_paths.append((pth.l1 * 25 + pth.l2 * 5 + pth.l3, pth.w_off, pth.out_off, bw_a_tiles))
            ent = [0] * 32
            ent[5], ent[6], ent[7] = pth.w_off, bw_a_tiles, n_mt
            bw_w_entries.append(ent)
            bw_a_tiles += n_mt
    # weight columns grouped by the input channel they read (stable: reference column order inside a group)
    in_order = np.argsort(col_meta[:, 0], kind="stable")
    xb_sorted = col_meta[in_order, 0]
    starts = np.nonzero(np.concatenate([[True], xb_sorted[1:] != xb_sorted[:-1]]))[0]
    bw_in_ptr = np.concatenate([starts, [len(in_order)]]).astype(np.int32)
    return UVUPlan(
        bw_blocks=np.array(bw_blocks, dtype=np.int32).reshape(-1, 4), bw_paths=np.array(bw_paths, dtype=np.int32).reshape(-1, 4),
        bw_max_mul=max(b[1] for b in bw_blocks),
        bw_w_entries=np.array(bw_w_entries, dtype=np.int32).reshape(-1, 32), bw_a_tiles=bw_a_tiles,
        bw_wfree_lds_floats=_bw_wfree_lds(bw_blocks), bw_max_l=max(max(p.l1, p.l2, p.l3) for p in paths),
        bw_sum_lanes=sum(1 << max(0, (int(b[1]) - 1).bit_length()) for b in bw_blocks),
        bw_in_ptr=bw_in_ptr, bw_in_cols=in_order.astype(np.int32),
        bw_col_meta=col_meta.astype(np.int32), bw_nnz_ijk=np.array(nnz_ijk, dtype=np.uint8).reshape(-1, 4),
        bw_nnz_c=np.array(nnz_c, dtype=np.float32),
        irreps_in1=irreps_in1, irreps_sh=irreps_sh, irreps_mid=irreps_mid, irreps_out=irreps_mid.simplify(),
        paths=paths, weight_numel=w_off, d_in=irreps_in1.dim, d_mid=d_mid, sh_dim=irreps_sh.dim,
        m_total=m_total, m_nterms=m_nterms, m_terms_idx=idx, m_terms_coef=coef, out_meta=meta, cg_nnz=cg_nnz,
        path_entries=np.array(entries, dtype=np.int32), unit_start=np.array(ustart, dtype=np.int32),
        units_per_tile=ustart[-1],
        group_entries=np.array(gentries, dtype=np.int64).astype(np.int32), group_unit_start=np.array(gstart, dtype=np.int32),
        group_units_per_tile=gstart[-1], fused_cols=np.array(fused_cols, dtype=np.int64),
        fused_lds_floats_per_wave=(lds_need + 3) // 4 * 4,
        fused_unit_map=fused_unit_map(np.array(gentries, dtype=np.int64)), fused_a_tiles=fused_a_tiles,
        group_entry_paths=gentry_paths, group_entry_u0=gentry_u0, group_entry_mul=gentry_mul,
    )


# ------------------------------------------------------------------------------------------
# species-indexed linear (FCTP with one-hot second operand) and plain o3.Linear
# ------------------------------------------------------------------------------------------
@dataclass
class LinearPlan:
    irreps_in: Irreps
    irreps_out: Irreps
    n_species: int
    weight_numel: int             # size of the flat reference parameter
    w_stride: int                 # packed floats per species
    gather: np.ndarray            # int64 [n_species, w_stride]: packed <- flat parameter index
    scale: np.ndarray             # f32 [w_stride]: path normalisation
    passes: List[np.ndarray]      # each int32 [n_segs, 8] segment table (matten_species_linear); pass p>0 accumulates
    fully_covered: bool = True    # False: some output irreps have no input path and must be zero-filled
    passes_t: List[np.ndarray] = None   # adjoint w.r.t. x: segment tables with input/output swapped
    perm_t: np.ndarray = None           # int64 [w_stride]: packed^T index j -> packed index (W^T per path)
    input_covered: bool = True          # False: some input irreps feed no output (their gradient is zero)
    d_in: int = 0
    d_out: int = 0
    flops_per_row: int = 0
    instr: List[Tuple[int, int, int, int]] = None   # (i_in, i_out, offset in the flat parameter, numel), reference order


def _plan_linear_like(irreps_in: Irreps, irreps_out: Irreps, n_species: int, paths) -> LinearPlan:
    """paths: list of (i_in, i_out) in the reference's instruction order; weights (mul_in, S, mul_out)."""
    x_offs, o_offs = irreps_in.offsets(), irreps_out.offsets()
    fan = {}
    for i_in, i_out in paths:
        fan[i_out] = fan.get(i_out, 0) + irreps_in[i_in].mul * n_species
    flat = 0
    packed = 0
    gather_cols: List[np.ndarray] = []
    scale_cols: List[np.ndarray] = []
    per_out: Dict[int, List[Tuple[int, int]]] = {}
    flops = 0
    instr = []
    for i_in, i_out in paths:
        mi, mo = irreps_in[i_in].mul, irreps_out[i_out].mul
        instr.append((i_in, i_out, flat, mi * n_species * mo))
        # flat index of W[u, s, w] = flat + (u*S + s)*mo + w ; packed index = packed + u*mo + w
        u = np.arange(mi)[:, None]
        w = np.arange(mo)[None, :]
        s = np.arange(n_species)[:, None, None]
        g = flat + (u[None] * n_species + s) * mo + w[None]
        gather_cols.append(g.reshape(n_species, mi * mo))
        scale_cols.append(np.full(mi * mo, fan[i_out] ** -0.5, dtype=np.float32))
        per_out.setdefault(i_out, []).append((i_in, packed))
        flat += mi * n_species * mo
        packed += mi * mo
        flops += 2 * mi * mo * irreps_out[i_out].ir.dim
    n_pass = max([len(v) for v in per_out.values()] + [1])
    passes = []
    for ps in range(n_pass):
        segs = []
        for i_out, lst in per_out.items():
            if ps >= len(lst):
                continue
            i_in, pk = lst[ps]
            mi, mo = irreps_in[i_in].mul, irreps_out[i_out].mul
            d = irreps_out[i_out].ir.dim
            segs.append((x_offs[i_in], d, mi, pk, mo, o_offs[i_out], 0, 0))
        passes.append(np.array(segs, dtype=np.int32).reshape(-1, 8))
    # output irreps without any input path are zero (e3nn output_mask): an empty segment (mul_in = 0) makes the
    # kernel store zeros (or the addend) there itself, so the caller never has to pre-fill the output
    uncovered = [i for i in range(len(irreps_out)) if irreps_out[i].dim > 0 and i not in per_out]
    if uncovered:
        zero = np.array([(0, irreps_out[i].ir.dim, 0, 0, irreps_out[i].mul, o_offs[i], 0, 0) for i in uncovered],
                        dtype=np.int32).reshape(-1, 8)
        passes[0] = np.concatenate([passes[0], zero]) if len(passes) else zero
        if not len(passes):
            passes = [zero]
    fully_covered = True
    # adjoint w.r.t. x: dX[x_off + u*d + k] = sum_w W[u,w] dY[o_off + w*d + k]  ==  the same operator with
    # (x_off, mul_in) <-> (o_off, mo) and each path's weight block transposed
    perm_t = np.zeros(packed, dtype=np.int64)
    segs_t_by_pass: List[List[Tuple[int, ...]]] = []
    used_inputs = set()
    for i_out, lst in per_out.items():
        for ps, (i_in, pk) in enumerate(lst):
            mi, mo = irreps_in[i_in].mul, irreps_out[i_out].mul
            d = irreps_out[i_out].ir.dim
            uu, ww = np.meshgrid(np.arange(mi), np.arange(mo), indexing="ij")
            perm_t[pk + ww * mi + uu] = pk + uu * mo + ww
            # several outputs may read one input block: spread them over passes so every pass writes each x once
            slot = 0
            while True:
                if slot == len(segs_t_by_pass):
                    segs_t_by_pass.append([])
                if all(sg[5] != x_offs[i_in] for sg in segs_t_by_pass[slot]):
                    break
                slot += 1
            segs_t_by_pass[slot].append((o_offs[i_out], d, mo, pk, mi, x_offs[i_in], 0, 0))
            used_inputs.add(i_in)
    passes_t = [np.array(sg, dtype=np.int32).reshape(-1, 8) for sg in segs_t_by_pass]
    unused = [i for i in range(len(irreps_in)) if irreps_in[i].dim > 0 and i not in used_inputs]
    if unused:  # inputs that feed no output: their gradient is zero, written by empty segments of the adjoint table
        zero_t = np.array([(0, irreps_in[i].ir.dim, 0, 0, irreps_in[i].mul, x_offs[i], 0, 0) for i in unused],
                          dtype=np.int32).reshape(-1, 8)
        if passes_t:
            passes_t[0] = np.concatenate([passes_t[0], zero_t])
        else:
            passes_t = [zero_t]
    input_covered = True
    gather = np.concatenate(gather_cols, axis=1) if gather_cols else np.zeros((n_species, 0), dtype=np.int64)
    scale = np.concatenate(scale_cols) if scale_cols else np.zeros(0, dtype=np.float32)
    return LinearPlan(irreps_in, irreps_out, n_species, flat, packed, gather.astype(np.int64), scale, passes,
                      fully_covered, passes_t, perm_t, input_covered, irreps_in.dim, irreps_out.dim, flops, instr)


def linear_flat_submap(view: LinearPlan, full: LinearPlan) -> Optional[np.ndarray]:
    """Index map flat parameter of ``view`` <- flat parameter of ``full`` when every instruction of ``view`` is an
    instruction of ``full`` between the same irreps with the same multiplicities (a plan for a subset of the input /
    output irreps); None when the instructions cannot be matched one to one."""
    table = {}
    for i, o, off, n in full.instr:
        key = (full.irreps_in[i].ir, full.irreps_out[o].ir)
        if key in table:
            return None
        table[key] = (off, n, full.irreps_in[i].mul, full.irreps_out[o].mul)
    idx = []
    for i, o, off, n in view.instr:
        f = table.get((view.irreps_in[i].ir, view.irreps_out[o].ir))
        if f is None or f[1:] != (n, view.irreps_in[i].mul, view.irreps_out[o].mul):
            return None
        idx.append(np.arange(f[0], f[0] + n, dtype=np.int64))
    return np.concatenate(idx) if idx else None


def plan_fctp(irreps_in1, n_species: int, irreps_out) -> LinearPlan:
    """FullyConnectedTensorProduct(in1, f"{S}x0e", out): instruction order for i_1, (i_2,) i_out."""
    irreps_in1 = Irreps(irreps_in1).simplify()
    irreps_out = Irreps(irreps_out).simplify()
    paths = [
        (i1, io)
        for i1, (_, ir1) in enumerate(irreps_in1)
        for io, (_, iro) in enumerate(irreps_out)
        if iro == ir1
    ]
    return _plan_linear_like(irreps_in1, irreps_out, n_species, paths)


def plan_linear(irreps_in, irreps_out) -> LinearPlan:
    """o3.Linear(irreps_in, irreps_out): no simplification, instruction order for i_in, i_out."""
    irreps_in, irreps_out = Irreps(irreps_in), Irreps(irreps_out)
    paths = [
        (ii, io)
        for ii, (_, iri) in enumerate(irreps_in)
        for io, (_, iro) in enumerate(irreps_out)
        if iri == iro
    ]
    return _plan_linear_like(irreps_in, irreps_out, 1, paths)


# ------------------------------------------------------------------------------------------
# Gate (+ BatchNorm indices)
# ------------------------------------------------------------------------------------------
@dataclass
class GatePlan:
    irreps_in: Irreps            # what the preceding conv must produce (sorted, simplified)
    irreps_out: Irreps           # irreps_scalars + irreps_gated
    irreps_scalars: Irreps
    irreps_gates: Irreps
    irreps_gated: Irreps
    meta: np.ndarray             # int32 [d_out, 4]
    n_bn_features: int           # irreps_out.num_irreps
    n_bn_scalars: int            # number of 0e channels in irreps_out


def plan_gate(tp_irreps_in1, tp_irreps_in2, tp_irreps_out, act_scalars: Dict[int, str],
              act_gates: Dict[int, str]) -> GatePlan:
    """act_* map parity (+1/-1) -> activation name (keys of ACT_CODE)."""
    tp_irreps_out = Irreps(tp_irreps_out).sort()[0].simplify()
    scalars = Irreps(
        [(m, ir) for m, ir in tp_irreps_out if ir.l == 0 and tp_path_exists(tp_irreps_in1, tp_irreps_in2, ir)]
    )
    gated = Irreps(
        [(m, ir) for m, ir in tp_irreps_out if ir.l > 0 and tp_path_exists(tp_irreps_in1, tp_irreps_in2, ir)]
    )
    if gated.dim > 0:
        if tp_path_exists(tp_irreps_in1, tp_irreps_in2, "0e"):
            gir = Irrep("0e")
        elif tp_path_exists(tp_irreps_in1, tp_irreps_in2, "0o"):
            gir = Irrep("0o")
        else:
            raise ValueError(
                f"tp_irreps_in1={tp_irreps_in1} times tp_irreps_in2={tp_irreps_in2} is unable to produce gates "
                f"needed for irreps_gated={gated}"
            )
        gates = Irreps([(m, gir) for m, _ in gated]).simplify()
    else:
        gates = Irreps([])
    scalars, gates, gated = scalars.simplify(), gates.simplify(), gated.simplify()

    # input layout: stable sort of scalars + gates + gated (e3nn nn.Gate / _Sortcut)
    cat = scalars + gates + gated
    sorted_ir, p, _ = cat.sort()
    offs = sorted_ir.offsets()
    pos = [offs[p[i]] for i in range(len(cat))]
    ns, ng = len(scalars), len(gates)
    irreps_in = sorted_ir.simplify()

    rows: List[Tuple[int, int, int, int]] = []  # (src, gate, act | gate_act << 8, bn_idx | mean_idx << 16)
    out_ir: List[Tuple[int, Irrep]] = []
    bn_idx = 0
    mean_idx = 0
    for b, (m, ir) in enumerate(scalars):
        name = act_scalars[ir.p]
        ir_o = Irrep(0, _scalar_out_parity(name, ir.p))
        for u in range(m):
            mi = (mean_idx + u) if ir_o.is_scalar() else 0xFFFF
            rows.append((pos[b] + u, -1, ACT_CODE[name], (bn_idx + u) | (mi << 16)))
        if ir_o.is_scalar():
            mean_idx += m
        bn_idx += m
        out_ir.append((m, ir_o))
    gate_chan: List[Tuple[int, int]] = []  # (position, parity) of every gate channel, in order
    for b, (m, ir) in enumerate(gates):
        gate_chan += [(pos[ns + b] + u, ir.p) for u in range(m)]
    gc = 0
    for b, (m, ir) in enumerate(gated):
        d = ir.dim
        p_gate = 1
        for u in range(m):
            gpos, gp = gate_chan[gc]
            gc += 1
            gname = act_gates[gp]
            p_gate = _scalar_out_parity(gname, gp)
            for k in range(d):
                rows.append((pos[ns + ng + b] + u * d + k, gpos, ACT_CODE[gname] << 8, (bn_idx + u) | (0xFFFF << 16)))
        bn_idx += m
        out_ir.append((m, Irrep(ir.l, ir.p * p_gate)))
    irreps_out = Irreps(out_ir)
    meta = np.array(rows, dtype=np.int64).reshape(-1, 4)
    meta = np.where(meta >= 2**31, meta - 2**32, meta).astype(np.int32)
    assert meta.shape[0] == irreps_out.dim
    return GatePlan(irreps_in, irreps_out, scalars, gates, gated, meta, irreps_out.num_irreps,
                    sum(m for m, ir in irreps_out if ir.is_scalar()))


@dataclass
class NormActPlan:
    """e3nn NormActivation as the reference builds it (nn/utils.py:142-150)"""
    irreps_in: Irreps
    irreps_out: Irreps
    chan: np.ndarray      # int32 [C, 4] as plan_batchnorm
    act_code: int
    epsilon: float = 1e-8


def plan_norm_act(tp_irreps_in1, tp_irreps_in2, tp_irreps_out, act_scalars: Dict[int, str]) -> NormActPlan:
    """irreps = (scalars + gated of the reachable, sorted, simplified target).simplify(); the EVEN-scalar activation
    acts on the norm of every channel ("norm is an even scalar, so activation_scalars[1]", nn/utils.py:145-146)."""
    tp_irreps_out = Irreps(tp_irreps_out).sort()[0].simplify()
    scalars = Irreps([(m, ir) for m, ir in tp_irreps_out if ir.l == 0 and tp_path_exists(tp_irreps_in1, tp_irreps_in2, ir)])
    gated = Irreps([(m, ir) for m, ir in tp_irreps_out if ir.l > 0 and tp_path_exists(tp_irreps_in1, tp_irreps_in2, ir)])
    irreps = (scalars + gated).simplify()
    chan, _ = plan_batchnorm(irreps)
    return NormActPlan(irreps, irreps, chan, ACT_CODE[act_scalars[1]])


_ACT_PARITY = {"silu": 0, "ssp": 0, "sigmoid": 0, "tanh": -1, "abs": 1}  # even(+1) / odd(-1) / neither(0)


def _scalar_out_parity(act_name: str, p_in: int) -> int:
    """Parity of act(x) for a scalar of parity p_in (e3nn nn.Activation): even input keeps +1."""
    if p_in == 1:
        return 1
    pa = _ACT_PARITY[act_name]
    if pa == 0:
        raise ValueError(f"activation {act_name} on an odd scalar violates parity")
    return pa


def plan_batchnorm(irreps, odd_scalars_too: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """(chan[C,4] int32 {column offset, 2l+1, is_0e, bias/mean index or -1}, col2chan[dim] int32) of e3nn BatchNorm.
    odd_scalars_too: every l = 0 channel is centred and biased, 0o included -- the reference's InstanceNorm tests
    `ir.l == 0` / `d == 1` (nn/utils.py:531,572) where e3nn's BatchNorm tests `ir.is_scalar()`."""
    irreps = Irreps(irreps)
    chan, col2chan = [], []
    off = 0
    mi = 0
    for mul, ir in irreps:
        for _ in range(mul):
            is0 = 1 if (ir.l == 0 if odd_scalars_too else ir.is_scalar()) else 0
            chan.append((off, ir.dim, is0, mi if is0 else -1))
            col2chan += [len(chan) - 1] * ir.dim
            off += ir.dim
            mi += is0
    return np.array(chan, dtype=np.int32).reshape(-1, 4), np.array(col2chan, dtype=np.int32)


# ------------------------------------------------------------------------------------------
# component-major neighbour sums + the streaming lin2 that reads them (matten_agg_linear)
# ------------------------------------------------------------------------------------------
AGG_CHUNK = 16  # floats per streamed piece of a row (one 16-byte load per lane group g = 0..3)
AGG_BLOCK = int(os.environ.get("MATTEN_AGG_BLOCK", "4"))   # chunks per block of matten_agg_linear (-DAL_BLK_CHUNKS; checked by ops.agg_linear)
AGG_STAGE_W = 32  # output floats per node of one io_table row (AL_STAGE_W)
AGG_ROW_CUT = os.environ.get("MATTEN_AGG_ROW_CUT", "component")   # "channel": the round-4 rows (A/B only; re-reads chunks)
AGG_MAX_MT = 2  # 16-channel output tiles per io_table row (AL_MAX_MT): wider irreps become several rows


@dataclass
class AggLinearPlan:
    """Layout contract between matten_tp_fused (writer) and matten_agg_linear (reader) for one conv layer
    (reference nn/conv.py:113-123: agg = scatter(tp(...)); out = lin2(agg, species) + self-connection).

    The reference's message row is "mul_ir": per path [channel u][component k].  lin2 contracts over the channels of ALL
    paths with the same output irrep, separately for every component k, so the row the two kernels exchange is instead
            region(io) = [component k][channel slot 0 .. Kpad_io)        one region per lin2 output irrep io
    with the channels of every (group entry, coupling) that feeds io side by side (widest pieces first: a piece of
    2^n channels starts at a multiple of 2^n floats) and Kpad_io = K_io rounded up to AGG_CHUNK.  The tensor-product
    wave stores its channel lanes contiguously; lin2 streams the row front to back in 64-byte pieces, every piece being
    the K-slice of ONE (io, k) = one accumulator, and needs no transposition.  Pad slots are never written (the
    reader masks them)."""
    entries: np.ndarray        # int32 [n_entries, 32]: UVUPlan.group_entries with out_off[c] / t_off[c] = (first float, k stride)
    ld: int                    # row stride in floats (multiple of 32: rows start on a 128-byte line)
    n_chunks: int              # AGG_CHUNK-float pieces per row that carry data
    io_table: np.ndarray       # int32 [n_io, 8] {chunk0, T, K, d3 | n_mt << 8 | cw << 16, a_off, out_off, mo, 0}
    blocks: np.ndarray         # int32 [n_blk, 4] {first chunk, n | first << 8 | last << 9 | k << 12 | io << 20, t0, 0}
    w_stride: int              # floats per species of the A table
    gather: np.ndarray         # int64 [S, w_stride] index into the flat lin2 weight, -1 = structural zero
    scale: np.ndarray          # f32 [w_stride]
    d_out: int
    max_mt: int


AGG_GATE_SETS = 3   # == matten_agg_linear_gate_sets(): table rows of lin2 that may hold gate scalars


def _bw_wfree_lds(bw_blocks, cap: int = 4096) -> int:
    """LDS floats per workgroup of the w-free adjoint: (256 / lanes per edge) edges x (paths x columns rounded to 4, + 4) for
    the block that needs most, between 2048 (the library's floor: 256 one-channel edges x 8 floats) and `cap` (blocks that
    need more walk their paths in rounds)"""
    need = 2048
    for _x, mul, _l, w in bw_blocks:
        cu = 1 << max(0, (int(mul) - 1).bit_length())
        need = max(need, (256 // cu) * ((int(w) >> 16) * ((int(mul) + 3) // 4 * 4) + 4))
    return int(min(cap, -(-need // 64) * 64))


def plan_agg_gate(ap: "AggLinearPlan", gate: "GatePlan") -> Optional[np.ndarray]:
    """Per column of lin2's output row (= the Gate's input row) what matten_agg_linear_gate's epilogue does with it:
    cmeta[d_conv, 4] int32 {type | act << 8, column in the activated row, gate lane, gate set}; type 1 = activated scalar,
    2 = gate scalar (its activated value stays in register set `gate set` of lane `gate lane` = its position inside the
    table row that produced it), 3 = gated component (multiplied by the gate found there).  None when the layer does
    not fit: more table rows with gates than the kernel has register sets, or a gate produced after a row that needs it."""
    meta = np.asarray(gate.meta).reshape(-1, 4)
    d_conv = int(ap.d_out)
    # table row = columns out_off + v * d3 + k of (v < mo, k0 <= k < k0 + kk); position in the row's stage = v * kk + (k - k0)
    rows = [(int(r[5]), int(r[6]), int(r[3] & 255), int(r[7]), int((r[3] >> 24) & 255))
            for r in np.asarray(ap.io_table).reshape(-1, 8)]

    def row_of(col):
        for ii, (o, mo, d3, k0, kk) in enumerate(rows):
            if o <= col < o + mo * d3 and k0 <= (col - o) % d3 < k0 + kk:
                return ii, (col - o) // d3 * kk + (col - o) % d3 - k0
        return None

    cm = np.zeros((d_conv, 4), dtype=np.int64)
    gate_cols = {}
    for o, (src, gcol, codes, _bn) in enumerate(meta):
        act, gact = int(codes) & 255, (int(codes) >> 8) & 255
        if gcol < 0:
            cm[src] = (1 | (act << 8), o, 0, 0)
        else:
            gate_cols.setdefault(int(gcol), gact)
            cm[src] = (3, o, int(gcol), -1)     # gate lane / set filled in below
    set_of_row = {}
    for gcol in sorted(gate_cols):
        where = row_of(gcol)
        if where is None:
            return None
        ii, lane = where
        st = set_of_row.setdefault(ii, len(set_of_row))
        if st >= AGG_GATE_SETS or lane >= 32:
            return None
        cm[gcol] = (2 | (gate_cols[gcol] << 8), 0, lane, st)
    first_user = {}
    for src in range(d_conv):
        if cm[src][0] & 255 == 3:
            gcol = int(cm[src][2])
            g_row = row_of(gcol)[0]
            u_row = row_of(src)
            if u_row is None or u_row[0] <= g_row:
                return None          # the gate must have left the wave before the first row that uses it
            cm[src][2], cm[src][3] = cm[gcol][2], cm[gcol][3]
    if any((cm[c][0] & 255) == 0 for c in range(d_conv)):
        return None                  # (every column of a Gate input is a scalar, a gate or gated)
    return cm.astype(np.int32)


def plan_agg_linear(uvu: UVUPlan, n_species: int, irreps_out) -> Optional[AggLinearPlan]:
    """None when an output irrep of lin2 has no input path (the mul_ir path handles those layers)."""
    irreps_out = Irreps(irreps_out).simplify()
    S = n_species
    ent = np.asarray(uvu.group_entries).reshape(-1, 32).copy()
    mid = uvu.irreps_mid
    # merged input blocks of lin2 = FullyConnectedTensorProduct(irreps_mid.simplify(), Sx0e, irreps_out)
    blk_of_slot, uoff_of_slot, blocks = [], [], []
    for mul, ir in mid:
        if blocks and blocks[-1][1] == ir:
            blk_of_slot.append(len(blocks) - 1)
            uoff_of_slot.append(blocks[-1][0])
            blocks[-1] = (blocks[-1][0] + mul, ir)
        else:
            blk_of_slot.append(len(blocks))
            uoff_of_slot.append(0)
            blocks.append((mul, ir))
    flat_of, fan, flat = {}, {}, 0
    for ib, (mi, ir) in enumerate(blocks):       # instruction order of the reference: for i_1, for i_out
        for io, (mo, iro) in enumerate(irreps_out):
            if iro == ir:
                flat_of[(ib, io)] = flat
                flat += mi * S * mo
                fan[io] = fan.get(io, 0) + mi * S
    o_offs = irreps_out.offsets()
    # pieces per output irrep: (entry, coupling, channels)
    pieces: Dict[int, List[Tuple[int, int, int]]] = {}
    for e in range(len(ent)):
        for c, pi in sorted(uvu.group_entry_paths[e].items()):
            pth = uvu.paths[pi]
            ir3 = Irrep(pth.l3, pth.p3)
            ios = [io for io, (_, iro) in enumerate(irreps_out) if iro == ir3]
            if len(ios) != 1:
                return None
            pieces.setdefault(ios[0], []).append((e, c, int(uvu.group_entry_mul[e])))
    if any(io not in pieces for io in range(len(irreps_out)) if irreps_out[io].dim > 0):
        return None
    io_rows, gather_parts, scale_parts = [], [], []
    chunk0, a_off, max_mt = 0, 0, 1
    for io, (mo, iro) in enumerate(irreps_out):
        if iro.dim == 0 or mo == 0:
            continue
        d3 = iro.dim
        plist = sorted(pieces[io], key=lambda t: (-(1 << max(0, (t[2] - 1).bit_length())), t[0], t[1]))
        slot_src: List[Tuple[int, int]] = []      # channel slot -> (merged input block, channel in it)
        piece_off = {}
        for (e, c, mul_c) in plist:
            width = 1 << max(0, (mul_c - 1).bit_length())
            while len(slot_src) % min(width, AGG_CHUNK):
                slot_src.append((-1, -1))
            piece_off[(e, c)] = len(slot_src)
            pth = uvu.paths[uvu.group_entry_paths[e][c]]
            ib = blk_of_slot[pth.slot]
            for uu in range(mul_c):
                slot_src.append((ib, uoff_of_slot[pth.slot] + uvu.group_entry_u0[e] + uu))
        if any(sb < 0 for sb, _ in slot_src):
            # alignment holes inside the region (a piece whose channel count is not a power of two): nobody writes those
            # slots of agg, and the reader only masks the slots PAST the region's end -- 0 x garbage is NaN when the
            # garbage is.  Such layers keep the mul_ir row and matten_species_linear.  (Found by the model fuzzer with
            # NaN-filled buffers; every shipped configuration has power-of-two multiplicities.)
            return None
        K = len(slot_src)
        T = -(-K // AGG_CHUNK)
        Kpad = T * AGG_CHUNK
        for (e, c, _) in plist:
            ent[e][8 + 12 + c] = chunk0 * AGG_CHUNK + piece_off[(e, c)]   # out_off[c]: first float of channel 0, k = 0
            ent[e][8 + c] = Kpad                                          # t_off[c]: floats between components
        # table rows of this irrep: <= AGG_STAGE_W output floats per node each (the kernel's output stage: one 128-byte
        # line per row).  <= 16 * AGG_MAX_MT channels per row; an irrep whose channels x components exceed the stage is cut
        # by COMPONENT range [k0, k0 + kk): the (io, k) units of different components are different chunks, so no chunk is
        # read twice and the rows share one set of A tiles (cutting by channel made every row re-read the whole region:
        # +18 % of the stream for 16x1o + 16x1e outputs).  Only irreps wider than a row's channels re-read (the scalars).
        step = 16 * AGG_MAX_MT
        while AGG_ROW_CUT == "channel" and step * d3 > AGG_STAGE_W and step > 1:
            step //= 2
        for v0 in range(0, mo, step):
            mo_p = min(step, mo - v0)
            n_mt = -(-mo_p // 16)
            cw = 16 if mo_p >= 16 else mo_p
            max_mt = max(max_mt, n_mt)
            # A tiles: [t][mt][g][c < cw][s]  <-  W[channel slot 16 t + 4 g + s][v = v0 + 16 mt + c]
            t_, mt_, g_, c_, s_ = np.meshgrid(np.arange(T), np.arange(n_mt), np.arange(4), np.arange(cw), np.arange(4), indexing="ij")
            slot = (16 * t_ + 4 * g_ + s_).reshape(-1)
            v = (v0 + 16 * mt_ + c_).reshape(-1)
            src = np.array(slot_src + [(-1, -1)] * (Kpad - K), dtype=np.int64)
            ib_, u_ = src[slot, 0], src[slot, 1]
            base = np.array([flat_of.get((int(b), io), 0) for b in ib_], dtype=np.int64)
            g0 = base + u_ * S * mo + v                     # W[u, s = 0, v]; + s * mo per species
            valid = (ib_ >= 0) & (v < mo)
            gather_parts.append(np.where(valid[None, :], g0[None, :] + np.arange(S)[:, None] * mo, -1))
            scale_parts.append(np.full(slot.size, fan[io] ** -0.5, dtype=np.float32))
            kmax = max(1, AGG_STAGE_W // mo_p)
            for k0 in range(0, d3, kmax):
                kk = min(kmax, d3 - k0)
                io_rows.append((chunk0, T, K, d3 | (n_mt << 8) | (cw << 16) | (kk << 24), a_off, o_offs[io] + v0 * d3, mo_p, k0))
            a_off += slot.size
        chunk0 += d3 * T
    ld = -(-(chunk0 * AGG_CHUNK) // 32) * 32
    # the row as the reader walks it: blocks of <= AGG_BLOCK chunks of one (io, k) unit
    blocks = []
    for ii, (c0, T, K, packed, _a, _o, _m, k0) in enumerate(io_rows):
        kk = (packed >> 24) & 255
        for k in range(k0, k0 + kk):
            for t0 in range(0, T, AGG_BLOCK):
                n = min(AGG_BLOCK, T - t0)
                last = t0 + n == T
                blocks.append((c0 + k * T + t0, n | (int(t0 == 0) << 8) | (int(last) << 9)
                               | (int(last and k == k0 + kk - 1) << 10) | (k << 12) | (ii << 20), t0, 0))
    gather = np.concatenate(gather_parts, axis=1).astype(np.int64)
    assert gather.max() < flat
    return AggLinearPlan(entries=ent.astype(np.int32), ld=ld, n_chunks=chunk0,
                         io_table=np.array(io_rows, dtype=np.int32).reshape(-1, 8),
                         blocks=np.array(blocks, dtype=np.int32).reshape(-1, 4), w_stride=a_off, gather=gather,
                         scale=np.concatenate(scale_parts), d_out=irreps_out.dim, max_mt=max_mt)
