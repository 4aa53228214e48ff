"""
Host planner of the conv-tile kernel (csrc/conv_tile.hip, ``matten_conv_tile``): tensor product + neighbour sum + lin2
(+ self-connection) of one conv layer with the neighbour sums ``agg[N, d_mid]`` never leaving the chip.

  reference nn/conv.py:113-123      msg = tp(x[src], edge_attrs, edge_embedding); agg = scatter(msg, dst) / sqrt(avg)
                                    out = lin2(agg, species) + self_connection

A workgroup (4 waves) owns a TILE of 16 destination nodes of ONE species (``matten_species_tiles`` cuts the batch into
blocks of ~32 crystals and pads every (block, species) run to a multiple of 16: the x[src] gathers of a tile stay inside
a ~2 MB L2 footprint, lin2's species-indexed weights are the same for the whole tile).  It walks the layer's group
entries in ROUNDS: four entries of one lanes-per-node class on one node group of the tile -- exactly the
workgroup-shared walk of matten_tp_fused (tp_walk.h), same contraction code, same per-node summation order.  After a
round the four waves park their neighbour sums in LDS, register by register (``dump[reg][lane]``, no conflicts, in at
most two PASSES cut at coupling boundaries so that the dump fits the LDS the walk has just released), and lin2 is
applied there on the matrix cores:

    out[n, io, v, k] += sum_u  A_p[u, v] * acc_e[n, u, (coupling c, k)]        p = path (entry e, coupling c) -> io

as v_mfma_f32_16x16x4_f32 with M = 16 output channels v, K = 4 channels u, N = 16 (node, component) columns of the round's
node group: the B operand is read straight from the dump (lane (g, col) takes the channels g*KS .. g*KS+KS-1 of its
column with one ds_read), the A operand is a per-species table of ready fragments (this planner's ``gather`` / ``scale``
over the flat lin2 parameter) fetched with one coalesced load per (piece, 16-channel output tile).  A UNIT = (output
irrep, 16-channel tile, <= 4 column tiles) is owned by one wave of the round (host-balanced), which sums all the round's
pieces that end in that irrep in its accumulators and adds them to the tile's output rows in LDS: fixed order, no
atomics.  The rows start as the self-connection and leave as lin2's output (or, with the Gate tables, activated).

Entries differ from matten_tp_fused's in two ways: scalar blocks are cut so that the rounds fill (the planner tries
both widths), and two ADJACENT input blocks of equal degree and multiplicity 2 (2x3o + 2x3e) share one four-lane entry
-- the dump is indexed by register and lane, so the two halves need no output offsets of their own; a coupling absent
for one half has zero radial-weight columns and no lin2 piece.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from .o3 import Irrep, Irreps
from .plan import TP_COMPACT, TP_GROUPS, TP_KIND_STRIDE, TP_MAX_COLS, TP_MAX_COLS_L0, TP_MAX_COLS_L1, UVUPlan

TILE_NODES = 16          # == matten_conv_tile_nodes()
DUMP_RS = 68             # floats between two registers of a wave's dump (64 lanes + 4: bank spread, 16-byte aligned)
DUMP_REGS = 28           # registers per lane and pass (== matten_conv_tile_dump_regs()): the dump of four waves overlays
                         # the walk's weight tiles and stage, 4 x 28 x 68 floats = 30.5 KB
UNIT_MAX_NT = 4          # column tiles (16 (node, component) pairs each) per lin2 unit: 16 accumulator registers
OUT_PAD = 4              # the tile's output rows are (d_out rounded up to 4) + OUT_PAD floats apart


def kind_combos(l1: int, gi: int) -> List[Tuple[int, int]]:
    lo, hi = TP_GROUPS[l1][gi]
    return [(l2, l3) for l2 in range(lo, hi + 1) for l3 in range(abs(l1 - l2), min(4, l1 + l2) + 1)]


def kind_passes(l1: int, gi: int, regs: int = DUMP_REGS) -> List[int]:
    """pass index of every coupling of Group<l1, gi> (cg_gen.h): couplings in order, a new pass whenever the next
    coupling's 2 l3 + 1 registers would not fit `regs` -- the same greedy cut the kernel makes at compile time"""
    out, used, p = [], 0, 0
    for (_, l3) in kind_combos(l1, gi):
        d3 = 2 * l3 + 1
        if used + d3 > regs:
            p, used = p + 1, 0
        out.append(p)
        used += d3
    return out


@dataclass
class ConvTilePlan:
    entries: np.ndarray        # int32 [n_entries, 32] GroupEntry records (tp_walk.h); t_off / out_off unused (0)
    fused_cols: np.ndarray     # int64 [W]: weight column of these entries' [u][c] order -> reference column, -1 = zero
    a_tiles: int               # 16-column tiles of all entries (pre-split A operand of the radial layer)
    lds_floats_per_wave: int   # the walk's weight tile
    quads: np.ndarray          # int32 [n_quads, 8] {e0, e1, e2, e3 (-1: loader only), cu_log2, n_pass, n_groups, wave_unit base}
    rounds: np.ndarray         # int32 [n_rounds, 2] {quad, node group} in walking order: class by class, node group by node
                               # group, so that the rounds that stage the SAME edge rows follow each other (L2 reuse)
    wave_units: np.ndarray     # int32 [sum n_pass * 4, 2] {first unit, n units} of (quad, pass, wave)
    units: np.ndarray          # int32 [n_units, 8] {out col of (v = 16 mt, k = 0), d3, valid v, nt0, n_nt, first piece, n pieces, 0}
    pieces: np.ndarray         # int32 [n_pieces, 4] {dump offset (floats), A offset (floats per species row), lanes per node (log2) of its entry, 0}
    # the same work lists packed for the kernel's LDS copy (one 32-bit record per fragment, in the order a wave meets them)
    frag_recs: np.ndarray      # int32 [n_frags]  A offset / 64 | first register << 14 | dump wave << 19 | lanes per node (log2) << 21 | last of its unit << 24
    unit_recs: np.ndarray      # int32 [n_units, 2] {out col | d3 << 12 | valid v << 16 | nt0 << 21 | n_nt << 25, log2 nodes per wave | class lanes << 4}
    phase_recs: np.ndarray     # int32 [sum n_pass * 4, 2] {first fragment | count << 16, first unit} of (quad, pass, wave)
    a_stride: int              # floats per species of the A table
    gather: np.ndarray         # int64 [S, a_stride] index into the flat lin2 weight, -1 = structural zero
    scale: np.ndarray          # f32 [a_stride]
    d_out: int
    out_ld: int                # row stride of the tile's output rows in LDS
    n_rounds: int
    lds_bytes: int
    idle_waves: int            # loader-only wave slots per tile (planning diagnostics)
    mfma_per_tile: int


def _cap(l1: int, n_combos: int, cols0: int, cols1: int) -> int:
    limit = cols0 if l1 == 0 else cols1 if l1 == 1 else TP_MAX_COLS
    cap = 64
    while cap > 1 and cap * n_combos > limit:
        cap //= 2
    return cap


def _build_entries(uvu: UVUPlan, cols0: int, merge: bool):
    """-> (rows [n,32] int64, per-entry list of halves [(u_lo, u_hi, {coupling: path index})], fused_cols, a_tiles, lds)"""
    by_block: Dict[int, list] = {}
    for p in uvu.paths:
        by_block.setdefault(p.i_in1, []).append(p)
    blocks = sorted(by_block.items(), key=lambda kv: kv[1][0].x_off)
    # adjacent blocks of equal degree and multiplicity 2 share an entry (see the module docstring)
    groups: List[List[int]] = []
    i = 0
    while i < len(blocks):
        cur = blocks[i][1][0]
        if merge and i + 1 < len(blocks):
            nxt = blocks[i + 1][1][0]
            if (cur.mul == 2 and nxt.mul == 2 and cur.l1 == nxt.l1 and cur.l1 >= 2
                    and nxt.x_off == cur.x_off + cur.mul * (2 * cur.l1 + 1)):
                groups.append([i, i + 1])
                i += 2
                continue
        groups.append([i])
        i += 1
    rows, halves, fused_cols = [], [], []
    a_tiles, lds_need = 0, 0
    for grp in groups:
        plists = [blocks[b][1] for b in grp]
        l1 = plists[0][0].l1
        d1 = 2 * l1 + 1
        mul_blk = plists[0][0].mul
        for gi, (lo, hi) in enumerate(TP_GROUPS[l1]):
            combos = kind_combos(l1, gi)
            present = [{(p.l2, p.l3): p for p in pl if lo <= p.l2 <= hi} for pl in plists]
            if not any(present):
                continue
            cap = _cap(l1, len(combos), cols0, TP_MAX_COLS_L1)
            if len(grp) == 2:
                chunks = [(0, 4)]                       # u 0,1 = first block, 2,3 = second block
            else:
                chunks = [(u0, min(cap, mul_blk - u0)) for u0 in range(0, mul_blk, cap)]
            for (u0, mul_c) in chunks:
                cu_log2 = max(1, (mul_c - 1).bit_length())
                mask = 0
                hv = []
                if len(grp) == 2:
                    for h, pr in enumerate(present):
                        hv.append((2 * h, 2 * h + 2, {c: uvu.paths.index(pr[key]) for c, key in enumerate(combos) if key in pr}, 0))
                else:
                    hv.append((0, mul_c, {c: uvu.paths.index(present[0][key]) for c, key in enumerate(combos) if key in present[0]}, u0))
                for (_, _, cmap, _) in hv:
                    for c in cmap:
                        mask |= 1 << c
                live = [c for c in range(len(combos)) if mask >> c & 1 or not TP_COMPACT]   # weight block [u][live c] (plan.py add_entry)
                n_mt = -(-(mul_c * len(live)) // 16)
                lds_need = max(lds_need, 16 * (16 * n_mt + 4))
                row = [l1 * TP_KIND_STRIDE + gi, plists[0][0].x_off + u0 * d1, mul_c, cu_log2, mask, len(fused_cols), a_tiles,
                       n_mt] + [0] * 24
                a_tiles += n_mt
                for uu in range(mul_c):
                    h = next(hh for hh in hv if hh[0] <= uu < hh[1])
                    for c in live:
                        if c in h[2]:
                            pth = uvu.paths[h[2][c]]
                            fused_cols.append(pth.w_off + h[3] + (uu - h[0]))
                        else:
                            fused_cols.append(-1)
                rows.append(row)
                halves.append(hv)
    return np.array(rows, dtype=np.int64), halves, np.array(fused_cols, dtype=np.int64), a_tiles, (lds_need + 3) // 4 * 4


def _quads_of(rows: np.ndarray):
    """entries that can share a round, sorted by kind, four at a time -> [(class cu_log2, [entry ids])].  A round's four
    waves walk the same destination nodes with the same slots per chunk: entries with 8 or 16 lanes per node form a class
    each (8 / 4 nodes per wave, 2 / 4 slots per chunk); entries with 4 and with 2 lanes per node walk ALL 16 nodes of the
    tile one slot per chunk, so they share rounds (class 2)."""
    classes: Dict[int, List[int]] = {}
    for e in range(len(rows)):
        classes.setdefault(max(2, int(rows[e][3])), []).append(e)
    out = []
    for cu_log2, members in sorted(classes.items(), reverse=True):
        run = sorted(members, key=lambda e: (int(rows[e][0]), e))
        nq = -(-len(run) // 4)
        cuts = [round(k * len(run) / nq) for k in range(nq + 1)]
        for k in range(nq):
            out.append((cu_log2, run[cuts[k]:cuts[k + 1]]))
    return out


def plan_conv_tile(uvu: UVUPlan, n_species: int, irreps_out) -> Optional[ConvTilePlan]:
    """None when the layer does not fit the kernel (an output irrep of lin2 without an input path or matched by several
    output blocks, a multiplicity the walk cannot take, too little LDS)."""
    irreps_out = Irreps(irreps_out).simplify()
    S = n_species
    mid = uvu.irreps_mid
    # merged input blocks of lin2 = FullyConnectedTensorProduct(irreps_mid.simplify(), Sx0e, irreps_out): the flat
    # parameter is W[(block ib, output io)][u, s, v] in instruction order (for i_1, for i_out), SURVEY appendix A.4
    blk_of_slot, uoff_of_slot, mblocks = [], [], []
    for mul, ir in mid:
        if mblocks and mblocks[-1][1] == ir:
            blk_of_slot.append(len(mblocks) - 1)
            uoff_of_slot.append(mblocks[-1][0])
            mblocks[-1] = (mblocks[-1][0] + mul, ir)
        else:
            blk_of_slot.append(len(mblocks))
            uoff_of_slot.append(0)
            mblocks.append((mul, ir))
    flat_of, fan, flat = {}, {}, 0
    for ib, (mi, ir) in enumerate(mblocks):
        for io, (mo, iro) in enumerate(irreps_out):
            if iro == ir:
                flat_of[(ib, io)] = flat
                flat += mi * S * mo
                fan[io] = fan.get(io, 0) + mi * S
    io_of_ir = {}
    for io, (mo, iro) in enumerate(irreps_out):
        if iro in io_of_ir:
            return None
        io_of_ir[iro] = io
    if any(io not in fan for io in range(len(irreps_out)) if irreps_out[io].dim > 0):
        return None
    o_offs = irreps_out.offsets()

    best = None
    for cols0 in sorted({TP_MAX_COLS_L0, TP_MAX_COLS}, reverse=True):
        rows, halves, fused_cols, a_tiles, lds_wave = _build_entries(uvu, cols0, merge=os.environ.get("MATTEN_CONV_TILE_MERGE", "1") != "0")
        if len(rows) == 0:
            return None
        quads = _quads_of(rows)
        idle = sum((4 - len(q)) * max(1, TILE_NODES // max(1, 64 >> cu)) for cu, q in quads)
        # half-idle waves of the two-lanes-per-node class (32 node slots per wave, 16 nodes per tile)
        idle2 = sum(1 for e in range(len(rows)) if int(rows[e][3]) == 1)
        key = (idle, -cols0)
        if best is None or key < best[0]:
            best = (key, rows, halves, fused_cols, a_tiles, lds_wave, quads, idle + idle2)
    _, rows, halves, fused_cols, a_tiles, lds_wave, quads, idle_waves = best

    quad_rows, wave_units, units, pieces = [], [], [], []
    gather_parts: List[np.ndarray] = []
    scale_parts: List[np.ndarray] = []
    a_off = 0
    n_rounds = 0
    mfma = 0
    for cu_log2, members in quads:
        npw = min(TILE_NODES, 64 >> cu_log2)          # nodes of the tile a wave covers (class 2: all 16)
        npw_log2 = npw.bit_length() - 1
        n_groups = TILE_NODES // npw
        kinds = [(int(rows[e][0]) // TP_KIND_STRIDE, int(rows[e][0]) % TP_KIND_STRIDE) for e in members]
        passes = [kind_passes(*k) for k in kinds]
        n_pass = max(max(p) for p in passes) + 1
        quad_rows.append(list(members) + [-1] * (4 - len(members)) + [cu_log2, n_pass, n_groups, len(wave_units)])
        n_rounds += n_groups
        for ps in range(n_pass):
            # pieces of this pass, by output irrep
            by_io: Dict[int, List[Tuple[int, int, int, int, int]]] = {}   # io -> [(wave, rel reg, half index, path, coupling)]
            for w, e in enumerate(members):
                combos = kind_combos(*kinds[w])
                rel = 0
                for c, (l2, l3) in enumerate(combos):
                    if passes[w][c] != ps:
                        continue
                    for h, (u_lo, u_hi, cmap, _) in enumerate(halves[e]):
                        if c in cmap:
                            pth = uvu.paths[cmap[c]]
                            io = io_of_ir.get(Irrep(pth.l3, pth.p3))
                            if io is None:
                                return None
                            by_io.setdefault(io, []).append((w, rel, h, cmap[c], c))
                    rel += 2 * l3 + 1
                if rel > DUMP_REGS:
                    return None
            # units: (io, 16-channel output tile, <= UNIT_MAX_NT column tiles); greedy balance over the four waves
            cand = []
            for io, plist in sorted(by_io.items()):
                mo, iro = irreps_out[io]
                d3 = iro.dim
                n_nt = -(-(npw * d3) // 16)
                for mt in range(-(-mo // 16)):
                    for nt0 in range(0, n_nt, UNIT_MAX_NT):
                        nn = min(UNIT_MAX_NT, n_nt - nt0)
                        cand.append((sum(max(1, (1 << int(rows[members[pw]][3])) // 4) for pw, *_ in plist) * nn + 2 * nn, io, mt, nt0, nn))
            load = [0, 0, 0, 0]
            per_wave: List[List[Tuple[int, int, int, int]]] = [[], [], [], []]
            for cost, io, mt, nt0, nn in sorted(cand, reverse=True):
                w = load.index(min(load))
                load[w] += cost
                per_wave[w].append((io, mt, nt0, nn))
            a_of: Dict[Tuple[int, int, int, int, int], int] = {}          # (io, mt, wave, coupling, half) -> A offset: shared by the nt splits
            for w in range(4):
                wave_units.append((len(units), len(per_wave[w])))
                for (io, mt, nt0, nn) in per_wave[w]:
                    mo, iro = irreps_out[io]
                    d3 = iro.dim
                    first_piece = len(pieces)
                    for (pw, rel, h, pi, c) in by_io[io]:
                        e = members[pw]
                        e_cu_log2 = int(rows[e][3])
                        ksv = max(1, (1 << e_cu_log2) // 4)   # channels per lane group g and contraction step
                        u_lo, u_hi, _, u0 = halves[e][h]
                        keyA = (io, mt, pw, c, h)
                        if keyA not in a_of:
                            a_of[keyA] = a_off
                            pth = uvu.paths[pi]
                            ib = blk_of_slot[pth.slot]
                            # A fragment [lane = (g, cc)][t < ksv] <- W[u, v]: u = entry channel g*ksv + t, v = 16 mt + cc
                            g_, c_, t_ = np.meshgrid(np.arange(4), np.arange(16), np.arange(ksv), indexing="ij")
                            ue = (g_ * ksv + t_).reshape(-1)
                            v = (16 * mt + c_).reshape(-1)
                            inside = (ue >= u_lo) & (ue < u_hi) & (v < mo)
                            ublk = uoff_of_slot[pth.slot] + u0 + (ue - u_lo)
                            g0 = flat_of[(ib, io)] + ublk * S * mo + v
                            gather_parts.append(np.where(inside[None, :], g0[None, :] + np.arange(S)[:, None] * mo, -1))
                            scale_parts.append(np.full(ue.size, fan[io] ** -0.5, dtype=np.float32))
                            a_off += ue.size
                        pieces.append((pw * DUMP_REGS * DUMP_RS + rel * DUMP_RS, a_of[keyA], e_cu_log2, 0))
                        mfma += ksv * nn * n_groups
                    units.append((o_offs[io] + 16 * mt * d3, d3, min(16, mo - 16 * mt), nt0, nn, first_piece,
                                  len(pieces) - first_piece, npw_log2 | (cu_log2 << 8)))
    rounds = []
    for cls in sorted({int(q[4]) for q in quad_rows}, reverse=True):
        qs = [qi for qi, q in enumerate(quad_rows) if int(q[4]) == cls]
        for r in range(int(quad_rows[qs[0]][6])):
            rounds += [(qi, r) for qi in qs]
    assert len(rounds) == n_rounds
    gather = np.concatenate(gather_parts, axis=1).astype(np.int64) if gather_parts else np.zeros((S, 0), np.int64)
    assert gather.size == 0 or gather.max() < flat
    d_out = irreps_out.dim
    out_ld = -(-d_out // 4) * 4 + OUT_PAD
    # LDS: max(walk: 4 weight tiles + double-buffered 16-row stage, dump) + output rows + node ids
    walk = 4 * lds_wave + 2 * 16 * 68
    dump = 4 * DUMP_REGS * DUMP_RS
    lds_bytes = 4 * (max(walk, dump) + TILE_NODES * out_ld + 32)
    if lds_bytes > 64 * 1024:
        return None
    # packed records: a wave's fragments of a (quad, pass), unit by unit
    frag_recs, unit_recs, phase_recs = [], [], []
    for (u0, nu) in wave_units:
        phase_recs.append((len(frag_recs), len(unit_recs)))
        n0 = len(frag_recs)
        for (col0, d3, vcount, nt0, nn, p0, npc, packed) in units[u0:u0 + nu]:
            assert col0 < 4096 and d3 < 16 and vcount <= 16 and nt0 < 16 and nn <= UNIT_MAX_NT and npc >= 1
            unit_recs.append((col0 | (d3 << 12) | (vcount << 16) | (nt0 << 21) | (nn << 25), (packed & 255) | ((packed >> 8) << 4)))
            for k, (doff, aoff, cul, _) in enumerate(pieces[p0:p0 + npc]):
                wv, rel = doff // (DUMP_REGS * DUMP_RS), (doff % (DUMP_REGS * DUMP_RS)) // DUMP_RS
                assert aoff % 64 == 0 and aoff // 64 < (1 << 14) and rel < 32 and wv < 4 and cul < 8
                frag_recs.append((aoff // 64) | (rel << 14) | (wv << 19) | (cul << 21) | (int(k == npc - 1) << 24))
        nf = len(frag_recs) - n0
        assert n0 < 65536 and nf < 65536
        phase_recs[-1] = (n0 | (nf << 16), phase_recs[-1][1])
    lds_bytes += 4 * (len(frag_recs) + 2 * len(unit_recs) + 2 * len(phase_recs))
    if lds_bytes > 64 * 1024:
        return None
    return ConvTilePlan(
        frag_recs=np.array(frag_recs, dtype=np.int64).astype(np.int32), unit_recs=np.array(unit_recs, dtype=np.int64).astype(np.int32).reshape(-1, 2),
        phase_recs=np.array(phase_recs, dtype=np.int64).astype(np.int32).reshape(-1, 2),
        entries=rows.astype(np.int32), fused_cols=fused_cols, a_tiles=a_tiles, lds_floats_per_wave=lds_wave,
        quads=np.array(quad_rows, dtype=np.int32).reshape(-1, 8), rounds=np.array(rounds, dtype=np.int32).reshape(-1, 2),
        wave_units=np.array(wave_units, dtype=np.int32).reshape(-1, 2),
        units=np.array(units, dtype=np.int32).reshape(-1, 8), pieces=np.array(pieces, dtype=np.int32).reshape(-1, 4),
        a_stride=a_off, gather=gather, scale=np.concatenate(scale_parts) if scale_parts else np.zeros(0, np.float32),
        d_out=d_out, out_ld=out_ld, n_rounds=n_rounds, lds_bytes=lds_bytes, idle_waves=idle_waves, mfma_per_tile=mfma)


def emulate_lin2(plan: ConvTilePlan, uvu: UVUPlan, lin2_weight: np.ndarray, species: int, acc_of_entry, add: np.ndarray) -> np.ndarray:
    """numpy walk through the tables exactly as the kernel reads them (tests/test_host.py): ``acc_of_entry(e)`` ->
    [16 nodes, lanes-per-node channels, NACC] neighbour sums of entry e for one tile; returns the tile's [16, d_out]
    output rows = add + lin2."""
    S = plan.gather.shape[0]
    atab = np.where(plan.gather[species] >= 0, lin2_weight[np.clip(plan.gather[species], 0, None)] * plan.scale, 0.0)
    out = add.astype(np.float64).copy()
    for q in plan.quads:
        members, cu_log2, n_pass, n_groups, base = q[:4], int(q[4]), int(q[5]), int(q[6]), int(q[7])
        npw = min(TILE_NODES, 64 >> cu_log2)
        for r in range(n_groups):
            for ps in range(n_pass):
                # the dump: [wave][reg][lane]
                dump = np.zeros((4, DUMP_REGS, 64))
                for w, e in enumerate(members):
                    if e < 0:
                        continue
                    kind = int(plan.entries[e][0])
                    l1, gi = kind // TP_KIND_STRIDE, kind % TP_KIND_STRIDE
                    pas = kind_passes(l1, gi)
                    acc = acc_of_entry(int(e))                     # [16, cu, NACC]
                    cu = 1 << int(plan.entries[e][3])
                    rel, off = 0, 0
                    for c, (l2, l3) in enumerate(kind_combos(l1, gi)):
                        d3 = 2 * l3 + 1
                        if pas[c] == ps:
                            for k in range(d3):
                                for j in range(npw):
                                    dump[w, rel + k, j * cu:(j + 1) * cu] = acc[r * npw + j, :, off + k]
                            rel += d3
                        off += d3
                flatd = np.zeros(4 * DUMP_REGS * DUMP_RS)
                for w in range(4):
                    for rg in range(DUMP_REGS):
                        flatd[(w * DUMP_REGS + rg) * DUMP_RS:(w * DUMP_REGS + rg) * DUMP_RS + 64] = dump[w, rg]
                for w in range(4):
                    u0, nu = plan.wave_units[base + ps * 4 + w]
                    for un in plan.units[u0:u0 + nu]:
                        col0, d3, vcount, nt0, n_nt, p0, npc, packed = (int(t) for t in un)
                        npw_log2 = packed & 255
                        D = np.zeros((n_nt, 16, 16))               # [nt][v][col]
                        for pc in plan.pieces[p0:p0 + npc]:
                            doff, aoff, cu = int(pc[0]), int(pc[1]), 1 << int(pc[2])
                            ksv = max(1, cu // 4)
                            A = atab[aoff:aoff + 64 * ksv].reshape(4, 16, ksv)       # [g][v][t]
                            for nt in range(n_nt):
                                for c in range(16):
                                    col = (nt0 + nt) * 16 + c
                                    j, k = col & (npw - 1), col >> npw_log2
                                    if k >= d3:
                                        continue
                                    for g in range(4):
                                        if g * ksv >= cu:        # (the kernel reads the next node's lanes there, times zero weights)
                                            continue
                                        b = flatd[doff + k * DUMP_RS + j * cu + g * ksv: doff + k * DUMP_RS + j * cu + g * ksv + ksv]
                                        D[nt, :, c] += A[g] @ b
                        for nt in range(n_nt):
                            for c in range(16):
                                col = (nt0 + nt) * 16 + c
                                j, k = col & (npw - 1), col >> npw_log2
                                if k >= d3:
                                    continue
                                for v in range(vcount):
                                    out[r * npw + j, col0 + v * d3 + k] += D[nt, v, c]
    return out
