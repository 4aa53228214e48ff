/*
 * matten_hip.h -- C ABI of libmatten_hip.so: MI355X (gfx950) kernels for MatTen's equivariant
 * message-passing hot path.
 *
 * The reference (wengroup/matten) has no FFI for this path: it calls e3nn / torch_scatter Python
 * ops.  Each entry point below replaces one of those call sites (cited as reference file:line,
 * relative to the reference repository root) and is what a ctypes / pybind stub on the reference
 * side would bind -- see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); no allocation inside
 *   - kernels are enqueued on `stream` and never synchronise; the library keeps no global state
 *   - return value: 0 on success, negative MATTEN_E* on a host-detectable error
 *     (data-dependent errors, e.g. an unsupported atomic number, are reported through a device
 *     `int32_t* err_flag` the caller reads back when it wants to)
 *   - fp32 data, int32 indices inside the library; the backbone boundary's int64 tensors
 *     (edge_index, atomic_numbers, batch, ptr -- reference data/_dtype.py:4) are read as int64
 *   - irreps data layout: blocks concatenated, each block [mul, 2l+1] row-major (e3nn "mul_ir")
 */
#ifndef MATTEN_HIP_H
#define MATTEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* matten_stream_t; /* hipStream_t */

#define MATTEN_OK 0
#define MATTEN_EINVAL (-1)  /* bad argument (null pointer, negative size, unsupported shape) */
#define MATTEN_ELAUNCH (-2) /* hipGetLastError() != hipSuccess after a launch */
#define MATTEN_ENOMEM (-3)  /* caller-provided workspace too small */

/* ABI version of this header; bumped on any signature change. */
int matten_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Graph indexing.  Replaces the implicit ordering torch_scatter.scatter(msg, edge_dst) relies
 * on (nn/conv.py:113-114): builds a destination-sorted CSR so that the neighbour sum is a
 * segmented reduction with a fixed (stable => deterministic) summation order.
 *   edge_index [2,E] int64: row 0 = centre i ("src"), row 1 = neighbour j ("dst") -- data/data.py:297-301
 *   perm[E]      : sorted position -> original edge id (stable sort by dst)
 *   rowptr[N+1]  : CSR offsets into the sorted edge list
 *   src_sorted[E]: edge_index[0][perm[e]]
 * workspace: matten_csr_workspace_bytes(E, N) bytes.
 * Out-of-range node ids set err_flag bit 0.
 * Sparse graphs (E <= 64 N) are built by counting (in-degree atomics, scan, rank of the edge ids inside every
 * segment: no device-wide sort), dense ones by a stable radix sort; the result is the same.  The ranking step costs
 * degree^2 / 16 comparisons per node (cutoff-radius graphs: degree ~ 10-200); a hub segment of more than 2048 edges in
 * an otherwise sparse graph is sorted by a whole workgroup instead (O(d log^2 d), same order).
 * ------------------------------------------------------------------------------------------ */
size_t matten_csr_workspace_bytes(int64_t n_edges, int64_t n_nodes);
int matten_csr_counting_max_avg_degree(void);   /* E <= this * N: counting build (no device-wide sort) */
int matten_csr_build(const int64_t* edge_index, int64_t n_edges, int64_t n_nodes, int32_t* perm,
                     int32_t* rowptr, int32_t* src_sorted, void* workspace, size_t workspace_bytes,
                     int32_t* err_flag, matten_stream_t stream);

/* Long CSR segments cut into VIRTUAL nodes of at most max_len edges (small batches, where one hub -- a one-atom cell has
 * hundreds of neighbours inside the cutoff -- would otherwise set the duration of every tensor-product launch: the kernels
 * walk a segment serially).  The pieces tile the sorted edge list in order:
 *   vrowptr[bound + 1] : CSR row pointer over virtual nodes (pieces past the real count are empty, = n_edges)
 *   vseg[n_nodes + 1]  : int64, virtual range of every real node (the `ptr` of matten_segment_reduce: the real node's
 *                        neighbour sum = ordered sum of its pieces' rows)
 *   vnn[bound]         : num_neigh of the piece's real node (optional, for per-node normalisation; both NULL or neither)
 * bound = matten_csr_split_bound(n_nodes, n_edges, max_len) = n_nodes + n_edges / max_len; no host sync.
 * Reference: the order torch_scatter.scatter adds messages in is unspecified (nn/conv.py:114); here it is fixed and
 * depends on the node's own segment only. */
int64_t matten_csr_split_bound(int64_t n_nodes, int64_t n_edges, int64_t max_len);
int matten_csr_split(const int32_t* rowptr, int64_t n_nodes, int64_t n_edges, int64_t max_len, int32_t* vrowptr,
                     int64_t* vseg, const float* num_neigh, float* vnn, matten_stream_t stream);

/* Grouping of n items by an integer key in [0, n_keys): order[n] = item positions stably sorted by key,
 * seg[n_keys+1] = first sorted position of each key.  This is the species grouping the species-indexed linears
 * (FullyConnectedTensorProduct with a one-hot operand, nn/conv.py:59-86) walk instead of evaluating densely over
 * the one-hot.  workspace: matten_group_workspace_bytes(n, n_keys) bytes.  A key out of range sets err_flag bit 0.
 * (n_keys <= 256: per-run histograms + scan + ballot ranks; more keys: stable radix sort.) */
size_t matten_group_workspace_bytes(int64_t n, int64_t n_keys);
int matten_group_by_key(const int64_t* key, int64_t n, int64_t n_keys, int32_t* order, int32_t* seg,
                        void* workspace, size_t workspace_bytes, int32_t* err_flag, matten_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SpeciesEmbedding.forward (nn/embedding.py:85-110) + _AtomicNumberToIndex.forward (:230-259)
 *   species_index[n] = z_to_index[Z[n] - min_z]           (int64 out, reference dtype)
 *   node_feats[n,:]  = W[:, species] + b                   (Linear on a one-hot == column lookup)
 *   node_attrs (one-hot [N,S]) is written only if node_attrs != NULL.
 * err_flag bit 1: Z outside [min_z,max_z]; bit 2: Z maps to -1 (unsupported species).
 * ------------------------------------------------------------------------------------------ */
int matten_species_embed(const int64_t* atomic_numbers, int64_t n_nodes, const int64_t* z_to_index,
                         int64_t min_z, int64_t max_z, int64_t n_species, const float* weight /*[dim,S]*/,
                         const float* bias /*[dim]*/, int64_t dim, int64_t* species_index, int32_t* species_i32,
                         float* node_feats, float* node_attrs, int32_t* err_flag, matten_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * with_edge_vectors (nn/_nequip.py:214-268) + o3.SphericalHarmonics (nn/_nequip.py:167-174)
 * + soft_one_hot_linspace bessel * sqrt(nb) (nn/embedding.py:185-199), fused, per edge.
 *   vec = pos[dst] - pos[src] + shift . cell[batch[src]]
 * Sorted-order outputs (consumed by the kernels below; e = sorted position, o = perm[e]):
 *   geom_sorted[E,4] = (vx, vy, vz, |v|)
 *   sh_sorted[E, sh_stride]   real SH of v/|v| in the first (lmax+1)^2 columns, l-major, m=-l..l,
 *                             'component' normalised; sh_stride >= (lmax+1)^2 (32 keeps rows on 128-B lines);
 *                             the remaining columns of every row are written as zeros (no pre-clearing needed)
 * Optional original-order outputs for the backbone's data dict (NULL to skip):
 *   edge_vectors[E,3], edge_lengths[E], edge_attrs[E,(lmax+1)^2], edge_embedding[E,nb]
 * cell is [B,3,3] (rows = lattice vectors) or NULL; n_cells==1 uses cell 0 for every edge.
 * Node ids outside [0, n_nodes) and crystal ids outside [0, n_cells) are clamped (memory safety only: matten_csr_build
 * flags such a batch, and the host may read that flag after this kernel was enqueued).
 * ------------------------------------------------------------------------------------------ */
int matten_edge_geom(const float* pos, const int64_t* edge_index, const float* edge_cell_shift,
                     const float* cell, int64_t n_cells, const int64_t* batch, const int32_t* perm,
                     int64_t n_edges, int64_t n_nodes, int lmax, int n_basis, float r_start, float r_end,
                     float* geom_sorted, float* sh_sorted, int sh_stride, float* edge_vectors, float* edge_lengths,
                     float* edge_attrs, float* edge_embedding, matten_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Radial MLP: e3nn FullyConnectedNet([nb, h, h, W], act=silu) applied to the Bessel embedding
 * (nn/utils.py:246-251,260 <- nn/conv.py:113).  fp32 MFMA (v_mfma_f32_16x16x4_f32).
 *   in : geom_sorted[E,4] (uses |v|); the Bessel basis is recomputed in the prologue
 *   w0p[nb_pad, h], w1p[h, h], w2p[h, w_pad]: weights pre-scaled by 1/sqrt(fan_in) and by the
 *        normalize2mom constant of the *previous* activation, row-major, zero padded
 *        (nb_pad multiple of 4, h == 32, w_pad multiple of 16)
 *   out: w_edge[E, w_pad] fp32, or bf16 (round to nearest even) when out_is_bf16
 * ------------------------------------------------------------------------------------------ */
int matten_radial_mlp(const float* geom_sorted, int64_t n_edges, int n_basis, float r_start, float r_end,
                      const float* w0p, int nb_pad, const float* w1p, const float* w2p, int hidden, int w_pad,
                      float act_cst, void* w_edge, int out_is_bf16, matten_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 'uvu' TensorProduct + gather + scatter-add + neighbour normalisation over MATERIALISED per-edge weights
 * (nn/utils.py:230-237,263; nn/conv.py:113-120):
 *   agg[n, slot(p,u,k)] = norm_n * sum_{e in in(n)} w[e, woff_p+u] * sqrt(2 l3+1) sum_ij C^{l1 l2 l3}_{ijk} x[src_e, xoff_p+u*d1+i] Y_{l2,j}(e)
 *   norm_n = 1/sqrt(avg_num_neighbors) if avg_num_neighbors > 0 else 1/sqrt(num_neigh[n])
 * One wave per (path, node group), a lane owns one output channel of one destination node, CG coefficients are
 * compile-time literals (l <= 4).  The training forward (its adjoint reads the same w), and an independent implementation
 * of the contraction the production kernel matten_tp_fused is tested against.
 *   path_entries[n_entries, 8] int32 {l1*25+l2*5+l3, x_off, w_off, out_off, mul(<=64), log2(lanes per node), 0, 0}
 *   unit_start[n_entries+1]   int32  prefix sum of waves per node tile (matten_tp_tile_nodes() nodes per tile)
 * ------------------------------------------------------------------------------------------ */
int matten_tp_tile_nodes(void);
/* (w_edge: fp32, or bf16 when w_is_bf16 -- the opt-in bf16 storage of the training step's per-edge tensors) */
int matten_tp_paths(const float* x, int64_t d_in, const void* w_edge, int64_t w_pad, const float* sh_sorted,
                    int64_t sh_dim, const int32_t* rowptr, const int32_t* src_sorted, int64_t n_nodes,
                    const int32_t* path_entries, const int32_t* unit_start, int64_t n_entries,
                    int64_t units_per_tile, int64_t d_mid, float avg_num_neighbors, const float* num_neigh,
                    float* agg /*[N,d_mid]*/, int w_is_bf16, matten_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * lin2 of a conv layer as a row stream over COMPONENT-MAJOR neighbour sums (reference nn/conv.py:84-86,123:
 * out = FullyConnectedTensorProduct(agg, one_hot(species)) + self-connection).
 *   agg[N, ld]: per output irrep io of lin2 a region [component k][channel slot 0..16 T_io): the channels of every
 *               path with that irrep side by side (written by matten_tp_fused when group_entries word 8+c (t_off[c])
 *               holds the component stride 16 T_io and word 20+c (out_off[c]) the first float of the entry's channel 0);
 *               regions back to back, so chunk f (16 floats) of a row belongs to exactly one (io, k)
 *   io_table[n_io, 8] int32: {first chunk, T, K (channel slots used; the rest are never read as data),
 *               d3 | n_mt << 8 | cw << 16, a_off, out_off, mul_out, 0}
 *   blocks[n_blocks, 4] int32: the row cut into runs of <= 8 chunks of one (io, k), front to back:
 *               {first chunk, n | first-of-unit << 8 | last-of-unit << 9 | last unit of the table row << 10 | k << 12 |
 *               io << 20, chunk index inside the unit, 0}; a table row has mul_out * d3 <= 32 (wider irreps: several rows)
 *   wtab[n_species, w_stride]: per species and irrep the weights as MFMA A fragments at a_off,
 *               [t < T][mt < n_mt][g < 4][c < cw][s < 4] = fan^-1/2 W_s[slot 16 t + 4 g + s][v = 16 mt + c] (cw = 16 or mul_out)
 *   order / seg: rows sorted by species + per-species offsets (matten_group_by_key), or NULL/NULL with n_species == 1
 *   out[N, d_out] in the reference's mul_ir layout: out[n, out_off + v d3 + k] = add[...] + sum
 * ------------------------------------------------------------------------------------------ */
int matten_agg_linear_max_mt(void);        /* column tiles per io_table row (mul_out of a row <= 16 * this) */
int matten_agg_linear_block_chunks(void);  /* chunks per block (the n of a blocks[] row is <= this) */
size_t matten_agg_linear_lds_bytes(int64_t w_stride, int64_t n_io, int64_t n_blocks);   /* LDS a launch needs ... */
size_t matten_agg_linear_max_lds_bytes(void);                                          /* ... and may have */
int matten_agg_linear(const float* agg, int64_t ld, const int32_t* order, const int32_t* seg, int64_t n_species,
                      const float* wtab, int64_t w_stride, const int32_t* io_table, int64_t n_io,
                      const int32_t* blocks, int64_t n_blocks, const float* add, int64_t add_ld, int64_t d_out,
                      int64_t n_rows, float* out, matten_stream_t stream);
/* matten_agg_linear with the conv layer's Gate (reference nn/utils.py:134-140) and eval-mode BatchNorm (:418,432)
 * applied in the epilogue: out [n_rows, d_act] is the ACTIVATED row, the conv output row [n_rows, d_out] never exists.
 *   cmeta[d_out, 4] int32 per column of lin2's output {type | act << 8, column in the activated row, gate lane, gate set}:
 *     type 1 activated scalar, 2 gate scalar (kept in registers: set < matten_agg_linear_gate_sets(), lane = its position
 *     in the table row that produced it), 3 gated component; gates are produced by table rows that precede their users
 *     (host: plan.plan_agg_gate);  act codes and act_cst[8] as matten_gate_bn;
 *   bn_scale / bn_shift [d_act] (both or NULL): weight / sqrt(running_var + eps) and, on 0e columns,
 *     bias - running_mean * scale */
int matten_agg_linear_gate(const float* agg, int64_t ld, const int32_t* order, const int32_t* seg, int64_t n_species,
                           const float* wtab, int64_t w_stride, const int32_t* io_table, int64_t n_io,
                           const int32_t* blocks, int64_t n_blocks, const float* add, int64_t add_ld, int64_t d_out,
                           int64_t n_rows, const int32_t* cmeta, const float* act_cst, const float* bn_scale,
                           const float* bn_shift, int64_t d_act, float* out, matten_stream_t stream);
int matten_agg_linear_gate_sets(void);
/* ------------------------------------------------------------------------------------------
 * Fused production path of one conv layer's edge work (reference nn/utils.py:246-251,260,263 +
 * nn/conv.py:113-120): the per-edge radial weights are never written to memory.
 *   matten_radial_hidden: rbf(|v|) -> 32 -> 32 (silu), fp32 MFMA; writes h2s[E,2,32] fp16 (128 B per edge): every
 *                         hidden feature v as two pieces, v = hi + 2^-11 lo (hi = fp16(v), lo = fp16(2^11 (v - hi)),
 *                         fp16 subnormals zeroed: v to 2^-24 relative); column g*8+kk of piece 0 (hi) / piece 1 (lo)
 *                         holds hidden feature 16*(kk>>2) + 4*g + (kk&3)
 *   matten_tp_fused:      per (input block, l2 group, node group) wave: last MLP layer on the matrix
 *                         cores (w = h2 . W2p as three fp16 products hi.hi + 2^-11 (hi.lo + lo.hi) accumulated in
 *                         fp32: fp32-class rounding at 1/5 of the fp32 MFMA time) into a wave-private LDS tile, consumed in place by the
 *                         literal-coefficient CG contraction + CSR neighbour sum
 *   w2p[32, w_pad]: last layer weights pre-scaled (1/sqrt(32) * normalize2mom(silu)), columns in the
 *                   fused [entry][u][coupling] order of group_entries, w_pad >= w_cols + 16
 *   group_entries[n_entries, 32] int32 {l1*8+g, x_off, mul, log2(lanes per node), coupling mask, w_base, first A tile,
 *                   A tiles, t_off[12], out_off[12]}: one entry per (input irrep block chunk, l2 group g); couplings in the
 *                   order of cg_gen.h Group<l1,g>, weight columns [u][c] (absent couplings: zero columns), mul * couplings
 *                   <= matten_tp_max_cols*(); out_off[c] = first float of channel 0 in the output row, t_off[c] = 0
 *                   (the reference's mul_ir row) or the floats between two components (component-major row)
 *   unit_map[units_per_tile]: wave index inside a node tile -> flags | entry << 8 | node group (of 64 >> cu_log2 nodes);
 *                   flags: bit 24 = the unit's workgroup (four consecutive units) shares an LDS stage: same node
 *                   group and lanes per node -- or, with bit 26 (paired), units 0,1 on node group r and units 2,3 the
 *                   same two entries on r + 1; bit 25 = loader-only unit (feeds the stage, contracts nothing).  Any
 *                   bijection is valid, the host orders it node group first (plan.fused_unit_map) so the waves of a
 *                   workgroup read the same hidden-feature / harmonics rows
 *   a_split / a_scale_inv (both or neither; NULL: the kernel splits w2p itself, ~400 instructions per wave):
 *                   w2p as ready-made MFMA A fragments.  Entry e owns tiles [a_tile, a_tile + n_mt) (group_entries
 *                   words 6, 7); tile t, lane (g = lane >> 4, c = lane & 15) holds 16 fp16: hi[kk], lo[kk], kk < 8, of
 *                   s_e * w2p[16 (kk >> 2) + 4 g + (kk & 3)][w_base + 16 (t - a_tile) + c] with s_e the power of two
 *                   that puts the entry's largest magnitude in [2^13, 2^14); a_scale_inv[e] = 1 / (s_e h_scale)
 *   lds_floats_per_wave: max over entries of 16*T*(16*ceil(mul*NC/16)+36), T = max(1, nodes_per_wave/16)
 *                   (sh_sorted rows must be 32 floats apart: sh_stride == 32)
 *   h_scale (device pointer to ONE float, or NULL = 1): power of two the hidden features are multiplied by before
 *                   the fp16 split, chosen by the host so that |h_scale * h2| < 2^15 for ANY edge (from the weights'
 *                   column sums: 1 for a normally scaled MLP); the caller multiplies a_scale_inv by 1 / h_scale
 */
int matten_radial_hidden(const float* geom_sorted, int64_t n_edges, int n_basis, float r_start, float r_end,
                         const float* w0p, int nb_pad, const float* w1p, int hidden, uint16_t* h2s,
                         const float* h_scale, matten_stream_t stream);
/* the same for n_layers <= 8 radial MLPs over one edge list in ONE launch (every conv layer of a model reads the same
 * edge lengths; the Bessel basis is evaluated once): w0p / w1p / h2s / h_scale (or NULL) are HOST arrays of n_layers
 * device pointers */
int matten_radial_hidden_multi(const float* geom_sorted, int64_t n_edges, int n_basis, float r_start, float r_end,
                               const float* const* w0p, int nb_pad, const float* const* w1p, int hidden,
                               uint16_t* const* h2s, const float* const* h_scale, int n_layers,
                               matten_stream_t stream);
int matten_tp_fused(const float* x, int64_t d_in, const uint16_t* h2s, const float* w2p, int64_t w_pad,
                    const float* sh_sorted, int64_t sh_stride, const int32_t* rowptr, const int32_t* src_sorted,
                    int64_t n_nodes, const int32_t* group_entries, const int32_t* unit_map, int64_t n_entries,
                    int64_t units_per_tile, int64_t lds_floats_per_wave, int64_t d_mid, float avg_num_neighbors,
                    const float* num_neigh, const uint16_t* a_split, const float* a_scale_inv,
                    float* agg /*[N,d_mid]*/, matten_stream_t stream);

int matten_tp_max_cols(void);   /* weight columns (mul * couplings) a group entry of matten_tp_fused may have */
int matten_tp_max_cols_l0(void);   /* the same for entries of scalar (l1 = 0) input blocks */
int matten_tp_max_cols_l1(void);   /* ... and of vector (l1 = 1) input blocks */
/* 31-bit hash of the coupling-group lists the library's generated coupling code (csrc/cg_gen.h) was built for; the host plans
 * entries for the lists of matten_amd/plan.py (plan.tp_groups_hash) and refuses a library built for others. */
int matten_tp_groups_hash(void);

/* ------------------------------------------------------------------------------------------
 * FullyConnectedTensorProduct(x, one_hot(species)) == species-indexed per-irrep linear
 * (nn/conv.py:59-61,77-79,84-86 called :109,112,123), and e3nn o3.Linear when species == NULL
 * (nn/nodewise.py:111-117, model_factory/tfn_scalar_tensor.py:49-51,68).
 *   out[n, o_off + w*d + k] = (add ? add[..] : 0) + sum_{u<mul_in} Wp[species(n), w_off + u*mo + w] * x[n, x_off + u*d + k]
 *   x rows are d_in floats apart, out rows d_out, add rows add_ld (>= d_out; a column slice of a wider matrix is fine:
 *   the conv layer evaluates lin1 and the self-connection as one call and hands the second half on as the addend)
 *   for every segment, w < mo, k < d.  fp32 MFMA over tiles of 16 rows x 16 output channels.
 *   segs[n_segs, 8] int32 {x_off, d, mul_in, w_off, mo, o_off, 0, 0}: one per (input irrep block ->
 *       output irrep block) path; outputs no segment covers are NOT written (the caller zero-fills
 *       unreachable irreps)
 *   Wp[n_species, w_stride]: weights repacked per species with the path normalisation folded in.
 *   order[N] / seg[n_species+1]: node ids sorted by species and the offsets of each species' run
 *   (e.g. perm / rowptr of matten_csr_build over {row0 = node id, row1 = species}); both NULL for a
 *   plain linear (n_species == 1).  Rows are processed species by species so the weight table of
 *   one species is staged on chip once per workgroup; outputs land at their original row.
 * ------------------------------------------------------------------------------------------ */
int matten_species_linear(const float* x, int64_t d_in, const int32_t* order, const int32_t* seg, int64_t n_species,
                          const float* wp, int64_t w_stride, const int32_t* segs, int64_t n_segs, int64_t d_out,
                          const float* add, int64_t add_ld, int64_t n_rows, float* out, matten_stream_t stream);

/* Same operator, same arguments, for SHORT rows (16 * (d_in | 1) + w_stride floats <= 64 KB of LDS, one segment table):
 * a workgroup keeps its 16 rows and the species' weights in LDS and walks the irrep blocks without the streaming
 * kernel's chunk pipeline (reference call sites nn/conv.py:109,112 lin1 / self-connection, nn/nodewise.py:116).
 * Returns MATTEN_EINVAL when the rows do not fit: call matten_species_linear then. */
int matten_species_linear_rows(const float* x, int64_t d_in, const int32_t* order, const int32_t* seg, int64_t n_species,
                               const float* wp, int64_t w_stride, const int32_t* segs, int64_t n_segs, int64_t d_out,
                               const float* add, int64_t add_ld, int64_t n_rows, float* out, matten_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * e3nn Gate (nn/utils.py:134-140,158-159 <- nn/conv.py:209) fused with e3nn BatchNorm in eval
 * mode (nn/utils.py:418,432-433 <- nn/conv.py:211).
 *   meta[d_out] int4 {src, gate(-1: scalar), act (0 none,1 silu,2 tanh,3 sigmoid,4 ssp,5 abs) | gate_act<<8, bn_idx | mean_idx<<16 (0xFFFF: none)}
 *   out = act(x[src])*c_act                      (scalars)
 *       = x[src] * gate_act(x[gate])*c_gate      (gated irreps)
 *   then, if bn_weight != NULL: (out - running_mean[mean_idx]) * rsqrt(running_var[bn_idx]+eps)*bn_weight[bn_idx] + bn_bias[mean_idx]
 *   act_cst[8]: normalize2mom constants indexed by act code.
 * ------------------------------------------------------------------------------------------ */
int matten_gate_bn(const float* x, int64_t d_in, const int32_t* meta, int64_t d_out, const float* act_cst,
                   const float* running_mean, const float* running_var, const float* bn_weight,
                   const float* bn_bias, float eps, int64_t n_rows, float* out, matten_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * NodewiseReduce: torch_scatter.scatter(x, batch, reduce=mean|sum) over contiguous crystals
 * (nn/nodewise.py:142-148).  ptr[B+1] int64 (PyG convention).  mean divides by max(count,1).
 * ------------------------------------------------------------------------------------------ */
int matten_segment_reduce(const float* x, int64_t dim, const int64_t* ptr, int64_t n_segments, int mean,
                          float* out, matten_stream_t stream);
/* reduce = min | max (the other two values nn/nodewise.py:131 accepts): per crystal and column the smallest / largest
 * value and, in arg [n_segments, dim] int64 (optional), the row it came from (an empty crystal: 0 and -1);
 * _bwd: dx[arg[b, c], c] = dy[b, c], dx zero-initialised */
int matten_segment_minmax(const float* x, int64_t dim, const int64_t* ptr, int64_t n_segments, int take_max, float* out,
                          int64_t* arg, matten_stream_t stream);
int matten_segment_minmax_bwd(const float* dy, int64_t dim, const int64_t* arg, int64_t n_segments, float* dx,
                              matten_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Adjoint of the radial MLP (training; reference nn/utils.py:246-251,260 differentiated by autograd there).
 * Inputs as matten_radial_mlp (packed weights; w2p in the column order of dw), plus dw[E, dw_ld] = dL/dw from
 * matten_tp_backward (fp32, or bf16 when dw_is_bf16; only the first w_cols columns are read as data).
 * Outputs are PARTIAL sums the caller adds up (fixed order, no atomics):
 *   part_small[matten_radial_mlp_bwd_small_slices(E)][nb_pad*32 + 32*32]: scale0 d/dW0p [nb_pad,32], scale1 d/dW1p [32,32]
 *   part_w2[matten_radial_mlp_bwd_w2_ranges(E, w_pad)][32][w_pad]:        scale2 d/dW2p
 *   h2_scratch[E, 32]: workspace (the recomputed hidden features, written by the first kernel, read by the second)
 * ------------------------------------------------------------------------------------------ */
int64_t matten_radial_mlp_bwd_small_slices(int64_t n_edges);
int64_t matten_radial_mlp_bwd_w2_ranges(int64_t n_edges, int64_t w_pad);
int matten_radial_mlp_bwd(const float* geom_sorted, int64_t n_edges, int n_basis, float r_start, float r_end,
                          const float* w0p, int nb_pad, const float* w1p, const float* w2p, int hidden, int w_pad,
                          int w_cols, const void* dw, int64_t dw_ld, int dw_is_bf16, float* h2_scratch,
                          float* part_small, float* part_w2, float scale0, float scale1, float scale2,
                          float* grad_small, float* grad_w2, matten_stream_t stream);
/* grad_small [nb_pad*32 + 32*32] / grad_w2 [32, w_pad] (both or neither, may be NULL): the partial sums added up in slice
 * order by a third launch of the same call */
/* the raw layers (w0 [nb,32], w1 [32,32], w2 [32,W]) -> packed operands w0p = scale0 w0 (rows padded to nb_pad),
 * w1p = scale1 w1, w2p = scale2 w2 (columns padded to w_pad), one launch; matten_radial_mlp_bwd multiplies its partial
 * sums by the same three factors, so they are gradients w.r.t. the RAW layers */
int matten_radial_pack(const float* w0, const float* w1, const float* w2, int n_basis, int nb_pad, int w_cols, int w_pad,
                       float scale0, float scale1, float scale2, float* w0p, float* w1p, float* w2p,
                       matten_stream_t stream);
/* Operands of matten_tp_fused derived from the RAW radial layers by kernels (training on the production kernel):
 * matten_radial_pack_cols: as matten_radial_pack, the last layer gathered through cols[n_cols] int64 (fused column order
 *   of plan.fused_cols, -1 = zero column), zero up to w_pad;
 * matten_radial_h_scale: out2 = [s, 1/s], the power of two <= 1 that keeps the hidden features inside the fp16 range
 *   (bound from the two layers' column-sum norms, as nn/utils.py RadialMLP._fp16_scale);
 * matten_split_a_tiles: w2p [32, w_pad] -> the fp16 hi / lo A fragments of every group entry (word 5 = first column,
 *   6 = first tile, 7 = tile count) + scale_inv[n_entries] = 1 / (entry scale) * h_scale[1] (h_scale may be NULL). */
int matten_radial_pack_cols(const float* w0, const float* w1, const float* w2, int n_basis, int nb_pad, int w_cols,
                            const int64_t* cols, int n_cols, int w_pad, float scale0, float scale1, float scale2, float* w0p,
                            float* w1p, float* w2p, matten_stream_t stream);
int matten_radial_h_scale(const float* w0, const float* w1, int n_basis, float r_start, float r_end, float act_cst,
                          float* out2, matten_stream_t stream);
int matten_split_a_tiles(const float* w2p, int64_t w_pad, const int32_t* group_entries, int64_t n_entries,
                         const float* h_scale, uint16_t* frag, float* scale_inv, matten_stream_t stream);
/* out[i] = src[idx[i]] * scale[(scale_by_source ? idx[i] : i) % scale_period]: the per-species re-packing of a flat e3nn
 * weight (idx = gather table, scale by output position) and its adjoint (idx = inverse permutation, scale by source) */
int matten_gather_scale(const float* src, const int64_t* idx, const float* scale, int64_t n, int64_t scale_period,
                        int scale_by_source, float* out, const int64_t* perm2, float* out2, matten_stream_t stream);
/* perm2 [scale_period] / out2 (both or neither): a second output with the columns of every period permuted,
 * out2[r, q] = out[r, perm2[q]] -- the transposed packed weights of the species linear's adjoint in the same launch */
int64_t matten_species_linear_wgrad_scratch_floats(int64_t n_rows, int64_t n_species, int64_t w_stride);   /* matten_species_linear_wgrad's scratch */

/* ------------------------------------------------------------------------------------------
 * CartesianTensor.to_cartesian (utils.py:123-124, predict.py:145): out[b,:] = x[b,:] @ Q
 *   Q [n_in, n_out] row-major (n_in = 21, n_out = 81 for ijkl=jikl=klij)
 * ------------------------------------------------------------------------------------------ */
int matten_dense_rows(const float* x, int64_t n_in, const float* q, int64_t n_out, int64_t n_rows, float* out,
                      matten_stream_t stream);


/* ==========================================================================================
 * Adjoint (backward) operators for the training step (reference: autograd through
 * model/model.py:276-372 shared_step -> loss.backward()).  fp32.
 * ========================================================================================== */

/* adjoint of matten_tp_paths (w_edge in the reference column order):
 *   dw[e,q]              = norm * sum_ijk C_ijk x[src,x_base+i] Y[e,y_off+j] G[dst,out_base+k]
 *   dx[src, x_base + i] += norm * w[e,q] * sum_jk C_ijk Y[e,y_off+j] G[dst,out_base+k]     (dx zero-initialised)
 *   col_meta[n_cols,4] int32 {x_base, out_base, nnz_begin, nnz_count | y_off<<16}; nnz_ijk[nnz,4] uint8 {i,j,k,0}
 *   (i-major per coupling); nnz_c[nnz] = sqrt(2 l3+1) C_ijk
 *   in_ptr[n_in+1] / in_cols[n_cols] (both or neither): the weight columns grouped by the input channel (x_base) they
 *   read.  With them a thread owns (edge, input channel) and adds its paths' contributions to dx in registers: one
 *   atomic per (edge, channel, component) instead of one per path; without, one thread per (edge, column). */
int matten_tp_backward(const float* x, int64_t d_in, const void* w_edge, int64_t w_ld, const float* sh_sorted,
                       int64_t sh_stride, const int32_t* src_sorted, const int32_t* dst_sorted,
                       const int32_t* col_meta, int64_t n_cols, const uint8_t* nnz_ijk, const float* nnz_c,
                       const float* g_agg, int64_t d_mid, float avg_num_neighbors, const float* num_neigh,
                       int64_t n_edges, float* dx, void* dw, int64_t dw_ld, const int32_t* in_ptr,
                       const int32_t* in_cols, int64_t n_in, int edge_is_bf16 /* w_edge AND dw are bf16 */,
                       matten_stream_t stream);

/* Adam (torch.optim.Adam semantics, amsgrad off, L2 weight decay added to the gradient) over ONE flat fp32 buffer of n
 * elements; step[0] (device) = the number of this step, already incremented by the caller; buffers 16-byte aligned. */
int matten_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* step,
                     float lr, float beta1, float beta2, float eps, float weight_decay, matten_stream_t stream);

/* the same adjoint with the literal-coefficient coupling code of the forward kernels (cg_gen.h): a thread owns (edge,
 * channel of one input block) and walks the block's paths; one atomic per (edge, channel, component) into dx.
 *   blocks[n_blocks,4] int32 {x_off, mul, l1, first path | n_paths << 16}; paths[n_paths,4] {l1*25+l2*5+l3, w_off, out_off, 0};
 *   sum_lanes = sum over the blocks of mul rounded up to a power of two (sizes the launch: the blocks' workgroup ranges
 *   are laid end to end); w_edge / dw fp32, or both bf16 when edge_is_bf16
 * dx: two modes.  dx_edges == NULL: dx [N, d_in] zero-initialised, contributions meet through fp32 atomics (order not
 * fixed).  dx_edges != NULL (scratch [E, d_in]; every input block's columns are written, columns of input irreps without
 * a path must be zero on entry): each edge's contribution is stored and dx[n] = the sum over the edges leaving n in the
 * order of out_perm[out_ptr[n] .. out_ptr[n+1]) -- sorted-edge indices grouped by source node (the CSR of src_sorted:
 * matten_csr_build on it) -- bitwise reproducible; dx [n_nodes, d_in] need not be initialised. */
int matten_tp_backward_lit(const float* x, int64_t d_in, const void* w_edge, int64_t w_ld, const float* sh_sorted,
                           int64_t sh_stride, const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* blocks,
                           int64_t n_blocks, int64_t max_mul, const int32_t* paths, int64_t n_paths, const float* g_agg,
                           int64_t d_mid, float avg_num_neighbors, const float* num_neigh, int64_t n_edges, float* dx,
                           void* dw, int64_t dw_ld, int edge_is_bf16, int64_t n_nodes, const int32_t* out_ptr,
                           const int32_t* out_perm, float* dx_edges, int max_l, matten_stream_t stream);
/* matten_tp_backward_lit for training on the fused forward: w[E, W] is never read -- every workgroup re-evaluates the weights of
 * its (edges, input block) on the matrix cores from h2s[E, 2, 32] (matten_radial_hidden: the rows the forward used) and the
 * A fragments of the last radial layer in REFERENCE column order: frag / w_inv = matten_split_a_tiles over one pseudo-entry per
 * path (int32 [n_paths, 32], words 5 / 6 / 7 = w_off, first tile, ceil(mul / 16); plan.bw_w_entries), paths[p][3] = that first
 * tile.  Same dw / dx contract as matten_tp_backward_lit (fixed-order sums with out_ptr / out_perm / dx_edges). */
int matten_tp_backward_lit_wfree(const float* x, int64_t d_in, const uint16_t* h2s, const uint16_t* frag, const float* w_inv,
                                 const float* sh_sorted, int64_t sh_stride, const int32_t* src_sorted,
                                 const int32_t* dst_sorted, const int32_t* blocks, int64_t n_blocks, int64_t sum_lanes,
                                 const int32_t* paths, int64_t n_paths, const float* g_agg, int64_t d_mid,
                                 float avg_num_neighbors, const float* num_neigh, int64_t n_edges, float* dx, void* dw,
                                 int64_t dw_ld, int edge_is_bf16, int64_t n_nodes, const int32_t* out_ptr,
                                 const int32_t* out_perm, float* dx_edges, int64_t lds_floats, int max_l, int max_mul,
                                 matten_stream_t stream);
/* max_mul (w-free only): the channel count of the widest block; a workgroup holds 256 / (lanes per edge) edges, so blocks of more
 * than 256 channels are refused (MATTEN_EINVAL) -- such a layer runs matten_tp_backward_lit on a materialised w. */
/* max_l (both functions): the largest degree among the paths' (l1, l2, l3); <= 2 selects the instantiation without the l = 3, 4
 * coupling code (about half the registers, twice the resident waves). */
/* lds_floats (2048 .. 15360): floats of LDS per workgroup for its [edge][path][column] weight tile; a block whose paths do not
 * fit takes them in rounds (plan.bw_wfree_lds_floats = what the widest block needs in one round, capped at 4096). */

/* adjoint of matten_species_linear w.r.t. the packed weights (the adjoint w.r.t. x is matten_species_linear
 * itself with the transposed segment table and transposed packed weights):
 *   dWp[s, w_off + u*mo + w] = sum_{rows n of species s} sum_k x[n, x_off+u*d+k] dy[n, o_off+w*d+k]
 * fp32 MFMA over (row, component); fixed summation order.  Every weight of the table's segments is written (dwp need
 * not be initialised).  scratch [matten_species_linear_wgrad_scratch_floats(n_rows, n_species, w_stride)], required when
 * that count is above 0 (large batches: a species' rows are cut into slices of 128 rows; a species of several slices
 * writes one scratch row per slice and a second launch adds them in slice order), else may be NULL */
int matten_species_linear_wgrad(const float* x, int64_t d_in, const float* dy, int64_t d_out, const int32_t* order,
                                const int32_t* seg, int64_t n_species, int64_t n_rows, const int32_t* segs,
                                int64_t n_segs, int64_t w_stride, float* dwp, float* scratch, matten_stream_t stream);

/* adjoint of the Gate part of matten_gate_bn (bn_weight == NULL forward); every column of dx is written (no atomics:
 * a gate's gradient is summed over its channel's components in order) */
int matten_gate_bwd(const float* x, int64_t d_in, const int32_t* meta, int64_t d_out, const float* act_cst,
                    const float* dy, int64_t n_rows, float* dx, matten_stream_t stream);

/* e3nn BatchNorm in training mode (batch statistics over all rows; reference nn/utils.py:418,432-433):
 *   chan[n_chan,4] int32 {column offset, 2l+1, is_0e, index into bias / running_mean or -1}; col2chan[dim]
 *   fwd: mean[c] (0e only, else 0), nu[c] = mean_n mean_k (x-mean)^2, y = (x-mean) rsqrt(nu+eps) weight[c] (+ bias)
 *   bwd: dx (A, B are [n_chan] scratch that return sum dy (x-mean) and sum dy: dweight = A rsqrt(nu+eps), dbias = B) */
int64_t matten_bn_scratch_floats(int64_t n_rows, int64_t dim);
int matten_bn_train_fwd(const float* x, int64_t dim, int64_t n_rows, const int32_t* col2chan, const int32_t* chan,
                        int64_t n_chan, const float* weight, const float* bias, float eps, float* mean, float* nu,
                        float* y, float* running_mean, float* running_var, float momentum, float* scratch,
                        matten_stream_t stream);
/* scratch [matten_bn_scratch_floats(n_rows, dim)] (fwd and bwd; may be NULL when that is 0): large batches reduce in two
 * stages, per-(16-row block, column) partial records first, merged per channel in a fixed order (0e channels by the
 * pairwise mean / squared-deviation update, so no E[x^2] - mean^2 cancellation) */
/* running_mean [number of 0e channels] / running_var [n_chan] (both or neither, may be NULL): updated in place by the
 * statistics kernel, running = (1 - momentum) running + momentum batch (e3nn BatchNorm in training mode) */
int matten_bn_train_bwd(const float* x, const float* dy, int64_t dim, int64_t n_rows, const int32_t* col2chan,
                        const int32_t* chan, int64_t n_chan, const float* mean, const float* nu, const float* weight,
                        float eps, float* A, float* B, float* dx, float* dweight, float* dbias, float* scratch,
                        matten_stream_t stream);
/* dweight [n_chan] = A rsqrt(nu + eps), dbias [number of 0e channels] = B of the 0e channels (both or neither, may be NULL) */

/* e3nn NormActivation as the reference configures it (nn/utils.py:142-150: nonlinearity_type "norm"; normalize = True,
 * epsilon = 1e-8, bias = False): every channel c (chan[c] = {column offset, 2l+1, is_0e, mean index}, as for
 * matten_bn_train_fwd; every column of the row belongs to one channel) is scaled by f(n) / n with
 * n = sqrt(max(sum_k x_k^2, epsilon^2)) and f the activation code `act` (1 silu, 2 tanh, 3 sigmoid, 4 shifted softplus,
 * 5 abs) applied as is (no second-moment normalisation).  With bn_weight != NULL the eval-mode BatchNorm that follows
 * the activation is folded in (weight / sqrt(running_var + bn_eps) per channel, bias - running_mean * scale on 0e).
 * matten_norm_act_bwd: the adjoint without BatchNorm (a clamped norm is a constant). */
int matten_norm_act(const float* x, int64_t dim, int64_t n_rows, const int32_t* chan, int64_t n_chan, int act,
                    float epsilon, const float* running_mean, const float* running_var, const float* bn_weight,
                    const float* bn_bias, float bn_eps, float* y, matten_stream_t stream);
int matten_norm_act_bwd(const float* x, const float* dy, int64_t dim, int64_t n_rows, const int32_t* chan, int64_t n_chan,
                        int act, float epsilon, float* dx, matten_stream_t stream);

/* Instance ("graph") normalisation, reference nn/utils.py:448-588 (the reference's own InstanceNorm: one set of
 * statistics per crystal, nodes as samples): matten_bn_train_fwd / _bwd with per-crystal statistics, in training and in
 * evaluation alike (it keeps no running averages).  Rows are grouped per crystal: seg_ptr[n_seg + 1] int64 row offsets,
 * seg_of_row[n_rows] int64 = the crystal of every row (the batch dict's `ptr` and `batch`).  chan as above, except that
 * the reference centres and biases every l = 0 channel (0e AND 0o: `ir.l == 0`, nn/utils.py:531,572), so chan[.][2]
 * is set for both.  mean / nu / A / B are [n_seg, n_chan]. */
int matten_instance_norm_fwd(const float* x, int64_t dim, int64_t n_rows, const int64_t* seg_ptr, const int64_t* seg_of_row,
                             int64_t n_seg, const int32_t* col2chan, const int32_t* chan, int64_t n_chan,
                             const float* weight, const float* bias, float eps, float* mean, float* nu, float* y,
                             matten_stream_t stream);
int matten_instance_norm_bwd(const float* x, const float* dy, int64_t dim, int64_t n_rows, const int64_t* seg_ptr,
                             const int64_t* seg_of_row, int64_t n_seg, const int32_t* col2chan, const int32_t* chan,
                             int64_t n_chan, const float* mean, const float* nu, const float* weight, float eps, float* A,
                             float* B, float* dx, matten_stream_t stream);

/* adjoint of matten_segment_reduce */
int matten_segment_reduce_bwd(const float* dy, int64_t dim, const int64_t* ptr, int64_t n_segments, int mean, float* dx,
                              matten_stream_t stream);


/* ==========================================================================================
 * Graph construction on the GPU (replaces neighbor_list_and_relative_vec,
 * data/data.py:285-413, i.e. ase.neighborlist.primitive_neighbor_list("ijS") + the self-edge filter).
 *   edges = all (i, j, S): | pos[j] + S.cell - pos[i] | < r_cut (strict, fp64), (i==j, S==0) excluded,
 *   emitted in lexicographic order (i, j, Sx, Sy, Sz); i, j are GLOBAL node ids (ptr-offset applied).
 *   pos[N,3] fp64; cell[B,9] fp64 (rows = lattice vectors); ptr[B+1] int64;
 *   pair_ptr[B+1] int64 = running sum of n_b^2: ordered atom pairs are numbered crystal by crystal, i-major.
 * matten_graph_prep (replaces the per-batch prologue of the reference's collate, data/dataset.py:150-152): per crystal
 *   the inverse cell in closed form, frac[N,3] = pos @ inv, bound[B,3] = r_cut |inv[:, k]| (a neighbour's shift
 *   along axis k lies within bound_k of -(frac_j - frac_i)_k), batch[N] int64, pos_f32[N,3], cell_f32[B,9].
 * Two passes: matten_neighbor_count -> counts[pair_ptr[B]] (edges of each ordered pair (i, j)); the caller scans
 * them into offsets[pair_ptr[B] + 1] (exclusive, int64); matten_neighbor_summary -> {n_edges, smallest edge count
 * of a crystal} (the builder's one read-back); matten_neighbor_fill writes edge_index[2,n_edges] (int64),
 * edge_cell_shift[n_edges,3] (fp32) and num_neigh[N] (fp32, optional: edges per centre atom, the reference's
 * bincount(i), data/data.py:401-411).  max_atoms = largest n_b; at most 65535 crystals per call.
 * The destination-sorted CSR of the list (what matten_csr_build derives from edge_index: destination = edge_index[1],
 * stable in the edge id) comes out of the same two passes when asked for: counts_t[pair_ptr[B]] (optional) receives each
 * pair's count at its j-major number pair_ptr[b] + (j - lo) n_b + (i - lo); the caller scans it into offsets_t like
 * offsets (offsets_t[0] is subtracted: the scan may continue the i-major one), and matten_neighbor_fill then also writes rowptr[n_atoms + 1], src_sorted[n_edges], perm[n_edges] (int32; all
 * four of offsets_t / rowptr / src_sorted / perm or none: NULL) -- bit for bit matten_csr_build's outputs.
 * ========================================================================================== */
int matten_graph_prep(const double* pos, const double* cell, const int64_t* ptr, int64_t n_crystals, double r_cut,
                      double* frac, double* bound, int64_t* batch, float* pos_f32, float* cell_f32,
                      matten_stream_t stream);
int matten_neighbor_count(const double* pos, const double* cell, const int64_t* ptr, const double* frac,
                          const double* bound, const int64_t* pair_ptr, double r_cut, int64_t n_crystals,
                          int64_t max_atoms, int32_t* counts, int32_t* counts_t, matten_stream_t stream);
int matten_neighbor_summary(const int64_t* offsets, const int64_t* pair_ptr, int64_t n_crystals, int64_t* out2,
                            matten_stream_t stream);
int matten_neighbor_fill(const double* pos, const double* cell, const int64_t* ptr, const double* frac,
                         const double* bound, const int64_t* pair_ptr, double r_cut, int64_t n_crystals,
                         int64_t max_atoms, const int64_t* offsets, int64_t n_edges, int64_t* edge_index,
                         float* edge_cell_shift, float* num_neigh, const int64_t* offsets_t, int64_t n_atoms,
                         int32_t* rowptr, int32_t* src_sorted, int32_t* perm, matten_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MATTEN_HIP_H */
