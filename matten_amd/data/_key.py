"""
Names of the entries of the backbone's data dict (the operator boundary).
Same strings as the reference's ``matten.data._key`` (data/_key.py:14-49) so dicts interoperate.
"""
from typing import Dict

import torch

Type = Dict[str, torch.Tensor]

# geometry / graph (inputs)
POSITIONS = "pos"
CELL = "cell"
EDGE_INDEX = "edge_index"
EDGE_CELL_SHIFT = "edge_cell_shift"
NUM_NEIGH = "num_neigh"
ATOMIC_NUMBERS = "atomic_numbers"
BATCH = "batch"
PTR = "ptr"

# written by the backbone
SPECIES_INDEX = "species_index"
NODE_ATTRS = "node_attrs"
NODE_FEATURES = "node_features"
EDGE_VECTORS = "edge_vectors"
EDGE_LENGTH = "edge_lengths"
EDGE_ATTRS = "edge_attrs"
EDGE_EMBEDDING = "edge_embedding"
EDGE_MESSAGE = "edge_message"

# kept for dict compatibility with reference-side code (unused on this path)
PER_ATOM_ENERGY = "atomic_energy"
TOTAL_ENERGY = "total_energy"

# private entries of the MI355X backbone (dst-sorted CSR view of the graph, shared by all layers)
AMD_PERM = "_amd_perm"            # [E] i32  sorted position -> original edge id
AMD_ROWPTR = "_amd_rowptr"        # [N+1] i32
AMD_SRC = "_amd_src_sorted"       # [E] i32
AMD_GEOM = "_amd_geom_sorted"     # [E,4] f32 (vx,vy,vz,|v|)
AMD_SH = "_amd_sh_sorted"         # [E,(lmax+1)^2] f32
AMD_SPECIES = "_amd_species_order"  # (order[N] i32 nodes sorted by species, seg[S+1] i32)
AMD_SPECIES_I32 = "_amd_species_i32"  # [N] i32 species index per node (the conv-fused kernel's per-node weight row)
AMD_RBF = "_amd_rbf_params"       # [3] f64 cpu tensor (num_basis, start, end)
