"""
ctypes binding of libmatten_hip.so (C ABI declared in include/matten_hip.h).

There is no fallback: if the shared library is missing or a symbol is absent, importing the
compute path fails loudly.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C matten_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmatten_hip.so")

ABI_VERSION = 44

# name -> (restype, argtypes); must match include/matten_hip.h
P = c_void_p
SIGNATURES = {
    "matten_abi_version": (c_int, []),
    "matten_csr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "matten_group_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "matten_csr_counting_max_avg_degree": (c_int, []),
    "matten_csr_build": (c_int, [P, c_int64, c_int64, P, P, P, P, c_size_t, P, P]),
    "matten_csr_split_bound": (c_int64, [c_int64, c_int64, c_int64]),
    "matten_csr_split": (c_int, [P, c_int64, c_int64, c_int64, P, P, P, P, P]),
    "matten_group_by_key": (c_int, [P, c_int64, c_int64, P, P, P, c_size_t, P, P]),
    "matten_species_embed": (c_int, [P, c_int64, P, c_int64, c_int64, c_int64, P, P, c_int64, P, P, P, P, P, P]),
    "matten_edge_geom": (c_int, [P, P, P, P, c_int64, P, P, c_int64, c_int64, c_int, c_int, c_float, c_float, P, P, c_int, P, P, P, P, P]),
    "matten_radial_mlp": (c_int, [P, c_int64, c_int, c_float, c_float, P, c_int, P, P, c_int, c_int, c_float, P, c_int, P]),
    "matten_tp_tile_nodes": (c_int, []),
    "matten_tp_paths": (c_int, [P, c_int64, P, c_int64, P, c_int64, P, P, c_int64, P, P, c_int64, c_int64, c_int64, c_float, P, P, c_int,
                                P]),
    "matten_radial_hidden_multi": (c_int, [P, c_int64, c_int, c_float, c_float, P, c_int, P, c_int, P, P, c_int, P]),
    "matten_agg_linear_max_mt": (c_int, []),
    "matten_agg_linear_block_chunks": (c_int, []),
    "matten_agg_linear_lds_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "matten_agg_linear_max_lds_bytes": (c_size_t, []),
    "matten_agg_linear": (c_int, [P, c_int64, P, P, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_int64, c_int64,
                                  c_int64, P, P]),
    "matten_agg_linear_gate": (c_int, [P, c_int64, P, P, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_int64, c_int64,
                                       c_int64, P, P, P, P, c_int64, P, P]),
    "matten_agg_linear_gate_sets": (c_int, []),
    "matten_radial_hidden": (c_int, [P, c_int64, c_int, c_float, c_float, P, c_int, P, c_int, P, P, P]),
    "matten_tp_max_cols": (c_int, []),
    "matten_tp_max_cols_l0": (c_int, []),
    "matten_tp_max_cols_l1": (c_int, []),
    "matten_tp_groups_hash": (c_int, []),
    "matten_tp_fused": (c_int, [P, c_int64, P, P, c_int64, P, c_int64, P, P, c_int64, P, P, c_int64, c_int64, c_int64, c_int64, c_float, P, P, P, P, P]),
    "matten_species_linear": (c_int, [P, c_int64, P, P, c_int64, P, c_int64, P, c_int64, c_int64, P, c_int64, c_int64, P, P]),
    "matten_species_linear_rows": (c_int, [P, c_int64, P, P, c_int64, P, c_int64, P, c_int64, c_int64, P, c_int64, c_int64, P, P]),
    "matten_radial_mlp_bwd_small_slices": (c_int64, [c_int64]),
    "matten_radial_mlp_bwd_w2_ranges": (c_int64, [c_int64, c_int64]),
    "matten_radial_mlp_bwd": (c_int, [P, c_int64, c_int, c_float, c_float, P, c_int, P, P, c_int, c_int, c_int, P, c_int64,
                                      c_int, P, P, P, c_float, c_float, c_float, P, P, P]),
    "matten_radial_pack": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, P, P, P, P]),
    "matten_radial_pack_cols": (c_int, [P, P, P, c_int, c_int, c_int, P, c_int, c_int, c_float, c_float, c_float, P, P, P, P]),
    "matten_radial_h_scale": (c_int, [P, P, c_int, c_float, c_float, c_float, P, P]),
    "matten_split_a_tiles": (c_int, [P, c_int64, P, c_int64, P, P, P, P]),
    "matten_gather_scale": (c_int, [P, P, P, c_int64, c_int64, c_int, P, P, P, P]),
    "matten_species_linear_wgrad_scratch_floats": (c_int64, [c_int64, c_int64, c_int64]),
    "matten_tp_backward": (c_int, [P, c_int64, P, c_int64, P, c_int64, P, P, P, c_int64, P, P, P, c_int64, c_float, P, c_int64, P, P, c_int64,
                                   P, P, c_int64, c_int, P]),
    "matten_tp_backward_lit": (c_int, [P, c_int64, P, c_int64, P, c_int64, P, P, P, c_int64, c_int64, P, c_int64, P, c_int64, c_float,
                                       P, c_int64, P, P, c_int64, c_int, c_int64, P, P, P, c_int, P]),
    "matten_tp_backward_lit_wfree": (c_int, [P, c_int64, P, P, P, P, c_int64, P, P, P, c_int64, c_int64, P, c_int64, P, c_int64, c_float,
                                             P, c_int64, P, P, c_int64, c_int, c_int64, P, P, P, c_int64, c_int, c_int, P]),
    "matten_adam_step": (c_int, [P, P, P, P, c_int64, P, c_float, c_float, c_float, c_float, c_float, P]),
    "matten_species_linear_wgrad": (c_int, [P, c_int64, P, c_int64, P, P, c_int64, c_int64, P, c_int64, c_int64, P, P, P]),
    "matten_gate_bwd": (c_int, [P, c_int64, P, c_int64, P, P, c_int64, P, P]),
    "matten_bn_scratch_floats": (c_int64, [c_int64, c_int64]),
    "matten_bn_train_fwd": (c_int, [P, c_int64, c_int64, P, P, c_int64, P, P, c_float, P, P, P, P, P, c_float, P, P]),
    "matten_bn_train_bwd": (c_int, [P, P, c_int64, c_int64, P, P, c_int64, P, P, P, c_float, P, P, P, P, P, P, P]),
    "matten_norm_act": (c_int, [P, c_int64, c_int64, P, c_int64, c_int, c_float, P, P, P, P, c_float, P, P]),
    "matten_norm_act_bwd": (c_int, [P, P, c_int64, c_int64, P, c_int64, c_int, c_float, P, P]),
    "matten_instance_norm_fwd": (c_int, [P, c_int64, c_int64, P, P, c_int64, P, P, c_int64, P, P, c_float, P, P, P, P]),
    "matten_instance_norm_bwd": (c_int, [P, P, c_int64, c_int64, P, P, c_int64, P, P, c_int64, P, P, P, c_float, P, P, P, P]),
    "matten_segment_reduce_bwd": (c_int, [P, c_int64, P, c_int64, c_int, P, P]),
    "matten_graph_prep": (c_int, [P, P, P, c_int64, ctypes.c_double, P, P, P, P, P, P]),
    "matten_neighbor_count": (c_int, [P, P, P, P, P, P, ctypes.c_double, c_int64, c_int64, P, P, P]),
    "matten_neighbor_summary": (c_int, [P, P, c_int64, P, P]),
    "matten_neighbor_fill": (c_int, [P, P, P, P, P, P, ctypes.c_double, c_int64, c_int64, P, c_int64, P, P, P, P, c_int64,
                                     P, P, P, P]),
    "matten_gate_bn": (c_int, [P, c_int64, P, c_int64, P, P, P, P, P, c_float, c_int64, P, P]),
    "matten_segment_reduce": (c_int, [P, c_int64, P, c_int64, c_int, P, P]),
    "matten_segment_minmax": (c_int, [P, c_int64, P, c_int64, c_int, P, P, P]),
    "matten_segment_minmax_bwd": (c_int, [P, c_int64, P, c_int64, P, P]),
    "matten_dense_rows": (c_int, [P, c_int64, P, c_int64, c_int64, P, P]),
}

_lib = None


class MattenHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and return the library with argtypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MattenHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. matten_amd has no CPU fallback; "
            "run `make -C matten_amd/csrc` (needs hipcc, --offload-arch=gfx950)."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MattenHipError(f"symbol {name} missing from {LIB_PATH}") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.matten_abi_version()
    if v != ABI_VERSION:
        raise MattenHipError(f"ABI version mismatch: library {v}, python binding {ABI_VERSION}")
    _lib = lib
    return lib


_ERRORS = {-1: "MATTEN_EINVAL (bad argument)", -2: "MATTEN_ELAUNCH (HIP launch failed)", -3: "MATTEN_ENOMEM (workspace too small)"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise MattenHipError(f"{what} failed: {_ERRORS.get(rc, rc)}")
