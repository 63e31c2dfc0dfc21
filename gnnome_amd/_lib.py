"""ctypes binding of libgnnome_hip.so (C ABI declared in include/gnnome_hip.h).

There is no CPU fallback: if the library is missing, or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GNNOME_HIP_LIB: another build of the SAME library (same ABI) for A/B measurements of two builds on one box - never a fallback
LIB_PATH = os.environ.get("GNNOME_HIP_LIB") or os.path.join(_HERE, "lib", "libgnnome_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gnnome_hip.h")

_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_int64
_sz = ctypes.c_size_t

# name -> argtypes, in the order of include/gnnome_hip.h
SIGNATURES = {
    "gnnome_abi_version": [],
    "gnnome_last_error": [],
    "gnnome_set_tuning": [_i, _i],
    "gnnome_debug_gate_profile": [_p],
    "gnnome_graph_views_workspace_bytes": [_l, _l, ctypes.POINTER(_sz)],
    "gnnome_build_graph_views": [_p, _p, _l, _l, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "gnnome_encode_f32": [_p, _l, _i, _p, _p, _p, _i, _p, _p, _i, _p, _p],
    "gnnome_linear_f32": [_p, _l, _i, _i, _p, _i, _p, _i, _p, _i, _p],
    "gnnome_linear_acc_f32": [_p, _l, _i, _i, _p, _i, _p, _i, _p, _i, _p],
    "gnnome_weight_planes_f16": [_p, _i, _i, _i, _p, _p],
    "gnnome_linear_planes_f32": [_p, _l, _i, _i, _p, _p, _i, _p, _i, _p],
    "gnnome_linear_planes_route": [_l, _i, _i, _i],
    "gnnome_edge_gate_f32": [_p, _p, _l, _i, _p, _p, _i, _p, _p, _p, _i, _i, _p, _p, _p],
    "gnnome_edge_gate_encode_f32": [_p, _p, _p, _p, _p, _p, _p, _l, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p],
    "gnnome_linear_ref_f32": [_p, _l, _i, _i, _p, _i, _p, _i, _p, _i, _p],
    "gnnome_edge_gate_ref_f32": [_p, _p, _l, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "gnnome_node_aggregate_f32": [_p, _i, _l, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p],
    "gnnome_build_node_records": [_p, _p, _p, _p, _p, _l, _p, _p],
    "gnnome_debug_node_records": [_p],
    "gnnome_node_aggregate_range_f32": [_p, _i, _l, _l, _l, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p],
    "gnnome_stream_schedule_sizes": [_l, _l, _i, ctypes.POINTER(_l), ctypes.POINTER(_sz)],
    "gnnome_build_stream_schedule": [_l, _l, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "gnnome_node_aggregate_stream_f32": [_p, _i, _l, _l, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p,
                                         _p, _l, _p, _p],
    "gnnome_edge_score_f32": [_p, _l, _i, _i, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p],
    "gnnome_debug_forward_events": [_p, _p, _p, _p, _i],
    "gnnome_model_forward_workspace_bytes": [_l, _l, _i, _i, ctypes.POINTER(_sz)],
    "gnnome_model_forward_f32": [_p, _p, _p, _p, _p, _p, _sz, _p],
    "gnnome_model_forward_buffers_f32": [_p, _p, _p, _p, _p, _p, _p],
    "gnnome_edge_gate_raw_f32": [_p, _p, _l, _i, _p, _p, _i, _p, _p, _p, _i, _p],
    "gnnome_edge_gate_raw_stats_rows": [_i, ctypes.POINTER(_i)],
    "gnnome_edge_gate_raw_stats_f32": [_p, _p, _l, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p],
    "gnnome_edge_gate_raw_stats_x16": [_p, _p, _l, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p],
    "gnnome_edge_gate_bn_f32": [_p, _p, _p, _l, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p],
    "gnnome_edge_gate_bn_x16": [_p, _p, _p, _l, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p],
    "gnnome_node_aggregate_raw_f32": [_p, _i, _l, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "gnnome_colsum_workspace_bytes": [ctypes.POINTER(_sz)],
    "gnnome_colsum2_f32": [_p, _p, _l, _i, _p, _p, _p, _p, _sz, _p],
    "gnnome_bn_relu_res_f32": [_p, _p, _p, _p, _l, _i, _p, _p],
    "gnnome_bn_relu_res_x16": [_p, _p, _p, _p, _l, _i, _p, _p],
    "gnnome_bn_bwd_stats_f32": [_p, _p, _p, _p, _p, _l, _i, _p, _p, _p, _sz, _p],
    "gnnome_bn_bwd_apply_f32": [_p, _p, _p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p],
    "gnnome_bn_bwd_apply_tables_f32": [_p, _p, _p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "gnnome_ln_relu_res_f32": [_p, _p, _p, _p, _l, _i, _i, _p, _p],
    "gnnome_ln_bwd_f32": [_p, _p, _p, _p, _l, _i, _i, _p, _p, _p, _p, _sz, _p],
    "gnnome_mul23_f32": [_p, _p, _p, _l, _p, _p, _p],
    "gnnome_add_f32": [_p, _p, _l, _p, _p],
    "gnnome_relu_bwd_f32": [_p, _p, _l, _p, _p],
    "gnnome_segment_sum_f32": [_p, _i, _p, _p, _l, _p, _i, _p],
    "gnnome_segment_sum2_f32": [_p, _i, _p, _p, _p, _l, _p, _i, _p, _i, _p],
    "gnnome_segment_sum2_amax_f32": [_p, _i, _p, _p, _p, _l, _p, _i, _p, _i, _p, _p],
    "gnnome_segment_sum2_x16": [_p, _i, _p, _p, _p, _l, _p, _i, _p, _i, _p],
    "gnnome_wgrad_workspace_bytes": [_l, _i, _i, ctypes.POINTER(_sz)],
    "gnnome_wgrad_f32": [_p, _i, _i, _p, _i, _i, _l, _p, _i, _p, _sz, _p],
    "gnnome_wgrad_scaled_f32": [_p, _i, _i, _p, _i, _i, _l, _p, _p, _i, _p, _sz, _p],
    "gnnome_wgrad_x16": [_p, _i, _i, _p, _i, _i, _l, _p, _i, _p, _sz, _p],
    "gnnome_wgrad_blocks_f32": [_p, _i, _i, _i, _p, _i, _i, _l, _p, _i, _p, _p, _sz, _p],
    "gnnome_wgrad_blocks_scaled_f32": [_p, _i, _i, _i, _p, _i, _i, _l, _p, _p, _i, _p, _p, _sz, _p],
    "gnnome_linear_blocks_f32": [_p, _i, _i, _l, _i, _p, _i, _i, _p, _i, _i, _p],
    "gnnome_linear_blocks_scaled_f32": [_p, _i, _i, _l, _i, _p, _i, _i, _p, _p, _i, _i, _p],
    "gnnome_score_tail_bwd_f32": [_p, _p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p],
    "gnnome_agg_edge_bwd_f32": [_p, _l, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p],
    "gnnome_encode_hidden_f32": [_p, _l, _i, _p, _p, _p, _i, _p, _p],
    "gnnome_gather_rows_f32": [_p, _i, _p, _l, _i, _p, _i, _p],
    "gnnome_scatter_add_rows_f32": [_p, _i, _p, _l, _i, _p, _i, _p],
    "gnnome_closure_workspace_bytes": [ctypes.POINTER(_sz)],
    "gnnome_degree_features_f32": [_p, _p, _l, _i, _p, _p, _sz, _p],
    "gnnome_edge_features_f32": [_p, _p, _l, _p, _p, _sz, _p],
    "gnnome_bn_bwd_dgrad_f32": [_p, _p, _l, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p],
    "gnnome_bn_bwd_dgrad_amax_f32": [_p, _p, _l, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p],
    "gnnome_bn_bwd_dgrad_out_f32": [_p, _p, _p, _l, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p],
    "gnnome_bn_bwd_dgrad_out_amax_f32": [_p, _p, _p, _l, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p],
    "gnnome_bn_bwd_dgrad_x16": [_p, _p, _l, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p],
    "gnnome_bn_bwd_dgrad_out_x16": [_p, _p, _p, _l, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p],
    "gnnome_agg_edge_bwd_stats_f32": [_p, _l, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "gnnome_agg_edge_bwd_stats_x16": [_p, _l, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "gnnome_agg_bwd_fused_f32": [_p, _l, _l, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "gnnome_agg_bwd_fused_x16": [_p, _l, _l, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "gnnome_bn_bwd_terms_f32": [_p, _p, _p, _l, _i, _p, _p, _p, _p],
    "gnnome_pack_layer_f32": [_p, _p, _p, _p, _i, _p, _p, _p, _p, _p],
    "gnnome_bn_train_finish_f32": [_p, _p, _p, _l, _i, _p, _p, _p, _p, _p, ctypes.c_float, ctypes.c_float, _i, _p, _p, _p, _p, _p],
    "gnnome_gate_center_f32": [_p, _l, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p],
    "gnnome_greedy_walks_workspace_bytes": [_l, _i, ctypes.POINTER(_sz)],
    "gnnome_greedy_walks": [_p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "gnnome_mark_walk_visited": [_p, _p, _p, _l, _p, _p],
    "gnnome_overlap_workspace_bytes": [ctypes.POINTER(_sz)],
    "gnnome_overlap_edit_distance": [_p, _p, _l, _p, _i, _p, _p, _p, _l, _p, _p, _p, _sz, _p],
    "gnnome_adjacency_support": [_p, _p, _p, _l, _l, _p, _p],
    "gnnome_bfs_levels": [_p, _p, _l, _p, _p, _p, _p, _p, _p, _p],
    "gnnome_hem_propose": [_p, _p, _p, _p, _p, _l, _i, _p, _p],
    "gnnome_kway_gains": [_p, _p, _p, _p, _l, _p, _p, _p],
    "gnnome_greedy_growing_host": [_p, _p, _p, _p, _l, _i, _p],
    "gnnome_edge_loss_f32": [_p, _p, _p, _l, _p, ctypes.c_float, ctypes.c_float, _p, _p, _p, _p, _p, _sz, _p],
}

ABI_VERSION = 17


# the parameter blocks of gnnome_model_forward_f32 (include/gnnome_hip.h), field for field
class LayerParams(ctypes.Structure):
    _fields_ = [(n, _p) for n in ("Wcat", "Wcat_planes", "bcat", "W3", "b3", "scale_e", "shift_e", "scale_h", "shift_h")] + [
        ("norm_kind", ctypes.c_int32), ("reference_order", ctypes.c_int32)]


class ModelParams(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ("hidden", "hidden_ne", "num_layers", "score_hidden", "node_features", "edge_features")]
                + [(n, _p) for n in ("node_W1", "node_b1", "node_W2", "node_b2", "edge_W1", "edge_b1", "edge_W2", "edge_b2")]
                + [("layers_host", ctypes.POINTER(LayerParams)), ("W_nodes", _p), ("W_nodes_planes", _p), ("b_nodes", _p), ("W1e", _p),
                   ("ld_w1e", ctypes.c_int32), ("reserved", ctypes.c_int32), ("W2", _p), ("b2", _p), ("W3", _p), ("b3", _p)])


class Views(ctypes.Structure):
    _fields_ = ([("num_nodes", ctypes.c_int64), ("num_edges", ctypes.c_int64)]
                + [(n, _p) for n in ("in_ptr", "srt_src", "srt_dst", "srt_eid", "out_ptr", "out_pos", "out_dst", "node_gather")]
                + [("transposed", ctypes.c_int32), ("reserved", ctypes.c_int32)])


class ForwardBuffers(ctypes.Structure):
    _fields_ = [("h", _p * 2), ("P", _p), ("e", _p * 2), ("PQ", _p)]


NORM_AFFINE = 0
NORM_LAYER = 1

_lib = None


class GnnomeHipError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise GnnomeHipError(
            f"{LIB_PATH} not found - the HIP extension is not built. Run `make -C gnnome_amd/csrc` "
            "(or __graft_entry__.build()). gnnome_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name == "gnnome_last_error" else _i
    got = lib.gnnome_abi_version()
    if got != ABI_VERSION:
        raise GnnomeHipError(f"libgnnome_hip.so ABI {got} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gnnome_last_error()
        raise GnnomeHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
