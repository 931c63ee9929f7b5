"""ctypes binding of libbliss_gnn.so (the C ABI in include/bliss_gnn.h).

There is NO fallback: if the HIP library is missing or a symbol is absent, importing this
module raises.  The product path never routes through oracle/ or torch re-implementations.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbliss_gnn.so")

EINVAL = -1
MODE_BANDIT, MODE_LADIES, MODE_UNIFORM_NODES, MODE_PARTIALS = 0, 1, 4, 8

ERR_BITS = {
    1: "frontier larger than 2^31-1 edges",
    2: "candidate capacity exceeded / seed id out of range",
    4: "kept-node capacity exceeded",
    8: "block-edge capacity exceeded",
    16: "non-finite or negative weight reached an exact reduction",
    32: "exact sum left its fixed-point range",
    64: "seed capacity exceeded",
    128: "random stream ran short / generator made no progress",
    256: "a cross-stream flag never came (bliss_flag_wait timed out)",
}


class Graph(C.Structure):
    _fields_ = [("indptr", C.c_void_p), ("indices", C.c_void_p), ("eid", C.c_void_p),
                ("num_nodes", C.c_int32), ("num_edges", C.c_int64)]


class NodeMaps(C.Structure):
    _fields_ = [("local_id", C.c_void_p), ("first_pos", C.c_void_p), ("acc_p2", C.c_void_p)]


class LayerCounts(C.Structure):
    _fields_ = [("S", C.c_int32), ("E", C.c_int32), ("C", C.c_int32), ("K", C.c_int32), ("B", C.c_int32),
                ("err", C.c_int32), ("iters", C.c_int32), ("all_one", C.c_int32), ("c", C.c_double)]


class LayerWs(C.Structure):
    _fields_ = [("counts", C.c_void_p), ("seg_ptr", C.c_void_p), ("seed_acc", C.c_void_p), ("chunk_cnt", C.c_void_p),
                ("cand_nid", C.c_void_p), ("p", C.c_void_p), ("P", C.c_void_p), ("new_id", C.c_void_p),
                ("kept_nid", C.c_void_p), ("node_prob", C.c_void_p), ("hist", C.c_void_p), ("src_cnt", C.c_void_p), ("cap_c", C.c_int32), ("cap_k", C.c_int32),
                ("n_bins", C.c_int32), ("bin_cap", C.c_int64), ("bin_cursor", C.c_void_p), ("bin_rec", C.c_void_p),
                ("bitmap", C.c_void_p), ("word_prefix", C.c_void_p), ("touched_key", C.c_void_p), ("touched_sum", C.c_void_p),
                ("span_seg", C.c_void_p), ("kept_rec", C.c_void_p), ("span_cnt", C.c_void_p), ("kept_rec_positions", C.c_int64),
                ("kept_map", C.c_void_p), ("entry_flag", C.c_void_p), ("w_pend", C.c_void_p),
                ("fs_ticket", C.c_void_p), ("fs_fanout", C.c_int32), ("fs_is_last", C.c_int32), ("fs_rng_cap", C.c_int32),
                ("fs_reserved", C.c_int32), ("fs_eps", C.c_double), ("fs_rng_ctl", C.c_void_p), ("fs_layer_off", C.c_void_p),
                ("block_ready_flag", C.c_void_p)]


class Exp3Block(C.Structure):
    _fields_ = [("w_pos", C.c_void_p), ("row_sum", C.c_void_p), ("scratch", C.c_void_p), ("norm_out", C.c_void_p),
                ("blk_indptr", C.c_void_p), ("blk_src", C.c_void_p), ("blk_dst", C.c_void_p), ("blk_pos", C.c_void_p),
                ("q_ij", C.c_void_p), ("node_prob", C.c_void_p), ("embed_norm", C.c_void_p), ("alpha_or_null", C.c_void_p),
                ("dst_nid", C.c_void_p), ("n_edges_dev", C.c_void_p), ("rewards_out", C.c_void_p), ("edges_bound", C.c_int32),
                ("norm_pend", C.c_void_p)]


EXP3_MAX_BLOCKS = 8


class Exp3RankLists(C.Structure):                      # bliss_exp3_rank_lists_t
    _fields_ = [("w_pos", C.c_void_p * EXP3_MAX_BLOCKS), ("row_sum", C.c_void_p * EXP3_MAX_BLOCKS),
                ("pos_off_words", C.c_int32 * EXP3_MAX_BLOCKS), ("factor_off_bf16", C.c_int32 * EXP3_MAX_BLOCKS),
                ("count_off_words", C.c_int32 * EXP3_MAX_BLOCKS), ("bound", C.c_int32 * EXP3_MAX_BLOCKS),
                ("n_blocks", C.c_int32), ("n_ranks", C.c_int32), ("rank_stride_words", C.c_int64)]


class PackLists(C.Structure):                          # bliss_pack_lists_t
    _fields_ = [("pos", C.c_void_p * EXP3_MAX_BLOCKS), ("n_dev", C.c_void_p * EXP3_MAX_BLOCKS),
                ("pos_off_words", C.c_int32 * EXP3_MAX_BLOCKS), ("count_off_words", C.c_int32 * EXP3_MAX_BLOCKS),
                ("bound", C.c_int32 * EXP3_MAX_BLOCKS), ("n_blocks", C.c_int32)]


class TileGemm(C.Structure):                           # bliss_tile_gemm_t
    _fields_ = [("a1", C.c_void_p), ("a1_stride", C.c_int64), ("ids", C.c_void_p),
                ("w1", C.c_void_p), ("w1_stride", C.c_int64), ("k1", C.c_int32),
                ("a2", C.c_void_p), ("a2_stride", C.c_int64), ("w2", C.c_void_p), ("w2_stride", C.c_int64), ("k2", C.c_int32),
                ("bias", C.c_void_p), ("m_bound", C.c_int32), ("m_dev", C.c_void_p), ("n", C.c_int32),
                ("out", C.c_void_p), ("out_stride", C.c_int64), ("a_copy", C.c_void_p), ("copy_stride", C.c_int64),
                ("in_norm", C.c_void_p), ("out_norm", C.c_void_p),
                ("relu", C.c_int32), ("drop_p", C.c_float), ("drop_seed", C.c_uint32), ("drop_ctr", C.c_void_p)]


class DGrad(C.Structure):                              # bliss_dgrad_t
    _fields_ = [("a1", C.c_void_p), ("a1_stride", C.c_int64), ("w1", C.c_void_p), ("w1_stride", C.c_int64), ("k1", C.c_int32),
                ("a2", C.c_void_p), ("a2_stride", C.c_int64), ("w2", C.c_void_p), ("w2_stride", C.c_int64), ("k2", C.c_int32),
                ("m2_bound", C.c_int32), ("m2_dev", C.c_void_p),
                ("m_bound", C.c_int32), ("m_dev", C.c_void_p), ("n", C.c_int32), ("out", C.c_void_p), ("out_stride", C.c_int64)]


class WGrad(C.Structure):                              # bliss_wgrad_t
    _fields_ = [("d", C.c_void_p), ("d_stride", C.c_int64), ("n_out", C.c_int32),
                ("x", C.c_void_p), ("x_stride", C.c_int64), ("k_in", C.c_int32),
                ("rows_bound", C.c_int32), ("rows_dev", C.c_void_p),
                ("dw", C.c_void_p), ("dw_stride", C.c_int64), ("db", C.c_void_p)]


class GatFused(C.Structure):                           # bliss_gat_fused_t
    _fields_ = [("indptr", C.c_void_p), ("src", C.c_void_p), ("n_dst", C.c_int32), ("n_dst_dev", C.c_void_p),
                ("feat", C.c_void_p), ("feat_stride", C.c_int64), ("attn", C.c_void_p), ("heads", C.c_int32), ("head_dim", C.c_int32),
                ("negative_slope", C.c_float), ("e", C.c_void_p), ("a", C.c_void_p), ("a_drop", C.c_void_p),
                ("rst", C.c_void_p), ("rst_stride", C.c_int64),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint32), ("drop_ctr", C.c_void_p), ("drop_ctr_used", C.c_void_p),
                ("g", C.c_void_p), ("g_stride", C.c_int64), ("de", C.c_void_p), ("d_er", C.c_void_p), ("d_er_stride", C.c_int64),
                ("dattn_part", C.c_void_p),
                ("wg_row", C.c_void_p), ("n_wg_dev", C.c_void_p), ("cap_wg", C.c_int32), ("row_ws", C.c_void_p), ("seg_part", C.c_void_p),
                ("err", C.c_void_p)]


ADAM_MAX_TENSORS = 32


class AdamTensors(C.Structure):                        # bliss_adam_t
    _fields_ = [("param", C.c_void_p * ADAM_MAX_TENSORS), ("grad", C.c_void_p * ADAM_MAX_TENSORS),
                ("exp_avg", C.c_void_p * ADAM_MAX_TENSORS), ("exp_avg_sq", C.c_void_p * ADAM_MAX_TENSORS),
                ("numel", C.c_int64 * ADAM_MAX_TENSORS), ("count", C.c_int32)]


class BlockOut(C.Structure):
    _fields_ = [("indptr", C.c_void_p), ("src", C.c_void_p), ("dst", C.c_void_p), ("pos", C.c_void_p),
                ("eid", C.c_void_p), ("edge_weights", C.c_void_p), ("q_ij", C.c_void_p), ("t_indptr", C.c_void_p),
                ("t_edge", C.c_void_p), ("t_scratch", C.c_void_p), ("cap_b", C.c_int32)]


_P, _I32, _I64, _F, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double

# name -> argtypes; every symbol declared in include/bliss_gnn.h must appear here (tests check)
SIGNATURES = {
    "bliss_layer_counts_bytes": [],
    "bliss_frontier_prob": [C.POINTER(Graph), C.POINTER(NodeMaps), _P, _P, _I32, _P, _I32, C.c_int, _F, _F, _I64, C.POINTER(LayerWs), _P],
    "bliss_rng_stream_begin": [_P, _P, _P, _P, _I32, _P],
    "bliss_rng_stream_wait": [_P, _P, _P, C.c_int, _I32, _P],
    "bliss_rng_stream_end": [_P, _P, _P, _I32, _P, _P],
    "bliss_rng_stream_chain": [_P, _P, _P, _P, _I32, _P, _I32, _P, _P],
    "bliss_rng_stream_ready": [_P],
    "bliss_rng_prepare": [_I32, _P],
    "bliss_mt_jump_poly": [_I64, _P],
    "bliss_flag_wait": [_P, _P, _P],
    "bliss_flag_set_spin_bound": [C.c_int64],
    "bliss_flag_raise": [_P, _P],
    "bliss_gather_rows": [_P, _I64, _P, _I32, _I32, _P, _I64, _P, _P],
    "bliss_exp3_apply_ranks": [C.POINTER(Exp3RankLists), _P, _P, _P, _P],
    "bliss_pack_lists": [C.POINTER(PackLists), _P, _P],
    "bliss_mt19937_uniform": [_P, _P, _I32, _P, _I32, _P],
    "bliss_poisson_select": [C.POINTER(LayerWs), _I32, _D, _P, _P, _P, C.c_int, _I32, _I64, _P],
    "bliss_multinomial_select": [C.POINTER(LayerWs), _P, _I32, _P],
    "bliss_tile_gemm": [C.POINTER(TileGemm), C.POINTER(TileGemm), _P],
    "bliss_cross_entropy": [_P, _I64, _P, _I32, _I32, _P, _P, _I64, _P, _P, _P, _P],
    "bliss_cross_entropy_sum": [_P, _I64, _P, _I64, _P, _P, _I32, _I32, _P, _P, _I64, _P, _P, _P, _P],
    "bliss_cross_entropy_masked": [_P, _I64, _P, _I64, _P, _I32, _P, _I32, _I32, _P, _F, _I32, _P, _P, _I64, _P, _P, _P, _P],
    "bliss_adam_step": [C.POINTER(AdamTensors), _P, _F, _F, _F, _F, _P],
    "bliss_cand_importance": [_P, _I32, C.c_int, _P, _P, _P],
    "bliss_poisson_scale": [_P, _P, _I32, _D, _P, _P],
    "bliss_keyed_select": [_P, _P, _P, _I32, _P, C.c_uint64, C.c_uint64, _I32, _P, _P, _P],
    "bliss_exp3_normalize_global": [_P, _I64, _P, _P, _P, _P, _P],
    "bliss_build_block": [C.POINTER(Graph), C.POINTER(NodeMaps), _P, _P, _I32, C.c_int, _F, _F, _I64, C.POINTER(LayerWs), C.POINTER(BlockOut), _P],
    "bliss_normalized_edata": [C.POINTER(Graph), _P, _P],
    "bliss_embed_norm": [_P, _I32, _I32, _I64, _P, _P],
    "bliss_spmm_fwd": [_P, _P, _P, _P, _P, _I64, _I32, _P, _I32, _I32, C.c_int, _P, _I64, C.c_int, _P, _P],
    "bliss_spmm_bwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _I32, _I32, C.c_int, _P, _I64, C.c_int, _P, _P],
    "bliss_block_transpose": [_P, _P, _I32, _I32, _I32, _P, _P, _P, _I64, _P],
    "bliss_spmm_chunk_edges": [_I32],
    "bliss_sage_epilogue_fwd": [_P, _I64, _P, _I64, _I32, _I32, _F, C.c_uint32, _P, _P, _I64, _P, _P],
    "bliss_sage_epilogue_bwd": [_P, _I64, _P, _I64, _I32, _I32, _F, _P, _I64, _P],
    "bliss_gat_chunk_edges": [],
    "bliss_graph_prepare": [_P, _P, _I64, _I32, C.c_int, _P, _P, _P, _P, _P, _P, _I64, _P],
    "bliss_exp3_update": [C.POINTER(Graph), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _I32, _F, _P, _P, C.c_int, _P, _P],
    "bliss_exp3_step": [C.POINTER(Graph), _P, C.POINTER(Exp3Block), _I32, _F, _P, _P],
    "bliss_exp3_normalize_global_rows": [C.POINTER(Exp3Block), _I32, _I64, _P, _I64, _P],
    "bliss_exp3_update_blocks": [C.POINTER(Graph), _P, C.POINTER(Exp3Block), _I32, _F, _P, _P],
    "bliss_exp3_step_deferred": [C.POINTER(Graph), _P, C.POINTER(Exp3Block), _I32, _F, _P, _P, _P],
    "bliss_exp3_normalize_pending": [C.POINTER(Exp3Block), _I32, _I64, _P],
    "bliss_exp3_apply": [_P, _P, _P, _P, _P, _I32, _P, _P],
    "bliss_exp3_normalize": [_P, _I64, _P, _P, _P, _P],
    "bliss_row_sum": [_P, _I64, _P, _P],
    "bliss_gat_logits": [_P, _P, _P, _I32, _P, _I64, _P, _I32, _I32, _F, _P, _P],
    "bliss_sage_dgrad": [C.POINTER(DGrad), _P],
    "bliss_sage_wgrad": [C.POINTER(WGrad), _I32, _P, _I64, _P],
    "bliss_shard_local_seeds": [_P, _I32, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P],
    "bliss_shard_scatter_partials": [_P, _P, _P, _P, _P, _P, _P, _I32, _P, _P],
    "bliss_shard_zero_dense": [_P, _I32, _P],
    "bliss_shard_pack_rows": [_P, _P, _I32, _I32, _I32, _P, _I64, _I32, _P, _I64, _P],
    "bliss_shard_place_rows": [_P, _I64, _P, _P, _I32, _P, _I64, _I32, _I32, _P],
    "bliss_shard_take_rows": [_P, _I32, _I64, _I32, _P, _P, _I32, _P, _I64, _I32, _P],
    "bliss_shard_candidates": [_P, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P, _P, _P],
    "bliss_shard_select_kept": [_P, _P, _P, _P, C.c_uint64, _P, _I32, _P, _I32, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _I32, _P, _P, _P],
    "bliss_gat_fused_supported": [_I32, _I32],
    "bliss_gat_segment_edges": [],
    "bliss_gat_fused_stamps": [_P, _P],
    "bliss_gat_segments": [_P, _I32, _I32, _P, _P, _P, _P],
    "bliss_gat_fused_fwd": [C.POINTER(GatFused), _P],
    "bliss_gat_fused_bwd_dst": [C.POINTER(GatFused), _P, _P, _P, _P],
    "bliss_gat_rows_src_fused": [_P, _I32, _P, _P, _P, _P, _I32, _P, _P, _P, _I64, _P, _I64, _P, _I32, _I32, _F, _P, _I64, _P, _P],
    "bliss_gat_logits_f32": [_P, _P, _P, _I32, _P, _I64, _P, _I32, _I32, _F, _P, _P],
    "bliss_gat_edge_dot": [_P, _P, _P, _I32, _P, _I64, _P, _I64, _I32, _I32, _P, _P],
    "bliss_gat_edge_softmax": [_P, _I32, _P, _P, _I32, C.c_int, _P, _P],
    "bliss_gat_rows": [C.c_int, _P, _I32, _P, _P, _P, _P, _I32, _P, _P, _I64, _P, _I32, _I32, _F, _P, _I64, _P, _P, _P],
    "bliss_gat_alpha": [_P, _I32, _P, _P, _P, _P, _P],
    "bliss_prof_enable": [C.c_int],
    "bliss_prof_reset": [],
    "bliss_prof_read": [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    "bliss_prof_kernel_count": [],
}


SPECIAL_SIGNATURES = ("bliss_prof_kernel_name", "bliss_block_transpose_temp_bytes", "bliss_graph_prepare_capacity",
                      "bliss_graph_prepare_temp_bytes", "bliss_rng_stream_handle", "bliss_sage_wgrad_workspace")   # non-int return types, set in _load()


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C bliss_gnn_amd/csrc`.  bliss_gnn_amd has no CPU or PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.bliss_block_transpose_temp_bytes.argtypes = [_I32, _I32]
    lib.bliss_block_transpose_temp_bytes.restype = C.c_int64
    lib.bliss_sage_wgrad_workspace.argtypes = [C.POINTER(WGrad), _I32]
    lib.bliss_sage_wgrad_workspace.restype = C.c_int64
    for name in ("bliss_graph_prepare_capacity", "bliss_graph_prepare_temp_bytes"):
        getattr(lib, name).argtypes = [_I64, _I32, C.c_int]
        getattr(lib, name).restype = C.c_int64
    lib.bliss_rng_stream_handle.argtypes = []
    lib.bliss_rng_stream_handle.restype = C.c_int64
    lib.bliss_prof_kernel_name.argtypes = [C.c_int]
    lib.bliss_prof_kernel_name.restype = C.c_char_p
    assert lib.bliss_layer_counts_bytes() == C.sizeof(LayerCounts), "LayerCounts layout mismatch"
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}" + (" (invalid argument)" if rc == EINVAL else " (hipError_t)"))


def err_string(bits):
    return "; ".join(msg for b, msg in ERR_BITS.items() if bits & b) or "unknown"
