"""ctypes binding of libsgf.so (the C ABI declared in include/sgf.h).

There is deliberately no fallback: if the shared library is missing or a call fails, the product
path raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsgf.so")

SGF_F32 = 0
SGF_BF16 = 1

_lib = None

# name -> (restype, argtypes).  Order and types mirror include/sgf.h exactly.
_P = c_void_p
SIGNATURES = {
    "sgf_version": (c_int32, []),
    "sgf_last_error": (c_char_p, []),
    "sgf_reload_env": (c_int32, []),
    "sgf_csr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sgf_csr_build": (c_int32, [_P, c_int64, c_int64, _P, _P, _P, _P, _P, c_size_t, _P]),
    "sgf_csr_transpose": (c_int32, [_P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "sgf_subgraph_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sgf_subgraph_plan": (c_int32, [_P, c_int64, c_int64, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "sgf_subgraph_emit": (c_int32, [_P, c_int64, c_int64, _P, c_int32, c_int64, _P, _P, _P, c_size_t, _P]),
    "sgf_subgraph_csr_plan_workspace_bytes": (c_size_t, [c_int64]),
    "sgf_subgraph_csr_emit_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sgf_subgraph_csr_plan": (c_int32, [_P, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_size_t, _P]),
    "sgf_subgraph_csr_emit": (c_int32, [_P, _P, c_int64, _P, c_int64, _P, _P, _P, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "sgf_graph_prologue_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sgf_graph_prologue_plan": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "sgf_graph_prologue_emit": (c_int32, [c_int64, c_int64, c_int32, c_int32, c_int64, _P, _P, c_size_t, _P]),
    "sgf_spmm": (c_int32, [_P, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P]),
    "sgf_spmm_segment_len": (c_int32, []),
    "sgf_spmm_split_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "sgf_spmm_split": (c_int32, [_P, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, c_int32, c_int32, c_int64,
                                 c_int64, _P, c_size_t, _P]),
    "sgf_spmm_stream": (c_int32, [_P, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, c_int32, c_int32, c_int64,
                                  c_int64, _P, c_size_t, _P]),
    "sgf_reorder_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sgf_reorder": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, _P, c_size_t, _P]),
    "sgf_spmm_plan_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32]),
    "sgf_spmm_plan": (c_int32, [_P, _P, _P, c_int64, c_int64, c_int32, c_int32, c_int64, _P, _P, _P, _P, _P, _P,
                                _P, c_size_t, _P]),
    "sgf_spmm_lds_rows_len": (c_int32, [c_int32]),
    "sgf_spmm_blocked": (c_int32, [_P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int32, c_int32,
                                   c_int32, c_int32, c_int64, c_int64, _P, c_size_t, _P]),
    "sgf_spmm_tile_supported": (c_int32, [c_int32, c_int32]),
    "sgf_spmm_tile_blocks": (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P, _P]),
    "sgf_spmm_tile_plan_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "sgf_spmm_tile_plan": (c_int32, [_P, _P, _P, c_int64, c_int64, _P, c_int64, c_int32, c_int32, c_int64, _P, _P, _P,
                                     _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "sgf_spmm_tile_fill": (c_int32, [_P, _P, _P, _P, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P]),
    "sgf_spmm_tile_sparse_len": (c_int32, []),
    "sgf_spmm_tile_pack_workspace_bytes": (c_size_t, [c_int64]),
    "sgf_spmm_tile_pack_layout": (c_int32, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "sgf_spmm_tile_pack": (c_int32, [_P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P]),
    "sgf_spmm_tile": (c_int32, [_P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, _P, c_int64,
                                c_int64, c_int32, c_int32, c_int64, c_int64, _P, c_size_t, _P]),
    "sgf_neighbor_sample_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "sgf_neighbor_sample_mark": (c_int32, [_P, _P, c_int64, c_int32, _P]),
    "sgf_neighbor_sample_batch_workspace_bytes": (c_size_t, [c_int64, _P, c_int32, _P, _P]),
    "sgf_neighbor_sample_batch": (c_int32, [_P, _P, _P, c_int64, _P, c_int32, c_uint64, c_uint64, _P, _P, c_int64, _P, _P,
                                            c_int64, _P, _P, c_size_t, _P]),
    "sgf_neighbor_sample_hop": (c_int32, [_P, _P, _P, c_int64, c_int32, c_int32, c_uint64, c_uint64, c_int32, _P, c_int32,
                                          c_int64, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "sgf_gather_rows": (c_int32, [_P, c_int64, c_int32, c_int64, _P, c_int32, c_int64, c_int32, _P, c_int64,
                                  c_int32, _P]),
    "sgf_pad_rows": (c_int32, [_P, c_int64, c_int32, c_int64, _P, c_int32, c_int64, c_int32, c_int32, _P, c_int64,
                               c_int32, _P]),
    "sgf_attn_stats_len": (c_int64, [c_int32, c_int32]),
    "sgf_attn_bstats_len": (c_int64, [c_int32, c_int32]),
    "sgf_attn_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "sgf_attn_fwd_reduce": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int32, c_int32,
                                      c_int32, c_int32, _P, _P, c_size_t, _P]),
    "sgf_attn_fwd_apply": (c_int32, [_P, c_int64, _P, c_int64, c_int64, c_double, c_int32, c_int32,
                                     c_int32, c_int32, _P, _P, c_int64, _P, _P, _P]),
    "sgf_attn_bwd_reduce": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int32,
                                      c_int32, c_int32, _P, _P, c_size_t, _P]),
    "sgf_attn_bwd_apply": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64,
                                     _P, c_int64, c_double, c_int32, c_int32, c_int32, c_int32, _P, _P,
                                     _P, c_int64, _P, c_int64, _P, c_int64, _P]),
    "sgf_attn_bwd_reduce_heads": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int32,
                                            c_int32, c_int32, _P, _P, c_size_t, _P]),
    "sgf_attn_bwd_apply_heads": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64,
                                           _P, c_int64, c_double, c_int32, c_int32, c_int32, c_int32, _P, _P,
                                           _P, c_int64, _P, c_int64, _P, c_int64, _P]),
    "sgf_attn_h_bstats_len": (c_int64, [c_int32]),
    "sgf_attn_h_fwd": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, c_int64, _P, _P]),
    "sgf_attn_h_bwd_reduce": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int32,
                                        _P, _P, c_size_t, _P]),
    "sgf_attn_h_bwd_apply": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int32,
                                       _P, _P, _P, _P, _P, c_int64, _P, c_size_t, _P]),
    "sgf_attn_h_bwd_apply_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "sgf_attn_h_bwd_split_supported": (c_int32, [c_int32, c_int32]),
    "sgf_attn_h_bwd_pre": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int32, _P, _P, _P, c_size_t, _P, _P]),
    "sgf_attn_h_bwd_reduce_scaled": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int32, _P, _P, c_size_t,
                                               _P]),
    "sgf_attn_h_bwd_post": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, c_size_t, _P, c_int64, _P,
                                      c_int64, _P]),
    "sgf_attn_h_small_saved_bytes": (c_size_t, [c_int32, c_int32]),
    "sgf_attn_h_small_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "sgf_attn_h_small_fwd": (c_int32, [_P, c_int64, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32,
                                       _P, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "sgf_attn_h_small_bwd": (c_int32, [_P, c_int64, _P, _P, _P, c_float, c_int32, c_int32, _P, c_size_t, _P, c_int64, _P,
                                       _P, _P, _P, _P, _P, _P, c_int64, _P, c_size_t, _P]),
    "sgf_gram_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "sgf_gram": (c_int32, [_P, c_int64, c_int32, _P, c_int64, c_int32, c_int64, c_int32, _P, c_int64, _P,
                           _P, c_size_t, _P]),
    "sgf_gram2": (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P, c_int64, c_int32, c_int64, c_int32, _P, c_int64, _P,
                            c_int64, _P, _P, c_size_t, _P]),
    "sgf_ln_fwd": (c_int32, [_P, c_int64, _P, c_int64, c_float, c_float, _P, _P, c_int32, c_float,
                             c_int64, c_int32, c_int32, _P, c_int64, _P, _P, _P]),
    "sgf_ln_bwd_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "sgf_ln_bwd": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_float, c_float, _P,
                             c_int32, _P, _P, c_int64, c_int32, c_int32, _P, c_int64, _P, c_int64, _P,
                             _P, _P, c_size_t, _P]),
    "sgf_colstats_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "sgf_colstats": (c_int32, [_P, c_int64, _P, c_int64, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "sgf_bn_finalize": (c_int32, [_P, _P, c_double, c_float, c_float, _P, _P, c_int32, _P, _P, _P]),
    "sgf_bn_apply": (c_int32, [_P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int32, c_int64, c_int32,
                               c_int32, _P, c_int64, _P]),
    "sgf_bn_bwd_stats": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int32, c_int64, c_int32,
                                   c_int32, _P, _P, c_size_t, _P]),
    "sgf_bn_bwd_stats2": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int32, c_int64, c_int32,
                                    c_int32, _P, _P, c_size_t, _P]),
    "sgf_gram_ln_bwd_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "sgf_gram_ln_bwd": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int32, c_int32, _P, c_int64, c_int32, c_int64,
                                  c_int32, _P, c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "sgf_gram_bn_bwd_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "sgf_gram_bn_bwd": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int32, _P, c_float, c_int32,
                                  c_int32, _P, c_int64, c_int32, c_int64, c_int32, _P, c_int64, _P, _P, c_size_t, _P]),
    "sgf_gram2_bn_bwd_supported": (c_int32, [c_int32, c_int32, c_int64, c_int32]),
    "sgf_gram2_bn_bwd": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int32, _P, c_float, c_int32, c_int32, _P, c_int64,
                                   _P, c_int64, c_int32, c_int64, c_int32, _P, c_int64, _P, c_int64, _P, c_int64, _P, _P,
                                   c_size_t, _P]),
    "sgf_bn_bwd_apply": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int32, _P, c_float,
                                   c_int32, c_int64, c_int32, c_int32, _P, c_int64, _P]),
    "sgf_dropout": (c_int32, [_P, c_int64, _P, c_int64, c_float, c_uint64, c_int64, c_int32, c_int32, _P,
                              c_int64, _P]),
    "sgf_nll_workspace_bytes": (c_size_t, [c_int64]),
    "sgf_nll_fwd": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, _P, _P, c_int64, _P, _P, c_size_t, _P]),
    "sgf_nll_bwd": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, _P, _P, c_int64, _P, c_float, _P,
                              c_int64, _P]),
    "sgf_sum_n": (c_int32, [_P, _P, c_int32, c_int64, c_int32, c_int32, _P, c_int64, _P]),
    "sgf_colsum_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "sgf_colsum": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "sgf_combine_fc_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "sgf_combine_fc_fwd": (c_int32, [_P, c_int64, c_float, _P, c_int64, c_float, _P, _P, c_int64, c_int32, c_int32,
                                     c_int32, _P, c_int64, _P]),
    "sgf_combine_fc_bwd": (c_int32, [_P, c_int64, _P, c_int64, c_int32, c_int32, c_float, c_float, c_int32, _P,
                                     c_int64, _P, c_int64, _P]),
    "sgf_combine_fc_fwd_mapped": (c_int32, [_P, c_int64, c_float, _P, c_int64, c_float, _P, _P, c_int64, c_int32, c_int32,
                                     c_int32, _P, c_int64, _P, _P]),
    "sgf_combine_fc_bwd_g": (c_int32, [_P, c_int64, _P, c_int64, c_int32, c_int32, c_float, c_float, c_int32, _P, c_int64, _P,
                                       c_int64, _P, _P, c_int64, _P]),
    "sgf_combine_fc_bwd_mapped": (c_int32, [_P, c_int64, _P, c_int64, c_int32, c_int32, c_float, c_float, c_int32, _P,
                                     c_int64, _P, c_int64, _P, _P]),
    "sgf_gcn_epilogue_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "sgf_gcn_epilogue_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "sgf_gcn_epilogue_stats": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int32, c_int32, _P, c_int64,
                                         _P, _P, _P, c_size_t, _P]),
    "sgf_gcn_epilogue_dx": (c_int32, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, c_int32, _P, c_int64, _P]),
    "sgf_gcn_epilogue_dx2": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, _P, c_int64,
                                       c_int32, _P]),
    "sgf_gcn_epilogue_dx2_acc_supported": (c_int32, [c_int32, c_int32]),
    "sgf_gcn_epilogue_dx2_acc": (c_int32, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, _P, c_int64, _P,
                                           c_int64, _P, c_int64, _P]),
    "sgf_gcn_epilogue_apply": (c_int32, [_P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int32, c_int64, c_int32,
                               c_int32, _P, c_int64, _P]),
    "sgf_gcn_bn_bwd_dx_supported": (c_int32, [c_int32, c_int32]),
    "sgf_gcn_bn_bwd_dx_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "sgf_gcn_bn_bwd_dx": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, c_int32, _P, c_float, c_int32, _P, c_int64,
                                    c_int64, c_int32, c_int32, _P, c_int64, _P, c_int64, _P, _P, c_size_t, _P, c_int64,
                                    c_int32, _P, c_size_t, _P]),
    "sgf_gcn_epilogue_cat_supported": (c_int32, [c_int32, c_int32]),
    "sgf_gcn_epilogue_cat": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int32, _P, c_int64,
                                       _P, _P, _P, c_size_t, _P]),
    "sgf_gcn_epilogue_partial_bytes": (c_size_t, [c_int64, c_int32]),
    "sgf_gcn_epilogue_dtype_partial_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "sgf_gcn_epilogue_partial": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int32, c_int32, _P,
                                           c_size_t, _P]),
    "sgf_gcn_epilogue_stats_add": (c_int32, [_P, c_int64, _P, c_int64, _P, c_size_t, c_int64, c_int32, c_int32,
                                             c_int32, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "sgf_stem_pair_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "sgf_stem_pair": (c_int32, [_P, c_int64, c_int64, c_int32, _P, c_int64, _P, _P, c_int64, _P, c_int32, c_int32, _P,
                                c_int64, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "sgf_gemm": (c_int32, [_P, c_int64, c_int64, c_int32, _P, c_int64, c_int64, c_int32, c_int64, c_int32, c_int64, c_float,
                           _P, _P, c_float, _P, c_int64, c_int32, _P, c_int64, c_int32, _P]),
    "sgf_comm_available": (c_int32, []),
    "sgf_comm_unique_id_bytes": (c_int32, []),
    "sgf_comm_unique_id": (c_int32, [_P]),
    "sgf_comm_create": (c_int32, [_P, c_int32, c_int32, _P]),
    "sgf_comm_destroy": (c_int32, [_P]),
    "sgf_comm_all_reduce_f32": (c_int32, [_P, _P, c_int64, _P]),
    "sgf_comm_all_gather": (c_int32, [_P, _P, _P, c_int64, _P]),
    "sgf_comm_all_to_all": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "sgf_axpby": (c_int32, [_P, c_int64, c_float, _P, c_int64, c_float, c_int64, c_int32, c_int32, _P,
                            c_int64, _P]),
}


class SgfError(RuntimeError):
    """A libsgf call returned a non-zero status (mirrors PyTorch: RuntimeError)."""


def load():
    """dlopen libsgf.so (after torch, so both bind the same libamdhip64.so.7) and type its symbols."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SgfError(
            f"libsgf.so not found at {LIB_PATH}: build it with `make` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`).  sgformer_amd has no CPU / "
            "eager fallback by design.")
    import torch  # noqa: F401  (loads torch's bundled HIP runtime first; same SONAME)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def available() -> bool:
    return os.path.exists(LIB_PATH)


def check(rc: int, what: str):
    if rc != 0:
        msg = load().sgf_last_error()
        raise SgfError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def call(name: str, *args):
    """Invoke an int-returning entry point and raise SgfError on failure."""
    rc = getattr(load(), name)(*args)
    check(rc, name)
