"""Loader for ``libdctr_hip.so``.

The product path has no CPU fallback: if the library is missing, fails to load, or a tensor is not
on a GPU, the call raises.  (The CPU restatement of these ops lives in ``oracle/`` and is test
infrastructure only.)
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdctr_hip.so")
ABI_VERSION = 25

c_float_p = ctypes.c_void_p  # device pointers travel as integers


class Field(ctypes.Structure):
    """``dctr_field_t`` (include/dctr.h) -- 64 bytes."""
    _fields_ = [("table", ctypes.c_void_p), ("gacc", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("vocab", ctypes.c_int64), ("dim", ctypes.c_int32), ("col", ctypes.c_int32),
                ("len", ctypes.c_int32), ("pool", ctypes.c_int32), ("len_col", ctypes.c_int32),
                ("out_off", ctypes.c_int32), ("ld", ctypes.c_int32), ("ld_state", ctypes.c_int32)]


class Plan(ctypes.Structure):
    """``dctr_plan_t`` (include/dctr.h) -- host struct, device arrays."""
    _fields_ = [("deep", ctypes.c_void_p), ("wide", ctypes.c_void_p), ("dense_cols", ctypes.c_void_p),
                ("wdense_cols", ctypes.c_void_p), ("wdense_w", ctypes.c_void_p),
                ("n_deep", ctypes.c_int32), ("n_deep_fixed", ctypes.c_int32), ("n_wide", ctypes.c_int32),
                ("n_dense", ctypes.c_int32), ("n_wdense", ctypes.c_int32), ("dense_off", ctypes.c_int32),
                ("emb_dim", ctypes.c_int32), ("n_xcols", ctypes.c_int32), ("n_wide_fixed", ctypes.c_int32),
                ("max_dim", ctypes.c_int32), ("vec", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("step_sync", ctypes.c_void_p), ("out_chunks", ctypes.c_void_p), ("chunk_rows", ctypes.c_int32),
                ("pad_", ctypes.c_int32), ("ext", ctypes.c_void_p)]


class USlot(ctypes.Structure):
    """``dctr_uslot_t`` (include/dctr.h): one X column feeding one general update unit -- 48 bytes."""
    _fields_ = [("col", ctypes.c_int32), ("goff", ctypes.c_int32), ("wide", ctypes.c_int32), ("pool", ctypes.c_int32),
                ("t", ctypes.c_int32), ("len", ctypes.c_int32), ("len_col", ctypes.c_int32), ("den", ctypes.c_int32),
                ("am_deep", ctypes.c_int32), ("am_wide", ctypes.c_int32), ("vu0", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class VUnit(ctypes.Structure):
    """``dctr_vunit_t``: (unit, group of P partitions) -- 40 bytes."""
    _fields_ = [("di", ctypes.c_int32), ("wi", ctypes.c_int32), ("c0", ctypes.c_int32), ("n_slots", ctypes.c_int32),
                ("k", ctypes.c_int32), ("j", ctypes.c_int32), ("kshift", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("kmagic", ctypes.c_uint64)]


class PlanExt(ctypes.Structure):
    """``dctr_plan_ext_t``: general update units of a plan (pooled VarLen fields, shared tables) + per-step side buffers."""
    _fields_ = [("slots", ctypes.c_void_p), ("vunits", ctypes.c_void_p), ("am_deep_off", ctypes.c_void_p),
                ("am_wide_off", ctypes.c_void_p), ("h_vunits", ctypes.c_void_p), ("h_vocab", ctypes.c_void_p),
                ("den_t", ctypes.c_void_p), ("amax", ctypes.c_void_p),
                ("n_vcols", ctypes.c_int32), ("n_vunits", ctypes.c_int32), ("n_units", ctypes.c_int32),
                ("max_unit_slots", ctypes.c_int32), ("n_den", ctypes.c_int32), ("ld_amax", ctypes.c_int32),
                ("gslot_deep", ctypes.c_void_p), ("gslot_wide", ctypes.c_void_p),
                ("n_gslot_deep", ctypes.c_int32), ("n_gslot_wide", ctypes.c_int32)]


MAX_UNIT_SLOTS = 128


MLP_MAX_LAYERS = 12


class MlpLayer(ctypes.Structure):
    """``dctr_mlp_layer_t`` (include/dctr.h)."""
    _fields_ = [("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("h", ctypes.c_void_p), ("dh", ctypes.c_void_p),
                ("gW", ctypes.c_void_p), ("gbias", ctypes.c_void_p), ("K", ctypes.c_int32), ("N", ctypes.c_int32),
                ("ld_w", ctypes.c_int32), ("ld_h", ctypes.c_int32), ("relu", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class Mlp(ctypes.Structure):
    """``dctr_mlp_t`` (include/dctr.h) -- host struct, device pointers."""
    _fields_ = [("layer", MlpLayer * MLP_MAX_LAYERS), ("w_out", ctypes.c_void_p), ("g_w_out", ctypes.c_void_p),
                ("n_layers", ctypes.c_int32), ("pad_", ctypes.c_int32), ("step_sync", ctypes.c_void_p)]


class DenseStep(ctypes.Structure):
    """``dctr_dense_step_t`` (include/dctr.h): an optimizer step applied by the kernel that finishes a gradient."""
    _fields_ = [("kind", ctypes.c_int32), ("lr", ctypes.c_float), ("eps", ctypes.c_float), ("pad_", ctypes.c_int32),
                ("grad_base", ctypes.c_void_p), ("param_base", ctypes.c_void_p), ("state_base", ctypes.c_void_p)]


class DenseItem(ctypes.Structure):
    """``dctr_dense_item_t`` (include/dctr.h): one tensor of a ``dctr_dense_opt_multi`` list."""
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("state", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("l2", ctypes.c_float), ("pad_", ctypes.c_float)]


PLAN_HAS_GACC, PLAN_HAS_STATE, PLAN_HAS_MAXPOOL = 1, 2, 4
SYNC_TOWER, SYNC_GATHER, SYNC_UPDATE, SYNC_ERR, SYNC_INTS = 0, 1, 2, 12, 32
LAZY_SGD, LAZY_ADAGRAD, LAZY_ADAM, LAZY_RMSPROP = 0, 1, 2, 3


class LazyUnit(ctypes.Structure):
    """dctr_lazy_unit_t (include/dctr.h)"""
    _fields_ = [("deep", ctypes.c_void_p), ("deep_s1", ctypes.c_void_p), ("deep_s2", ctypes.c_void_p),
                ("deep_g", ctypes.c_void_p), ("wide", ctypes.c_void_p), ("wide_s1", ctypes.c_void_p),
                ("wide_s2", ctypes.c_void_p), ("wide_g", ctypes.c_void_p), ("stamp", ctypes.c_void_p),
                ("vocab", ctypes.c_int64), ("dim", ctypes.c_int32), ("col", ctypes.c_int32),
                ("l2_deep", ctypes.c_float), ("l2_wide", ctypes.c_float),
                ("ld_deep", ctypes.c_int32), ("ld_deep_s1", ctypes.c_int32),
                ("ld_wide", ctypes.c_int32), ("ld_wide_s1", ctypes.c_int32)]


class LazyOpt(ctypes.Structure):
    """dctr_lazy_opt_t"""
    _fields_ = [("kind", ctypes.c_int32), ("lr", ctypes.c_float), ("eps", ctypes.c_float),
                ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
                ("n_ss", ctypes.c_int32), ("n_bc", ctypes.c_int32), ("any_l2", ctypes.c_int32),
                ("adam_ss", ctypes.c_void_p), ("adam_bc", ctypes.c_void_p), ("adam_rbc", ctypes.c_void_p)]
POOL_CODE = {None: 0, "sum": 1, "mean": 2, "max": 3}
BWD_ACCUM, BWD_SGD = 0, 1
OPT_SGD, OPT_ADAGRAD = 0, 1
UPD_SGD, UPD_ADAGRAD, UPD_ACCUM, UPD_LAZY = 0, 1, 2, 3
EINVAL, ENOSUP = -1, -2          # (include/dctr.h DCTR_EINVAL / DCTR_ENOSUP)

_I32, _I64, _F32, _P = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes): every symbol include/dctr.h declares.  tests/test_abi.py parses the
# header and checks this table and the built library against it.
SIGNATURES = {
    "dctr_abi_version": (ctypes.c_int, []),
    "dctr_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "dctr_sizeof_field": (ctypes.c_size_t, []),
    "dctr_sizeof_plan": (ctypes.c_size_t, []),
    "dctr_sizeof_uslot": (ctypes.c_size_t, []),
    "dctr_sizeof_vunit": (ctypes.c_size_t, []),
    "dctr_sizeof_plan_ext": (ctypes.c_size_t, []),
    "dctr_embed_fwd": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _I64, _I32, _P, _I64, _P, _I64, _P, _P, _P, _I32, _P,
                                      _P, _P, _I64, _P]),
    "dctr_embed_update_supported": (ctypes.c_int, [ctypes.POINTER(Plan), _I64, _I32]),
    "dctr_embed_update_partitions": (ctypes.c_int32, [ctypes.POINTER(Plan), _I32]),
    "dctr_embed_ids": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _I32, _P, _I64, _I32, _P, _P, _P]),
    "dctr_embed_update": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _I32, _I64, _P, _P, _I32, _P, _I64, _P, _I64, _P, _I64,
                                         _P, _P, _I64, _I32, _F32, _F32, _P, _I64, _P, _P, _P, _I64, _I32, _P]),
    "dctr_embed_update_lazy": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _I32, _I64, _P, _P, _I32, _P, _I64, _P, _I64, _P, _I64,
                                              _P, _P, _I64, _P, _I64, _P, _P, _I64, _P, _P, _P, _P]),
    "dctr_embed_segments": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _I32, _I64, _P, _P, _I32, _P, _I64, _P]),
    "dctr_embed_update_workspace_ints": (ctypes.c_int64, [ctypes.POINTER(Plan), _I32, _I32]),
    "dctr_embed_bwd": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _I64, _I32, _P, _I64, _P, _I64, _P, _P,
                                      _I32, _F32, _P]),
    "dctr_embed_apply": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _I64, _I32, _I32, _F32, _F32, _P]),
    "dctr_sizeof_lazy_unit": (ctypes.c_size_t, []),
    "dctr_sizeof_lazy_opt": (ctypes.c_size_t, []),
    "dctr_lazy_sweep": (ctypes.c_int, [_P, _I32, _I64, _I32, _P, _P, _I32, _I32, _P]),
    "dctr_lazy_catchup": (ctypes.c_int, [_P, _I32, _P, _I32, _P, _P, _I32, _I32, _P, _P]),
    "dctr_lazy_apply": (ctypes.c_int, [_P, _I32, _P, _I32, _P, _P, _I32, _I32, _P]),
    "dctr_lazy_flush": (ctypes.c_int, [_P, _I32, _I64, _P, _P, _I32, _I32, _P]),
    "dctr_lazy_step_inc": (ctypes.c_int, [_P, _P]),
    "dctr_dense_opt_reg": (ctypes.c_int, [_P, _P, _P, _P, _P, _I64, _P, _P, _P]),
    "dctr_cin_workspace_floats": (ctypes.c_size_t, [_I32, _I32, _I32]),
    "dctr_cin_layer_fwd": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I64, _P,
                                          _P]),
    "dctr_cin_bwd_workspace_floats": (ctypes.c_size_t, [_I32, _I32, _I32, _I32, _I32]),
    "dctr_cin_layer_bwd": (ctypes.c_int, [_P, _P, _I64, _I32, _P, _I64, _P, _I64, _P, _I32, _I32, _I32, _I32, _I32, _P, _I64,
                                          _P, _I64, _I32, _P, _P, _P, _P]),
    "dctr_cin_pool_fwd": (ctypes.c_int, [_P, _I32, _I32, _I32, _I32, _P, _I64, _P]),
    "dctr_cin_pool_bwd": (ctypes.c_int, [_P, _P, _I64, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "dctr_rows_dot": (ctypes.c_int, [_P, _I64, _P, _I32, _I32, _P, _P]),
    "dctr_rows_tdot": (ctypes.c_int, [_P, _I64, _P, _I32, _I32, _P, _P, _P]),
    "dctr_senet_fwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P, _I32, _P, _P, _P, _P]),
    "dctr_senet_bwd_workspace_floats": (ctypes.c_size_t, [_I32, _I32, _I32]),
    "dctr_senet_bwd": (ctypes.c_int, [_P, _P, _I64, _I32, _I32, _I32, _P, _P, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "dctr_bilinear_fwd": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _I64, _P, _I64,
                                         _I32, _I32, _P]),
    "dctr_bilinear_bwd_workspace_floats": (ctypes.c_size_t, [_I32, _I32, _I32]),
    "dctr_bilinear_bwd": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I32, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P,
                                         _I64, _P, _P, _P, _P, _P, _I32, _P]),
    "dctr_bilinear_wide_bwd_workspace_floats": (ctypes.c_size_t, [_I32, _I32]),
    "dctr_bilinear_wide_fwd_workspace_floats": (ctypes.c_size_t, [_I32, _I32]),
    "dctr_bilinear_wide_fwd": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I32, _I32, _I32, _I32, _P, _I64, _I32, _P, _I64,
                                              _I32, _P, _I32, _P, _I64, _P, _I64, _P, _P, _P]),
    "dctr_bilinear_wide_bwd": (ctypes.c_int, [_P, _I64, _P, _I64, _P, _P, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P, _I64,
                                              _P, _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "dctr_bilinear_wide_pack_floats": (ctypes.c_size_t, [_I32]),
    "dctr_inner_product_fwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _P, _I64, _P]),
    "dctr_inner_product_bwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _P, _I64, _P, _I64, _P]),
    "dctr_crossnet_vec_fwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P, _P, _I64, _P]),
    "dctr_crossnet_vec_bwd_workspace_floats": (ctypes.c_size_t, [_I32, _I32, _I32]),
    "dctr_crossnet_vec_bwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P]),
    "dctr_crossnet_mat_supported": (ctypes.c_int, [_I32, _I32]),
    "dctr_crossnet_mat_fwd": (ctypes.c_int, [ctypes.POINTER(Mlp), _P, _I64, _I32, _P]),
    "dctr_crossnet_mat_bwd_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(Mlp), _I32]),
    "dctr_crossnet_mat_bwd": (ctypes.c_int, [ctypes.POINTER(Mlp), _P, _I64, _I32, _P, _I64, _P, _I64, _P, _P]),
    "dctr_crossnet_mix_supported": (ctypes.c_int, [_I32, _I32, _I32, _I32]),
    "dctr_crossnet_mix_fwd": (ctypes.c_int, [ctypes.POINTER(Mlp), _I32, _I32, _P, _I64, _I32, _P]),
    "dctr_crossnet_mix_bwd_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(Mlp), _I32]),
    "dctr_crossnet_mix_bwd": (ctypes.c_int, [ctypes.POINTER(Mlp), _I32, _I32, _P, _I64, _I32, _P, _I64, _P, _I64, _P, _P]),
    "dctr_sizeof_mlp": (ctypes.c_size_t, []),
    "dctr_mlp_fwd": (ctypes.c_int, [ctypes.POINTER(Mlp), _P, _I64, _I32, _P, _P]),
    "dctr_mlp_bwd_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(Mlp), _I32]),
    "dctr_mlp_bwd": (ctypes.c_int, [ctypes.POINTER(Mlp), _P, _I64, _I32, _P, _I64, _P, _I64, _P, _P]),
    "dctr_mlp_train_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(Mlp), _I32]),
    "dctr_mlp_train_step": (ctypes.c_int, [ctypes.POINTER(Mlp), _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64,
                                           _P, _I32, _P, _P]),
    "dctr_mlp_train_wgrad": (ctypes.c_int, [ctypes.POINTER(Mlp), _P, _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "dctr_embed_tower_train_supported": (ctypes.c_int, [ctypes.POINTER(Plan), ctypes.POINTER(Mlp), _I32]),
    "dctr_embed_tower_train_step": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _I64, ctypes.POINTER(Mlp), _I32, _I32, _P, _P,
                                                   _P, _P, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P]),
    "dctr_mlp_train_wgrad_counters": (ctypes.c_size_t, [ctypes.POINTER(Mlp), _I32]),
    "dctr_mlp_train_wgrad_sync": (ctypes.c_int, [ctypes.POINTER(Mlp), _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dctr_embed_tower_train_step_sync": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _I64, ctypes.POINTER(Mlp), _I32, _I32, _P,
                                                        _P, _P, _P, _P, _I64, _P, _I64, _P, _I64, _P, _P, _P, _I32, _P]),
    "dctr_sizeof_dense_step": (ctypes.c_size_t, []),
    "dctr_step_wait": (ctypes.c_int, [_P, _I32, _I32, _P]),
    "dctr_step_signal": (ctypes.c_int, [_P, _I32, _P]),
    "dctr_stamp": (ctypes.c_int, [_P, _P]),
    "dctr_copy_async": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P]),
    "dctr_enable_peer_access": (ctypes.c_int, [_I32]),
    "dctr_exchange_post": (ctypes.c_int, [_P, _I32, _I32, _P, _P]),
    "dctr_exchange_wait": (ctypes.c_int, [_P, _I32, _P, _I32, _P, _P]),
    "dctr_exchange_next": (ctypes.c_int, [_P, _P]),
    "dctr_sum_ranks": (ctypes.c_int, [_P, _P, _P, _I32, _I64, _I64, _P, _I32, _P]),
    "dctr_exchange_sync": (ctypes.c_int, [_P, _P, _I32, _I32, _P, _I32, _I32, _P, _P]),
    "dctr_shard_stage": (ctypes.c_int, [_P, _I64, _P, _I32, _I32, _P, _I64, _P, _P, _I64, _P, _I32, _I32, _P, _I64, _I32,
                                        _P]),
    "dctr_bce_head": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _P, _P]),
    "dctr_dense_opt": (ctypes.c_int, [_P, _P, _P, _I64, _I32, _F32, _F32, _P]),
    "dctr_rows_join": (ctypes.c_int, [_P, _I64, _P, _I64, _I32, _P, _I64, _I32, _P, _I64, _I32, _P]),
    "dctr_relu_bwd_bias_workspace_floats": (ctypes.c_size_t, [_I32, _I32]),
    "dctr_relu_bwd_bias": (ctypes.c_int, [_P, _I64, _P, _I64, _I32, _I32, _P, _I64, _P, _P, _P]),
    "dctr_sizeof_dense_item": (ctypes.c_size_t, []),
    "dctr_dense_opt_multi": (ctypes.c_int, [ctypes.POINTER(DenseItem), _I32, _I32, _F32, _F32, _P]),
    "dctr_l2_value_multi": (ctypes.c_int, [ctypes.POINTER(DenseItem), _I32, _P, _P]),
    "dctr_shard_assemble_fwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _I64, _P, _I32, _I32, _P, _P,
                                               _I32, _P, _I64, _P, _P, _P, _I64, _P]),
    "dctr_shard_assemble_bwd": (ctypes.c_int, [_P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _I64, _P, _P, _P, _I64, _P,
                                               _I64, _P, _I64, _P, _I32, _P, _P]),
    "dctr_shard_assemble_bwd_next": (ctypes.c_int, [_P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _I64, _P, _P, _P, _I64,
                                                    _P, _I64, _P, _I64, _P, _I32, _P, _P, _I64, _P, _P]),
    "dctr_fm_fwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P]),
    "dctr_fm_bwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _P, _P, _I64, _I32, _P]),
    "dctr_interacting_supported": (ctypes.c_int, [_I32, _I32, _I32]),
    "dctr_interacting_bwd_workspace_floats": (ctypes.c_size_t, [_I32, _I32]),
    "dctr_interacting_fwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I64, _P]),
    "dctr_interacting_bwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I64, _P, _I64,
                                            _P, _P, _P, _P, _P, _P]),
    "dctr_afm_bwd_workspace_floats": (ctypes.c_size_t, [_I32, _I32, _I32]),
    "dctr_afm_fwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "dctr_afm_bwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _P, _P,
                                    _P]),
    "dctr_bi_pooling_fwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _I64, _P]),
    "dctr_bi_pooling_bwd": (ctypes.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _I64, _P, _I64, _P]),
}

# entry points that exist only in the diagnostics build (make -C csrc diag -> libdctr_hip_diag.so; include/dctr.h
# declares them under DCTR_DIAG): tools/upd_trace.py and tools/mlp_trace.py load that library with use_diag_library()
DIAG_SIGNATURES = {
    "dctr_dbg_update_trace": (None, [_P, _I32]),
    "dctr_dbg_mlp_trace": (None, [_P]),
}
DIAG_LIB_PATH = os.path.join(_HERE, "libdctr_hip_diag.so")

_lib = None
_lock = threading.Lock()


def use_diag_library():
    """Profiling tools only: make lib() load the diagnostics build (same ABI + the dctr_dbg_* hooks)."""
    global LIB_PATH, _lib
    if not os.path.exists(DIAG_LIB_PATH):
        raise RuntimeError("%s is not built: make -C deepctr-torch_amd/csrc diag" % DIAG_LIB_PATH)
    LIB_PATH, _lib = DIAG_LIB_PATH, None
    SIGNATURES.update(DIAG_SIGNATURES)


def available():
    return os.path.exists(LIB_PATH)


def lib():
    """The loaded library; raises (never falls back) when it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libdctr_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C deepctr-torch_amd/csrc`; there is no CPU fallback for the HIP hot path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here == header and library disagree
            fn.restype, fn.argtypes = res, args
        if handle.dctr_abi_version() != ABI_VERSION:
            raise RuntimeError("libdctr_hip.so ABI %d != binding ABI %d" % (handle.dctr_abi_version(), ABI_VERSION))
        if handle.dctr_sizeof_field() != ctypes.sizeof(Field) or handle.dctr_sizeof_plan() != ctypes.sizeof(Plan) \
                or handle.dctr_sizeof_mlp() != ctypes.sizeof(Mlp) \
                or handle.dctr_sizeof_uslot() != ctypes.sizeof(USlot) \
                or handle.dctr_sizeof_vunit() != ctypes.sizeof(VUnit) \
                or handle.dctr_sizeof_plan_ext() != ctypes.sizeof(PlanExt) \
                or handle.dctr_sizeof_lazy_unit() != ctypes.sizeof(LazyUnit) \
                or handle.dctr_sizeof_lazy_opt() != ctypes.sizeof(LazyOpt) \
                or handle.dctr_sizeof_dense_step() != ctypes.sizeof(DenseStep) \
                or handle.dctr_sizeof_dense_item() != ctypes.sizeof(DenseItem):
            raise RuntimeError("dctr_field_t / dctr_plan_t / dctr_mlp_t layout mismatch between header and binding")
        _lib = handle
    return _lib


def check(rc, what="dctr call"):
    if rc != 0:
        msg = lib().dctr_strerror(int(rc))
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", rc))


def stream_handle(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            "%s: tensor lives on %s -- the DeepCTR hot path runs only on an AMD GPU through libdctr_hip.so; "
            "there is no CPU fallback (build the model with device='cuda:0')." % (what, t.device))
