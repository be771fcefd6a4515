"""ctypes view of libps_amd.so -- exactly the C ABI of include/ps_native.h.

The HIP library is the product; there is no CPU fallback.  Loading fails
loudly when the shared object is missing, and every compute entry point
fails with PS_E_HIP when no MI355X is visible.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PS_AMD_LIB: another build of the SAME library (tools/ab_bench.sh compares two builds on one GPU box)
LIB_PATH = os.environ.get("PS_AMD_LIB") or os.path.join(HERE, "lib", "libps_amd.so")

PS_OK, PS_MISSING, PS_NO_UPDATER = 0, 204, 500
PS_E_BAD_ARG, PS_E_HIP, PS_E_UNSUPPORTED, PS_E_STATE = -1, -2, -3, -4
PS_UPD_ADAM, PS_UPD_FTRL, PS_UPD_SIMPLE = 0, 1, 2
PS_ROUTE_ID_MOD, PS_ROUTE_JAVA_STRING = 0, 1
PS_MODEL_DNN, PS_MODEL_WIDEDEEP = 0, 1
PS_GRAD_COMPAT, PS_GRAD_INTENDED = 0, 1
PS_ACT_NONE, PS_ACT_RELU, PS_ACT_SIGMOID = 0, 1, 2
PS_SUM_AUTO, PS_SUM_SEQUENTIAL, PS_SUM_CHUNKED = 0, 1, 2


class PsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ps_amd error %d: %s" % (code, msg))
        self.code = code


class ps_updater_t(C.Structure):
    _fields_ = [("kind", C.c_int), ("alfa", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float), ("beta", C.c_float), ("l1", C.c_float), ("l2", C.c_float),
                ("eta", C.c_float)]


class ps_model_config_t(C.Structure):
    _fields_ = [("kind", C.c_int), ("F", C.c_int), ("D", C.c_int), ("X", C.c_int), ("nfc", C.c_int),
                ("fc_dims", C.c_int * 8), ("wide_size", C.c_int64), ("max_batch", C.c_int),
                ("max_nnz", C.c_int64), ("emb_grad_mode", C.c_int), ("wide_grad_mode", C.c_int),
                ("use_graph", C.c_int), ("emb_sum_order", C.c_int)]


class ps_batch_t(C.Structure):
    _fields_ = [("B", C.c_int), ("ids", C.c_void_p), ("offsets", C.c_void_p), ("dense", C.c_void_p),
                ("labels", C.c_void_p), ("wide_ids", C.c_void_p), ("on_device", C.c_int), ("nnz", C.c_int64)]


_vp, _i, _i64, _f, _cp = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_char_p
_pi, _pi64, _pf, _pd = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_double)
_pvp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes): one entry per function declared in include/ps_native.h
class ps_ingest_config_t(C.Structure):
    _fields_ = [("F", C.c_int), ("X", C.c_int), ("batch", C.c_int), ("threads", C.c_int), ("offset", C.c_int),
                ("step", C.c_int), ("ids_via_float", C.c_int), ("wide_size", C.c_int64)]


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_TO_ALL_V_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.c_size_t, C.c_void_p)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class ps_comm_ops_t(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("nranks", C.c_int), ("rank", C.c_int), ("all_gather", ALL_GATHER_FN),
                ("all_to_all_v", ALL_TO_ALL_V_FN), ("all_reduce_sum_f32", ALL_REDUCE_FN), ("flags", C.c_int)]


PS_COMM_OWN_IN_PLACE = 1


SIGNATURES = {
    "ps_last_error": (_cp, []),
    "ps_version": (_cp, []),
    "ps_device_count": (_i, [_pi]),
    "ps_updater_default_adam": (None, [C.POINTER(ps_updater_t)]),
    "ps_updater_default_ftrl": (None, [C.POINTER(ps_updater_t)]),
    "ps_updater_name": (_i, [C.POINTER(ps_updater_t), _cp, _i]),
    "ps_updater_from_name": (_i, [_cp, C.POINTER(ps_updater_t)]),
    "ps_java_string_hash": (C.c_int32, [_cp]),
    "ps_router_shard_key": (_i, [_cp, _i]),
    "ps_router_shard_id": (_i, [_i, _i, _i64, _i]),
    "ps_store_create": (_i, [_i, C.c_uint64, _pvp]),
    "ps_store_destroy": (_i, [_vp]),
    "ps_store_device": (_i, [_vp]),
    "ps_store_create_embedding": (_i, [_vp, _i, _pi64, _i, _i, _i, _i, _i]),
    "ps_store_create_wide": (_i, [_vp, _i64]),
    "ps_store_create_fc": (_i, [_vp, _i, _i, _i]),
    "ps_store_set_updater": (_i, [_vp, _cp, C.POINTER(ps_updater_t)]),
    "ps_store_get": (_i, [_vp, _cp, _pf, _i, _pi]),
    "ps_store_put": (_i, [_vp, _cp, _pf, _i]),
    "ps_store_get_rows": (_i, [_vp, _i, _pi64, _i64, _i, _pf]),
    "ps_store_put_rows": (_i, [_vp, _i, _pi64, _i64, _i, _pf]),
    "ps_store_get_wide": (_i, [_vp, _pi64, _i64, _i, _pf]),
    "ps_store_put_wide": (_i, [_vp, _pi64, _i64, _i, _pf]),
    "ps_store_global_step": (_i64, [_vp]),
    "ps_store_advance_global_step": (_i, [_vp, _i64]),
    "ps_store_key_length": (_i, [_vp, _cp, _pi]),
    "ps_store_push_update": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_pf), _pi, _i]),
    "ps_store_bytes": (_i64, [_vp]),
    "ps_store_sync": (_i, [_vp]),
    "ps_stream_sync": (_i, [_vp, _vp]),
    "ps_model_create": (_i, [_vp, C.POINTER(ps_model_config_t), _pvp]),
    "ps_model_destroy": (_i, [_vp]),
    "ps_model_train": (_i, [_vp, C.POINTER(ps_batch_t), _pf]),
    "ps_model_forward": (_i, [_vp, C.POINTER(ps_batch_t), _pf]),
    "ps_model_backward": (_i, [_vp]),
    "ps_model_update": (_i, [_vp]),
    "ps_model_predict": (_i, [_vp, C.POINTER(ps_batch_t), _pf]),
    "ps_model_sync": (_i, [_vp]),
    "ps_model_last_loss": (_i, [_vp, _pf]),
    "ps_model_get_act": (_i, [_vp, _i, _pf, _i64, _pi, _pi]),
    "ps_model_get_delta": (_i, [_vp, _i, _pf, _i64, _pi, _pi]),
    "ps_model_get_p": (_i, [_vp, _pf, _i]),
    "ps_model_get_emb_grads": (_i, [_vp, _i, _pi64, _pf, _i64, _pi64]),
    "ps_model_get_fc_grad": (_i, [_vp, _i, _i, _pf, _i]),
    "ps_dev_alloc": (_i, [_vp, C.c_size_t, _pvp]),
    "ps_dev_free": (_i, [_vp, _vp]),
    "ps_dev_upload": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "ps_dev_download": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "ps_emb_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i]),
    "ps_fc_forward": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i]),
    "ps_fc_backward": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _i]),
    "ps_fc_pending_grad": (_i, [_vp, _i, _i, _pf, _i, _pi]),
    "ps_dense_update": (_i, [_vp, _i]),
    "ps_emb_backward_update": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i, _vp, _i, _i, _i, _i]),
    "ps_emb_last_grads": (_i, [_vp, _pi64, _pf, _i64, _pi64]),
    "ps_store_set_stream": (_i, [_vp, _vp]),
    "ps_shard_plan": (_i, [_vp, C.POINTER(ps_batch_t), _i, _vp, _pi64, _pvp, _pi64]),
    "ps_shard_plan_launch": (_i, [_vp, C.POINTER(ps_batch_t), _i, _vp]),
    "ps_shard_plan_finish": (_i, [_vp, _pi64, _pvp, _pi64]),
    "ps_shard_serve_pull": (_i, [_vp, _vp, _i64, _vp]),
    "ps_shard_forward_backward": (_i, [_vp, _vp, _pf]),
    "ps_shard_grads": (_i, [_vp, _pvp, _pi64]),
    "ps_shard_apply_push": (_i, [_vp, _vp, _vp, _i64, _pi64, _i, _i]),
    "ps_shard_flat_grad": (_i, [_vp, _pvp, _pi64]),
    "ps_shard_apply_flat": (_i, [_vp, _i]),
    "ps_libsvm_count": (_i, [_cp, C.c_size_t, _i, _i, _pi64]),
    "ps_libsvm_parse": (_i, [_cp, C.c_size_t, C.POINTER(ps_ingest_config_t), _i64, _i64, _pi64, _pf, _pf, _pi64, _pi64]),
    "ps_ingest_create": (_i, [_vp, C.POINTER(ps_ingest_config_t), _pvp]),
    "ps_ingest_destroy": (_i, [_vp]),
    "ps_ingest_open_file": (_i, [_vp, _cp]),
    "ps_ingest_open_memory": (_i, [_vp, _cp, C.c_size_t]),
    "ps_ingest_lines": (_i, [_vp, _pi64]),
    "ps_ingest_next": (_i, [_vp, C.POINTER(ps_batch_t)]),
    "ps_ingest_reset": (_i, [_vp]),
    "ps_ingest_stats": (_i, [_vp, _pd, _pi64, _pi64]),
    "ps_store_save": (_i, [_vp, _cp]),
    "ps_store_load": (_i, [_vp, _cp]),
    "ps_comm_rccl_unique_id": (_i, [C.c_char_p]),
    "ps_comm_rccl_create": (_i, [_vp, _i, _i, C.c_char_p, C.POINTER(ps_comm_ops_t)]),
    "ps_shard_exchange_stats": (_i, [_vp, C.POINTER(C.c_int64), _i]),
    "ps_shard_collective_times": (_i, [_vp, _pd]),
    "ps_comm_rccl_info": (_i, [C.POINTER(ps_comm_ops_t), _pi, _pi, _pi]),
    "ps_comm_rccl_destroy": (_i, [C.POINTER(ps_comm_ops_t)]),
    "ps_comm_rccl_calls": (_i, [C.POINTER(ps_comm_ops_t), _pi64]),
    "ps_comm_selfcheck": (_i, [_vp, C.POINTER(ps_comm_ops_t)]),
    "ps_shard_step": (_i, [_vp, C.POINTER(ps_batch_t), C.POINTER(ps_comm_ops_t), _i, _pf]),
    "ps_shard_step_begin": (_i, [_vp, C.POINTER(ps_batch_t), C.POINTER(ps_comm_ops_t), _i]),
    "ps_shard_step_finish": (_i, [_vp, C.POINTER(ps_comm_ops_t), _i, _pf]),
    "ps_shard_step_finish_begin": (_i, [_vp, C.POINTER(ps_comm_ops_t), _i, C.POINTER(ps_batch_t), _pf]),
    "ps_auc_compute": (_i, [_vp, _vp, _vp, _i64, _i, _pd, _pi64, _pi64]),
    "ps_bench_gather": (_i, [_vp, _i64, _i, _i64, _i, _i, C.c_uint64, _pd, _pd, _pd]),
    "ps_bench_gather_check": (_i, [_vp, _i64, _i, _i64, _i, C.c_uint64, _i64, _pi64, _pi64, _pf]),
    "ps_bench_gemm": (_i, [_vp, _i, _i, _i, _i, _i, _i, _pd]),
    "ps_tune_set": (_i, [_cp, _i]),
    "ps_store_join_mode": (_i, [_vp, _cp, _i]),
    "ps_store_wait_timeouts": (C.c_int64, [_vp]),
    "ps_model_time_steps": (_i, [_vp, C.POINTER(ps_batch_t), _i, _pd]),
    "ps_model_set_keep_grads": (_i, [_vp, _i]),
    "ps_ingest_train": (_i, [_vp, _vp, _i64, _pi64]),
    "ps_shard_mapped_info": (_i, [_vp, _pi64]),
    "ps_model_set_profile": (_i, [_vp, _i]),
    "ps_model_set_profile_filter": (_i, [_vp, _cp]),
    "ps_model_profile_report": (_i, [_vp, _cp, _i]),
}

_lib = None


def lib():
    """Load libps_amd.so (built by `python -m ps_amd.build`).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build the HIP extension with `python -m ps_amd.build` "
                          "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)     # AttributeError here = header and library out of step
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != PS_OK:
        raise PsError(rc, lib().ps_last_error().decode(errors="replace"))
    return rc
