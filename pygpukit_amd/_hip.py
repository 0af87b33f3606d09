"""ctypes binding of libpgk_hip.so (the C ABI declared in include/pgk_hip.h).

This replaces the reference's pybind11 module loader (src/pygpukit/_native_loader.py:1-193,
src/pygpukit/core/backend.py:197-202).  There is exactly one backend: if the shared library
cannot be loaded, or a call fails, a RuntimeError is raised - nothing falls back to the CPU.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGK_LIB") or os.path.join(_HERE, "libpgk_hip.so")     # PGK_LIB: a diagnostic build of the same library

c_void_pp = C.POINTER(C.c_void_p)
c_i32_p = C.POINTER(C.c_int32)


class PgkError(RuntimeError):
    """A libpgk_hip call returned a non-zero status (reference: std::runtime_error / CudaError
    surfacing as RuntimeError, native/core/types.hpp:108-111)."""


class DeviceProps(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 32), ("total_mem", C.c_size_t),
                ("cu_count", C.c_int), ("wavefront_size", C.c_int), ("clock_khz", C.c_int),
                ("lds_per_cu", C.c_int), ("l2_bytes", C.c_int)]


class PoolStats(C.Structure):
    _fields_ = [("bytes_in_use", C.c_size_t), ("bytes_cached", C.c_size_t), ("bytes_reserved_peak", C.c_size_t),
                ("n_alloc", C.c_uint64), ("n_pool_hit", C.c_uint64), ("n_device_malloc", C.c_uint64),
                ("n_free", C.c_uint64)]


class ModelConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("hidden_size", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int),
                ("num_kv_heads", C.c_int), ("head_dim", C.c_int), ("intermediate_size", C.c_int),
                ("max_seq_len", C.c_int), ("max_batch", C.c_int), ("norm_eps", C.c_float), ("rope_theta", C.c_float),
                ("weight_format", C.c_int), ("use_qk_norm", C.c_int)]


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("attn_norm", "w_qkv", "s_qkv", "q_norm", "k_norm", "w_o", "s_o",
                                          "mlp_norm", "w_gate_up", "s_gate_up", "w_down", "s_down")]


# name -> (argtypes); every function returns pgk_status (int) unless listed in _NON_STATUS
_V, _I, _F, _Z, _D, _I64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_double, C.c_int64
_PROTOS = {
    "pgk_device_count": [C.POINTER(_I)], "pgk_device_set": [_I], "pgk_device_get": [C.POINTER(_I)],
    "pgk_device_sync": [], "pgk_device_props": [_I, C.POINTER(DeviceProps)],
    "pgk_mem_info": [C.POINTER(_Z), C.POINTER(_Z)],
    "pgk_malloc": [c_void_pp, _Z], "pgk_free": [_V], "pgk_pool_stats": [C.POINTER(PoolStats)], "pgk_pool_trim": [],
    "pgk_host_alloc": [c_void_pp, _Z], "pgk_host_free": [_V],
    "pgk_memcpy_h2d": [_V, _V, _Z, _V], "pgk_memcpy_d2h": [_V, _V, _Z, _V],
    "pgk_memcpy_h2d_async": [_V, _V, _Z, _V], "pgk_memcpy_d2h_async": [_V, _V, _Z, _V],
    "pgk_memcpy_d2d": [_V, _V, _Z, _V], "pgk_memset": [_V, _I, _Z, _V], "pgk_fill": [_V, _D, _Z, _I, _V],
    "pgk_stream_create": [c_void_pp, _I], "pgk_stream_destroy": [_V], "pgk_stream_sync": [_V],
    "pgk_stream_set_current": [_V], "pgk_stream_get_current": [c_void_pp],
    "pgk_event_create": [c_void_pp], "pgk_event_destroy": [_V], "pgk_event_record": [_V, _V],
    "pgk_event_sync": [_V], "pgk_stream_wait_event": [_V, _V], "pgk_event_query": [_V, C.POINTER(_I)], "pgk_event_elapsed_ms": [_V, _V, C.POINTER(_F)],
    "pgk_graph_begin_capture": [_V], "pgk_graph_end_capture": [_V, c_void_pp], "pgk_graph_launch": [_V, _V],
    "pgk_graph_num_nodes": [_V, C.POINTER(_Z)], "pgk_graph_destroy": [_V], "pgk_stream_is_capturing": [_V, C.POINTER(_I)],
    "pgk_binary": [_V, _V, _V, _Z, _I, _I, _V], "pgk_binary_inplace": [_V, _V, _Z, _I, _I, _V],
    "pgk_bias_add_inplace": [_V, _V, _I, _I, _I, _V], "pgk_activation": [_V, _V, _Z, _I, _I, _V],
    "pgk_reduce": [_V, _V, _Z, _I, _I, _V], "pgk_softmax_rows": [_V, _V, _I, _I, _I, _V], "pgk_sum_axis": [_V, _V, _I, _I, _I, _I, _V],
    "pgk_clamp": [_V, _V, _Z, _F, _F, _I, _V], "pgk_where": [_V, _V, _V, _V, _Z, _I, _V], "pgk_widen_i32_i64": [_V, _V, _Z, _V],
    "pgk_glu": [_V, _V, _V, _Z, _I, _I, _V], "pgk_glu_packed": [_V, _V, _I, _I, _I, _I, _V], "pgk_cast": [_V, _I, _V, _I, _Z, _V],
    "pgk_rmsnorm": [_V, _V, _V, _I, _I, _F, _I, _V], "pgk_rmsnorm_residual": [_V, _V, _V, _V, _I, _I, _F, _I, _V],
    "pgk_layernorm": [_V, _V, _V, _V, _I, _I, _F, _I, _V],
    "pgk_rope_inplace": [_V, _V, _V, _V, _I, _I, _I, _I, _I, _I, _V],
    "pgk_transpose_2d": [_V, _V, _I, _I, _I, _V], "pgk_transpose_3d_021": [_V, _V, _I, _I, _I, _I, _V], "pgk_transpose_batched": [_V, _V, _I, _I, _I, _I, _V], "pgk_transpose_4d_0213": [_V, _V, _I, _I, _I, _I, _I, _V],
    "pgk_repeat_interleave_axis1": [_V, _V, _I, _I, _I, _I, _I, _V],
    "pgk_split_qkv_batch": [_V, _V, _V, _V, _I, _I, _I, _I, _I, _V],
    "pgk_embedding_lookup": [_V, _V, _I, _I, _I, _V, _I, _V],
    "pgk_slice_rows_range_ptr": [_V, _V, _V, _I, _I, _I, _V],
    "pgk_kv_cache_write": [_V, _V, _I, _I, _I, _I, _I, _I, _I, _V, _V],
    "pgk_argmax": [_V, _I, _I, _I, _V, _V],
    "pgk_gemm_nn": [_V, _V, _V, _I, _I, _I, _I, _V], "pgk_gemm_nt": [_V, _V, _V, _V, _I, _I, _I, _I, _V],
    "pgk_gemv": [_V, _V, _V, _I, _I, _I, _V], "pgk_gemv_fp8_bf16": [_V, _V, _V, _V, _I, _I, _I, _V],
    "pgk_w8a16_gemm_kn": [_V, _V, _V, _V, _I, _I, _I, _V], "pgk_w8a16_gemm_nk": [_V, _V, _V, _V, _I, _I, _I, _V],
    "pgk_gemm_fp8_nt": [_V, _V, _V, _V, _V, _I, _I, _I, _V], "pgk_quantize_fp8_rows": [_V, _V, _V, _I, _I, _I, _V],
    "pgk_quantize_fp8_blocks": [_V, _V, _V, _I, _I, _V],
    "pgk_paged_attention_v1": [_V, _V, _V, _V, _V, _V, _I, _I, _I, _I, _I, _I, _I, _F, _V, _I, _V],
    "pgk_paged_cache_write": [_V, _V, _V, _V, _V, _I, _I, _I, _I, _I, _V],
    "pgk_scatter_last_token_logits": [_V, _V, _V, _V, _I, _I, _I, _V], "pgk_prepare_position_ids": [_V, _V, _V, _V, _V, _I, _V],
    "pgk_check_eos": [_V, _V, _I, _I, _V], "pgk_exclusive_cumsum_i32": [_V, _V, _I, _V],
    "pgk_st_open": [C.c_char_p, c_void_pp],
    "pgk_st_tensor_info": [_V, C.c_char_p, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "pgk_st_tensor_data": [_V, C.c_char_p, c_void_pp, C.POINTER(C.c_uint64)], "pgk_st_upload": [_V, C.c_char_p, _V, C.c_uint64, _V],
    "pgk_jit_version": [C.POINTER(_I), C.POINTER(_I)],
    "pgk_jit_compile": [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), _I, c_void_pp, C.POINTER(_I)],
    "pgk_jit_program_code": [_V, c_void_pp, C.POINTER(_Z)], "pgk_jit_kernel_create": [_V, C.c_char_p, c_void_pp, C.POINTER(_I)],
    "pgk_jit_suggested_block_size": [_V, _Z, C.POINTER(_I)],
    "pgk_jit_launch": [_V, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, c_void_pp, _V],
    "pgk_sample_token": [_V, _I, _I, _I, _F, _I, _F, _F, _V, _V, _V],
    "pgk_sdpa_causal": [_V, _V, _V, _V, _I, _I, _I, _I, _I, _F, _I64, _I64, _I64, _I64, _I64, _I64, _I, _V],
    "pgk_sdpa_fixed_cache": [_V, _V, _V, _V, _I, _I, _I, _I, _I, _F, _I, _V, _V, _I, _V],
    "pgk_engine_create": [C.POINTER(ModelConfig), _V, _V, _V, C.POINTER(LayerWeights), c_void_pp],
    "pgk_engine_destroy": [_V], "pgk_engine_bytes": [_V, C.POINTER(_Z), C.POINTER(_Z)],
    "pgk_engine_prefill": [_V, _I, c_i32_p, _I, _I, _V, C.POINTER(_F), _V],
    "pgk_engine_set_state": [_V, c_i32_p, c_i32_p, _I, _V], "pgk_engine_decode_step": [_V, _I, _V],
    "pgk_engine_profile_step": [_V, _I, _I, C.POINTER(_F), C.POINTER(_I), _V],
    "pgk_engine_timeline": [_V, _I, _I, C.POINTER(C.c_uint64), _I, C.POINTER(_I), _V],
    "pgk_engine_capture": [_V, _I, _V], "pgk_engine_replay": [_V, _I, _V], "pgk_engine_logits_ptr": [_V, c_void_pp],
    "pgk_engine_read_tokens": [_V, c_i32_p, _I, _I, _V], "pgk_engine_reset_log": [_V, _V],
    "pgk_engine_set_sampling": [_V, _F, _I, _F, C.POINTER(_F), _I, _V],
    "pgk_engine_read_clock": [_V, C.POINTER(C.c_uint64), _I, _V],
    "pgk_engine_kv_ptr": [_V, _I, c_void_pp, c_void_pp], "pgk_engine_state_ptr": [_V, c_void_pp, c_void_pp], "pgk_engine_launches_per_step": [_V, C.POINTER(_I)],
    "pgk_comm_unique_id": [C.c_char_p], "pgk_comm_init": [c_void_pp, C.c_char_p, _I, _I], "pgk_comm_destroy": [_V],
    "pgk_comm_broadcast": [_V, _V, _Z, _I, _V], "pgk_comm_all_gather": [_V, _V, _V, _Z, _V],
    "pgk_comm_all_reduce_max_f64": [_V, _V, _I, _V], "pgk_comm_barrier": [_V, _V],
}
_NON_STATUS = {"pgk_last_error": ([], C.c_char_p), "pgk_version": ([], C.c_char_p),
               "pgk_sdpa_decode_workspace_bytes": ([_I, _I, _I], C.c_size_t),
               "pgk_paged_attention_workspace_bytes": ([_I, _I, _I, _I], C.c_size_t),
               "pgk_st_close": ([_V], None), "pgk_st_num_tensors": ([_V], C.c_int), "pgk_st_file_size": ([_V], C.c_uint64),
               "pgk_st_tensor_name": ([_V, _I], C.c_char_p),
               "pgk_jit_available": ([], C.c_int), "pgk_jit_library_path": ([], C.c_char_p),
               "pgk_jit_program_log": ([_V], C.c_char_p), "pgk_jit_program_destroy": ([_V], None),
               "pgk_jit_kernel_destroy": ([_V], None)}

EXPORTED_SYMBOLS = sorted(list(_PROTOS) + list(_NON_STATUS))

_lib = None


def load():
    """Load libpgk_hip.so once; raise loudly when it is missing (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libpgk_hip.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pygpukit_amd/csrc`). pygpukit_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, (argtypes, restype) in _NON_STATUS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def call(name: str, *args):
    """Invoke a status-returning entry point; raise PgkError with the library's message on failure."""
    lib = load()
    st = getattr(lib, name)(*args)
    if st != 0:
        msg = lib.pgk_last_error()
        raise PgkError(f"{name} failed (status {st}): {msg.decode(errors='replace') if msg else ''}")


import ctypes as C  # noqa: E402,F811  (re-export for callers that build ctypes arguments)
_device_checked = False


def device_count() -> int:
    lib = load()
    n = C.c_int(0)
    lib.pgk_device_count(C.byref(n))
    return n.value


def require_device() -> None:
    """Every array/op entry point goes through here: no GPU -> RuntimeError."""
    global _device_checked
    if _device_checked:
        return
    if device_count() < 1:
        raise RuntimeError("pygpukit_amd: no HIP device visible (this backend has no CPU fallback)")
    _device_checked = True
