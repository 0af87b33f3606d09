from pygpukit_amd.ops.nn.activation import gelu, relu2, sigmoid, silu, tanh
from pygpukit_amd.ops.nn.attention import (sdpa_causal, sdpa_causal_fixed_cache, sdpa_causal_fixed_cache_ptr,
                                          sdpa_causal_strided)
from pygpukit_amd.ops.nn.fused import geglu, glu_packed, rmsnorm_residual, swiglu
from pygpukit_amd.ops.nn.linear import bias_add_inplace, slice_rows_range_ptr, split_qkv_batch
from pygpukit_amd.ops.nn.norm import layernorm, rmsnorm
from pygpukit_amd.ops.nn.rope import rope_inplace, rope_inplace_f32table

__all__ = ["gelu", "silu", "sigmoid", "tanh", "relu2", "sdpa_causal", "sdpa_causal_fixed_cache",
           "sdpa_causal_fixed_cache_ptr", "sdpa_causal_strided", "rmsnorm_residual", "swiglu", "geglu", "glu_packed",
           "bias_add_inplace", "split_qkv_batch", "slice_rows_range_ptr", "layernorm", "rmsnorm", "rope_inplace",
           "rope_inplace_f32table"]
