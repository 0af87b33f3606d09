"""pygpukit_amd.ops: operator surface mirroring pygpukit.ops on the LLM-inference hot path."""

from pygpukit_amd.ops.elementwise import add, add_inplace, clamp, copy_to, div, mul, mul_inplace, sub, where
from pygpukit_amd.ops.embedding import (embedding_lookup, embedding_lookup_batch, embedding_lookup_ptr,
                                       kv_cache_prefill, kv_cache_prefill_gqa, kv_cache_update, kv_cache_update_gqa, kv_cache_update_gqa_ptr)
from pygpukit_amd.ops.matmul import (batched_matmul, gemm_w8a16_init_lut, gemv_bf16, gemv_bf16_opt_available, gemv_fp8_bf16,
                                    gemv_fp8_bf16_batched, linear_bias_gelu, matmul, matmul_nt, transpose, w8a16_gemm,
                                    w8a16_gemm_nk, w8a16_gemm_sm120, matmul_fp8, matmul_fp8_sm120, gemm_fp8_fp8_blockwise_nt,
                                    quantize_fp8_rows, quantize_fp8_blocks, fp8_available, fp8_init_lut)
from pygpukit_amd.ops.nn import (bias_add_inplace, geglu, gelu, glu_packed, layernorm, relu2, rmsnorm, rmsnorm_residual, rope_inplace,
                                rope_inplace_f32table, sdpa_causal, sdpa_causal_fixed_cache, sdpa_causal_fixed_cache_ptr,
                                sdpa_causal_strided, sigmoid, silu, slice_rows_range_ptr, split_qkv_batch, swiglu, tanh)
from pygpukit_amd.ops.reduction import argmax, argmax_int, argmax_rows, max, mean, min, softmax, sum, sum_axis
from pygpukit_amd.ops.unary import abs, cos, exp, log, neg, relu, rsqrt, sin, sqrt
from pygpukit_amd.ops.paged import (allocate_kv_cache, argmax_sample, check_eos, compute_cumsum, copy_to_paged_cache, gather_embeddings,
                                    paged_attention_v1, prepare_batch_inputs, prepare_position_ids, reshape_and_cache,
                                    scatter_last_token_logits)
from pygpukit_amd.ops.sampling import (sample_greedy, sample_multinomial, sample_token_gpu, sample_topk, sample_topk_to_buf_ptr,
                                       sample_topp, set_sampling_seed)
from pygpukit_amd.ops.tensor import (cast_bf16_to_f32, cast_f16_to_f32, cast_f32_to_bf16, cast_f32_to_f16, concat_axis0,
                                    repeat_interleave_axis1, reshape_copy, transpose_3d_012, transpose_3d_021, transpose_4d_0132,
                                    transpose_4d_0213)

__all__ = [n for n in dir() if not n.startswith("_")]
