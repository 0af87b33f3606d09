from pygpukit_amd.ops.matmul.fp8 import (fp8_available, fp8_init_lut, gemm_fp8_fp8_blockwise_nt, matmul_fp8, matmul_fp8_sm120,
                                         quantize_fp8_blocks, quantize_fp8_rows)
from pygpukit_amd.ops.matmul.gemv import gemv_bf16, gemv_bf16_opt_available, gemv_fp8_bf16, gemv_fp8_bf16_batched
from pygpukit_amd.ops.matmul.generic import batched_matmul, linear_bias_gelu, matmul, matmul_nt, transpose
from pygpukit_amd.ops.matmul.w8a16 import gemm_w8a16_init_lut, w8a16_gemm, w8a16_gemm_nk, w8a16_gemm_sm120

__all__ = ["matmul", "matmul_nt", "transpose", "batched_matmul", "linear_bias_gelu", "gemv_bf16",
           "gemv_bf16_opt_available", "gemv_fp8_bf16", "gemv_fp8_bf16_batched", "w8a16_gemm_sm120", "w8a16_gemm", "w8a16_gemm_nk",
           "gemm_w8a16_init_lut", "matmul_fp8", "matmul_fp8_sm120", "gemm_fp8_fp8_blockwise_nt", "quantize_fp8_rows",
           "quantize_fp8_blocks", "fp8_available", "fp8_init_lut"]
