// Packed-weight skinny GEMM (ops_pkgemm.hip): arguments and entry points shared with the engine.
#pragma once

#include "pgk_device.hip.h"
#include "pgk_internal.h"

namespace pgk {

enum { PK_EPI_BF16 = 0, PK_EPI_SLAB = 1, PK_EPI_ACCUM = 2, PK_EPI_SWIGLU = 3, PK_EPI_QKV = 4 };

struct PkArgs {
    const bf16* a;
    int lda;
    const bf16* wp;        // packed [N/16][K/32][64][8]
    int M, N, K;
    int ksteps;            // k-steps (32 k) per K split; multiple of 4
    int nblk, mblk, splits;
    int tiles_per_cb;      // n-tiles between consecutive n-blocks (4; 8 for the head epilogue)
    int tile_b_off;        // NTW == 2: the wave's second n-tile is this many tiles after its first
    void* c;
    int ldc;
    // head epilogue
    const bf16 *q_gamma, *k_gamma;
    float eps;
    const float *rope_cos, *rope_sin;
    bf16 *kcache, *vcache;
    int hq, hkv, max_seq, start_pos;
    // RMSNorm carried across projections (pkgemm_resid_nt produces, the SwiGLU / QKV / bf16 epilogues consume): the rows in `a`
    // are bf16(h * gamma), NOT normalised; ss_in[m][0 .. ss_n) are partial sums of squares of row m of h; the consumer scales
    // its result rows by 1 / sqrt(sum / K + ss_eps) - a per-row scalar commutes with the product.  Null: rows are normalised.
    const float* ss_in;
    int ss_n;
    float ss_eps;
};

constexpr int PK_SS_LD = 64;    // floats per row of a sum-of-squares table

pgk_status pack_weights_bf16(const void* w, void* wp, int N, int K, hipStream_t st);
// the same working copy for fp8 (e4m3) weights with 128 x 128 bf16 block scales: wp = bf16(code * scale), fragment-major
pgk_status pack_weights_fp8(const void* w, const void* scale, void* wp, int N, int K, hipStream_t st);
int pkgemm_pick_splits(int M, int N, int K);
bool pkgemm_shape_ok(int N, int K, bool splittable);
// h[M][N] += A[M][K] . W^T (no K split over workgroups: the K quarters of a workgroup's waves meet in LDS), and for the next
// RMSNorm: xpre[M][N] = bf16(h_new * gamma_next), ss_out[m][n-block] = sum of h_new^2 over the block's columns.
// gamma_next == nullptr: residual update only.  Returns the number of partial sums per row through *ss_n.
bool pkgemm_resid_ok(int N, int K);
pgk_status pkgemm_resid_nt(const bf16* a, int lda, const void* wp, float* h, int M, int N, int K, const bf16* gamma_next, bf16* xpre,
                           float* ss_out, int* ss_n, hipStream_t st);
pgk_status pkgemm_nt(const bf16* a, int lda, const void* wp, void* c, int ldc, int epi, int splits, int M, int N, int K, const PkArgs* head,
                     hipStream_t st);

}  // namespace pgk
