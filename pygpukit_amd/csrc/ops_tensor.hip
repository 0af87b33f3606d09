// Layout shuffles, embedding gathers, KV-cache scatter, argmax.  Pure byte movers:
// 16-byte accesses when alignment allows, LDS-tiled 2-D transpose.

#include "pgk_device.hip.h"
#include "pgk_internal.h"

namespace pgk {

static inline int cap_grid(size_t items, int block = 256) {
    size_t g = (items + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > 4096 ? 4096 : g);
}

// ---- 2-D transpose through a padded 64x64 LDS tile ----------------------------------------
template <class E>
__global__ __launch_bounds__(256) void transpose2d_kernel(const E* in, E* out, int rows, int cols) {
    __shared__ E tile[64][65];
    in += (size_t)blockIdx.z * rows * cols;    // batch of independent [rows, cols] matrices (transpose_3d_012 / 4d_0132)
    out += (size_t)blockIdx.z * rows * cols;
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    for (int r = ty; r < 64; r += 4) {
        const int gr = by + r, gc = bx + tx;
        if (gr < rows && gc < cols) tile[r][tx] = in[(size_t)gr * cols + gc];
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int oc = by + tx, orow = bx + r;  // out is [cols, rows]
        if (orow < cols && oc < rows) out[(size_t)orow * rows + oc] = tile[tx][r];
    }
}

// ---- row-granular copies: out_row(i) = in_row(map(i)), rows of `row_bytes` bytes -----------
// MODE 0: transpose_3d_021 [d0,d1,d2] -> [d1,d0,d2]
// MODE 1: repeat_interleave_axis1 [d0,d1,d2] -> [d0,d1*r,d2]
template <int MODE, class V>
__global__ void row_map_kernel(const V* in, V* out, int d0, int d1, int rep, int row_vecs) {
    const size_t out_rows = (MODE == 0) ? (size_t)d0 * d1 : (size_t)d0 * d1 * rep;
    const size_t total = out_rows * row_vecs;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += stride) {
        const int v = (int)(i % row_vecs);
        const size_t orow = i / row_vecs;
        size_t irow;
        if (MODE == 0) {
            const size_t j = orow / d0, ii = orow % d0;  // out[j][ii] = in[ii][j]
            irow = ii * d1 + j;
        } else {
            const size_t a = orow / ((size_t)d1 * rep), b = (orow / rep) % d1;
            irow = a * d1 + b;
        }
        out[orow * row_vecs + v] = in[irow * row_vecs + v];
    }
}

template <class V>
__global__ void split_qkv_kernel(const V* qkv, V* q, V* k, V* v, int rows, int qv, int kv, int vv) {
    const int tot = qv + kv + vv;
    const size_t total = (size_t)rows * tot;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % tot);
        const size_t r = i / tot;
        const V val = qkv[i];
        if (c < qv) q[r * qv + c] = val;
        else if (c < qv + kv) k[r * kv + (c - qv)] = val;
        else v[r * vv + (c - qv - kv)] = val;
    }
}

// out[i,:] = table[ids[i],:] ; ids from device memory, or the single host id when ids == nullptr
template <class V>
__global__ void gather_rows_kernel(const V* table, V* out, const int32_t* ids, int host_id, int row_vecs) {
    const int r = blockIdx.x;
    const int id = ids ? ids[r] : host_id;
    const V* src = table + (size_t)id * row_vecs;
    V* dst = out + (size_t)r * row_vecs;
    for (int v = threadIdx.x; v < row_vecs; v += blockDim.x) dst[v] = src[v];
}

// out[0:count,:] = table[start:start+count,:] with start read from device memory
template <class V>
__global__ void slice_rows_kernel(const V* table, V* out, const int32_t* start_buf, int count, int row_vecs) {
    const int start = start_buf[0];
    const size_t total = (size_t)count * row_vecs;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += stride)
        out[i] = table[(size_t)start * row_vecs + i];
}

// cache[hc, pos+s, :] = new_kv[s, hc / (Hc/Hkv), :]
template <class V>
__global__ void kv_write_kernel(const V* new_kv, V* cache, int seq, int hkv, int hc, int max_seq, int dvecs,
                                int host_pos, const int32_t* pos_buf) {
    const int pos = pos_buf ? pos_buf[0] : host_pos;
    const int rep = hc / hkv;
    const size_t total = (size_t)seq * hc * dvecs;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += stride) {
        const int v = (int)(i % dvecs);
        const size_t t = i / dvecs;
        const int h = (int)(t % hc);
        const int s = (int)(t / hc);
        const int row = pos + s;
        if (row < 0 || row >= max_seq) continue;  // never write outside the cache
        cache[((size_t)h * max_seq + row) * dvecs + v] = new_kv[((size_t)s * hkv + h / rep) * dvecs + v];
    }
}

// ---- argmax over each row; ties -> lowest index -------------------------------------------
__device__ __forceinline__ void argmax_combine(float& bv, int& bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

template <class T>
__global__ __launch_bounds__(1024) void argmax_kernel(const T* x, int n, int32_t* out_idx) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const T* row = x + (size_t)blockIdx.x * n;
    float bv = -INFINITY;
    int bi = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = to_f(row[i]);
        if (v > bv) { bv = v; bi = i; }  // ascending i per thread: strict > keeps the lowest index
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        argmax_combine(bv, bi, ov, oi);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w) argmax_combine(bv, bi, sv[w], si[w]);
        out_idx[blockIdx.x] = (bi == 0x7FFFFFFF) ? 0 : bi;
    }
}

// pick the widest vector type that divides row_bytes and keeps both pointers aligned
static inline int pick_vec_bytes(size_t row_bytes, std::initializer_list<const void*> ptrs) {
    for (int w : {16, 8, 4, 2, 1}) {
        bool ok = (row_bytes % w) == 0;
        for (const void* p : ptrs) ok = ok && ((reinterpret_cast<uintptr_t>(p) % w) == 0);
        if (ok) return w;
    }
    return 1;
}

}  // namespace pgk

using namespace pgk;

#define PGK_BY_VEC(w, ...)                                           \
    switch (w) {                                                     \
        case 16: { using V = uint4; __VA_ARGS__; } break;            \
        case 8: { using V = uint2; __VA_ARGS__; } break;             \
        case 4: { using V = uint32_t; __VA_ARGS__; } break;          \
        case 2: { using V = uint16_t; __VA_ARGS__; } break;          \
        default: { using V = uint8_t; __VA_ARGS__; } break;          \
    }

extern "C" {

pgk_status pgk_transpose_2d(const void* in, void* out, int rows, int cols, int itemsize, pgk_stream s) {
    PGK_REQUIRE(in && out, "pgk_transpose_2d: null pointer");
    PGK_REQUIRE(rows >= 0 && cols >= 0, "pgk_transpose_2d: bad shape");
    if (!rows || !cols) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    dim3 grid(ceil_div(cols, 64), ceil_div(rows, 64));
    switch (itemsize) {
        case 1: transpose2d_kernel<uint8_t><<<grid, 256, 0, st>>>((const uint8_t*)in, (uint8_t*)out, rows, cols); break;
        case 2: transpose2d_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t*)in, (uint16_t*)out, rows, cols); break;
        case 4: transpose2d_kernel<uint32_t><<<grid, 256, 0, st>>>((const uint32_t*)in, (uint32_t*)out, rows, cols); break;
        case 8: transpose2d_kernel<uint2><<<grid, 256, 0, st>>>((const uint2*)in, (uint2*)out, rows, cols); break;
        default: return set_error(PGK_ERR_INVALID, "pgk_transpose_2d: itemsize %d", itemsize);
    }
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_transpose_batched(const void* in, void* out, int batch, int rows, int cols, int itemsize, pgk_stream s) {
    PGK_REQUIRE(in && out, "pgk_transpose_batched: null pointer");
    PGK_REQUIRE(batch >= 0 && batch <= 65535 && rows >= 0 && cols >= 0, "pgk_transpose_batched: bad shape (batch %d <= 65535)", batch);
    if (!batch || !rows || !cols) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    dim3 grid(ceil_div(cols, 64), ceil_div(rows, 64), batch);
    switch (itemsize) {
        case 1: transpose2d_kernel<uint8_t><<<grid, 256, 0, st>>>((const uint8_t*)in, (uint8_t*)out, rows, cols); break;
        case 2: transpose2d_kernel<uint16_t><<<grid, 256, 0, st>>>((const uint16_t*)in, (uint16_t*)out, rows, cols); break;
        case 4: transpose2d_kernel<uint32_t><<<grid, 256, 0, st>>>((const uint32_t*)in, (uint32_t*)out, rows, cols); break;
        case 8: transpose2d_kernel<uint2><<<grid, 256, 0, st>>>((const uint2*)in, (uint2*)out, rows, cols); break;
        default: return set_error(PGK_ERR_INVALID, "pgk_transpose_batched: itemsize %d", itemsize);
    }
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_transpose_4d_0213(const void* in, void* out, int d0, int d1, int d2, int d3, int itemsize, pgk_stream s) {
    PGK_REQUIRE(in && out, "pgk_transpose_4d_0213: null pointer");
    if (!d0 || !d1 || !d2 || !d3) return PGK_OK;
    // each of the d0 slabs is a [d1, d2, d3] -> [d2, d1, d3] row permutation
    const size_t slab = (size_t)d1 * d2 * d3 * itemsize;
    for (int b = 0; b < d0; ++b)
        if (pgk_status r = pgk_transpose_3d_021((const char*)in + b * slab, (char*)out + b * slab, d1, d2, d3, itemsize, s)) return r;
    return PGK_OK;
}

pgk_status pgk_transpose_3d_021(const void* in, void* out, int d0, int d1, int d2, int itemsize, pgk_stream s) {
    PGK_REQUIRE(in && out, "pgk_transpose_3d_021: null pointer");
    if (!d0 || !d1 || !d2) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    const size_t row_bytes = (size_t)d2 * itemsize;
    const int w = pick_vec_bytes(row_bytes, {in, out});
    const int row_vecs = (int)(row_bytes / w);
    const int grid = cap_grid((size_t)d0 * d1 * row_vecs);
    PGK_BY_VEC(w, row_map_kernel<0, V><<<grid, 256, 0, st>>>((const V*)in, (V*)out, d0, d1, 1, row_vecs));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_repeat_interleave_axis1(const void* in, void* out, int d0, int d1, int d2, int repeats, int itemsize,
                                       pgk_stream s) {
    PGK_REQUIRE(in && out, "pgk_repeat_interleave_axis1: null pointer");
    PGK_REQUIRE(repeats >= 1, "pgk_repeat_interleave_axis1: repeats=%d", repeats);
    if (!d0 || !d1 || !d2) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    const size_t row_bytes = (size_t)d2 * itemsize;
    const int w = pick_vec_bytes(row_bytes, {in, out});
    const int row_vecs = (int)(row_bytes / w);
    const int grid = cap_grid((size_t)d0 * d1 * repeats * row_vecs);
    PGK_BY_VEC(w, row_map_kernel<1, V><<<grid, 256, 0, st>>>((const V*)in, (V*)out, d0, d1, repeats, row_vecs));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_split_qkv_batch(const void* qkv, void* q, void* k, void* v, int rows, int q_dim, int k_dim, int v_dim,
                               int itemsize, pgk_stream s) {
    PGK_REQUIRE(qkv && q && k && v, "pgk_split_qkv_batch: null pointer");
    if (!rows) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    int w = 16;
    for (; w > 1; w >>= 1) {
        if (w < itemsize) { w = itemsize; break; }
        const bool ok = ((size_t)q_dim * itemsize) % w == 0 && ((size_t)k_dim * itemsize) % w == 0 &&
                        ((size_t)v_dim * itemsize) % w == 0 && (uintptr_t)qkv % w == 0 && (uintptr_t)q % w == 0 &&
                        (uintptr_t)k % w == 0 && (uintptr_t)v % w == 0;
        if (ok) break;
    }
    if (w < itemsize) w = itemsize;
    const int qv = q_dim * itemsize / w, kv = k_dim * itemsize / w, vv = v_dim * itemsize / w;
    const int grid = cap_grid((size_t)rows * (qv + kv + vv));
    PGK_BY_VEC(w, split_qkv_kernel<V><<<grid, 256, 0, st>>>((const V*)qkv, (V*)q, (V*)k, (V*)v, rows, qv, kv, vv));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_embedding_lookup(const void* table, void* out, int hidden, int itemsize, int h_id, const int32_t* ids,
                                int n_ids, pgk_stream s) {
    PGK_REQUIRE(table && out, "pgk_embedding_lookup: null pointer");
    PGK_REQUIRE(hidden > 0 && n_ids >= 1, "pgk_embedding_lookup: bad shape");
    PGK_REQUIRE(ids || h_id >= 0, "pgk_embedding_lookup: negative token id %d", h_id);
    hipStream_t st = resolve_stream(s);
    const size_t row_bytes = (size_t)hidden * itemsize;
    const int w = pick_vec_bytes(row_bytes, {table, out});
    const int row_vecs = (int)(row_bytes / w);
    PGK_BY_VEC(w, gather_rows_kernel<V><<<n_ids, 256, 0, st>>>((const V*)table, (V*)out, ids, h_id, row_vecs));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_slice_rows_range_ptr(const void* table, void* out, const int32_t* start_buf, int count, int row_elems,
                                    int itemsize, pgk_stream s) {
    PGK_REQUIRE(table && out && start_buf, "pgk_slice_rows_range_ptr: null pointer");
    if (!count) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    const size_t row_bytes = (size_t)row_elems * itemsize;
    const int w = pick_vec_bytes(row_bytes, {table, out});
    const int row_vecs = (int)(row_bytes / w);
    const int grid = cap_grid((size_t)count * row_vecs);
    PGK_BY_VEC(w, slice_rows_kernel<V><<<grid, 256, 0, st>>>((const V*)table, (V*)out, start_buf, count, row_vecs));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_kv_cache_write(const void* new_kv, void* cache, int seq, int hkv, int hc, int max_seq, int d,
                              int itemsize, int h_pos, const int32_t* pos_buf, pgk_stream s) {
    PGK_REQUIRE(new_kv && cache, "pgk_kv_cache_write: null pointer");
    PGK_REQUIRE(hkv > 0 && hc > 0 && hc % hkv == 0, "pgk_kv_cache_write: cache heads %d not a multiple of kv heads %d", hc, hkv);
    PGK_REQUIRE(pos_buf || (h_pos >= 0 && h_pos + seq <= max_seq), "pgk_kv_cache_write: rows %d..%d outside cache of %d",
                h_pos, h_pos + seq, max_seq);
    if (!seq) return PGK_OK;
    hipStream_t st = resolve_stream(s);
    const size_t row_bytes = (size_t)d * itemsize;
    const int w = pick_vec_bytes(row_bytes, {new_kv, cache});
    const int dvecs = (int)(row_bytes / w);
    const int grid = cap_grid((size_t)seq * hc * dvecs);
    PGK_BY_VEC(w, kv_write_kernel<V><<<grid, 256, 0, st>>>((const V*)new_kv, (V*)cache, seq, hkv, hc, max_seq, dvecs,
                                                          h_pos, pos_buf));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

pgk_status pgk_argmax(const void* x, int rows, int n, pgk_dtype dt, int32_t* out_idx, pgk_stream s) {
    PGK_REQUIRE(x && out_idx, "pgk_argmax: null pointer");
    PGK_REQUIRE(rows >= 1 && n >= 1, "pgk_argmax: bad shape [%d,%d]", rows, n);
    hipStream_t st = resolve_stream(s);
    const int block = n >= 16384 ? 1024 : 256;
    PGK_DISPATCH_FLOAT(dt, "pgk_argmax", argmax_kernel<T><<<rows, block, 0, st>>>((const T*)x, n, out_idx));
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

}  // extern "C"
