/*
 * pgk_hip.h - C ABI of libpgk_hip.so, the MI355X (gfx950) native backend for the
 * PyGPUkit LLM-inference hot path.
 *
 * This is the drop-in boundary.  In the reference the same boundary is the pybind11
 * module `_pygpukit_native` (native/bindings/module.cpp:10-21) over the C++ operator
 * API native/ops/ops.cuh; here it is a plain C ABI (no C++ types, no exceptions, no
 * torch types) bound from Python with ctypes (pygpukit_amd/_hip.py).  Every entry
 * point names the reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - every function returns pgk_status (0 = OK); pgk_last_error() gives the message
 *     for the calling thread (reference: C++ exceptions -> Python RuntimeError,
 *     native/core/types.hpp:108-111).
 *   - pointers are DEVICE pointers unless the parameter name starts with `h_`.
 *   - arrays are dense, row-major, C-contiguous (reference: core/array.py:197-215).
 *   - `dt` is the element type of the floating operands (PGK_F32 / PGK_F16 / PGK_BF16);
 *     bf16 travels as raw 16-bit words (reference: core/dtypes.py:54).
 *   - `stream` may be NULL = the calling thread's current stream
 *     (pgk_stream_set_current; default: one library-owned stream per device).  Ops
 *     never synchronise; only D2H copies, pgk_stream_sync and pgk_device_sync do.
 *     (The reference syncs after every op, native/ops/common/error.cuh:28-37.)
 *   - ops never allocate; callers pass outputs and workspaces.  Safe to capture.
 */
#ifndef PGK_HIP_H
#define PGK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int pgk_status;
#define PGK_OK 0
#define PGK_ERR_INVALID 1   /* bad shape / dtype / argument (reference: std::runtime_error) */
#define PGK_ERR_HIP 2       /* HIP runtime failure (reference: CudaError) */
#define PGK_ERR_UNSUPPORTED 3
#define PGK_ERR_RCCL 4
#define PGK_ERR_JIT 5        /* runtime compilation / module load / launch of a user kernel (reference: NvrtcError) */

/* Same order as the reference's DataType enum (native/bindings/core_bindings.cpp:19-30). */
typedef enum {
    PGK_F64 = 0, PGK_F32 = 1, PGK_F16 = 2, PGK_BF16 = 3, PGK_I64 = 4,
    PGK_I32 = 5, PGK_I16 = 6, PGK_I8 = 7, PGK_U8 = 8, PGK_I4 = 9
} pgk_dtype;

typedef void* pgk_stream;
typedef void* pgk_event;
typedef void* pgk_graph;
typedef void* pgk_comm;
typedef void* pgk_engine;

/* ---------------------------------------------------------------- errors / device -- */
const char* pgk_last_error(void);
const char* pgk_version(void);
/* native/bindings/core_bindings.cpp:52-57 (get_device_count, set_device, get_current_device,
 * device_synchronize, get_device_properties) */
pgk_status pgk_device_count(int* n);
pgk_status pgk_device_set(int dev);
pgk_status pgk_device_get(int* dev);
pgk_status pgk_device_sync(void);
typedef struct {
    char name[128];
    char arch[32];
    size_t total_mem;
    int cu_count;
    int wavefront_size;
    int clock_khz;
    int lds_per_cu;
    int l2_bytes;
} pgk_device_props_t;
pgk_status pgk_device_props(int dev, pgk_device_props_t* out);
pgk_status pgk_mem_info(size_t* free_bytes, size_t* total_bytes);

/* ---------------------------------------------------------------------- memory ------ */
/* Pooled device allocator.  Replaces per-array cuMemAlloc/cuMemFree
 * (native/core/memory.cpp:127-133) and the size-class pool of rust/pygpukit-core
 * (memory/pool.rs:106-422): power-of-two-ish size classes, cached free lists, no
 * hipMalloc on a pool hit (so allocation is legal while a stream is capturing). */
pgk_status pgk_malloc(void** ptr, size_t nbytes);
pgk_status pgk_free(void* ptr);
typedef struct {
    size_t bytes_in_use, bytes_cached, bytes_reserved_peak;
    uint64_t n_alloc, n_pool_hit, n_device_malloc, n_free;
} pgk_pool_stats_t;
pgk_status pgk_pool_stats(pgk_pool_stats_t* out);
pgk_status pgk_pool_trim(void);               /* return cached blocks to the driver */
pgk_status pgk_host_alloc(void** h_ptr, size_t nbytes);   /* pinned host memory */
pgk_status pgk_host_free(void* h_ptr);
/* GPUArray::copy_from_host / copy_to_host / fill_zeros (native/core/memory.hpp:87-91) and
 * memcpy_device_to_device_async (core_bindings.cpp:317).  h2d/d2h return after the copy
 * has completed (ordered after prior work on `stream`); *_async need pinned host memory. */
pgk_status pgk_memcpy_h2d(void* dst, const void* h_src, size_t nbytes, pgk_stream stream);
pgk_status pgk_memcpy_d2h(void* h_dst, const void* src, size_t nbytes, pgk_stream stream);
pgk_status pgk_memcpy_h2d_async(void* dst, const void* h_src, size_t nbytes, pgk_stream stream);
pgk_status pgk_memcpy_d2h_async(void* h_dst, const void* src, size_t nbytes, pgk_stream stream);
pgk_status pgk_memcpy_d2d(void* dst, const void* src, size_t nbytes, pgk_stream stream);
pgk_status pgk_memset(void* dst, int value, size_t nbytes, pgk_stream stream);
pgk_status pgk_fill(void* dst, double value, size_t n, pgk_dtype dt, pgk_stream stream); /* ones() */

/* ------------------------------------------------------- streams / events / graphs -- */
/* Stream(priority) native/core/stream.hpp:21-35 */
pgk_status pgk_stream_create(pgk_stream* out, int high_priority);
pgk_status pgk_stream_destroy(pgk_stream s);
pgk_status pgk_stream_sync(pgk_stream s);
pgk_status pgk_stream_set_current(pgk_stream s);     /* NULL restores the default stream */
pgk_status pgk_stream_get_current(pgk_stream* out);
/* CudaEvent native/core/event.hpp:13-38, event_elapsed_ms core_bindings.cpp:259 */
pgk_status pgk_event_create(pgk_event* out);
pgk_status pgk_event_destroy(pgk_event e);
pgk_status pgk_event_record(pgk_event e, pgk_stream s);
pgk_status pgk_event_sync(pgk_event e);
/* cudaStreamWaitEvent: work queued on `s` (NULL: the current stream) after this call runs after `e` */
pgk_status pgk_stream_wait_event(pgk_stream s, pgk_event e);
pgk_status pgk_event_query(pgk_event e, int* done);
pgk_status pgk_event_elapsed_ms(pgk_event start, pgk_event stop, float* ms);
/* CudaGraph native/core/cuda_graph.hpp:31-88: begin_capture / end_capture / replay /
 * synchronize / reset / is_ready / num_nodes.  Capture is on `stream` (thread-local
 * capture in the reference, cuda_graph.cu:84-107).  Pool blocks allocated by the capturing thread between begin and end
 * are baked into the graph: freeing one only parks it until pgk_graph_destroy, so replays never find them recycled. */
pgk_status pgk_graph_begin_capture(pgk_stream s);
pgk_status pgk_graph_end_capture(pgk_stream s, pgk_graph* out);
pgk_status pgk_graph_launch(pgk_graph g, pgk_stream s);
pgk_status pgk_graph_num_nodes(pgk_graph g, size_t* n);
pgk_status pgk_graph_destroy(pgk_graph g);
pgk_status pgk_stream_is_capturing(pgk_stream s, int* yes);

/* ------------------------------------------------------------------ elementwise ----- */
/* ops.cuh:24-37 add/mul/sub/div (out-of-place, same shape); op: 0 add, 1 sub, 2 mul, 3 div */
pgk_status pgk_binary(const void* a, const void* b, void* c, size_t n, int op, pgk_dtype dt, pgk_stream s);
/* ops.cuh:413,416 add_inplace / mul_inplace: a (op)= b */
pgk_status pgk_binary_inplace(void* a, const void* b, size_t n, int op, pgk_dtype dt, pgk_stream s);
/* ops.cuh:139 bias_add_inplace: out[rows,features] += bias[features] */
pgk_status pgk_bias_add_inplace(void* out, const void* bias, int rows, int features, pgk_dtype dt, pgk_stream s);
/* ops.cuh:136 gelu (tanh form 0.7978845608/0.044715), :176-179 silu; act: 0 silu, 1 gelu, 2 sigmoid, 3 tanh, 4 relu2 */
pgk_status pgk_activation(const void* x, void* y, size_t n, int act, pgk_dtype dt, pgk_stream s);
/* ... and the unary family of ops.cuh:60-101 (src/pygpukit/ops/unary.py:16-260) through the same entry:
 * act 5 exp, 6 log, 7 relu, 8 sin, 9 cos, 10 sqrt, 11 rsqrt, 12 abs, 13 neg */
/* ops.cuh:92-131 (src/pygpukit/ops/reduction.py:16-130,227): whole-array reduction to one element of the input dtype;
 * op: 0 sum, 1 mean, 2 max, 3 min.  Fixed two-level tree in fp32: the same bits on every run. */
pgk_status pgk_reduce(const void* x, void* out, size_t n, int op, pgk_dtype dt, pgk_stream s);
/* reduction.py:133-224 softmax over the last axis of [rows, n] (max-subtracted, fp32 math) */
pgk_status pgk_softmax_rows(const void* x, void* y, int rows, int n, pgk_dtype dt, pgk_stream s);
/* reduction.py:271-300 sum_axis of a 2-D [m, n]: axis 0 -> out[n], axis 1 -> out[m] */
pgk_status pgk_sum_axis(const void* x, void* out, int m, int n, int axis, pgk_dtype dt, pgk_stream s);
/* ops/elementwise.py:254-276 clamp to [lo, hi]; :279-308 where(cond != 0 ? a : b), cond one byte per element */
pgk_status pgk_clamp(const void* x, void* y, size_t n, float lo, float hi, pgk_dtype dt, pgk_stream s);
pgk_status pgk_where(const uint8_t* cond, const void* a, const void* b, void* y, size_t n, pgk_dtype dt, pgk_stream s);
/* int32 -> int64 (the reference's ops.argmax returns an int64 [1] array, reduction.py:249-268) */
pgk_status pgk_widen_i32_i64(const int32_t* src, int64_t* dst, size_t n, pgk_stream s);
/* ops.cuh:206-212 swiglu / geglu: out = act(gate) * up ; act: 0 silu, 1 gelu */
pgk_status pgk_glu(const void* gate, const void* up, void* out, size_t n, int act, pgk_dtype dt, pgk_stream s);
/* Row-packed GLU: gate_up[rows, 2*inter] (gate columns then up columns, the fused gate_up projection of
 * src/pygpukit/llm/layers/mlp.py:84-86) -> out[rows, inter] = act(gate) * up.  Replaces the reference's
 * narrow + silu + mul_inplace sequence (models/causal.py:537-549), which is only exact for rows == 1. */
pgk_status pgk_glu_packed(const void* gate_up, void* out, int rows, int inter, int act, pgk_dtype dt, pgk_stream s);
/* ops.cuh:426-436 cast_f32_to_bf16 / f32_to_f16 / bf16_to_f32 / f16_to_f32 (RNE) */
pgk_status pgk_cast(const void* src, pgk_dtype src_dt, void* dst, pgk_dtype dst_dt, size_t n, pgk_stream s);

/* ------------------------------------------------------------------------ norms ----- */
/* ops.cuh:152-155 rmsnorm(input[rows,features], gamma[features]) -> out (may alias input) */
pgk_status pgk_rmsnorm(const void* x, const void* gamma, void* out, int rows, int features, float eps,
                       pgk_dtype dt, pgk_stream s);
/* ops.cuh:199-201 rmsnorm_residual: out = rmsnorm(x + residual) * gamma */
pgk_status pgk_rmsnorm_residual(const void* x, const void* residual, const void* gamma, void* out, int rows,
                                int features, float eps, pgk_dtype dt, pgk_stream s);
/* ops.cuh:143 layernorm (population variance) */
pgk_status pgk_layernorm(const void* x, const void* gamma, const void* beta, void* out, int rows, int features,
                         float eps, pgk_dtype dt, pgk_stream s);

/* ------------------------------------------------------------------------- rope ----- */
/* ops.cuh:218 rope_inplace (table in dt) and :224 rope_inplace_f32table (fp32 table):
 * q[S,Hq,D], k[S,Hk,D], cos/sin[S,D]; rotate-half, table column d < D/2. */
pgk_status pgk_rope_inplace(void* q, void* k, const void* cos, const void* sin, int seq, int hq, int hk, int d,
                            pgk_dtype dt, int f32_table, pgk_stream s);

/* ------------------------------------------------------------- layout shuffles ------ */
/* ops.cuh:132 transpose (2-D) */
pgk_status pgk_transpose_2d(const void* in, void* out, int rows, int cols, int itemsize, pgk_stream s);
/* ops.cuh:352-354 transpose_3d_021: [d0,d1,d2] -> [d1,d0,d2] */
pgk_status pgk_transpose_3d_021(const void* in, void* out, int d0, int d1, int d2, int itemsize, pgk_stream s);
/* src/pygpukit/ops/tensor.py:256-318 transpose_3d_012 ([d0,d1,d2] -> [d0,d2,d1]) and :320-380 transpose_4d_0132
 * ([d0,d1,d2,d3] -> [d0,d1,d3,d2]): `batch` independent [rows, cols] matrices, each transposed */
pgk_status pgk_transpose_batched(const void* in, void* out, int batch, int rows, int cols, int itemsize, pgk_stream s);
/* tensor.py:191-254 transpose_4d_0213: [d0,d1,d2,d3] -> [d0,d2,d1,d3] */
pgk_status pgk_transpose_4d_0213(const void* in, void* out, int d0, int d1, int d2, int d3, int itemsize, pgk_stream s);
/* ops.cuh:349 repeat_interleave_axis1: [d0,d1,d2] -> [d0,d1*r,d2] */
pgk_status pgk_repeat_interleave_axis1(const void* in, void* out, int d0, int d1, int d2, int repeats,
                                       int itemsize, pgk_stream s);
/* ops.cuh:271 split_qkv_batch: qkv[rows,q+k+v] -> q[rows,q], k[rows,k], v[rows,v] */
pgk_status pgk_split_qkv_batch(const void* qkv, void* q, void* k, void* v, int rows, int q_dim, int k_dim,
                               int v_dim, int itemsize, pgk_stream s);
/* (concat_axis0 ops.cuh:345, reshape_copy :375-377, copy_to :419 are pgk_memcpy_d2d.) */

/* ------------------------------------------------------ embedding / KV cache -------- */
/* ops.cuh:403-405 embedding_lookup / _ptr / _batch: out[i,:] = table[ids[i],:].
 * `ids` is a device int32 array when ids_on_device, else h_id is used for row 0. */
pgk_status pgk_embedding_lookup(const void* table, void* out, int hidden, int itemsize, int h_id,
                                const int32_t* ids, int n_ids, pgk_stream s);
/* ops.cuh:410 slice_rows_range_ptr: out[0:count,:] = table[start:start+count,:], start from device int32 */
pgk_status pgk_slice_rows_range_ptr(const void* table, void* out, const int32_t* start_buf, int count, int row_elems,
                                    int itemsize, pgk_stream s);
/* ops.cuh:397-399 kv_cache_update_gqa / _ptr / kv_cache_prefill_gqa: scatter new_kv[S,Hkv,D]
 * into cache[Hc,max_seq,D] rows pos..pos+S-1; cache head h reads kv head h/(Hc/Hkv).  Hc == Hq is
 * the reference's GQA-expanded layout; Hc == Hkv the un-expanded MI355X layout.  pos_buf (device
 * int32) overrides h_pos when non-NULL (graph replay). */
pgk_status pgk_kv_cache_write(const void* new_kv, void* cache, int seq, int hkv, int hc, int max_seq, int d,
                              int itemsize, int h_pos, const int32_t* pos_buf, pgk_stream s);

/* -------------------------------------------------------------------- sampling ------ */
/* ops.cuh:104 argmax, :572 sample_greedy: index of the max over n elements of each of `rows`
 * rows; ties resolve to the LOWEST index (np.argmax; src/pygpukit/llm/sampling.py:60-61).
 * out_idx: device int32[rows]. */
pgk_status pgk_argmax(const void* x, int rows, int n, pgk_dtype dt, int32_t* out_idx, pgk_stream s);

/* sample_multinomial / sample_topk / sample_topp / sample_topk_to_buf_ptr (src/pygpukit/ops/sampling.py:11-141,
 * native/ops/sampling/sampling.cu): one token per logits row, temperature > 0, top_k = 0 disables top-k,
 * top_p = 1 disables the nucleus; the uniform random number comes from `u` or, when `u_buf` is non-NULL, from device
 * memory (graph-replay compatible).  Deterministic function of its inputs, defined in csrc/ops_sampling.hip and
 * restated by oracle/cpu_ref.py sample_token_u.  Result: int32 per row in device memory. */
pgk_status pgk_sample_token(const void* logits, int rows, int vocab, pgk_dtype dt, float temperature, int top_k,
                            float top_p, float u, const float* u_buf, int32_t* out_tokens, pgk_stream s);

/* ------------------------------------------------------------------ safetensors reader ------ */
/* SafeTensorsFile (src/pygpukit/llm/safetensors.py:122-235 over rust/pygpukit-core/src/llm/tensor_loader.rs): the file
 * is mmap'ed read-only and its JSON header parsed once.  dtype ids are safetensors.py:28-43 (0 F32, 1 F16, 2 BF16, 3 F64,
 * 4 F8_E4M3, 5 F8_E5M2, 6 I32, 7 I64, 8 I16, 9 I8, 10 U8, 11 BOOL); `offset` is from the start of the file. */
pgk_status pgk_st_open(const char* path, void** handle);
void pgk_st_close(void* handle);
int pgk_st_num_tensors(void* handle);
uint64_t pgk_st_file_size(void* handle);
const char* pgk_st_tensor_name(void* handle, int i);
pgk_status pgk_st_tensor_info(void* handle, const char* name, int* dtype, int* ndim, int64_t* shape8, uint64_t* offset,
                              uint64_t* nbytes);
pgk_status pgk_st_tensor_data(void* handle, const char* name, const void** ptr, uint64_t* nbytes);
/* mapped file -> device without a host copy (loader.py:160-175 memcpy_ptr_to_device) */
pgk_status pgk_st_upload(void* handle, const char* name, void* dst_device, uint64_t dst_bytes, pgk_stream s);

/* ----------------------------------------------------------------- runtime compilation ------ */
/* native/jit/compiler.hpp + kernel.hpp (bound in native/bindings/jit_bindings.cpp:65-122): NVRTC -> PTX ->
 * cuModuleLoadData -> cuLaunchKernel becomes hiprtc -> gfx950 code object -> hipModuleLoadData ->
 * hipModuleLaunchKernel.  libhiprtc is dlopen'ed on first use (is_nvrtc_available's contract).  `rtc_code` outputs
 * are hiprtcResult values (numerically nvrtcResult's) or 1000 NotLoaded / 1001 load failed / 1002 function not found /
 * 1003 launch failed, as in src/pygpukit/jit/compiler.py:20-43. */
int pgk_jit_available(void);
const char* pgk_jit_library_path(void);
pgk_status pgk_jit_version(int* major, int* minor);
/* compile_to_ptx: NVRTC arch flags in `options` are dropped, --offload-arch=gfx950 is supplied; a program handle is
 * returned even on a compilation error so that its log can be read. */
pgk_status pgk_jit_compile(const char* source, const char* name, const char* const* options, int n_options,
                           void** program_out, int* rtc_code);
const char* pgk_jit_program_log(void* program);
pgk_status pgk_jit_program_code(void* program, const void** code, size_t* size);
void pgk_jit_program_destroy(void* program);
/* JITKernel: module + function handle for one extern "C" __global__ function of a compiled program */
pgk_status pgk_jit_kernel_create(void* program, const char* func_name, void** kernel_out, int* rtc_code);
void pgk_jit_kernel_destroy(void* kernel);
pgk_status pgk_jit_suggested_block_size(void* kernel, size_t dynamic_smem, int* block_size);
/* args[i] points at the value of kernel argument i (cuLaunchKernel's kernelParams convention) */
pgk_status pgk_jit_launch(void* kernel, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                          unsigned shared_bytes, void** args, pgk_stream s);

/* ------------------------------------------------- paged KV cache / continuous batching ------ */
/* ops.cuh:466-478 paged_attention_v1 (native/ops/attention/paged_attention.cuh:46-200): single-query attention over a
 * paged cache.  Q/out [num_seqs, Hq, D]; K/V cache [num_blocks, Hkv, block_size, D]; block_tables [num_seqs,
 * max_blocks_per_seq] int32; context_lens [num_seqs] int32 (device).  scale <= 0 -> 1/sqrt(D).  max_context bounds the
 * context lengths (it sizes the KV split); contexts beyond 512 need `workspace` of
 * pgk_paged_attention_workspace_bytes bytes.  dt = PGK_BF16 / PGK_F16 (the reference is f16-only). */
size_t pgk_paged_attention_workspace_bytes(int num_seqs, int num_heads, int head_dim, int max_context);
pgk_status pgk_paged_attention_v1(const void* q, const void* k_cache, const void* v_cache, const int32_t* block_tables,
                                  const int32_t* context_lens, void* out, int num_seqs, int num_heads, int num_kv_heads,
                                  int head_dim, int block_size, int max_blocks_per_seq, int max_context, float scale,
                                  void* workspace, pgk_dtype dt, pgk_stream s);
/* ops.cuh:480-502 copy_to_paged_cache / reshape_and_cache (paged_attention.cuh:206-283): K_new/V_new
 * [n_tokens, Hkv, D] rows scattered to slot_mapping[token] = physical_block * block_size + offset (negative: skip). */
pgk_status pgk_paged_cache_write(const void* k_new, const void* v_new, void* k_cache, void* v_cache,
                                 const int32_t* slot_mapping, int n_tokens, int num_kv_heads, int block_size, int head_dim,
                                 int itemsize, pgk_stream s);
/* ops.cuh:520-528 scatter_last_token_logits (continuous_batching.cuh:103-133): out[b] = logits[seq_start[b] + seq_lens[b] - 1] */
pgk_status pgk_scatter_last_token_logits(const void* logits, void* out, const int32_t* seq_start, const int32_t* seq_lens,
                                         int batch, int vocab, int itemsize, pgk_stream s);
/* ops.cuh:530-539 prepare_position_ids (continuous_batching.cuh:139-165) */
pgk_status pgk_prepare_position_ids(const int32_t* seq_start, const int32_t* seq_ctx, const int32_t* is_prefill,
                                    const int32_t* input_lens, int32_t* position_ids, int batch, pgk_stream s);
/* ops.cuh:550-553 check_eos (continuous_batching.cuh:231-241), :555-556 compute_cumsum (exclusive) */
pgk_status pgk_check_eos(const int32_t* tokens, int32_t* finished, int n, int eos_token_id, pgk_stream s);
pgk_status pgk_exclusive_cumsum_i32(const int32_t* in, int32_t* out, int n, pgk_stream s);

/* ---------------------------------------------------------------------- matmul ------ */
/* ops.cuh:119-124 matmul: C[M,N] = A[M,K] B[K,N]  (row-major, fp32 accumulate, dt out). */
pgk_status pgk_gemm_nn(const void* a, const void* b, void* c, int m, int n, int k, pgk_dtype dt, pgk_stream s);
/* LinearBF16 (src/pygpukit/llm/layers/linear.py:46-99): C[M,N] = A[M,K] W[N,K]^T (+bias[N]) on the
 * PyTorch-layout weight directly - no transposed copy of W is ever made (the reference keeps W and W^T). */
pgk_status pgk_gemm_nt(const void* a, const void* w, const void* bias, void* c, int m, int n, int k,
                       pgk_dtype dt, pgk_stream s);
/* pygpukit_gemv_bf16_opt_sm120 (native/ops/matmul/gemv/bf16_bf16/sm120/bf16_opt.cu:36-47):
 * C[N] = A[K] . B[N,K]^T.  dt = PGK_BF16 / PGK_F16 / PGK_F32 (all fp32 accumulate). */
pgk_status pgk_gemv(const void* a, const void* b_nk, void* c, int k, int n, pgk_dtype dt, pgk_stream s);
/* gemv_fp8_bf16_sm120 / _batched (native/ops/matmul/gemv/w8a16_bf16/sm120/fp8_opt_kernels.cu:27-64):
 * C[M,N] = A[M,K] . (E4M3[B[N,K]] * scale[N/128,K/128])^T ; A, scale, C bf16; M >= 1. */
pgk_status pgk_gemv_fp8_bf16(const void* a, const uint8_t* b_nk, const void* scale, void* c, int m, int k, int n,
                             pgk_stream s);
/* pygpukit_w8a16_gemm_sm120 (native/bindings/gemm/fp8xbf16_bf16.cpp:9-12):
 * C[M,N] = A[M,K] . dequant(B_fp8[K,N], scale[K/128,N/128]) ; note the [K,N] layout. */
pgk_status pgk_w8a16_gemm_kn(const void* a, const uint8_t* b_kn, const void* scale, void* c, int m, int n, int k,
                             pgk_stream s);
/* Same product on the PyTorch-layout weight W_fp8[N,K] with scale[N/128,K/128] (the layout LinearFP8 stores,
 * src/pygpukit/llm/layers/linear.py:149-160): avoids the transposed fp8 + scale copies the reference
 * makes for its M > 1 path (linear.py:173-179). */
pgk_status pgk_w8a16_gemm_nk(const void* a, const uint8_t* w_nk, const void* scale, void* c, int m, int n, int k,
                             pgk_stream s);
/* gemm_fp8_fp8_blockwise_sm120 (src/pygpukit/ops/matmul/fp8.py:288-343; its native side is CUTLASS and absent
 * from the checkout): fp8 x fp8 MFMA GEMM with 128-wide block scales,
 *   C[m][n] = sum_kb a_scale[m][kb] * w_scale[n/128][kb] * sum_{k in kb} E4M3(a[m][k]) * E4M3(w[n][k]).
 * a_fp8 [M,K] codes with a_scale [M, K/128] fp32 (one scale per row per 128 k); w_fp8_nk [N,K] codes with
 * w_scale [ceil(N/128), K/128] bf16 (the LinearFP8 layout, linear.py:149-160); C bf16 [M,N].  K % 128 == 0. */
pgk_status pgk_gemm_fp8_nt(const uint8_t* a_fp8, const float* a_scale, const uint8_t* w_fp8_nk, const void* w_scale,
                           void* c, int m, int n, int k, pgk_stream s);
/* Activation quantiser for pgk_gemm_fp8_nt (the fp32-in "auto-quantise" half of matmul_fp8, fp8.py:20-83):
 * per (row, 128-k block) scale = absmax/448 (1 for an all-zero block), codes = RNE e4m3 of x/scale.
 * dt = PGK_BF16 / PGK_F16 / PGK_F32. */
pgk_status pgk_quantize_fp8_rows(const void* x, uint8_t* out_fp8, float* out_scale, int m, int k, pgk_dtype dt,
                                 pgk_stream s);
/* Weight quantiser: per 128x128 block scale = bf16(absmax/448), codes = RNE e4m3 of w/scale - the LinearFP8
 * storage format (loader.py:228-252 reads it from checkpoints; this builds it from bf16 weights on device). */
pgk_status pgk_quantize_fp8_blocks(const void* w_bf16, uint8_t* out_fp8, void* out_scale_bf16, int n, int k,
                                   pgk_stream s);

/* ------------------------------------------------------------------- attention ------ */
/* ops.cuh:287-290 sdpa_causal(Q[Hq,q,D], K[Hkv,kv,D], V, scale<=0 -> 1/sqrt(D)), mask
 * kv_pos < (kv_len - q_len) + q_pos + 1.  Strides are in elements so both the reference's
 * [H,S,D] layout and the projection's native [S,H,D] layout work without transposes; Hkv may be a
 * divisor of Hq (GQA without repeat_interleave). */
pgk_status pgk_sdpa_causal(const void* q, const void* k, const void* v, void* out, int hq, int hkv, int q_len,
                           int kv_len, int d, float scale, int64_t q_stride_h, int64_t q_stride_s,
                           int64_t kv_stride_h, int64_t kv_stride_s, int64_t o_stride_h, int64_t o_stride_s,
                           pgk_dtype dt, pgk_stream s);
/* ops.cuh:294-300 sdpa_causal_fixed_cache / _ptr: Q[Hq,q_len,D] over the first context_len rows of
 * cache[Hc,max_seq,D].  ctx_buf (device int32) overrides h_context_len when non-NULL.  q_len == 1
 * uses split-KV flash-decoding (replaces native/ops/nn/flash_decoding.cuh:75-377, fp16-only there);
 * `workspace` must hold pgk_sdpa_decode_workspace_bytes(). */
size_t pgk_sdpa_decode_workspace_bytes(int hq, int d, int max_seq);
pgk_status pgk_sdpa_fixed_cache(const void* q, const void* k_cache, const void* v_cache, void* out, int hq, int hc,
                                int q_len, int max_seq, int d, float scale, int h_context_len,
                                const int32_t* ctx_buf, void* workspace, pgk_dtype dt, pgk_stream s);

/* ---------------------------------------------------------------------- engine ------ */
/* Native decode/prefill launcher: one C call (or one hipGraph launch) per token step instead of
 * ~21 Python-dispatched launches per layer.  Replaces the device-dispatch half
 * of rust/pygpukit-core (dispatch/controller.rs:267-530) and the 2L+2 captured graphs of
 * src/pygpukit/llm/decode/m1_graph.py:248-589 with ONE whole-step graph whose token id, position
 * and context length live in device memory. */
typedef struct {
    int vocab_size, hidden_size, num_layers, num_heads, num_kv_heads, head_dim, intermediate_size;
    int max_seq_len;      /* KV-cache rows per sequence */
    int max_batch;        /* independent sequences resident on this GPU */
    float norm_eps, rope_theta;
    int weight_format;    /* 0 = bf16 linears, 1 = fp8-e4m3 linears with 128x128 bf16 block scales (w8a16),
                           * 2 = as 1, and prefill of > 128 tokens also quantises activations per row per 128 k
                           *     and runs the projections on the fp8 x fp8 MFMA GEMM (decode stays w8a16) */
    int use_qk_norm;
} pgk_model_config_t;

typedef struct {
    const void* attn_norm;          /* [H] bf16 */
    const void* w_qkv;              /* [(Hq+2Hkv)*D, H] bf16 or u8 */
    const void* s_qkv;              /* fp8 block scales or NULL */
    const void* q_norm; const void* k_norm;   /* [D] bf16 or NULL */
    const void* w_o;  const void* s_o;        /* [H, Hq*D] */
    const void* mlp_norm;           /* [H] */
    const void* w_gate_up; const void* s_gate_up;   /* [2I, H]: rows 0..I-1 gate, I..2I-1 up */
    const void* w_down; const void* s_down;         /* [H, I] */
} pgk_layer_weights_t;

pgk_status pgk_engine_create(const pgk_model_config_t* cfg, const void* embed, const void* lm_head,
                             const void* final_norm, const pgk_layer_weights_t* layers, pgk_engine* out);
pgk_status pgk_engine_destroy(pgk_engine e);
/* bytes the engine allocated from the pool (KV caches, activations, rope tables) */
pgk_status pgk_engine_bytes(pgk_engine e, size_t* kv_bytes, size_t* workspace_bytes);
/* Prefill `n` tokens of sequence slot `seq` starting at position start_pos; writes K/V rows, and if
 * h_logits_out != NULL copies the last row's logits (fp32 [V]) to host.  all_logits (device bf16
 * [n,V]) may be NULL. */
pgk_status pgk_engine_prefill(pgk_engine e, int seq, const int32_t* h_tokens, int n, int start_pos,
                              void* all_logits, float* h_last_logits, pgk_stream s);
/* Set the per-sequence decode state (token to feed, its position) for `batch` sequences. */
pgk_status pgk_engine_set_state(pgk_engine e, const int32_t* h_tokens, const int32_t* h_positions, int batch,
                                pgk_stream s);
/* Enqueue ONE decode step for the first `batch` sequences: embeds state tokens, runs all layers
 * (KV write at position, attention over position+1 rows), lm_head, greedy argmax; then
 * state.token = argmax, state.position += 1, and the token is appended to the engine's device
 * token log.  No host interaction: any number of steps can be queued back to back. */
pgk_status pgk_engine_decode_step(pgk_engine e, int batch, pgk_stream s);
/* Eager steps whose every launch carries its own start/stop hipEvent (hipExtLaunchKernelGGL: the dispatch's begin -> end
 * interval, what rocprofv3 --kernel-trace reports): per-kernel-class time sums (ms) and launch counts for the 8 classes
 * embed, norm_qkv, attn, oproj, gateup, down, lmhead, argmax (in that order).  Advances the decode state like n_iters
 * ordinary steps.  (Measurement only: the reference's counterpart is its KernelProfiler, native/core/profiler.hpp.) */
pgk_status pgk_engine_profile_step(pgk_engine e, int batch, int n_iters, float* h_ms_sum, int* h_count, pgk_stream s);
/* Timeline of ONE graph-replayed step (diagnostic): every workgroup of every kernel stamps the 100 MHz s_memrealtime
 * counter at its first and last instruction; per launch, in launch order,
 *   h_out[6 i .. 6 i + 5] = { kernel class, workgroups, first start, last start, first end, last end }
 * (times in 10 ns ticks from the step's first start).  `warm` replays precede the measured one; the state advances by
 * warm + 1 steps; the engine's own captured graph is untouched. */
pgk_status pgk_engine_timeline(pgk_engine e, int batch, int warm, uint64_t* h_out, int max_launches, int* n_launches, pgk_stream s);
/* Capture decode_step(batch) into hipGraphs owned by the engine / replay them.  A step has two launch sequences - the
 * short-context one (contexts <= 512, a single sequence <= 384: whole-context attention kernels, 4L+2 launches at batch 1)
 * and the split-KV one (5L+2), whose slices are cut for a context tier (1024, 2048, ... positions, the cache length) - all
 * correct at ANY context they cover; capture records the short sequence and one split-KV graph per tier the cache can hold,
 * and replay picks per step by the CONTEXT the step will see, from the positions last given to pgk_engine_set_state plus
 * the steps enqueued since (a host-side bound, a speed hint only).  The reference's fixed cache takes any max_seq_len with one code
 * path (src/pygpukit/llm/layers/attention.py:128-146, llm/decode/m1_graph.py:248-325). */
pgk_status pgk_engine_capture(pgk_engine e, int batch, pgk_stream s);
pgk_status pgk_engine_replay(pgk_engine e, int n_steps, pgk_stream s);
/* Read back: logits of the last step (device pointer, fp32 [batch,V]), token log (host copy). */
pgk_status pgk_engine_logits_ptr(pgk_engine e, void** logits_f32);
pgk_status pgk_engine_read_tokens(pgk_engine e, int32_t* h_out, int batch, int n_steps, pgk_stream s);
pgk_status pgk_engine_reset_log(pgk_engine e, pgk_stream s);
/* In-graph stochastic sampling (the graph-compatible sample_topk_to_buf_ptr of src/pygpukit/ops/sampling.py:40-71, for the
 * whole-step graph): each step draws one token per sequence from the step's fp32 logits with pgk_sample_token's
 * semantics; the uniform numbers are row (step counter % n_rows) of `h_uniforms` [n_rows][max_batch], copied to the device
 * here.  temperature <= 0 restores greedy argmax.  The sampling node's arguments are baked into a captured graph, so a
 * call that changes on/off, temperature, top_k, top_p or n_rows (or has to grow a buffer) DROPS the engine's captured
 * graph: pgk_engine_replay fails until pgk_engine_capture is called again.  A refill with identical parameters and
 * n_rows keeps the graph and only queues fresh uniforms. */
pgk_status pgk_engine_set_sampling(pgk_engine e, float temperature, int top_k, float top_p, const float* h_uniforms, int n_rows,
                                   pgk_stream s);
/* Diagnostic: per logged step, {s_memtime (shader clock ticks), s_memrealtime (100 MHz ticks)} stamped by the
 * step's last kernel: the in-kernel shader clock between two steps is d(memtime)/d(memrealtime) x 100 MHz
 * (MI355X_MICROARCH.md, DVFS give-back item 6).  h_out: uint64[2 * n_steps]. */
pgk_status pgk_engine_read_clock(pgk_engine e, uint64_t* h_out, int n_steps, pgk_stream s);
/* KV cache access for parity tests: pointers to layer `l`'s K and V caches [max_batch,Hkv,max_seq,D] bf16 */
pgk_status pgk_engine_kv_ptr(pgk_engine e, int layer, void** k, void** v);
/* device int32[max_batch] arrays holding each sequence's current token and position (the step's inputs and,
 * after it, its sampled tokens): what the data-parallel harness all-gathers over RCCL */
pgk_status pgk_engine_state_ptr(pgk_engine e, void** tokens, void** positions);
/* number of kernel launches one decode step enqueues (for reporting) */
pgk_status pgk_engine_launches_per_step(pgk_engine e, int* n);

/* ------------------------------------------------------------------------ RCCL ------ */
/* New functionality (the reference is single-GPU, docs/scheduler.md:358): data-parallel batch
 * decode over one 8xMI355X node.  One process per GPU; RCCL over xGMI only for the one-time weight
 * broadcast and the per-step gather of sampled tokens / logits. */
pgk_status pgk_comm_unique_id(char* h_id128);                       /* 128-byte ncclUniqueId */
pgk_status pgk_comm_init(pgk_comm* out, const char* h_id128, int rank, int world);
pgk_status pgk_comm_destroy(pgk_comm c);
pgk_status pgk_comm_broadcast(pgk_comm c, void* buf, size_t nbytes, int root, pgk_stream s);
pgk_status pgk_comm_all_gather(pgk_comm c, const void* send, void* recv, size_t nbytes_per_rank, pgk_stream s);
pgk_status pgk_comm_all_reduce_max_f64(pgk_comm c, double* buf, int n, pgk_stream s);
pgk_status pgk_comm_barrier(pgk_comm c, pgk_stream s);

#ifdef __cplusplus
}
#endif
#endif /* PGK_HIP_H */
