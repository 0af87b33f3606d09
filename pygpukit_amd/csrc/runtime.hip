// Runtime core of libpgk_hip.so: errors, device, pooled allocator, streams, events,
// hipGraph capture/replay, copies.  MI355X counterpart of the reference's native/core
// (memory.cpp, stream.cpp, event.cpp, cuda_graph.cu) plus the device half of the Rust
// memory pool; see include/pgk_hip.h for the per-function citations.

#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <tuple>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "pgk_device.hip.h"
#include "pgk_internal.h"

namespace pgk {

static thread_local std::string g_last_error;
static thread_local hipStream_t g_current_stream = nullptr;
static thread_local bool g_has_current = false;

int set_error(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

static void register_stream(hipStream_t s);   // pool registry (below)

// One library-owned (non-blocking) stream per device, created on first use.
static std::mutex g_stream_mu;
static hipStream_t g_default_streams[64] = {nullptr};

static hipStream_t default_stream() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_stream_mu);
    if (!g_default_streams[dev]) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
        g_default_streams[dev] = s;
        register_stream(s);
    }
    return g_default_streams[dev];
}

hipStream_t resolve_stream(pgk_stream s) {
    if (s) return static_cast<hipStream_t>(s);
    if (g_has_current) return g_current_stream;
    return default_stream();
}

// ------------------------------------------------------------------------------------------
// Pooled allocator.  Size classes: powers of two from 512 B to 1 MiB, then multiples of 2 MiB.
// A hit costs no driver call, which also makes allocation legal while a stream is being captured.
//
// Every op is asynchronous, so a freed block may still be in use by work already queued.  The pool is therefore stream-
// aware (the caching-allocator scheme): a block belongs to the stream that was current when it was allocated; a free puts
// it on THAT stream's list, where the same stream may reuse it at once (stream order protects it); another stream takes
// it only when the owning stream has drained (hipStreamQuery), i.e. when everything queued before the free has finished.
// Blocks allocated while a graph is being captured (pgk_graph_begin_capture .. end_capture on this thread) are baked into
// the graph's nodes: freeing one parks it until that graph is destroyed, so no replay can find its temporaries recycled.
// (A block used on a stream other than its owner's is the caller's to order - an event or a sync before dropping it.)
// ------------------------------------------------------------------------------------------
struct Block {
    int dev;
    size_t cls;
    hipStream_t stream;      // owner
    uint64_t capture_id;     // != 0: allocated during that capture
};
struct Pool {
    std::mutex mu;
    std::map<std::tuple<int, size_t, hipStream_t>, std::vector<void*>> free_lists;
    std::unordered_map<void*, Block> live;
    std::map<uint64_t, std::vector<std::pair<void*, Block>>> parked;   // capture id -> freed blocks the graph may still write
    std::set<hipStream_t> streams;   // streams this library created and has not destroyed: the only ones it may query
    uint64_t next_capture_id = 1;
    pgk_pool_stats_t st{};
    size_t reserved = 0;
};
static thread_local uint64_t g_capture_id = 0;   // the capture this thread is recording (0: none)
static Pool& pool() {
    static Pool* p = new Pool();  // intentionally leaked: frees may run during interpreter teardown
    return *p;
}

static void register_stream(hipStream_t s) {
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    P.streams.insert(s);
}

static size_t size_class(size_t n) {
    if (n <= 512) return 512;
    if (n <= (1u << 20)) {
        size_t c = 512;
        while (c < n) c <<= 1;
        return c;
    }
    const size_t g = 2u << 20;
    return (n + g - 1) / g * g;
}

template <class T>
__global__ void fill_kernel(T* p, T v, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

}  // namespace pgk

using namespace pgk;

extern "C" {

const char* pgk_last_error(void) { return g_last_error.c_str(); }
const char* pgk_version(void) { return "pgk_hip 0.1 (gfx950)"; }

pgk_status pgk_device_count(int* n) {
    PGK_REQUIRE(n, "pgk_device_count: null output");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return set_error(PGK_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n = c;
    return PGK_OK;
}
pgk_status pgk_device_set(int dev) { PGK_CHECK_HIP(hipSetDevice(dev)); return PGK_OK; }
pgk_status pgk_device_get(int* dev) { PGK_REQUIRE(dev, "null"); PGK_CHECK_HIP(hipGetDevice(dev)); return PGK_OK; }
pgk_status pgk_device_sync(void) { PGK_CHECK_HIP(hipDeviceSynchronize()); return PGK_OK; }

pgk_status pgk_device_props(int dev, pgk_device_props_t* out) {
    PGK_REQUIRE(out, "pgk_device_props: null output");
    hipDeviceProp_t p;
    PGK_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    memset(out, 0, sizeof(*out));
    strncpy(out->name, p.name, sizeof(out->name) - 1);
    strncpy(out->arch, p.gcnArchName, sizeof(out->arch) - 1);
    out->total_mem = p.totalGlobalMem;
    out->cu_count = p.multiProcessorCount;
    out->wavefront_size = p.warpSize;
    out->clock_khz = p.clockRate;
    out->lds_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    out->l2_bytes = p.l2CacheSize;
    return PGK_OK;
}

pgk_status pgk_mem_info(size_t* free_bytes, size_t* total_bytes) {
    size_t f = 0, t = 0;
    PGK_CHECK_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return PGK_OK;
}

// ------------------------------------------------------------------------------ memory ----
pgk_status pgk_malloc(void** ptr, size_t nbytes) {
    PGK_REQUIRE(ptr, "pgk_malloc: null output");
    *ptr = nullptr;
    int dev = 0;
    PGK_CHECK_HIP(hipGetDevice(&dev));
    const size_t cls = size_class(nbytes ? nbytes : 1);
    hipStream_t owner = resolve_stream(nullptr);
    Pool& P = pool();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        P.st.n_alloc++;
        auto take = [&](std::vector<void*>& list) {
            void* p = list.back();
            list.pop_back();
            P.live[p] = Block{dev, cls, owner, g_capture_id};
            P.st.n_pool_hit++;
            P.st.bytes_cached -= cls;
            P.st.bytes_in_use += cls;
            *ptr = p;
        };
        auto it = P.free_lists.find({dev, cls, owner});
        if (it != P.free_lists.end() && !it->second.empty()) { take(it->second); return PGK_OK; }
        // another stream's cached block of this class: only once that stream has drained (never while capturing: a
        // query on a capturing stream is an error, and the answer would say nothing about replays)
        // key stream == nullptr: blocks whose owner was destroyed (after a sync) - free for anyone
        if (g_capture_id == 0) {
            for (auto lo = P.free_lists.lower_bound({dev, cls, nullptr}); lo != P.free_lists.end(); ++lo) {
                if (std::get<0>(lo->first) != dev || std::get<1>(lo->first) != cls) break;
                if (lo->second.empty()) continue;
                hipStream_t os = std::get<2>(lo->first);
                if (os != nullptr) {
                    if (!P.streams.count(os)) continue;                     // not ours to query
                    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                    if (hipStreamIsCapturing(os, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); continue; }
                    if (hipStreamQuery(os) != hipSuccess) { (void)hipGetLastError(); continue; }
                }
                take(lo->second);
                return PGK_OK;
            }
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, cls);
    if (e != hipSuccess) {
        // Out of memory: give cached blocks back to the driver and retry once.
        (void)hipGetLastError();
        pgk_pool_trim();
        e = hipMalloc(&p, cls);
    }
    if (e != hipSuccess)
        return set_error(PGK_ERR_HIP, "pgk_malloc(%zu bytes): %s", nbytes, hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(P.mu);
    P.live[p] = Block{dev, cls, owner, g_capture_id};
    P.st.n_device_malloc++;
    P.st.bytes_in_use += cls;
    P.reserved += cls;
    if (P.reserved > P.st.bytes_reserved_peak) P.st.bytes_reserved_peak = P.reserved;
    *ptr = p;
    return PGK_OK;
}

pgk_status pgk_free(void* ptr) {
    if (!ptr) return PGK_OK;
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.live.find(ptr);
    PGK_REQUIRE(it != P.live.end(), "pgk_free: pointer %p was not allocated by pgk_malloc", ptr);
    const Block b = it->second;
    P.live.erase(it);
    P.st.n_free++;
    P.st.bytes_in_use -= b.cls;
    P.st.bytes_cached += b.cls;
    auto pk = b.capture_id ? P.parked.find(b.capture_id) : P.parked.end();
    if (pk != P.parked.end()) pk->second.push_back({ptr, b});                    // the graph that captured it is still alive
    else P.free_lists[{b.dev, b.cls, P.streams.count(b.stream) ? b.stream : nullptr}].push_back(ptr);
    return PGK_OK;
}

pgk_status pgk_pool_stats(pgk_pool_stats_t* out) {
    PGK_REQUIRE(out, "pgk_pool_stats: null output");
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    *out = P.st;
    return PGK_OK;
}

pgk_status pgk_pool_trim(void) {
    Pool& P = pool();
    std::vector<std::pair<int, void*>> victims;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (auto& kv : P.free_lists) {
            for (void* p : kv.second) {
                victims.push_back({std::get<0>(kv.first), p});
                P.st.bytes_cached -= std::get<1>(kv.first);
                P.reserved -= std::get<1>(kv.first);
            }
            kv.second.clear();
        }
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& v : victims) {
        (void)hipSetDevice(v.first);
        (void)hipFree(v.second);
    }
    (void)hipSetDevice(cur);
    return PGK_OK;
}

pgk_status pgk_host_alloc(void** h_ptr, size_t nbytes) {
    PGK_REQUIRE(h_ptr, "pgk_host_alloc: null output");
    PGK_CHECK_HIP(hipHostMalloc(h_ptr, nbytes ? nbytes : 1, hipHostMallocDefault));
    return PGK_OK;
}
pgk_status pgk_host_free(void* h_ptr) {
    if (h_ptr) PGK_CHECK_HIP(hipHostFree(h_ptr));
    return PGK_OK;
}

pgk_status pgk_memcpy_h2d(void* dst, const void* h_src, size_t nbytes, pgk_stream stream) {
    if (!nbytes) return PGK_OK;
    hipStream_t s = resolve_stream(stream);
    PGK_CHECK_HIP(hipMemcpyAsync(dst, h_src, nbytes, hipMemcpyHostToDevice, s));
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cs);
    if (cs == hipStreamCaptureStatusNone) PGK_CHECK_HIP(hipStreamSynchronize(s));
    return PGK_OK;
}
pgk_status pgk_memcpy_d2h(void* h_dst, const void* src, size_t nbytes, pgk_stream stream) {
    if (!nbytes) return PGK_OK;
    hipStream_t s = resolve_stream(stream);
    PGK_CHECK_HIP(hipMemcpyAsync(h_dst, src, nbytes, hipMemcpyDeviceToHost, s));
    PGK_CHECK_HIP(hipStreamSynchronize(s));
    return PGK_OK;
}
pgk_status pgk_memcpy_h2d_async(void* dst, const void* h_src, size_t nbytes, pgk_stream stream) {
    if (!nbytes) return PGK_OK;
    PGK_CHECK_HIP(hipMemcpyAsync(dst, h_src, nbytes, hipMemcpyHostToDevice, resolve_stream(stream)));
    return PGK_OK;
}
pgk_status pgk_memcpy_d2h_async(void* h_dst, const void* src, size_t nbytes, pgk_stream stream) {
    if (!nbytes) return PGK_OK;
    PGK_CHECK_HIP(hipMemcpyAsync(h_dst, src, nbytes, hipMemcpyDeviceToHost, resolve_stream(stream)));
    return PGK_OK;
}
pgk_status pgk_memcpy_d2d(void* dst, const void* src, size_t nbytes, pgk_stream stream) {
    if (!nbytes) return PGK_OK;
    PGK_CHECK_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, resolve_stream(stream)));
    return PGK_OK;
}
pgk_status pgk_memset(void* dst, int value, size_t nbytes, pgk_stream stream) {
    if (!nbytes) return PGK_OK;
    PGK_CHECK_HIP(hipMemsetAsync(dst, value, nbytes, resolve_stream(stream)));
    return PGK_OK;
}

pgk_status pgk_fill(void* dst, double value, size_t n, pgk_dtype dt, pgk_stream stream) {
    if (!n) return PGK_OK;
    hipStream_t s = resolve_stream(stream);
    const int block = 256;
    const int grid = (int)((n + block - 1) / block < 2048 ? (n + block - 1) / block : 2048);
    switch (dt) {
        case PGK_F64: fill_kernel<double><<<grid, block, 0, s>>>((double*)dst, value, n); break;
        case PGK_F32: fill_kernel<float><<<grid, block, 0, s>>>((float*)dst, (float)value, n); break;
        case PGK_F16: fill_kernel<_Float16><<<grid, block, 0, s>>>((_Float16*)dst, (_Float16)value, n); break;
        case PGK_BF16: {
            // host-side RNE of the fill value
            float f = (float)value;
            uint32_t u;
            memcpy(&u, &f, 4);
            uint16_t b = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
            fill_kernel<uint16_t><<<grid, block, 0, s>>>((uint16_t*)dst, b, n);
        } break;
        case PGK_I64: fill_kernel<long long><<<grid, block, 0, s>>>((long long*)dst, (long long)value, n); break;
        case PGK_I32: fill_kernel<int><<<grid, block, 0, s>>>((int*)dst, (int)value, n); break;
        case PGK_I16: fill_kernel<short><<<grid, block, 0, s>>>((short*)dst, (short)value, n); break;
        case PGK_I8: fill_kernel<signed char><<<grid, block, 0, s>>>((signed char*)dst, (signed char)value, n); break;
        case PGK_U8: fill_kernel<unsigned char><<<grid, block, 0, s>>>((unsigned char*)dst, (unsigned char)value, n); break;
        default: return set_error(PGK_ERR_INVALID, "pgk_fill: unsupported dtype %d", (int)dt);
    }
    PGK_LAUNCH_CHECK();
    return PGK_OK;
}

// ------------------------------------------------------------ streams / events / graphs ----
pgk_status pgk_stream_create(pgk_stream* out, int high_priority) {
    PGK_REQUIRE(out, "pgk_stream_create: null output");
    int lo = 0, hi = 0;
    PGK_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s = nullptr;
    PGK_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? hi : lo));
    {
        Pool& P = pool();
        std::lock_guard<std::mutex> lk(P.mu);
        P.streams.insert(s);
    }
    *out = s;
    return PGK_OK;
}
// Drains the stream first: the pool blocks it owns (cached or still live) then belong to nobody and may be reused by anyone.
pgk_status pgk_stream_destroy(pgk_stream s) {
    if (!s) return PGK_OK;
    hipStream_t st = (hipStream_t)s;
    bool ours = false;
    {
        Pool& P = pool();
        std::lock_guard<std::mutex> lk(P.mu);
        ours = P.streams.count(st) != 0;
    }
    if (!ours) return PGK_OK;             // a wrapped foreign handle: not ours to destroy (or already destroyed)
    (void)hipStreamSynchronize(st);
    {
        Pool& P = pool();
        std::lock_guard<std::mutex> lk(P.mu);
        P.streams.erase(st);
        for (auto it = P.free_lists.begin(); it != P.free_lists.end();) {
            if (std::get<2>(it->first) == st) {
                auto& orphan = P.free_lists[{std::get<0>(it->first), std::get<1>(it->first), nullptr}];
                orphan.insert(orphan.end(), it->second.begin(), it->second.end());
                it = P.free_lists.erase(it);
            } else {
                ++it;
            }
        }
    }
    if (g_has_current && g_current_stream == st) { g_current_stream = nullptr; g_has_current = false; }
    PGK_CHECK_HIP(hipStreamDestroy(st));
    return PGK_OK;
}
pgk_status pgk_stream_sync(pgk_stream s) { PGK_CHECK_HIP(hipStreamSynchronize(resolve_stream(s))); return PGK_OK; }
pgk_status pgk_stream_set_current(pgk_stream s) {
    g_current_stream = (hipStream_t)s;
    g_has_current = (s != nullptr);
    return PGK_OK;
}
pgk_status pgk_stream_get_current(pgk_stream* out) {
    PGK_REQUIRE(out, "pgk_stream_get_current: null output");
    *out = resolve_stream(nullptr);
    return PGK_OK;
}

pgk_status pgk_event_create(pgk_event* out) {
    PGK_REQUIRE(out, "pgk_event_create: null output");
    hipEvent_t e = nullptr;
    PGK_CHECK_HIP(hipEventCreate(&e));
    *out = e;
    return PGK_OK;
}
pgk_status pgk_event_destroy(pgk_event e) { if (e) PGK_CHECK_HIP(hipEventDestroy((hipEvent_t)e)); return PGK_OK; }
pgk_status pgk_event_record(pgk_event e, pgk_stream s) {
    PGK_CHECK_HIP(hipEventRecord((hipEvent_t)e, resolve_stream(s)));
    return PGK_OK;
}
// queue a wait for `e` on stream `s` (NULL: the current stream): everything enqueued on s afterwards runs after e
pgk_status pgk_stream_wait_event(pgk_stream s, pgk_event e) {
    PGK_REQUIRE(e, "pgk_stream_wait_event: null event");
    PGK_CHECK_HIP(hipStreamWaitEvent(resolve_stream(s), (hipEvent_t)e, 0));
    return PGK_OK;
}
pgk_status pgk_event_sync(pgk_event e) { PGK_CHECK_HIP(hipEventSynchronize((hipEvent_t)e)); return PGK_OK; }
pgk_status pgk_event_query(pgk_event e, int* done) {
    PGK_REQUIRE(done, "pgk_event_query: null output");
    hipError_t r = hipEventQuery((hipEvent_t)e);
    if (r == hipSuccess) { *done = 1; return PGK_OK; }
    if (r == hipErrorNotReady) { (void)hipGetLastError(); *done = 0; return PGK_OK; }
    return set_error(PGK_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(r));
}
pgk_status pgk_event_elapsed_ms(pgk_event start, pgk_event stop, float* ms) {
    PGK_REQUIRE(ms, "pgk_event_elapsed_ms: null output");
    PGK_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PGK_OK;
}

struct GraphObj {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t capture_id = 0;     // pool blocks allocated during the capture stay parked while this object lives
};

// give the blocks a destroyed (or failed) capture was holding back to their owners' lists
static void release_capture(uint64_t id) {
    if (!id) return;
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.parked.find(id);
    if (it == P.parked.end()) return;
    for (auto& pb : it->second)
        P.free_lists[{pb.second.dev, pb.second.cls, P.streams.count(pb.second.stream) ? pb.second.stream : nullptr}].push_back(pb.first);
    P.parked.erase(it);
}

pgk_status pgk_graph_begin_capture(pgk_stream s) {
    PGK_REQUIRE(g_capture_id == 0, "pgk_graph_begin_capture: this thread is already capturing");
    PGK_CHECK_HIP(hipStreamBeginCapture(resolve_stream(s), hipStreamCaptureModeRelaxed));
    Pool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    g_capture_id = P.next_capture_id++;
    P.parked[g_capture_id];      // exists <=> the capture / its graph is alive
    return PGK_OK;
}
pgk_status pgk_graph_end_capture(pgk_stream s, pgk_graph* out) {
    PGK_REQUIRE(out, "pgk_graph_end_capture: null output");
    GraphObj* g = new GraphObj();
    g->capture_id = g_capture_id;
    g_capture_id = 0;
    hipError_t e = hipStreamEndCapture(resolve_stream(s), &g->graph);
    if (e != hipSuccess || !g->graph) {
        release_capture(g->capture_id);
        delete g;
        return set_error(PGK_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    }
    e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g->graph);
        release_capture(g->capture_id);
        delete g;
        return set_error(PGK_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    *out = g;
    return PGK_OK;
}
pgk_status pgk_graph_launch(pgk_graph g, pgk_stream s) {
    PGK_REQUIRE(g, "pgk_graph_launch: null graph");
    PGK_CHECK_HIP(hipGraphLaunch(((GraphObj*)g)->exec, resolve_stream(s)));
    return PGK_OK;
}
pgk_status pgk_graph_num_nodes(pgk_graph g, size_t* n) {
    PGK_REQUIRE(g && n, "pgk_graph_num_nodes: null argument");
    PGK_CHECK_HIP(hipGraphGetNodes(((GraphObj*)g)->graph, nullptr, n));
    return PGK_OK;
}
pgk_status pgk_graph_destroy(pgk_graph g) {
    if (!g) return PGK_OK;
    GraphObj* o = (GraphObj*)g;
    if (o->exec) (void)hipGraphExecDestroy(o->exec);
    if (o->graph) (void)hipGraphDestroy(o->graph);
    release_capture(o->capture_id);
    delete o;
    return PGK_OK;
}
pgk_status pgk_stream_is_capturing(pgk_stream s, int* yes) {
    PGK_REQUIRE(yes, "pgk_stream_is_capturing: null output");
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    PGK_CHECK_HIP(hipStreamIsCapturing(resolve_stream(s), &cs));
    *yes = (cs != hipStreamCaptureStatusNone);
    return PGK_OK;
}

}  // extern "C"
