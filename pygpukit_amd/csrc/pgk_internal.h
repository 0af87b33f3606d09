// Host-side internals shared by every translation unit of libpgk_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "pgk_hip.h"

namespace pgk {

// Records the message for pgk_last_error() (thread-local) and returns `code`.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// NULL -> the calling thread's current stream (default: the device's library stream).
hipStream_t resolve_stream(pgk_stream s);

inline size_t dtype_size(pgk_dtype dt) {
    switch (dt) {
        case PGK_F64: case PGK_I64: return 8;
        case PGK_F32: case PGK_I32: return 4;
        case PGK_F16: case PGK_BF16: case PGK_I16: return 2;
        default: return 1;
    }
}

inline bool is_float_dtype(pgk_dtype dt) { return dt == PGK_F32 || dt == PGK_F16 || dt == PGK_BF16; }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int ceil_div(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace pgk

#define PGK_CHECK_HIP(expr)                                                                         \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return pgk::set_error(PGK_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                                  __FILE__, __LINE__);                                              \
    } while (0)

#define PGK_REQUIRE(cond, ...)                                         \
    do {                                                               \
        if (!(cond)) return pgk::set_error(PGK_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define PGK_LAUNCH_CHECK() PGK_CHECK_HIP(hipGetLastError())

// Dispatch a callable templated on the device element type for the three float dtypes.
#define PGK_DISPATCH_FLOAT(dt, NAME, ...)                                             \
    switch (dt) {                                                                     \
        case PGK_F32: { using T = float; __VA_ARGS__; } break;                        \
        case PGK_F16: { using T = pgk::f16; __VA_ARGS__; } break;                     \
        case PGK_BF16: { using T = pgk::bf16; __VA_ARGS__; } break;                   \
        default: return pgk::set_error(PGK_ERR_INVALID, NAME ": unsupported dtype %d", (int)(dt)); \
    }
