// fp8 x fp8 MFMA GEMM (config 5: Llama-3-8B-shape prefill).  Placeholder entry points: they fail
// loudly until the kernel lands (no silent fallback to another precision).
#include "pgk_internal.h"

using namespace pgk;

extern "C" {

pgk_status pgk_gemm_fp8_nt(const uint8_t*, const float*, const uint8_t*, const void*, void*, int, int, int, pgk_stream) {
    return set_error(PGK_ERR_UNSUPPORTED, "pgk_gemm_fp8_nt: not implemented in this build");
}

pgk_status pgk_quantize_fp8_rows(const void*, uint8_t*, float*, int, int, pgk_stream) {
    return set_error(PGK_ERR_UNSUPPORTED, "pgk_quantize_fp8_rows: not implemented in this build");
}

}  // extern "C"
