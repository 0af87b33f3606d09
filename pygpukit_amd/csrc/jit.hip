// Runtime compilation of user kernels: hiprtc -> code object -> hipModule -> launch.
//
// Replaces native/jit/{compiler,kernel}.hpp + nvrtc_loader (NVRTC -> PTX -> cuModuleLoadData -> cuLaunchKernel,
// bound in native/bindings/jit_bindings.cpp:65-122) with the HIP runtime-compilation path.  Like the reference's
// NVRTC loader, libhiprtc is opened lazily with dlopen, so the pre-compiled operators work on a machine without
// it and `pgk_jit_available` reports which is the case.  Result codes 0-11 are hiprtcResult (numerically identical to
// nvrtcResult, which the reference's NvrtcErrorCode mirrors); 1000+ are this layer's, as in the reference
// (src/pygpukit/jit/compiler.py:20-43).

#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <string>
#include <vector>

#include "pgk_internal.h"

namespace {

struct Rtc {
    void* lib = nullptr;
    std::string path;
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    decltype(&hiprtcVersion) version = nullptr;
    decltype(&hiprtcGetErrorString) err_string = nullptr;
    bool ok = false;
};

Rtc& rtc() {
    static Rtc r = [] {
        Rtc x;
        for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) { x.path = name; break; }
        }
        if (!x.lib) return x;
#define PGK_SYM(field, sym) x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.lib, #sym))
        PGK_SYM(create, hiprtcCreateProgram); PGK_SYM(compile, hiprtcCompileProgram); PGK_SYM(log_size, hiprtcGetProgramLogSize);
        PGK_SYM(log, hiprtcGetProgramLog); PGK_SYM(code_size, hiprtcGetCodeSize); PGK_SYM(code, hiprtcGetCode);
        PGK_SYM(destroy, hiprtcDestroyProgram); PGK_SYM(version, hiprtcVersion); PGK_SYM(err_string, hiprtcGetErrorString);
#undef PGK_SYM
        x.ok = x.create && x.compile && x.log_size && x.log && x.code_size && x.code && x.destroy && x.version;
        Dl_info info;
        if (x.ok && dladdr(reinterpret_cast<void*>(x.create), &info) && info.dli_fname) x.path = info.dli_fname;
        return x;
    }();
    return r;
}

struct Program {
    std::vector<char> code;   // the gfx950 code object (what PTX is to the reference)
    std::string log;
};

struct Kernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    std::string name;
};

}  // namespace

using namespace pgk;

extern "C" {

enum { PGK_JIT_NOT_LOADED = 1000, PGK_JIT_LOAD_FAILED = 1001, PGK_JIT_FUNCTION_NOT_FOUND = 1002, PGK_JIT_LAUNCH_FAILED = 1003 };

int pgk_jit_available(void) { return rtc().ok ? 1 : 0; }

const char* pgk_jit_library_path(void) { return rtc().ok ? rtc().path.c_str() : ""; }

pgk_status pgk_jit_version(int* major, int* minor) {
    PGK_REQUIRE(major && minor, "pgk_jit_version: null output");
    PGK_REQUIRE(rtc().ok, "pgk_jit_version: libhiprtc is not loadable");
    const hiprtcResult r = rtc().version(major, minor);
    PGK_REQUIRE(r == HIPRTC_SUCCESS, "hiprtcVersion failed (%d)", (int)r);
    return PGK_OK;
}

// Compile HIP C++ `source` for gfx950.  *rtc_code receives the hiprtcResult / 1000+ code (0 on success); the program
// handle is returned even when compilation fails, so the caller can read the log.
pgk_status pgk_jit_compile(const char* source, const char* name, const char* const* options, int n_options, void** program_out,
                           int* rtc_code) {
    PGK_REQUIRE(source && program_out && rtc_code, "pgk_jit_compile: null argument");
    *program_out = nullptr;
    if (!rtc().ok) {
        *rtc_code = PGK_JIT_NOT_LOADED;
        return set_error(PGK_ERR_UNSUPPORTED, "pgk_jit_compile: libhiprtc is not loadable on this machine");
    }
    hiprtcProgram prog = nullptr;
    hiprtcResult r = rtc().create(&prog, source, name ? name : "kernel.hip", 0, nullptr, nullptr);
    if (r != HIPRTC_SUCCESS) {
        *rtc_code = (int)r;
        return set_error(PGK_ERR_JIT, "hiprtcCreateProgram failed: %s", rtc().err_string ? rtc().err_string(r) : "?");
    }
    std::vector<const char*> opts;
    bool has_arch = false;
    for (int i = 0; i < n_options; ++i) {
        if (!options[i]) continue;
        const std::string o = options[i];
        // the reference's callers pass NVRTC architecture flags; there is one target here
        if (o.rfind("-arch=", 0) == 0 || o.rfind("--gpu-architecture", 0) == 0 || o.rfind("-gencode", 0) == 0) continue;
        if (o.rfind("--offload-arch", 0) == 0) has_arch = true;
        opts.push_back(options[i]);
    }
    if (!has_arch) opts.push_back("--offload-arch=gfx950");
    r = rtc().compile(prog, (int)opts.size(), opts.data());
    Program* p = new Program();
    size_t n = 0;
    if (rtc().log_size(prog, &n) == HIPRTC_SUCCESS && n > 1) {
        p->log.resize(n);
        rtc().log(prog, p->log.data());
        while (!p->log.empty() && p->log.back() == '\0') p->log.pop_back();
    }
    *rtc_code = (int)r;
    if (r == HIPRTC_SUCCESS) {
        size_t cs = 0;
        if (rtc().code_size(prog, &cs) == HIPRTC_SUCCESS && cs) {
            p->code.resize(cs);
            rtc().code(prog, p->code.data());
        }
    }
    rtc().destroy(&prog);
    *program_out = p;
    if (r != HIPRTC_SUCCESS)
        return set_error(PGK_ERR_JIT, "hiprtc compilation failed (%s): %.600s", rtc().err_string ? rtc().err_string(r) : "?", p->log.c_str());
    return PGK_OK;
}

const char* pgk_jit_program_log(void* program) { return program ? ((Program*)program)->log.c_str() : ""; }

pgk_status pgk_jit_program_code(void* program, const void** code, size_t* size) {
    PGK_REQUIRE(program && code && size, "pgk_jit_program_code: null argument");
    Program* p = (Program*)program;
    *code = p->code.data();
    *size = p->code.size();
    return PGK_OK;
}

void pgk_jit_program_destroy(void* program) { delete (Program*)program; }

pgk_status pgk_jit_kernel_create(void* program, const char* func_name, void** kernel_out, int* rtc_code) {
    PGK_REQUIRE(program && func_name && kernel_out && rtc_code, "pgk_jit_kernel_create: null argument");
    Program* p = (Program*)program;
    *kernel_out = nullptr;
    PGK_REQUIRE(!p->code.empty(), "pgk_jit_kernel_create: the program did not compile");
    Kernel* k = new Kernel();
    k->name = func_name;
    hipError_t e = hipModuleLoadData(&k->module, p->code.data());
    if (e != hipSuccess) {
        (void)hipGetLastError();   // do not leave the failure in the runtime's sticky last-error slot
        delete k;
        *rtc_code = PGK_JIT_LOAD_FAILED;
        return set_error(PGK_ERR_JIT, "hipModuleLoadData failed: %s", hipGetErrorString(e));
    }
    e = hipModuleGetFunction(&k->fn, k->module, func_name);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // (a later kernel-launch check would otherwise report this lookup failure)
        (void)hipModuleUnload(k->module);
        delete k;
        *rtc_code = PGK_JIT_FUNCTION_NOT_FOUND;
        return set_error(PGK_ERR_INVALID, "kernel '%s' not found in the compiled module (declare it extern \"C\"): %s", func_name,
                         hipGetErrorString(e));
    }
    *rtc_code = 0;
    *kernel_out = k;
    return PGK_OK;
}

void pgk_jit_kernel_destroy(void* kernel) {
    Kernel* k = (Kernel*)kernel;
    if (!k) return;
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
}

pgk_status pgk_jit_suggested_block_size(void* kernel, size_t dynamic_smem, int* block_size) {
    PGK_REQUIRE(kernel && block_size, "pgk_jit_suggested_block_size: null argument");
    int grid = 0, block = 0;
    PGK_CHECK_HIP(hipModuleOccupancyMaxPotentialBlockSize(&grid, &block, ((Kernel*)kernel)->fn, dynamic_smem, 0));
    *block_size = block;
    return PGK_OK;
}

// args: array of n pointers, each to one kernel argument's value (the cuLaunchKernel kernelParams convention)
pgk_status pgk_jit_launch(void* kernel, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                          unsigned shared_bytes, void** args, pgk_stream s) {
    PGK_REQUIRE(kernel, "pgk_jit_launch: null kernel");
    PGK_REQUIRE(gx && gy && gz && bx && by && bz && (unsigned long long)bx * by * bz <= 1024, "pgk_jit_launch: bad launch shape");
    const hipError_t e = hipModuleLaunchKernel(((Kernel*)kernel)->fn, gx, gy, gz, bx, by, bz, shared_bytes, resolve_stream(s), args, nullptr);
    if (e != hipSuccess) (void)hipGetLastError();
    if (e != hipSuccess) return set_error(PGK_ERR_JIT, "hipModuleLaunchKernel(%s) failed: %s", ((Kernel*)kernel)->name.c_str(), hipGetErrorString(e));
    return PGK_OK;
}

}  // extern "C"
