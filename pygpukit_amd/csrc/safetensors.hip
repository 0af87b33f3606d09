// Memory-mapped .safetensors reader.
//
// Replaces rust/pygpukit-core/src/llm/tensor_loader.rs as used by src/pygpukit/llm/safetensors.py:122-235
// (SafeTensorsFile: names, info, bytes, data pointer) and the direct file -> device upload of
// src/pygpukit/llm/loader.py:160-175 (memcpy_ptr_to_device).  Format: u64 little-endian header length, a JSON object
// {"tensor": {"dtype": "BF16", "shape": [...], "data_offsets": [begin, end]}, ..., "__metadata__": {...}}, then the
// raw data; offsets are relative to the end of the header.  The file is mapped read-only; nothing is copied until a
// tensor is uploaded.

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "pgk_internal.h"

namespace {

struct TensorRec {
    std::string name;
    int dtype = -1;            // src/pygpukit/llm/safetensors.py:28-43 Dtype ids
    std::vector<int64_t> shape;
    uint64_t begin = 0, end = 0;
};

struct StFile {
    int fd = -1;
    const uint8_t* map = nullptr;
    size_t size = 0, data_start = 0;
    std::vector<TensorRec> tensors;
    std::map<std::string, size_t> index;
};

// bytes per element for the Dtype ids below
size_t dtype_bytes(int id) {
    static const size_t b[12] = {4, 2, 2, 8, 1, 1, 4, 8, 2, 1, 1, 1};
    return id >= 0 && id < 12 ? b[id] : 0;
}

int dtype_id(const std::string& s) {
    static const std::map<std::string, int> m = {{"F32", 0}, {"F16", 1}, {"BF16", 2}, {"F64", 3}, {"F8_E4M3", 4}, {"F8_E5M2", 5},
                                                  {"I32", 6}, {"I64", 7}, {"I16", 8}, {"I8", 9}, {"U8", 10}, {"BOOL", 11}};
    const auto it = m.find(s);
    return it == m.end() ? -1 : it->second;
}

// minimal JSON walker for the header's fixed schema
struct Json {
    const char* p;
    const char* e;
    bool fail = false;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
    std::string str() {
        ws();
        std::string out;
        if (p >= e || *p != '"') { fail = true; return out; }
        ++p;
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                ++p;
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'u': out += '?'; p += 4; break;   // names are ASCII in practice; keep the walker in sync
                    default: out += *p;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= e) { fail = true; return out; }
        ++p;
        return out;
    }
    // non-negative integer below 2^62 (shape dimensions, byte offsets): a sign, a fraction, an exponent or more digits
    // than that fail the parse - the header comes from an untrusted file
    int64_t num() {
        ws();
        if (p >= e || *p < '0' || *p > '9') { fail = true; return 0; }
        int64_t v = 0;
        while (p < e && *p >= '0' && *p <= '9') {
            if (v > ((int64_t)1 << 62) / 10 - 1) { fail = true; return 0; }
            v = v * 10 + (*p++ - '0');
        }
        if (p < e && (*p == '.' || *p == 'e' || *p == 'E')) { fail = true; return 0; }
        return v;
    }
    void skip_value() {   // any JSON value (used for __metadata__ and unknown keys)
        ws();
        if (p >= e) { fail = true; return; }
        if (*p == '"') { str(); return; }
        if (*p == '{' || *p == '[') {
            const char open = *p, close = open == '{' ? '}' : ']';
            ++p;
            ws();
            if (eat(close)) return;
            do {
                if (open == '{') { str(); if (!eat(':')) { fail = true; return; } }
                skip_value();
                if (fail) return;
            } while (eat(','));
            if (!eat(close)) fail = true;
            return;
        }
        while (p < e && *p != ',' && *p != '}' && *p != ']') ++p;   // number / true / false / null
    }
};

}  // namespace

using namespace pgk;

extern "C" {

pgk_status pgk_st_open(const char* path, void** handle) {
    PGK_REQUIRE(path && handle, "pgk_st_open: null argument");
    *handle = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return set_error(PGK_ERR_INVALID, "pgk_st_open: cannot open '%s': %s", path, strerror(errno));
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 8) {
        close(fd);
        return set_error(PGK_ERR_INVALID, "pgk_st_open: '%s' is not a safetensors file (size %lld)", path, (long long)sb.st_size);
    }
    void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
        close(fd);
        return set_error(PGK_ERR_INVALID, "pgk_st_open: mmap of '%s' failed: %s", path, strerror(errno));
    }
    StFile* f = new StFile();
    f->fd = fd; f->map = (const uint8_t*)m; f->size = (size_t)sb.st_size;
    uint64_t hlen = 0;
    memcpy(&hlen, f->map, 8);
    auto bail = [&](const char* why) {
        munmap((void*)f->map, f->size); close(f->fd); delete f;
        return set_error(PGK_ERR_INVALID, "pgk_st_open: '%s': %s", path, why);
    };
    if (hlen == 0 || hlen > f->size - 8 || hlen > (100ull << 20)) return bail("header length out of range");
    f->data_start = 8 + (size_t)hlen;
    Json j{(const char*)f->map + 8, (const char*)f->map + 8 + hlen};
    if (!j.eat('{')) return bail("header is not a JSON object");
    j.ws();
    if (!j.eat('}')) {
        do {
            const std::string name = j.str();
            if (j.fail || !j.eat(':')) return bail("malformed header (key)");
            if (name == "__metadata__") { j.skip_value(); if (j.fail) return bail("malformed __metadata__"); continue; }
            TensorRec t;
            t.name = name;
            if (!j.eat('{')) return bail("malformed header (tensor entry)");
            do {
                const std::string key = j.str();
                if (j.fail || !j.eat(':')) return bail("malformed header (field)");
                if (key == "dtype") {
                    t.dtype = dtype_id(j.str());
                } else if (key == "shape") {
                    if (!j.eat('[')) return bail("malformed shape");
                    j.ws();
                    if (!j.eat(']')) {
                        do { t.shape.push_back(j.num()); } while (j.eat(','));
                        if (!j.eat(']')) return bail("malformed shape");
                    }
                } else if (key == "data_offsets") {
                    if (!j.eat('[')) return bail("malformed data_offsets");
                    t.begin = (uint64_t)j.num();
                    if (!j.eat(',')) return bail("malformed data_offsets");
                    t.end = (uint64_t)j.num();
                    if (!j.eat(']')) return bail("malformed data_offsets");
                } else {
                    j.skip_value();
                }
                if (j.fail) return bail("malformed header (value)");
            } while (j.eat(','));
            if (!j.eat('}')) return bail("malformed header (tensor entry end)");
            if (t.dtype < 0) return bail("unknown dtype string");
            // subtraction form: no sum that could wrap (the reference's Rust reader rejects the same inputs)
            if (t.end < t.begin || t.end > f->size - f->data_start) return bail("tensor data outside the file");
            if (t.shape.size() > 8) return bail("tensor rank above 8");
            {
                uint64_t elems = 1;
                for (int64_t d : t.shape) {
                    if (d < 0) return bail("negative dimension");
                    if (d != 0 && elems > (uint64_t)f->size / (uint64_t)d + 1) { elems = ~0ull; break; }   // already larger than the file
                    elems *= (uint64_t)d;
                }
                if (elems == ~0ull || elems > (uint64_t)f->size || elems * dtype_bytes(t.dtype) != t.end - t.begin)
                    return bail("shape x dtype size does not match data_offsets");
            }
            f->index[t.name] = f->tensors.size();
            f->tensors.push_back(std::move(t));
        } while (j.eat(','));
        if (!j.eat('}')) return bail("malformed header (end)");
    }
    *handle = f;
    return PGK_OK;
}

void pgk_st_close(void* handle) {
    StFile* f = (StFile*)handle;
    if (!f) return;
    if (f->map) munmap((void*)f->map, f->size);
    if (f->fd >= 0) close(f->fd);
    delete f;
}

int pgk_st_num_tensors(void* handle) { return handle ? (int)((StFile*)handle)->tensors.size() : 0; }
uint64_t pgk_st_file_size(void* handle) { return handle ? ((StFile*)handle)->size : 0; }
const char* pgk_st_tensor_name(void* handle, int i) {
    StFile* f = (StFile*)handle;
    return f && i >= 0 && i < (int)f->tensors.size() ? f->tensors[i].name.c_str() : "";
}

pgk_status pgk_st_tensor_info(void* handle, const char* name, int* dtype, int* ndim, int64_t* shape8, uint64_t* offset, uint64_t* nbytes) {
    PGK_REQUIRE(handle && name && dtype && ndim && shape8 && offset && nbytes, "pgk_st_tensor_info: null argument");
    StFile* f = (StFile*)handle;
    const auto it = f->index.find(name);
    if (it == f->index.end()) return set_error(PGK_ERR_INVALID, "tensor '%s' not found", name);
    const TensorRec& t = f->tensors[it->second];
    *dtype = t.dtype;
    *ndim = (int)t.shape.size();
    for (size_t i = 0; i < t.shape.size(); ++i) shape8[i] = t.shape[i];
    *offset = f->data_start + t.begin;
    *nbytes = t.end - t.begin;
    return PGK_OK;
}

pgk_status pgk_st_tensor_data(void* handle, const char* name, const void** ptr, uint64_t* nbytes) {
    PGK_REQUIRE(handle && name && ptr && nbytes, "pgk_st_tensor_data: null argument");
    StFile* f = (StFile*)handle;
    const auto it = f->index.find(name);
    if (it == f->index.end()) return set_error(PGK_ERR_INVALID, "tensor '%s' not found", name);
    const TensorRec& t = f->tensors[it->second];
    *ptr = f->map + f->data_start + t.begin;
    *nbytes = t.end - t.begin;
    return PGK_OK;
}

// mapped file -> device, no intermediate host copy (loader.py:160-175 memcpy_ptr_to_device)
pgk_status pgk_st_upload(void* handle, const char* name, void* dst_device, uint64_t dst_bytes, pgk_stream s) {
    const void* src = nullptr;
    uint64_t n = 0;
    if (pgk_status r = pgk_st_tensor_data(handle, name, &src, &n)) return r;
    PGK_REQUIRE(dst_device && dst_bytes == n, "pgk_st_upload: '%s' holds %llu bytes, destination %llu", name, (unsigned long long)n,
                (unsigned long long)dst_bytes);
    hipStream_t st = resolve_stream(s);
    PGK_CHECK_HIP(hipMemcpyAsync(dst_device, src, n, hipMemcpyHostToDevice, st));
    PGK_CHECK_HIP(hipStreamSynchronize(st));   // the source is pageable file memory: complete before returning
    return PGK_OK;
}

}  // extern "C"
