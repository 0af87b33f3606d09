// RCCL communicator wrappers (data-parallel batch decode over one 8xMI355X node).
// One process per GPU; the only traffic is the one-time weight broadcast and the per-step gather of
// sampled tokens / logits.  No counterpart in the reference (single GPU).

#include <rccl/rccl.h>

#include <cstring>

#include "pgk_internal.h"

#define PGK_CHECK_NCCL(expr)                                                                          \
    do {                                                                                              \
        ncclResult_t r_ = (expr);                                                                     \
        if (r_ != ncclSuccess)                                                                        \
            return pgk::set_error(PGK_ERR_RCCL, "%s: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

using namespace pgk;

struct CommObj {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    void* scratch = nullptr;  // one double for barriers
};

extern "C" {

pgk_status pgk_comm_unique_id(char* h_id128) {
    PGK_REQUIRE(h_id128, "pgk_comm_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    PGK_CHECK_NCCL(ncclGetUniqueId(&id));
    memcpy(h_id128, &id, 128);
    return PGK_OK;
}

pgk_status pgk_comm_init(pgk_comm* out, const char* h_id128, int rank, int world) {
    PGK_REQUIRE(out && h_id128, "pgk_comm_init: null argument");
    PGK_REQUIRE(world >= 1 && rank >= 0 && rank < world, "pgk_comm_init: rank %d / world %d", rank, world);
    ncclUniqueId id;
    memcpy(&id, h_id128, 128);
    CommObj* c = new CommObj();
    c->rank = rank;
    c->world = world;
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return set_error(PGK_ERR_RCCL, "ncclCommInitRank: %s", ncclGetErrorString(r));
    }
    if (pgk_status s = pgk_malloc(&c->scratch, 64)) { ncclCommDestroy(c->comm); delete c; return s; }
    *out = c;
    return PGK_OK;
}

pgk_status pgk_comm_destroy(pgk_comm ch) {
    if (!ch) return PGK_OK;
    CommObj* c = (CommObj*)ch;
    if (c->scratch) pgk_free(c->scratch);
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
    return PGK_OK;
}

pgk_status pgk_comm_broadcast(pgk_comm ch, void* buf, size_t nbytes, int root, pgk_stream s) {
    PGK_REQUIRE(ch && buf, "pgk_comm_broadcast: null argument");
    CommObj* c = (CommObj*)ch;
    PGK_CHECK_NCCL(ncclBroadcast(buf, buf, nbytes, ncclChar, root, c->comm, resolve_stream(s)));
    return PGK_OK;
}

pgk_status pgk_comm_all_gather(pgk_comm ch, const void* send, void* recv, size_t nbytes_per_rank, pgk_stream s) {
    PGK_REQUIRE(ch && send && recv, "pgk_comm_all_gather: null argument");
    CommObj* c = (CommObj*)ch;
    PGK_CHECK_NCCL(ncclAllGather(send, recv, nbytes_per_rank, ncclChar, c->comm, resolve_stream(s)));
    return PGK_OK;
}

pgk_status pgk_comm_all_reduce_max_f64(pgk_comm ch, double* buf, int n, pgk_stream s) {
    PGK_REQUIRE(ch && buf, "pgk_comm_all_reduce_max_f64: null argument");
    CommObj* c = (CommObj*)ch;
    PGK_CHECK_NCCL(ncclAllReduce(buf, buf, (size_t)n, ncclDouble, ncclMax, c->comm, resolve_stream(s)));
    return PGK_OK;
}

pgk_status pgk_comm_barrier(pgk_comm ch, pgk_stream s) {
    PGK_REQUIRE(ch, "pgk_comm_barrier: null communicator");
    CommObj* c = (CommObj*)ch;
    hipStream_t st = resolve_stream(s);
    PGK_CHECK_HIP(hipMemsetAsync(c->scratch, 0, 8, st));
    PGK_CHECK_NCCL(ncclAllReduce(c->scratch, c->scratch, 1, ncclDouble, ncclSum, c->comm, st));
    PGK_CHECK_HIP(hipStreamSynchronize(st));
    return PGK_OK;
}

}  // extern "C"
