// Caption-batch sharding and the ONE exchange of the path: an all-gather of the generated token ids (RCCL over xGMI,
// librccl dlopen'ed on first use so single-GPU hosts need no RCCL).
#include <dlfcn.h>

#include "context.h"

using namespace capdec;

// ---- RCCL (dlopen'ed): only the five entry points the path needs
namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
};
// (lib: CAPDEC_RCCL_LIB of the first context that needs the library; the process loads it once)
Rccl *rccl(const char *lib = nullptr) {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {lib, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (r.h) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
            r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
            r.CommCount = (decltype(r.CommCount))dlsym(r.h, "ncclCommCount");
            r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.h, "ncclCommUserRank");
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) r.h = nullptr;
        }
    }
    return r.h ? &r : nullptr;
}
#define CAPDEC_NCCL(expr)                                                                               \
    do {                                                                                                \
        ncclResult_t _r = (expr);                                                                       \
        if (_r != ncclSuccess) {                                                                        \
            capdec::set_error(std::string(#expr) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(_r) : "RCCL error")); \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)
}  // namespace


namespace capdec {
void comm_release(capdec_ctx *c) {
    if (c->comm && rccl()) { (void)rccl()->CommDestroy(c->comm); c->comm = nullptr; }
}
}  // namespace capdec

extern "C" {

int capdec_comm_unique_id(char *id) {
    static_assert(CAPDEC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    CAPDEC_CHECK(id != nullptr, "comm_unique_id: null id");
    const char *lib = getenv("CAPDEC_RCCL_LIB");      // (no context here: the one knob read outside capdec_create)
    CAPDEC_CHECK(rccl(lib && *lib ? lib : nullptr) != nullptr, "comm: librccl.so.1 could not be loaded (set CAPDEC_RCCL_LIB)");
    ncclUniqueId u;
    CAPDEC_NCCL(rccl()->GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int capdec_comm_init(capdec_ctx *c, int rank, int nranks, const char *id) {
    CAPDEC_CHECK(c && id && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad argument");
    CAPDEC_CHECK(c->comm == nullptr, "comm_init: this context already has a communicator");
    CAPDEC_CHECK(rccl(c->tune.rccl_lib.empty() ? nullptr : c->tune.rccl_lib.c_str()) != nullptr,
                 "comm: librccl.so.1 could not be loaded (set CAPDEC_RCCL_LIB)");
    CAPDEC_HIP(hipSetDevice(c->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    CAPDEC_NCCL(rccl()->CommInitRank(&c->comm, nranks, u, rank));
    c->comm_rank = rank;
    c->comm_world = nranks;
    return 0;
}

int capdec_comm_destroy(capdec_ctx *c) {
    CAPDEC_CHECK(c, "null context");
    if (c->comm) {
        (void)hipStreamSynchronize(c->stream);
        CAPDEC_NCCL(rccl()->CommDestroy(c->comm));
        c->comm = nullptr;
    }
    c->comm_rank = 0;
    c->comm_world = 1;
    return 0;
}

int capdec_comm_info(capdec_ctx *c, int *rank, int *nranks) {
    CAPDEC_CHECK(c && rank && nranks, "comm_info: null argument");
    *rank = 0;
    *nranks = 1;
    if (c->comm == nullptr) return 0;                       // no communicator: a single-GPU context
    CAPDEC_CHECK(rccl() && rccl()->CommCount && rccl()->CommUserRank, "comm_info: librccl lacks ncclCommCount / ncclCommUserRank");
    CAPDEC_NCCL(rccl()->CommCount(c->comm, nranks));        // what RCCL itself says, not what the host passed in
    CAPDEC_NCCL(rccl()->CommUserRank(c->comm, rank));
    return 0;
}

int capdec_shard_bounds(int n_total, int rank, int nranks, int *lo, int *hi) {
    CAPDEC_CHECK(lo && hi && n_total >= 0 && nranks >= 1 && rank >= 0 && rank < nranks, "shard_bounds: bad argument");
    const int per = (n_total + nranks - 1) / nranks;
    *lo = std::min(rank * per, n_total);
    *hi = std::min(*lo + per, n_total);
    return 0;
}

int capdec_gather_rows(capdec_ctx *c, const void *d_local, int n_local, int row_elems, int n_total, void *d_global) {
    CAPDEC_CHECK(c && n_local >= 0 && row_elems >= 1 && n_total >= 0 && (n_total == 0 || d_global) &&
                     (n_local == 0 || d_local), "gather_rows: bad argument");
    CAPDEC_HIP(hipSetDevice(c->device));
    const size_t row_b = (size_t)row_elems * 4;
    if (c->comm == nullptr) {
        CAPDEC_CHECK(n_local == n_total, "gather_rows: this rank holds only part of the rows but the context has no "
                                         "communicator (capdec_comm_init)");
        if (n_total && d_local != d_global)
            CAPDEC_HIP(hipMemcpyAsync(d_global, d_local, (size_t)n_total * row_b, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    int lo = 0, hi = 0;
    CAPDEC_TRY(capdec_shard_bounds(n_total, c->comm_rank, c->comm_world, &lo, &hi));
    CAPDEC_CHECK(n_local == hi - lo, "gather_rows: n_local is not this rank's shard of n_total (capdec_shard_bounds)");
    if (n_total == 0) return 0;
    const int per = (n_total + c->comm_world - 1) / c->comm_world;
    CAPDEC_TRY(c->g_pad.ensure((size_t)per * row_b));
    CAPDEC_TRY(c->g_all.ensure((size_t)per * c->comm_world * row_b));
    CAPDEC_HIP(hipMemsetAsync(c->g_pad.p, 0, (size_t)per * row_b, c->stream));
    if (n_local)
        CAPDEC_HIP(hipMemcpyAsync(c->g_pad.p, d_local, (size_t)n_local * row_b, hipMemcpyDeviceToDevice, c->stream));
    CAPDEC_NCCL(rccl()->AllGather(c->g_pad.p, c->g_all.p, (size_t)per * row_elems, ncclInt32, c->comm, c->stream));
    CAPDEC_HIP(hipMemcpyAsync(d_global, c->g_all.p, (size_t)n_total * row_b, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int capdec_gather_ids(capdec_ctx *c, const int32_t *d_ids, const int32_t *d_lens, const float *d_scores, int n_local,
                      int T, int n_total, int32_t *d_ids_global, int32_t *d_lens_global, float *d_scores_global) {
    CAPDEC_CHECK(c && T >= 1, "gather_ids: bad argument");
    CAPDEC_TRY(capdec_gather_rows(c, d_ids, n_local, T, n_total, d_ids_global));
    CAPDEC_TRY(capdec_gather_rows(c, d_lens, n_local, 1, n_total, d_lens_global));
    if (d_scores && d_scores_global) CAPDEC_TRY(capdec_gather_rows(c, d_scores, n_local, 1, n_total, d_scores_global));
    CAPDEC_HIP(hipStreamSynchronize(c->stream));
    return 0;
}


}  // extern "C"
