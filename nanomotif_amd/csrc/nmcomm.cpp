// Multi-GPU exchange step of the scoring path: sum of the per-rank int64 count tables (SURVEY.md §8(b), §8(e);
// reference: motif_model_bin accumulates over contigs, find_motifs_bin.py:1273-1283 — with contigs sharded over GPUs
// that sum becomes ONE all-reduce per scoring step).  RCCL is reached through dlopen, so the library itself has no
// link-time dependency on it and a single-GPU host needs no RCCL at all; in a process that already carries torch's
// RCCL (same SONAME) the loader hands back that copy.
//
// The collective runs on the ctx's own communication stream: nm_allreduce_counts_async orders it after the work
// queued so far on the scoring stream and returns; the scoring stream goes on with the next batch and is made to wait
// (nm_comm_wait, device side) only when it wants to overwrite or read that table again.  xGMI links are point-to-point
// and a 160 KB table is latency-bound: hiding the ring latency behind the next launch is the point of the two streams.
#include <dlfcn.h>

#include "nmscan_internal.h"

namespace {

typedef int ncclResult_t;
typedef void *ncclComm_t;
struct UniqueId { char internal[NM_COMM_ID_BYTES]; };

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(UniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, UniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;        // optional: what the communicator itself reports
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string why;
};

Rccl &rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
        r.why = dlerror();
    }
    if (!r.handle) return r;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.handle, "ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.handle, "ncclCommUserRank"));
    r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(dlsym(r.handle, "ncclCommCuDevice"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.handle, "ncclGetVersion"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
        r.why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
        r.handle = nullptr;
    }
    return r;
}

int need_rccl() {
    if (!rccl().handle) return fail(NM_ESTATE, "RCCL is not available (%s): multi-GPU count tables need librccl.so.1", rccl().why.c_str());
    return NM_OK;
}

const char *nccl_text(ncclResult_t e) { return rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error"; }

#define RCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t e_ = (expr);                                                                       \
        if (e_ != 0) return fail(NM_ESTATE, "%s failed: %s (%d)", #expr, nccl_text(e_), (int)e_);       \
    } while (0)

constexpr int NCCL_INT64 = 4, NCCL_SUM = 0;     // rccl.h: ncclInt64, ncclSum

}  // namespace

extern "C" {

int nm_comm_unique_id(uint8_t id[NM_COMM_ID_BYTES]) {
    if (!id) return fail(NM_EINVAL, "id is NULL");
    int rc = need_rccl();
    if (rc) return rc;
    UniqueId u;
    RCCL_TRY(rccl().GetUniqueId(&u));
    memcpy(id, u.internal, NM_COMM_ID_BYTES);
    return NM_OK;
}

int nm_comm_init(nm_ctx *c, int rank, int world, const uint8_t id[NM_COMM_ID_BYTES]) {
    if (!c || !id) return fail(NM_EINVAL, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(NM_EINVAL, "rank %d outside world of %d", rank, world);
    if (c->comm) return fail(NM_ESTATE, "this ctx already has a communicator (nm_comm_destroy first)");
    int rc = need_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    UniqueId u;
    memcpy(u.internal, id, NM_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    // one process per GPU: RCCL refuses two ranks of a communicator on one device (ncclInvalidUsage)
    RCCL_TRY(rccl().CommInitRank(&comm, world, u, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    if (!c->comm_stream) HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    for (auto &e : c->comm_done)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!c->comm_ready) HIP_TRY(hipEventCreateWithFlags(&c->comm_ready, hipEventDisableTiming));
    return NM_OK;
}

static int allreduce_after(nm_ctx *c, int64_t *d_counts, uint64_t n, int buffer_slot, hipStream_t producer) {
    if (!c || (n && !d_counts)) return fail(NM_EINVAL, "NULL argument");
    if (!c->comm) return fail(NM_ESTATE, "nm_comm_init has not been called on this ctx");
    if (buffer_slot < 0 || buffer_slot >= NM_COMM_SLOTS) return fail(NM_EINVAL, "buffer_slot %d outside 0..%d", buffer_slot, NM_COMM_SLOTS - 1);
    HIP_TRY(hipSetDevice(c->device));
    // the table is complete once everything queued so far on the scoring stream has run (with two scoring lanes:
    // on the lane of the most recent launch — the table the caller passes is that launch's)
    HIP_TRY(hipEventRecord(c->comm_ready, producer));
    HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->comm_ready, 0));
    if (n) RCCL_TRY(rccl().AllReduce(d_counts, d_counts, (size_t)n, NCCL_INT64, NCCL_SUM, static_cast<ncclComm_t>(c->comm), c->comm_stream));
    HIP_TRY(hipEventRecord(c->comm_done[buffer_slot], c->comm_stream));
    c->comm_pending[buffer_slot] = true;
    return NM_OK;
}

int nm_allreduce_counts_async(nm_ctx *c, int64_t *d_counts, uint64_t n, int buffer_slot) {
    if (!c) return fail(NM_EINVAL, "NULL argument");
    return allreduce_after(c, d_counts, n, buffer_slot, c->last_score_stream ? c->last_score_stream : c->stream);
}

int nm_comm_wait(nm_ctx *c, int buffer_slot) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (buffer_slot < 0 || buffer_slot >= NM_COMM_SLOTS) return fail(NM_EINVAL, "buffer_slot %d outside 0..%d", buffer_slot, NM_COMM_SLOTS - 1);
    if (!c->comm_pending[buffer_slot]) return NM_OK;
    HIP_TRY(hipStreamWaitEvent(c->stream, c->comm_done[buffer_slot], 0));
    if (c->lane_stream) HIP_TRY(hipStreamWaitEvent(c->lane_stream, c->comm_done[buffer_slot], 0));
    c->comm_pending[buffer_slot] = false;
    return NM_OK;
}

int nm_allreduce_counts(nm_ctx *c, int64_t *d_counts, uint64_t n) {
    int rc = nm_allreduce_counts_async(c, d_counts, n, 0);
    if (rc) return rc;
    return nm_comm_wait(c, 0);
}

int nm_allreduce_counts_host(nm_ctx *c, int64_t *counts, uint64_t n) {
    if (!c || (n && !counts)) return fail(NM_EINVAL, "NULL argument");
    if (!c->comm) return fail(NM_ESTATE, "nm_comm_init has not been called on this ctx");
    if (n == 0) return NM_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = nmdetail::ensure_stage(c, n * sizeof(int64_t));
    if (rc) return rc;
    memcpy(c->h_stage, counts, n * sizeof(int64_t));
    HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, n * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    rc = allreduce_after(c, static_cast<int64_t *>(c->d_stage), n, 0, c->stream);
    if (rc) return rc;
    rc = nm_comm_wait(c, 0);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->h_stage, c->d_stage, n * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    rc = nmdetail::release_stage(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(counts, c->h_stage, n * sizeof(int64_t));
    return NM_OK;
}

int nm_comm_info(nm_ctx *c, int32_t info[4]) {
    if (!c || !info) return fail(NM_EINVAL, "NULL argument");
    info[0] = info[3] = 0;
    info[1] = info[2] = -1;
    if (!c->comm) return NM_OK;
    // read back from the communicator, not from what nm_comm_init was told: the run reports the world RCCL really built
    ncclComm_t comm = static_cast<ncclComm_t>(c->comm);
    int v = 0;
    if (rccl().CommCount) {
        RCCL_TRY(rccl().CommCount(comm, &v));
        info[0] = v;
    } else {
        info[0] = c->comm_world;
    }
    if (rccl().CommUserRank) {
        RCCL_TRY(rccl().CommUserRank(comm, &v));
        info[1] = v;
    } else {
        info[1] = c->comm_rank;
    }
    if (rccl().CommCuDevice) {
        RCCL_TRY(rccl().CommCuDevice(comm, &v));
        info[2] = v;
    }
    v = 0;
    if (rccl().GetVersion) (void)rccl().GetVersion(&v);
    info[3] = v > 0 ? v : 1;
    return NM_OK;
}

int nm_comm_sync(nm_ctx *c) {
    if (!c) return fail(NM_EINVAL, "ctx is NULL");
    if (c->comm_stream) HIP_TRY(hipStreamSynchronize(c->comm_stream));
    return NM_OK;
}

int nm_comm_destroy(nm_ctx *c) {
    if (!c) return NM_OK;
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->comm && rccl().handle) (void)rccl().CommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    c->comm_world = 0;
    for (auto &e : c->comm_done) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    for (auto &p : c->comm_pending) p = false;
    if (c->comm_ready) (void)hipEventDestroy(c->comm_ready);
    c->comm_ready = nullptr;
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    c->comm_stream = nullptr;
    return NM_OK;
}

}  // extern "C"
