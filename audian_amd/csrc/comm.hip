// Multi-GPU exchange step of libhip_dsp: the all-gather of per-rank spectrogram tiles
// (channel-sharded) as plain C entry points over RCCL.
//
// RCCL is bound at run time (dlopen) so that single-GPU hosts do not need it and so that a
// process that already carries a copy (PyTorch bundles one under the same SONAME) keeps
// exactly one.  Only the four calls used here are resolved.
#include "common.h"
#include <dlfcn.h>

namespace {

struct NcclUniqueId { char internal[128]; };
typedef void *NcclComm;
enum { NCCL_FLOAT32 = 7 };

struct Rccl {
    void *handle;
    int (*GetUniqueId)(NcclUniqueId *);
    int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int);
    int (*CommDestroy)(NcclComm);
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t);
    const char *(*GetErrorString)(int);
};

Rccl g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

int load_rccl()
{
    if (g_rccl.handle) return HIPDSP_OK;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        hipdsp_set_error("cannot load RCCL (librccl.so.1): %s", dlerror());
        return HIPDSP_ERR_UNSUPPORTED;
    }
    g_rccl.GetUniqueId = (int (*)(NcclUniqueId *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(NcclComm *, int, NcclUniqueId, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(NcclComm))dlsym(h, "ncclCommDestroy");
    g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, NcclComm, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) {
        hipdsp_set_error("librccl lacks an expected symbol");
        dlclose(h);
        return HIPDSP_ERR_UNSUPPORTED;
    }
    g_rccl.handle = h;
    return HIPDSP_OK;
}

int nccl_fail(const char *what, int rc)
{
    hipdsp_set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return HIPDSP_ERR_HIP;
}

}  // namespace

struct hipdsp_comm {
    NcclComm comm;
    int rank, nranks;
};

extern "C" {

int hipdsp_comm_unique_id(void *id_out)
{
    HD_REQUIRE(id_out != nullptr, "id_out is NULL");
    int rc = load_rccl();
    if (rc != HIPDSP_OK) return rc;
    NcclUniqueId id;
    int n = g_rccl.GetUniqueId(&id);
    if (n != 0) return nccl_fail("ncclGetUniqueId", n);
    memcpy(id_out, &id, sizeof(id));
    return HIPDSP_OK;
}

int hipdsp_comm_create(hipdsp_ctx *ctx, const void *unique_id, int rank, int nranks, hipdsp_comm **out)
{
    HD_REQUIRE(ctx != nullptr && unique_id != nullptr && out != nullptr, "NULL argument");
    HD_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    *out = nullptr;
    int rc = load_rccl();
    if (rc != HIPDSP_OK) return rc;
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    NcclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    NcclComm c = nullptr;
    int n = g_rccl.CommInitRank(&c, nranks, id, rank);
    if (n != 0) return nccl_fail("ncclCommInitRank", n);
    hipdsp_comm *h = new hipdsp_comm();
    h->comm = c; h->rank = rank; h->nranks = nranks;
    *out = h;
    return HIPDSP_OK;
}

int hipdsp_comm_destroy(hipdsp_ctx *ctx, hipdsp_comm *comm)
{
    HD_REQUIRE(ctx != nullptr, "ctx is NULL");
    if (comm) {
        (void)hipStreamSynchronize(ctx->stream);
        if (g_rccl.handle) (void)g_rccl.CommDestroy(comm->comm);
        delete comm;
    }
    return HIPDSP_OK;
}

int hipdsp_allgather_f32(hipdsp_ctx *ctx, hipdsp_comm *comm, const float *send, float *recv,
                         int64_t count_per_rank)
{
    HD_REQUIRE(ctx != nullptr && comm != nullptr, "NULL argument");
    HD_REQUIRE(count_per_rank >= 0, "negative count");
    if (count_per_rank == 0) return HIPDSP_OK;
    HD_REQUIRE(send != nullptr && recv != nullptr, "NULL data pointer");
    HD_CHECK_HIP(hipSetDevice(ctx->device));
    int n = g_rccl.AllGather(send, recv, (size_t)count_per_rank, NCCL_FLOAT32, comm->comm, ctx->stream);
    if (n != 0) return nccl_fail("ncclAllGather", n);
    return HIPDSP_OK;
}

}  // extern "C"
