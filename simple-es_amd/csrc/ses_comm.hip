// ses_comm.hip -- the one exchange step of a multi-GPU generation: the all-gather of the per-offspring fitness
// over RCCL (xGMI inside a node).  Replaces the gather half of `results = p.map(RolloutWorker, arguments)`
// (learning_strategies/evolution/loop.py:66-79) when the population is sharded over one process per GPU.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the SONAME of both /opt/rocm's and the PyTorch wheel's copy:
// inside a process that has torch loaded the already-mapped library is reused, so there is never a second RCCL in
// the address space).  A single-GPU user never loads it.  Only the six entry points below are used; their C
// signatures are part of NCCL's stable API (ncclUniqueId = 128 opaque bytes passed by value, ncclFloat32 = 7).
#include <dlfcn.h>

#include <cstring>

#include "ses_internal.h"

namespace ses {

struct NcclId {
    char internal[SES_COMM_ID_BYTES];
};
typedef void *NcclComm;
constexpr int NCCL_FLOAT32 = 7;

struct Rccl {
    void *lib;
    int (*GetVersion)(int *);
    int (*GetUniqueId)(NcclId *);
    int (*CommInitRank)(NcclComm *, int, NcclId, int);
    int (*CommDestroy)(NcclComm);
    int (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t);
    const char *(*GetErrorString)(int);
};

static Rccl g_rccl;

static int rccl_error(const char *what, int rc)
{
    return set_error(SES_ERR_COMM, "%s failed: %s (rccl code %d)", what,
                     g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", rc);
}

static int load_rccl()
{
    if (g_rccl.lib) return SES_OK;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return set_error(SES_ERR_COMM, "librccl.so.1 not loadable: %s", dlerror());
    Rccl r;
    std::memset(&r, 0, sizeof r);
#define SES_RCCL_SYM(field, name)                                                                   \
    do {                                                                                            \
        *(void **)(&r.field) = dlsym(lib, name);                                                    \
        if (!r.field) return set_error(SES_ERR_COMM, "librccl.so.1 does not export %s", name);     \
    } while (0)
    SES_RCCL_SYM(GetVersion, "ncclGetVersion");
    SES_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    SES_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    SES_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    SES_RCCL_SYM(AllGather, "ncclAllGather");
    SES_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef SES_RCCL_SYM
    r.lib = lib;
    g_rccl = r;
    return SES_OK;
}

int comm_release(ses_handle *h)
{
    if (h->comm && g_rccl.CommDestroy) {
        (void)hipStreamSynchronize(h->stream);
        (void)g_rccl.CommDestroy((NcclComm)h->comm);
    }
    h->comm = nullptr;
    h->comm_world = 0;
    h->comm_rank = 0;
    return SES_OK;
}

}  // namespace ses

extern "C" {

int ses_comm_unique_id(void *id)
{
    using namespace ses;
    SES_REQUIRE(id, "ses_comm_unique_id: null argument");
    int rc = load_rccl();
    if (rc != SES_OK) return rc;
    NcclId nid;
    int nrc = g_rccl.GetUniqueId(&nid);
    if (nrc != 0) return rccl_error("ncclGetUniqueId", nrc);
    std::memcpy(id, nid.internal, SES_COMM_ID_BYTES);
    return SES_OK;
}

int ses_comm_init(ses_handle *h, int32_t rank, int32_t world, const void *id)
{
    using namespace ses;
    SES_REQUIRE(h && id, "ses_comm_init: null argument");
    SES_REQUIRE(world >= 1 && rank >= 0 && rank < world, "ses_comm_init: rank %d not in [0, %d)", rank, world);
    SES_REQUIRE(!h->comm, "ses_comm_init: this handle already has a communicator");
    int rc = load_rccl();
    if (rc != SES_OK) return rc;
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    NcclId nid;
    std::memcpy(nid.internal, id, SES_COMM_ID_BYTES);
    NcclComm comm = nullptr;
    int nrc = g_rccl.CommInitRank(&comm, world, nid, rank);   // collective: returns once every rank has joined
    if (nrc != 0) return rccl_error("ncclCommInitRank", nrc);
    h->comm = comm;
    h->comm_rank = rank;
    h->comm_world = world;
    return SES_OK;
}

int ses_comm_info(ses_handle *h, int32_t *rank, int32_t *world, int32_t *rccl_version)
{
    using namespace ses;
    SES_REQUIRE(h, "ses_comm_info: null handle");
    if (rank) *rank = h->comm ? h->comm_rank : 0;
    if (world) *world = h->comm ? h->comm_world : 0;      // 0: no communicator
    if (rccl_version) {
        int v = 0;
        if (g_rccl.lib) (void)g_rccl.GetVersion(&v);
        *rccl_version = v;
    }
    return SES_OK;
}

int ses_comm_destroy(ses_handle *h)
{
    SES_REQUIRE(h, "ses_comm_destroy: null handle");
    return ses::comm_release(h);
}

int ses_allgather_fitness(ses_handle *h, const float *local, int32_t n_per_rank, float *all)
{
    using namespace ses;
    SES_REQUIRE(h && local && all, "ses_allgather_fitness: null argument");
    SES_REQUIRE(n_per_rank >= 1, "ses_allgather_fitness: n_per_rank must be >= 1");
    SES_REQUIRE(h->comm, "ses_allgather_fitness: no communicator (call ses_comm_init first)");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    int nrc = g_rccl.AllGather(local, all, (size_t)n_per_rank, NCCL_FLOAT32, (NcclComm)h->comm, h->stream);
    if (nrc != 0) return rccl_error("ncclAllGather", nrc);
    return SES_OK;
}

}  // extern "C"
