// ses_comm.hip -- the one exchange step of a multi-GPU generation: the all-gather of the per-offspring fitness
// over RCCL (xGMI inside a node).  Replaces the gather half of `results = p.map(RolloutWorker, arguments)`
// (learning_strategies/evolution/loop.py:66-79) when the population is sharded over one process per GPU.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the SONAME of both /opt/rocm's and the PyTorch wheel's copy:
// inside a process that has torch loaded the already-mapped library is reused, so there is never a second RCCL in
// the address space).  A single-GPU user never loads it.  Only the six entry points below are used; their C
// signatures are part of NCCL's stable API (ncclUniqueId = 128 opaque bytes passed by value, ncclFloat32 = 7).
//
// Second transport, inside one node: PEER STORES.  The exchange is 16-256 KB in total and happens once per ~0.25 ms
// generation, so what it costs is latency, not bandwidth: a ring all-gather is W - 1 dependent hops behind a library
// launch.  xGMI is point-to-point and every GPU can store into every other GPU's memory, so each rank owns a MAILBOX
// (fine-grained device memory, exported with hipIpcGetMemHandle and mapped by the W - 1 peers) and ONE small kernel
// per rank does the whole exchange: workgroup b stores the local shard into peer b's mailbox, publishes a sequence
// number there (system-scope release), waits for peer b's sequence number in its own mailbox (acquire) and copies
// peer b's shard out -- one store, one flag, no hops, no intermediate rank.  Two slots alternate by sequence parity:
// a peer can only be one exchange ahead (its next exchange needs this rank's next flag, which is stream-ordered after
// this rank's reads), so slot (seq & 1) is never overwritten while it is still being read.
// A wait gives up after the handle's time-out (ses_set_tuning "comm_p2p_timeout_ms", default 60 s -- RCCL would wait for
// ever; a stalled rank is usually a checkpoint write or a page-in, not a dead peer) of the 100 MHz real-time counter: the
// shard is filled with NaN and a bit is set in a host-visible error word (ses_comm_p2p_status reads it without touching
// the stream) -- no kernel of this library spins for ever.  A sequence word that is already PAST the awaited exchange
// (the peer gave up on this rank earlier and moved on) is an error at once, without waiting.  By default the next
// ses_allgather_fitness then fails with SES_ERR_COMM; with "comm_p2p_keep_going" = 1 later exchanges still run (every
// rank keeps publishing, so nobody waits a time-out per generation) and the HOST is expected to poll the status word,
// agree with the other ranks and roll back -- ESLoop.run() does exactly that at its checkpoint boundaries.
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

// ---- peer-store transport ---------------------------------------------------------------------------------------
constexpr int P2P_MAX_WORLD = 16;
constexpr unsigned long long P2P_TICKS_PER_MS = 100000ull;       // the real-time counter runs at 100 MHz

}  // namespace ses

struct ses_p2p {
    int rank, world, max_per_rank, splits_max;
    bool attached;
    bool local;                             // attached to handles of this process (ses_comm_p2p_attach_local): nothing mapped, nothing to close
    size_t bytes, flag_offset;              // mailbox layout: float data[2][world][max_per_rank] | uint32 flags[2][world][splits_max] | seen mask
                                            //                 | u64 granules[2][world][max_per_rank / 2]
    size_t gran_offset;                     // the granule area: {sequence, value} words of the exchanges kernels do themselves (comm_p2p_granules_begin);
                                            // an area and a sequence of its own, so that no float of a flag-based exchange is ever read as a tag
    uint32_t gseq;                          // granule exchanges issued so far
    void *own;                              // this rank's mailbox (device memory, fine-grained)
    void *peer[ses::P2P_MAX_WORLD];         // peer[r]: rank r's mailbox as mapped here (peer[rank] == own)
    uint32_t seq;                           // exchanges issued so far
    uint32_t *err_host;                     // pinned host word set by a kernel that timed out
    uint32_t *err_dev;                      // its device alias
    uint32_t *err_seen;                     // the same mask in this rank's own device memory (behind the flags of the mailbox): what
                                            // the exchange kernel reads to know that a peer has already cost it a time-out
    unsigned long long timeout_ticks;       // how long a workgroup waits for a peer's sequence word
};

namespace ses {

struct P2pPeers {
    float *data[P2P_MAX_WORLD];
    uint32_t *flags[P2P_MAX_WORLD];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int P2P_SPLIT = 4096;          // floats one workgroup moves; a shard of n floats takes ceil(n / P2P_SPLIT) workgroups per peer

// workgroup (b, s): slice s of the local shard -> peer b's mailbox, then slice s of peer b's shard (from this rank's
// mailbox) -> out.  Every (source rank, slice) has its own sequence word, so no workgroup waits for another one of
// its own grid.  16-byte accesses when n and the pointers allow (the mailbox is uncached memory: wide and few).
__global__ __launch_bounds__(256) void k_allgather_p2p(const float *__restrict__ local, int n, int max_per_rank, int splits_max,
                                                       int rank, int world, uint32_t seq, P2pPeers peers,
                                                       float *__restrict__ out, uint32_t *err, int vec4,
                                                       unsigned long long timeout_ticks, uint32_t *err_seen)
{
    const int b = blockIdx.x, sp = blockIdx.y, slot = (int)(seq & 1u);
    // a peer that has already cost this rank a time-out is waited for 2 ms at most from then on (a stalled peer would
    // otherwise cost every later exchange the full time-out until the host's next recovery boundary); the mask is read from
    // device memory, early, so that its latency hides behind the stores
    const uint32_t seen = threadIdx.x == 0 ? __hip_atomic_load(err_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const int i0 = sp * P2P_SPLIT, i1 = i0 + P2P_SPLIT < n ? i0 + P2P_SPLIT : n;
    float *dst = peers.data[b] + ((size_t)slot * world + rank) * max_per_rank;
    if (vec4) {
        for (int i = i0 + 4 * threadIdx.x; i < i1; i += 1024)
            __builtin_nontemporal_store(*reinterpret_cast<const f32x4 *>(local + i), reinterpret_cast<f32x4 *>(dst + i));
    } else {
        for (int i = i0 + threadIdx.x; i < i1; i += 256) __builtin_nontemporal_store(local[i], dst + i);
    }
    __threadfence_system();                                           // the slice is visible to peer b ...
    __syncthreads();
    if (threadIdx.x == 0)                                              // ... before its sequence number is
        __hip_atomic_store(peers.flags[b] + ((size_t)slot * world + rank) * splits_max + sp, seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const uint32_t *flag = peers.flags[rank] + ((size_t)slot * world + b) * splits_max + sp;
        const unsigned long long t0 = real_time();
        const unsigned long long short_ticks = 2ull * P2P_TICKS_PER_MS;
        if (((seen >> (b & 31)) & 1u) && timeout_ticks > short_ticks) timeout_ticks = short_ticks;
        int good = 1;
        for (;;) {
            const uint32_t f = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (f == seq) break;
            // a word from a LATER exchange of the same slot: the peer stopped waiting for this rank at some point and has
            // overwritten the slice -- nothing to wait for
            if ((int32_t)(f - seq) > 0 || real_time() - t0 > timeout_ticks) { good = 0; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        ok = good;
        if (!good) {
            atomicOr_system(err, 1u << (b & 31));
            atomicOr(err_seen, 1u << (b & 31));
        }
    }
    __syncthreads();
    const float *src = peers.data[rank] + ((size_t)slot * world + b) * max_per_rank;
    float *o = out + (size_t)b * n;
    if (vec4) {
        const float nan = __builtin_nanf("");
        for (int i = i0 + 4 * threadIdx.x; i < i1; i += 1024) {
            const f32x4 v = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + i)) : f32x4{nan, nan, nan, nan};
            *reinterpret_cast<f32x4 *>(o + i) = v;
        }
    } else {
        for (int i = i0 + threadIdx.x; i < i1; i += 256) o[i] = ok ? __builtin_nontemporal_load(src + i) : __builtin_nanf("");
    }
}

// The same all-gather with the data as its own flag: workgroup (b, s) stores slice s of the local shard into rank b's
// mailbox as 8-byte {exchange number, value} granules -- one store per float, no fence, no flag word -- and then polls rank
// b's granules of the slice in its own mailbox.  Twice the bytes of the flag-based kernel over the links and no release /
// acquire round -- and, measured between two ranks of one GPU, 12.4 us per exchange of 4096 floats against 6.3: sixteen 8-byte
// uncached stores and sixteen polled loads per thread cost more than the one fence they save, so ses_allgather_fitness uses it
// only on request ("comm_granule_allgather").  It shares the granule area and its sequence with the exchanges kernels do
// themselves (ses_openai_generation_sharded: a few granules per WORKGROUP there, which is where granules pay), so a rank that
// does nothing else can answer such an exchange with it.
__global__ __launch_bounds__(256) void k_allgather_granules(const float *__restrict__ local, int n, P2pGranuleView gv,
                                                            float *__restrict__ out)
{
    const int b = blockIdx.x, sp = blockIdx.y;
    const int i0 = sp * P2P_SPLIT, i1 = i0 + P2P_SPLIT < n ? i0 + P2P_SPLIT : n;
    unsigned long long *dst = gv.dst[b];
    for (int i = i0 + threadIdx.x; i < i1; i += 256) granule_store(dst + i, gv.seq, __builtin_bit_cast(uint32_t, local[i]));
    const unsigned long long *src = gv.src + (size_t)b * gv.section;
    float *o = out + (size_t)b * n;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) o[i] = __builtin_bit_cast(float, granule_wait(src + i, gv, b));
}

static void p2p_free(ses_p2p *p)
{
    if (!p) return;
    for (int r = 0; r < p->world && p->attached && !p->local; ++r)
        if (r != p->rank && p->peer[r]) (void)hipIpcCloseMemHandle(p->peer[r]);
    if (p->own) (void)hipFree(p->own);
    if (p->err_host) (void)hipHostFree(p->err_host);
    delete p;
}

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

void comm_p2p_set_timeout(ses_handle *h)
{
    if (h->p2p)
        h->p2p->timeout_ticks = (unsigned long long)(h->tune_comm_p2p_timeout_ms > 0 ? h->tune_comm_p2p_timeout_ms : 60000) * P2P_TICKS_PER_MS;
}

int comm_p2p_granules_begin(ses_handle *comm, int granules, P2pGranuleView *v)
{
    static_assert(P2P_GRANULE_MAX_WORLD == P2P_MAX_WORLD, "world limit");
    ses_p2p *p = comm ? comm->p2p : nullptr;
    if (!p || !p->attached || granules < 1 || granules > p->max_per_rank / 2 || (comm->tune_comm_force_rccl && comm->comm) ||
        !comm->tune_comm_granules_enabled)
        return set_error(SES_ERR_UNSUPPORTED, "granule exchange: no peer-store transport for %d granules per rank", granules);
    if (*(volatile uint32_t *)p->err_host != 0u && !comm->tune_comm_p2p_keep_going)
        return set_error(SES_ERR_COMM, "peer-store exchange: an earlier exchange timed out waiting for rank mask 0x%x (its output was "
                         "NaN-filled); detach the transport (ses_comm_p2p_detach) to continue over RCCL", *(volatile uint32_t *)p->err_host);
    if (p->gseq == 0x7FFFFFFFu)
        return set_error(SES_ERR_COMM, "peer-store sequence exhausted; detach and attach the transport again");
    p->gseq += 1;
    const int slot = (int)(p->gseq & 1u);
    // granules[2][world][max_per_rank / 2]: section (slot, source rank).  Two slots alternate as in the flag-based exchange: a
    // peer's granules of exchange s + 2 can only come after it has finished s + 1, which needed this rank's granules of s + 1,
    // which this rank's stream issues after its reads of s.
    const size_t section = (size_t)p->max_per_rank / 2;
    for (int r = 0; r < p->world; ++r)
        v->dst[r] = (unsigned long long *)((char *)p->peer[r] + p->gran_offset) + ((size_t)slot * p->world + p->rank) * section;
    v->src = (const unsigned long long *)((char *)p->own + p->gran_offset) + (size_t)slot * p->world * section;
    v->section = (int)section;
    v->rank = p->rank; v->world = p->world;
    v->seq = p->gseq;
    v->timeout_ticks = p->timeout_ticks;
    v->err = p->err_dev; v->err_seen = p->err_seen;
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
    if (h->p2p) {
        (void)hipStreamSynchronize(h->stream);
        p2p_free(h->p2p);
        h->p2p = nullptr;
    }
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

int ses_comm_p2p_export(ses_handle *h, int32_t rank, int32_t world, int32_t max_per_rank, void *handle)
{
    using namespace ses;
    SES_REQUIRE(h && handle, "ses_comm_p2p_export: null argument");
    SES_REQUIRE(world >= 2 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world,
                "ses_comm_p2p_export: rank %d / world %d (2 <= world <= %d)", rank, world, P2P_MAX_WORLD);
    SES_REQUIRE(max_per_rank >= 1, "ses_comm_p2p_export: max_per_rank must be >= 1");
    SES_REQUIRE(!h->p2p, "ses_comm_p2p_export: this handle already has a mailbox");
    static_assert(sizeof(hipIpcMemHandle_t) <= SES_COMM_P2P_HANDLE_BYTES, "handle size");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    ses_p2p *p = new ses_p2p();
    std::memset(p, 0, sizeof *p);
    p->rank = rank; p->world = world;
    p->timeout_ticks = (unsigned long long)(h->tune_comm_p2p_timeout_ms > 0 ? h->tune_comm_p2p_timeout_ms : 60000) * P2P_TICKS_PER_MS;
    p->max_per_rank = (max_per_rank + 3) / 4 * 4;                     // slots stay 16-byte aligned
    p->splits_max = ceil_div(p->max_per_rank, P2P_SPLIT);
    p->flag_offset = (sizeof(float) * 2 * (size_t)world * p->max_per_rank + 255) / 256 * 256;
    const size_t seen_offset = (p->flag_offset + sizeof(uint32_t) * 2 * (size_t)world * p->splits_max + 255) / 256 * 256;
    p->gran_offset = seen_offset + 256;
    p->bytes = p->gran_offset + sizeof(unsigned long long) * 2 * (size_t)world * (p->max_per_rank / 2);
    hipError_t e = hipExtMallocWithFlags(&p->own, p->bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p->own, p->bytes, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) {
        p->own = nullptr; p2p_free(p);
        return set_error(SES_ERR_COMM, "ses_comm_p2p_export: no fine-grained device memory for the mailbox: %s", hipGetErrorString(e));
    }
    e = hipMemset(p->own, 0, p->bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();                  // zeroed before any peer can learn the handle
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->err_host, sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess) { *p->err_host = 0u; e = hipHostGetDevicePointer((void **)&p->err_dev, p->err_host, 0); }
    hipIpcMemHandle_t ipc;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&ipc, p->own);
    if (e != hipSuccess) {
        p2p_free(p);
        return set_error(SES_ERR_COMM, "ses_comm_p2p_export failed: %s", hipGetErrorString(e));
    }
    std::memset(handle, 0, SES_COMM_P2P_HANDLE_BYTES);
    std::memcpy(handle, &ipc, sizeof ipc);
    p->peer[rank] = p->own;
    p->err_seen = (uint32_t *)((char *)p->own + seen_offset);
    h->p2p = p;
    return SES_OK;
}

int ses_comm_p2p_attach(ses_handle *h, const void *handles)
{
    using namespace ses;
    SES_REQUIRE(h && handles, "ses_comm_p2p_attach: null argument");
    SES_REQUIRE(h->p2p && !h->p2p->attached, "ses_comm_p2p_attach: export a mailbox first (once)");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    ses_p2p *p = h->p2p;
    for (int r = 0; r < p->world; ++r) {
        if (r == p->rank) continue;
        hipIpcMemHandle_t ipc;
        std::memcpy(&ipc, (const char *)handles + (size_t)r * SES_COMM_P2P_HANDLE_BYTES, sizeof ipc);
        const hipError_t e = hipIpcOpenMemHandle(&p->peer[r], ipc, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int q = 0; q < r; ++q)
                if (q != p->rank && p->peer[q]) { (void)hipIpcCloseMemHandle(p->peer[q]); p->peer[q] = nullptr; }
            p->peer[r] = nullptr;
            return set_error(SES_ERR_COMM, "ses_comm_p2p_attach: cannot map the mailbox of rank %d: %s", r, hipGetErrorString(e));
        }
    }
    p->attached = true;
    return SES_OK;
}

int ses_comm_p2p_attach_local(ses_handle *h, ses_handle *const *peers)
{
    using namespace ses;
    SES_REQUIRE(h && peers, "ses_comm_p2p_attach_local: null argument");
    SES_REQUIRE(h->p2p && !h->p2p->attached, "ses_comm_p2p_attach_local: export a mailbox first (once)");
    ses_p2p *p = h->p2p;
    SES_REQUIRE(peers[p->rank] == h, "ses_comm_p2p_attach_local: peers[%d] must be this handle", p->rank);
    for (int r = 0; r < p->world; ++r) {
        const ses_handle *q = peers[r];
        SES_REQUIRE(q && q->p2p && q->p2p->own && q->p2p->rank == r && q->p2p->world == p->world &&
                        q->p2p->max_per_rank == p->max_per_rank,
                    "ses_comm_p2p_attach_local: peers[%d] has not exported a mailbox of the same shape as rank %d of %d", r, r, p->world);
    }
    // handles of one process on DIFFERENT devices (a host that drives several GPUs from one process): this device needs peer
    // access to the device that owns the mailbox before a kernel may store into it (the IPC route asks for it when it maps the
    // handle, hipIpcMemLazyEnablePeerAccess; here the pointer is used as it is)
    for (int r = 0; r < p->world; ++r) {
        const int peer_dev = peers[r]->cfg.device;
        if (peer_dev == h->cfg.device) continue;
        SES_HIP_TRY(hipSetDevice(h->cfg.device));
        int can = 0;
        SES_HIP_TRY(hipDeviceCanAccessPeer(&can, h->cfg.device, peer_dev));
        SES_REQUIRE(can, "ses_comm_p2p_attach_local: device %d cannot access the memory of device %d (rank %d's mailbox)",
                    h->cfg.device, peer_dev, r);
        const hipError_t e = hipDeviceEnablePeerAccess(peer_dev, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            return set_error(SES_ERR_COMM, "ses_comm_p2p_attach_local: hipDeviceEnablePeerAccess(%d): %s", peer_dev, hipGetErrorString(e));
        (void)hipGetLastError();
    }
    for (int r = 0; r < p->world; ++r) p->peer[r] = peers[r]->p2p->own;
    p->attached = true;
    p->local = true;
    return SES_OK;
}

int ses_comm_p2p_info(ses_handle *h, int32_t *world, int32_t *max_per_rank, int32_t *exchanges)
{
    SES_REQUIRE(h, "ses_comm_p2p_info: null handle");
    const bool on = h->p2p && h->p2p->attached;
    if (world) *world = on ? h->p2p->world : 0;                     // 0: the transport is not attached
    if (max_per_rank) *max_per_rank = on ? h->p2p->max_per_rank : 0;
    if (exchanges) *exchanges = on ? (int32_t)h->p2p->seq : 0;
    return SES_OK;
}

int ses_comm_p2p_counts(ses_handle *h, int32_t *flag_exchanges, int32_t *granule_exchanges)
{
    SES_REQUIRE(h, "ses_comm_p2p_counts: null handle");
    const bool on = h->p2p && h->p2p->attached;
    if (flag_exchanges) *flag_exchanges = on ? (int32_t)h->p2p->seq : 0;
    if (granule_exchanges) *granule_exchanges = on ? (int32_t)h->p2p->gseq : 0;
    return SES_OK;
}

int ses_comm_p2p_status(ses_handle *h, uint32_t *timed_out_mask)
{
    SES_REQUIRE(h && timed_out_mask, "ses_comm_p2p_status: null argument");
    *timed_out_mask = (h->p2p && h->p2p->err_host) ? *(volatile uint32_t *)h->p2p->err_host : 0u;
    return SES_OK;
}

int ses_comm_p2p_reset_status(ses_handle *h)
{
    SES_REQUIRE(h, "ses_comm_p2p_reset_status: null handle");
    if (h->p2p && h->p2p->err_host) {
        SES_HIP_TRY(hipSetDevice(h->cfg.device));
        SES_HIP_TRY(hipStreamSynchronize(h->stream));                // no exchange of this handle is in flight
        *(volatile uint32_t *)h->p2p->err_host = 0u;
        SES_HIP_TRY(hipMemset(h->p2p->err_seen, 0, sizeof(uint32_t)));
    }
    return SES_OK;
}

int ses_stream_create_exclusive(int32_t device, void **stream)
{
    SES_REQUIRE(stream, "ses_stream_create_exclusive: null argument");
    int ndev = 0;
    SES_HIP_TRY(hipGetDeviceCount(&ndev));
    SES_REQUIRE(device >= 0 && device < ndev, "ses_stream_create_exclusive: device %d not in [0,%d)", device, ndev);
    SES_HIP_TRY(hipSetDevice(device));
    int cus = 0;
    SES_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    SES_REQUIRE(cus >= 1 && cus <= 1024, "ses_stream_create_exclusive: %d compute units?", cus);
    uint32_t mask[32];
    const int words = (cus + 31) / 32;
    for (int w = 0; w < words; ++w) mask[w] = (w + 1) * 32 <= cus ? 0xFFFFFFFFu : ((1u << (cus - w * 32)) - 1u);
    hipStream_t s = nullptr;
    SES_HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
    *stream = (void *)s;
    return SES_OK;
}

int ses_stream_destroy(void *stream)
{
    SES_REQUIRE(stream, "ses_stream_destroy: null stream");
    SES_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return SES_OK;
}

int ses_comm_p2p_detach(ses_handle *h)
{
    SES_REQUIRE(h, "ses_comm_p2p_detach: null handle");
    if (h->p2p) {
        (void)hipStreamSynchronize(h->stream);
        ses::p2p_free(h->p2p);
        h->p2p = nullptr;
    }
    return SES_OK;
}

int ses_allgather_fitness(ses_handle *h, const float *local, int32_t n_per_rank, float *all)
{
    using namespace ses;
    SES_REQUIRE(h && local && all, "ses_allgather_fitness: null argument");
    SES_REQUIRE(n_per_rank >= 1, "ses_allgather_fitness: n_per_rank must be >= 1");
    if (h->p2p && h->p2p->attached && n_per_rank <= h->p2p->max_per_rank / 2 && !(h->tune_comm_force_rccl && h->comm) &&
        h->tune_comm_granules && h->tune_comm_granules_enabled) {
        // granules: the data is the flag (k_allgather_granules); shards beyond half a mailbox section take the flag-based kernel;
        // with the granules switched off for this transport (comm_granules_enabled = 0 after a failed self-test) the branch is
        // not entered at all -- comm_p2p_granules_begin would format an "unsupported" message on every exchange and leave it in
        // ses_last_error behind a call that succeeds
        SES_HIP_TRY(hipSetDevice(h->cfg.device));
        P2pGranuleView gv;
        const int grc = comm_p2p_granules_begin(h, n_per_rank, &gv);
        if (grc == SES_OK) {
            hipLaunchKernelGGL(k_allgather_granules, dim3(gv.world, ceil_div(n_per_rank, P2P_SPLIT)), dim3(256), 0, h->stream, local,
                               (int)n_per_rank, gv, all);
            SES_HIP_TRY(hipGetLastError());
            return SES_OK;
        }
        if (grc != SES_ERR_UNSUPPORTED) return grc;
        // granules switched off for this transport (comm_granules_enabled = 0 after a failed self-test): the flag-based kernel below
    }
    if (h->p2p && h->p2p->attached && n_per_rank <= h->p2p->max_per_rank && !(h->tune_comm_force_rccl && h->comm)) {
        ses_p2p *p = h->p2p;
        if (*(volatile uint32_t *)p->err_host != 0u && !h->tune_comm_p2p_keep_going)
            return set_error(SES_ERR_COMM, "ses_allgather_fitness: an earlier peer-store exchange timed out waiting for rank mask 0x%x "
                             "(its output was NaN-filled); detach the transport (ses_comm_p2p_detach) to continue over RCCL",
                             *(volatile uint32_t *)p->err_host);
        SES_HIP_TRY(hipSetDevice(h->cfg.device));
        P2pPeers peers;
        std::memset(&peers, 0, sizeof peers);
        for (int r = 0; r < p->world; ++r) {
            peers.data[r] = (float *)p->peer[r];
            peers.flags[r] = (uint32_t *)((char *)p->peer[r] + p->flag_offset);
        }
        if (p->seq == 0xFFFFFFFFu)                                       // 2^32 exchanges (weeks): the sequence words would wrap onto their initial 0
            return set_error(SES_ERR_COMM, "ses_allgather_fitness: peer-store sequence exhausted; detach and attach the transport again");
        p->seq += 1;
        const int vec4 = (n_per_rank % 4 == 0) && ((uintptr_t)local % 16 == 0) && ((uintptr_t)all % 16 == 0);
        hipLaunchKernelGGL(k_allgather_p2p, dim3(p->world, ceil_div(n_per_rank, P2P_SPLIT)), dim3(256), 0, h->stream, local,
                           (int)n_per_rank, p->max_per_rank, p->splits_max, p->rank, p->world, p->seq, peers, all, p->err_dev, vec4,
                           p->timeout_ticks, p->err_seen);
        SES_HIP_TRY(hipGetLastError());
        return SES_OK;
    }
    SES_REQUIRE(h->comm, "ses_allgather_fitness: no communicator (call ses_comm_init or ses_comm_p2p_attach first)");
    SES_HIP_TRY(hipSetDevice(h->cfg.device));
    int nrc = g_rccl.AllGather(local, all, (size_t)n_per_rank, NCCL_FLOAT32, (NcclComm)h->comm, h->stream);
    if (nrc != 0) return rccl_error("ncclAllGather", nrc);
    return SES_OK;
}

}  // extern "C"
