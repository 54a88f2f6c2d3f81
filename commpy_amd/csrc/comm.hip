// Collectives of the sharded decode path -- RCCL over xGMI, owned by the library (no torch.distributed).
//
// The hot path shards by codeword and has no exchange step inside any decoder; what crosses GPUs is
//   * one all-gather of the decoded bits (uint8) when a caller wants the whole batch on every GPU -- the reference's
//     counterpart is simply the returned array of viterbi_decode / ldpc_bp_decode (convcode.py:749, ldpc.py:251-254);
//   * one all-reduce (sum, int64) of the error / bit counters of a Monte-Carlo sweep (links.py:252-260).
// Two ways to form a communicator:
//   * cpx_comm_init_all  -- ONE process drives several GPUs (ncclCommInitAll); every collective call takes one buffer per
//     local device and is issued inside ncclGroupStart/End.  This is the mode a CommPy user gets from
//     commpy_amd.parallel.DeviceGroup: a plain Python script, no launcher.
//   * cpx_comm_init_rank -- one process per GPU (bench.py under torch.distributed.run): rank 0 creates a 128-byte id
//     (cpx_comm_unique_id), the launcher plumbing hands it to the other ranks, every rank joins with its current device.
// xGMI is point to point (7 links x ~153 GB/s per GPU): payloads here are small against the decode time (config 2: 8.4 MB
// of bits per GPU and step, config 4: 64 MB), so one plain ncclAllGather per step is issued on the decode stream and
// no bucketing is needed.
//
// librccl is loaded lazily with dlopen at the first communicator: a process that never asks for a collective never
// loads it, and when torch is already in the process its copy of librccl.so.1 (same SONAME) is the one that is found.
#include "cpx_internal.h"

#include <dlfcn.h>

#include <mutex>
#include <vector>

using namespace cpx;

namespace {

// the slice of the NCCL API that is used, declared here so that rccl.h is not needed at build time
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { NCCL_SUCCESS = 0 };
enum { NCCL_INT8 = 0, NCCL_UINT8 = 1, NCCL_INT64 = 4, NCCL_FLOAT64 = 8 };   // ncclDataType_t
enum { NCCL_SUM = 0, NCCL_MAX = 2 };                                        // ncclRedOp_t

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return CPX_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    CPX_REQUIRE(h, CPX_ENODEV, "cannot load librccl.so.1: %s", dlerror());
    Rccl r;
    r.handle = h;
#define SYM(field, name)                                                              \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                    \
    CPX_REQUIRE(r.field, CPX_ENODEV, "librccl lacks %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommCount, "ncclCommCount");
    SYM(CommUserRank, "ncclCommUserRank");
    SYM(AllGather, "ncclAllGather");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl = r;
    return CPX_OK;
}

#define CPX_NCCL(call)                                                                                 \
    do {                                                                                               \
        ncclResult_t _r = (call);                                                                      \
        if (_r != NCCL_SUCCESS) {                                                                      \
            set_error("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
            return CPX_EHIP;                                                                           \
        }                                                                                              \
    } while (0)

}  // namespace

struct cpx_comm {
    __attribute__((visibility("hidden"))) ~cpx_comm() = default;   // (see cpx_trellis, cpx_internal.h)
    int nranks = 0;                     // ranks of the whole communicator
    int nlocal = 0;                     // ranks driven by this process (1 for init_rank)
    int rank0 = 0;                      // rank of local device 0 (init_all: 0)
    std::vector<int> devices;           // [nlocal]
    std::vector<ncclComm_t> comms;      // [nlocal]
};

namespace {

// RAII: restore the caller's current device after a loop over the local devices
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { (void)hipGetDevice(&dev); }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

hipStream_t stream_of(const cpx_comm *c, void *const *streams, int i) {
    if (streams && streams[i]) return reinterpret_cast<hipStream_t>(streams[i]);
    (void)hipSetDevice(c->devices[i]);
    return lib_stream();
}

}  // namespace

extern "C" {

int cpx_comm_unique_id(void *id128) {
    CPX_REQUIRE(id128, CPX_EINVAL, "cpx_comm_unique_id: null pointer");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    CPX_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return CPX_OK;
}

int cpx_comm_init_rank(const void *id128, int nranks, int rank, cpx_comm **out) {
    CPX_REQUIRE(id128 && out, CPX_EINVAL, "cpx_comm_init_rank: null pointer");
    CPX_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, CPX_EINVAL, "cpx_comm_init_rank: bad rank %d of %d", rank, nranks);
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = load_rccl())) return rc;
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    cpx_comm *c = new cpx_comm;
    c->nranks = nranks; c->nlocal = 1; c->rank0 = rank;
    c->devices.resize(1); c->comms.resize(1);
    (void)hipGetDevice(&c->devices[0]);
    ncclResult_t r = g_rccl.CommInitRank(&c->comms[0], nranks, id, rank);
    if (r != NCCL_SUCCESS) {
        set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
        delete c;
        return CPX_EHIP;
    }
    *out = c;
    return CPX_OK;
}

int cpx_comm_init_all(const int *devices, int ndev, cpx_comm **out) {
    CPX_REQUIRE(out && ndev >= 1, CPX_EINVAL, "cpx_comm_init_all: bad arguments");
    int rc = ensure_device();
    if (rc) return rc;
    int have = 0;
    CPX_HIP(hipGetDeviceCount(&have));
    cpx_comm *c = new cpx_comm;
    c->nranks = ndev; c->nlocal = ndev; c->rank0 = 0;
    c->devices.resize(ndev); c->comms.resize(ndev);
    for (int i = 0; i < ndev; i++) {
        c->devices[i] = devices ? devices[i] : i;
        if (c->devices[i] < 0 || c->devices[i] >= have) {
            set_error("cpx_comm_init_all: device %d not present (%d visible)", c->devices[i], have);
            delete c;
            return CPX_EINVAL;
        }
    }
    if ((rc = load_rccl())) { delete c; return rc; }
    DeviceGuard guard;
    ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), ndev, c->devices.data());
    if (r != NCCL_SUCCESS) {
        set_error("ncclCommInitAll(%d devices) failed: %s", ndev, g_rccl.GetErrorString(r));
        delete c;
        return CPX_EHIP;
    }
    *out = c;
    return CPX_OK;
}

/* What RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank of the first local rank), not the numbers
 * the caller passed in: bench.py prints them as `comm_world`, so "did RCCL see N ranks" is answerable from its line. */
int cpx_comm_info(const cpx_comm *c, int *nranks, int *nlocal, int *first_rank) {
    CPX_REQUIRE(c && c->nlocal >= 1 && c->comms[0], CPX_EINVAL, "cpx_comm_info: null communicator");
    int count = -1, urank = -1;
    CPX_NCCL(g_rccl.CommCount(c->comms[0], &count));
    CPX_NCCL(g_rccl.CommUserRank(c->comms[0], &urank));
    CPX_REQUIRE(count == c->nranks && urank == c->rank0, CPX_EHIP,
                "cpx_comm_info: RCCL reports rank %d of %d, the communicator was formed as rank %d of %d", urank, count,
                c->rank0, c->nranks);
    if (nranks) *nranks = count;
    if (nlocal) *nlocal = c->nlocal;
    if (first_rank) *first_rank = urank;
    return CPX_OK;
}

int cpx_comm_destroy(cpx_comm *c) {
    if (!c) return CPX_OK;
    DeviceGuard guard;
    for (int i = 0; i < c->nlocal; i++)
        if (c->comms[i]) {
            (void)hipSetDevice(c->devices[i]);
            (void)g_rccl.CommDestroy(c->comms[i]);
        }
    delete c;
    return CPX_OK;
}

/* d_send[i] -> bytes_per_rank bytes of local device i; d_recv[i] -> nranks * bytes_per_rank bytes on local device i,
 * rank r's block at offset r * bytes_per_rank.  In place is allowed (d_send[i] == d_recv[i] + rank_i * bytes_per_rank). */
int cpx_comm_allgather_u8(cpx_comm *c, const void *const *d_send, void *const *d_recv, size_t bytes_per_rank,
                          void *const *streams) {
    CPX_REQUIRE(c && d_send && d_recv, CPX_EINVAL, "cpx_comm_allgather_u8: null pointer");
    if (bytes_per_rank == 0) return CPX_OK;
    DeviceGuard guard;
    std::vector<hipStream_t> st(c->nlocal);
    for (int i = 0; i < c->nlocal; i++) st[i] = stream_of(c, streams, i);
    CPX_NCCL(g_rccl.GroupStart());
    for (int i = 0; i < c->nlocal; i++) {
        ncclResult_t r = g_rccl.AllGather(d_send[i], d_recv[i], bytes_per_rank, NCCL_UINT8, c->comms[i], st[i]);
        if (r != NCCL_SUCCESS) {
            (void)g_rccl.GroupEnd();
            set_error("ncclAllGather failed: %s", g_rccl.GetErrorString(r));
            return CPX_EHIP;
        }
    }
    CPX_NCCL(g_rccl.GroupEnd());
    return CPX_OK;
}

static int allreduce(cpx_comm *c, const void *const *d_send, void *const *d_recv, size_t count, int dtype, int op,
                     void *const *streams) {
    CPX_REQUIRE(c && d_send && d_recv, CPX_EINVAL, "cpx_comm_allreduce: null pointer");
    CPX_REQUIRE(op == 0 || op == 1, CPX_EINVAL, "cpx_comm_allreduce: op must be 0 (sum) or 1 (max)");
    if (count == 0) return CPX_OK;
    DeviceGuard guard;
    std::vector<hipStream_t> st(c->nlocal);
    for (int i = 0; i < c->nlocal; i++) st[i] = stream_of(c, streams, i);
    CPX_NCCL(g_rccl.GroupStart());
    for (int i = 0; i < c->nlocal; i++) {
        ncclResult_t r = g_rccl.AllReduce(d_send[i], d_recv[i], count, dtype, op == 0 ? NCCL_SUM : NCCL_MAX, c->comms[i], st[i]);
        if (r != NCCL_SUCCESS) {
            (void)g_rccl.GroupEnd();
            set_error("ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
            return CPX_EHIP;
        }
    }
    CPX_NCCL(g_rccl.GroupEnd());
    return CPX_OK;
}

int cpx_comm_allreduce_i64(cpx_comm *c, const void *const *d_send, void *const *d_recv, size_t count, int op,
                           void *const *streams) {
    return allreduce(c, d_send, d_recv, count, NCCL_INT64, op, streams);
}

int cpx_comm_allreduce_f64(cpx_comm *c, const void *const *d_send, void *const *d_recv, size_t count, int op,
                           void *const *streams) {
    return allreduce(c, d_send, d_recv, count, NCCL_FLOAT64, op, streams);
}

}  // extern "C"
