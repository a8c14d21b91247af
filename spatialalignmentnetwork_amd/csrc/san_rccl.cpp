// Native RCCL binding for the gradient exchange (round 5): the data-parallel step's all-reduces become plain C-ABI calls
// (san_rccl_allreduce_sum_f32 on the communication stream), so a recorded step's replay tape (san_replay.cpp) walks them like
// any kernel launch -- no Python callable per collective.  RCCL is not linked: the library the process already has
// (torch's librccl.so, found by the caller in /proc/self/maps) is opened by path and six entry points are resolved.
// The communicator is this package's own (ncclCommInitRank with an id the caller broadcasts over torch.distributed);
// torch's process group stays in charge of everything outside the step (broadcasts, barriers, scalar reductions).
#include <dlfcn.h>
#include <mutex>
#include <vector>

#include "san_common.h"

namespace {

struct UniqueId {
    char internal[128];        // NCCL_UNIQUE_ID_BYTES
};
typedef int (*get_unique_id_t)(UniqueId*);
typedef int (*comm_init_rank_t)(void**, int, UniqueId, int);
typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*reduce_scatter_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*all_gather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*comm_destroy_t)(void*);
typedef const char* (*get_error_string_t)(int);
typedef int (*get_version_t)(int*);

struct Rccl {
    void* dl = nullptr;
    get_unique_id_t get_unique_id = nullptr;
    comm_init_rank_t comm_init_rank = nullptr;
    all_reduce_t all_reduce = nullptr;
    reduce_scatter_t reduce_scatter = nullptr;
    all_gather_t all_gather = nullptr;
    comm_destroy_t comm_destroy = nullptr;
    get_error_string_t get_error_string = nullptr;
    get_version_t get_version = nullptr;
} g;
std::mutex g_mu;
std::vector<void*> g_comms;    // handle = index

const char* err(int rc) { return g.get_error_string ? g.get_error_string(rc) : "?"; }

}  // namespace

extern "C" {

// path: the RCCL shared object to bind (NULL: "librccl.so.1" by the loader's search).  Returns the library's version code.
int san_rccl_load(const char* path, int* version) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.dl) {
        const char* p = path && path[0] ? path : "librccl.so.1";
        void* dl = dlopen(p, RTLD_NOW | RTLD_NOLOAD);           // the copy the process already has, if any
        if (!dl) dl = dlopen(p, RTLD_NOW | RTLD_LOCAL);
        if (!dl) {
            san_set_error("san_rccl_load: cannot open %s: %s", p, dlerror());
            return SAN_E_UNSUPPORTED;
        }
        g.get_unique_id = (get_unique_id_t)dlsym(dl, "ncclGetUniqueId");
        g.comm_init_rank = (comm_init_rank_t)dlsym(dl, "ncclCommInitRank");
        g.all_reduce = (all_reduce_t)dlsym(dl, "ncclAllReduce");
        g.reduce_scatter = (reduce_scatter_t)dlsym(dl, "ncclReduceScatter");
        g.all_gather = (all_gather_t)dlsym(dl, "ncclAllGather");
        g.comm_destroy = (comm_destroy_t)dlsym(dl, "ncclCommDestroy");
        g.get_error_string = (get_error_string_t)dlsym(dl, "ncclGetErrorString");
        g.get_version = (get_version_t)dlsym(dl, "ncclGetVersion");
        if (!g.get_unique_id || !g.comm_init_rank || !g.all_reduce || !g.comm_destroy) {
            san_set_error("san_rccl_load: %s lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy", p);
            return SAN_E_UNSUPPORTED;
        }
        g.dl = dl;
    }
    if (version) {
        *version = 0;
        if (g.get_version) g.get_version(version);
    }
    return SAN_OK;
}

// 128 bytes identifying a new communicator (rank 0 makes it, every rank gets it through torch.distributed)
int san_rccl_unique_id(void* id128) {
    SAN_CHECK_ARG(id128, "null pointer");
    SAN_CHECK_ARG(g.dl, "san_rccl_load first");
    const int rc = g.get_unique_id(static_cast<UniqueId*>(id128));
    if (rc != 0) {
        san_set_error("ncclGetUniqueId: %s", err(rc));
        return SAN_E_UNSUPPORTED;
    }
    return SAN_OK;
}

// Collective over the ranks of the new communicator (blocks until all of them called it); the calling thread's current HIP
// device is the rank's GPU.  *handle identifies the communicator in the calls below.
int san_rccl_comm_init(const void* id128, int world, int rank, int* handle) {
    SAN_CHECK_ARG(id128 && handle, "null pointer");
    SAN_CHECK_ARG(g.dl, "san_rccl_load first");
    SAN_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    UniqueId id = *static_cast<const UniqueId*>(id128);
    void* comm = nullptr;
    const int rc = g.comm_init_rank(&comm, world, id, rank);
    if (rc != 0 || !comm) {
        san_set_error("ncclCommInitRank(world %d, rank %d): %s", world, rank, err(rc));
        return SAN_E_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_comms.push_back(comm);
    *handle = (int)g_comms.size() - 1;
    return SAN_OK;
}

// In-place sum of buf[0 .. count) over the communicator's ranks, enqueued on `stream` (stream-ordered like a kernel launch).
// Replaces torch.distributed.all_reduce(flat_gradients) inside the training step (the reference has no data parallelism:
// SURVEY section 8(e); the 1 / world factor rides in the AdamW kernel).
int san_rccl_allreduce_sum_f32(int handle, float* buf, size_t count, void* stream) {
    SAN_CHECK_ARG(buf, "null pointer");
    void* comm = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (handle >= 0 && handle < (int)g_comms.size()) comm = g_comms[handle];
    }
    SAN_CHECK_ARG(comm, "no such communicator");
    if (count == 0) return SAN_OK;
    const int rc = g.all_reduce(buf, buf, count, /* ncclFloat32 */ 7, /* ncclSum */ 0, comm, (hipStream_t)stream);
    if (rc != 0) {
        san_set_error("ncclAllReduce(%zu floats): %s", count, err(rc));
        return SAN_E_UNSUPPORTED;
    }
    return SAN_OK;
}

static void* comm_of(int handle) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (handle >= 0 && handle < (int)g_comms.size()) ? g_comms[handle] : nullptr;
}

// recv[0 .. recv_count) = sum over ranks of their send[rank * recv_count ..): the first half of the two-phase exchange
int san_rccl_reduce_scatter_sum_f32(int handle, const float* send, float* recv, size_t recv_count, void* stream) {
    SAN_CHECK_ARG(send && recv, "null pointer");
    void* comm = comm_of(handle);
    SAN_CHECK_ARG(comm, "no such communicator");
    SAN_CHECK_ARG(g.reduce_scatter, "the bound RCCL has no ncclReduceScatter");
    if (recv_count == 0) return SAN_OK;
    const int rc = g.reduce_scatter(send, recv, recv_count, /* ncclFloat32 */ 7, /* ncclSum */ 0, comm, (hipStream_t)stream);
    if (rc != 0) {
        san_set_error("ncclReduceScatter(%zu floats per rank): %s", recv_count, err(rc));
        return SAN_E_UNSUPPORTED;
    }
    return SAN_OK;
}

// recv[r * send_count ..) = rank r's send[0 .. send_count): the second half
int san_rccl_allgather_f32(int handle, const float* send, float* recv, size_t send_count, void* stream) {
    SAN_CHECK_ARG(send && recv, "null pointer");
    void* comm = comm_of(handle);
    SAN_CHECK_ARG(comm, "no such communicator");
    SAN_CHECK_ARG(g.all_gather, "the bound RCCL has no ncclAllGather");
    if (send_count == 0) return SAN_OK;
    const int rc = g.all_gather(send, recv, send_count, /* ncclFloat32 */ 7, comm, (hipStream_t)stream);
    if (rc != 0) {
        san_set_error("ncclAllGather(%zu floats per rank): %s", send_count, err(rc));
        return SAN_E_UNSUPPORTED;
    }
    return SAN_OK;
}

int san_rccl_comm_destroy(int handle) {
    void* comm = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (handle >= 0 && handle < (int)g_comms.size()) {
            comm = g_comms[handle];
            g_comms[handle] = nullptr;
        }
    }
    if (comm && g.comm_destroy) g.comm_destroy(comm);
    return SAN_OK;
}

}  // extern "C"
