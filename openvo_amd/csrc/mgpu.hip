// Multi-GPU exchange of the path: frame pairs shard across the GPUs of one node, one process
// per GPU, and the ONLY exchange is the gather of the relative poses at the end of a batch -- 17 float64 per
// frame.  This file binds RCCL directly (librccl.so.1 is loaded at run time with dlopen, so single-GPU users
// never need it): ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclAllReduce on a HIP stream of its
// own.  The caller moves the 128-byte unique id from rank 0 to the other ranks (openvo_amd/sharding.py does it
// over a TCP socket on the node); no PyTorch, no MPI.
// The reference has no counterpart (single process, single thread): include/vo355.h, "multi-GPU".
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include "../../include/vo355.h"

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { NCCL_MAX = 2, NCCL_FLOAT64 = 8 };

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int*) = nullptr;
    int (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

static RcclApi g_rccl;
static std::string g_mgpu_err;

static bool rccl_load()
{
    if (g_rccl.handle) return true;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) { g_mgpu_err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define BIND(field, sym)                                                        \
    *(void**)(&g_rccl.field) = dlsym(g_rccl.handle, sym);                       \
    if (!g_rccl.field) { g_mgpu_err = std::string("librccl lacks ") + sym; dlclose(g_rccl.handle); g_rccl.handle = nullptr; return false; }
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(CommCount, "ncclCommCount");
    BIND(CommUserRank, "ncclCommUserRank");
    BIND(AllGather, "ncclAllGather");
    BIND(AllReduce, "ncclAllReduce");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    return true;
}

struct vo_mgpu {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double* d_send = nullptr;
    double* d_recv = nullptr;
    size_t cap = 0;        // doubles per rank the device buffers hold
    std::string err;
};

static int mg_fail(vo_mgpu* g, int code, const std::string& msg)
{
    if (g) g->err = msg;
    g_mgpu_err = msg;
    return code;
}

#define MG_HIP(g, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e__ = (call);                                                                          \
        if (e__ != hipSuccess) return mg_fail(g, VO_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define MG_NCCL(g, call)                                                                                  \
    do {                                                                                                  \
        int r__ = (call);                                                                                 \
        if (r__ != 0) return mg_fail(g, VO_E_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r__));  \
    } while (0)

extern "C" const char* vo_mgpu_last_error(const vo_mgpu* g) { return g ? g->err.c_str() : g_mgpu_err.c_str(); }

extern "C" int vo_device_count(int* n_out)
{
    if (!n_out) return VO_E_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *n_out = n;
    return VO_OK;
}

extern "C" int vo_mgpu_unique_id(uint8_t* id128)
{
    if (!id128) return VO_E_ARG;
    if (!rccl_load()) return VO_E_STATE;
    ncclUniqueId id;
    MG_NCCL(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return VO_OK;
}

extern "C" int vo_mgpu_create(int device, int rank, int world, const uint8_t* id128, vo_mgpu** out)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return VO_E_ARG;
    *out = nullptr;
    if (!rccl_load()) return VO_E_STATE;
    vo_mgpu* g = new vo_mgpu();
    g->device = device; g->rank = rank; g->world = world;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_mgpu_err = hipGetErrorString(e); delete g; return VO_E_HIP; }
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    int r = g_rccl.CommInitRank(&g->comm, world, id, rank);
    if (r != 0) {
        g_mgpu_err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r);
        (void)hipStreamDestroy(g->stream);
        delete g;
        return VO_E_HIP;
    }
    *out = g;
    return VO_OK;
}

extern "C" void vo_mgpu_destroy(vo_mgpu* g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->comm) (void)g_rccl.CommDestroy(g->comm);
    if (g->d_send) (void)hipFree(g->d_send);
    if (g->d_recv) (void)hipFree(g->d_recv);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

// what the communicator itself reports: number of ranks and this process's rank in it (ncclCommCount / ncclCommUserRank)
extern "C" int vo_mgpu_info(vo_mgpu* g, int* n_ranks, int* user_rank, int* device)
{
    if (!g || !n_ranks || !user_rank || !device) return mg_fail(g, VO_E_ARG, "vo_mgpu_info: bad argument");
    MG_NCCL(g, g_rccl.CommCount(g->comm, n_ranks));
    MG_NCCL(g, g_rccl.CommUserRank(g->comm, user_rank));
    *device = g->device;
    return VO_OK;
}

static int mg_reserve(vo_mgpu* g, size_t n)
{
    if (n <= g->cap) return VO_OK;
    if (g->d_send) (void)hipFree(g->d_send);
    if (g->d_recv) (void)hipFree(g->d_recv);
    g->d_send = g->d_recv = nullptr; g->cap = 0;
    MG_HIP(g, hipMalloc((void**)&g->d_send, n * sizeof(double)));
    MG_HIP(g, hipMalloc((void**)&g->d_recv, n * g->world * sizeof(double)));
    g->cap = n;
    return VO_OK;
}

// all ranks contribute n float64 (n identical on every rank); all receive world*n in rank order
extern "C" int vo_mgpu_all_gather_f64(vo_mgpu* g, const double* local, int n, double* all)
{
    if (!g || !local || !all || n <= 0) return mg_fail(g, VO_E_ARG, "vo_mgpu_all_gather_f64: bad argument");
    MG_HIP(g, hipSetDevice(g->device));
    int rc = mg_reserve(g, (size_t)n);
    if (rc) return rc;
    MG_HIP(g, hipMemcpyAsync(g->d_send, local, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g->stream));
    MG_NCCL(g, g_rccl.AllGather(g->d_send, g->d_recv, (size_t)n, NCCL_FLOAT64, g->comm, g->stream));
    MG_HIP(g, hipMemcpyAsync(all, g->d_recv, (size_t)n * g->world * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    MG_HIP(g, hipStreamSynchronize(g->stream));
    return VO_OK;
}

// the path's exchange: n frames x 17 float64 (row-major 4x4 relative transform + accept flag) per rank
extern "C" int vo_mgpu_gather_poses(vo_mgpu* g, const double* local_n17, int n_frames, double* all_n17)
{
    return vo_mgpu_all_gather_f64(g, local_n17, n_frames * 17, all_n17);
}

// element-wise max over ranks, in place (the bench's "slowest rank" time); doubles as a barrier
extern "C" int vo_mgpu_all_reduce_max_f64(vo_mgpu* g, double* v, int n)
{
    if (!g || !v || n <= 0) return mg_fail(g, VO_E_ARG, "vo_mgpu_all_reduce_max_f64: bad argument");
    MG_HIP(g, hipSetDevice(g->device));
    int rc = mg_reserve(g, (size_t)n);
    if (rc) return rc;
    MG_HIP(g, hipMemcpyAsync(g->d_send, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, g->stream));
    MG_NCCL(g, g_rccl.AllReduce(g->d_send, g->d_recv, (size_t)n, NCCL_FLOAT64, NCCL_MAX, g->comm, g->stream));
    MG_HIP(g, hipMemcpyAsync(v, g->d_recv, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, g->stream));
    MG_HIP(g, hipStreamSynchronize(g->stream));
    return VO_OK;
}
