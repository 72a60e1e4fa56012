// emi_comm.cpp -- the one collective of the path: the gather of solved trajectories to one rank over
// RCCL (xGMI inside a node), behind the C ABI (include/emi355x.h, emi_comm_*).
//
// instances shard over the 8 GPUs of a node with no communication while they are
// evaluated or solved; afterwards every rank hands its block of results (a few MB) to the root.  On the
// xGMI full mesh that is one direct transfer per peer, so the gather is a single group of point-to-point
// ncclSend / ncclRecv calls (no ring, no staging through other GPUs): latency-bound, one call per batch.
//
// librccl is opened at the first emi_comm_* call (dlopen), not linked: libemi355x.so then loads on hosts
// without RCCL, and inside a PyTorch process the RCCL already loaded there is the one that gets used.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include <type_traits>

#include "emi355x.h"

// The prototypes of the eight RCCL entry points this file calls.  Where <rccl/rccl.h> is installed (this image: ROCm 7.2)
// the real header is included and every function-pointer type of the table below is CHECKED against it at compile time;
// where it is not, the declarations that follow stand in (opaque handle, 128-byte id, int enums: the ABI-stable part).
// librccl itself is never linked: decltype(&ncclSend) names a type, not a symbol.
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define EMI_HAVE_RCCL_H 1
#endif
#endif
#ifndef EMI_HAVE_RCCL_H
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0 };
#endif

namespace {

static_assert(sizeof(ncclUniqueId) == EMI_COMM_ID_BYTES, "id size");
static_assert((int)ncclSuccess == 0 && (int)ncclInt8 == 0 && sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int),
              "RCCL enum values / sizes this file relies on");

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
#ifdef EMI_HAVE_RCCL_H
#define EMI_SAME(member, fn) static_assert(std::is_same<decltype(Rccl::member), decltype(&fn)>::value, #fn ": prototype differs from rccl.h")
EMI_SAME(GetUniqueId, ncclGetUniqueId);
EMI_SAME(CommInitRank, ncclCommInitRank);
EMI_SAME(CommDestroy, ncclCommDestroy);
EMI_SAME(GroupStart, ncclGroupStart);
EMI_SAME(GroupEnd, ncclGroupEnd);
EMI_SAME(Send, ncclSend);
EMI_SAME(Recv, ncclRecv);
EMI_SAME(GetErrorString, ncclGetErrorString);
#undef EMI_SAME
#endif

Rccl* rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (R.h) break;
        }
        if (!R.h) { R.err = std::string("librccl not found: ") + dlerror(); return; }
        auto sym = [&](const char* n) { void* p = dlsym(R.h, n); if (!p) R.err = std::string("librccl lacks ") + n; return p; };
        R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
        R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
        R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
        R.Send = (decltype(R.Send))sym("ncclSend");
        R.Recv = (decltype(R.Recv))sym("ncclRecv");
        R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    });
    return R.err.empty() ? &R : nullptr;
}

thread_local std::string g_err;   // errors of calls that have no handle yet

}  // namespace

struct emi_comm_s {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    std::string err;
};

namespace {
int cfail(emi_comm_t c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    (c ? c->err : g_err) = buf;
    return code;
}
}  // namespace

extern "C" {

int emi_comm_unique_id(void* id) {
    if (!id) return cfail(nullptr, EMI_ERR_ARG, "emi_comm_unique_id: null id");
    Rccl* R = rccl();
    if (!R) return cfail(nullptr, EMI_ERR_COMM, "librccl unavailable");
    ncclUniqueId u;
    const ncclResult_t st = R->GetUniqueId(&u);
    if (st != ncclSuccess) return cfail(nullptr, EMI_ERR_COMM, "ncclGetUniqueId: %s", R->GetErrorString(st));
    memcpy(id, &u, sizeof u);
    return EMI_OK;
}

int emi_comm_create(int device_id, int world, int rank, const void* id, emi_comm_t* out) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return cfail(nullptr, EMI_ERR_ARG, "emi_comm_create: bad argument");
    Rccl* R = rccl();
    if (!R) return cfail(nullptr, EMI_ERR_COMM, "librccl unavailable");
    if (hipSetDevice(device_id) != hipSuccess) return cfail(nullptr, EMI_ERR_NO_DEVICE, "emi_comm_create: no device %d", device_id);
    emi_comm_t c = new emi_comm_s;
    c->world = world; c->rank = rank; c->device = device_id;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const ncclResult_t st = R->CommInitRank(&c->comm, world, u, rank);
    if (st != ncclSuccess) {
        cfail(nullptr, EMI_ERR_COMM, "ncclCommInitRank(world %d, rank %d): %s", world, rank, R->GetErrorString(st));
        delete c;
        return EMI_ERR_COMM;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        R->CommDestroy(c->comm);
        delete c;
        return cfail(nullptr, EMI_ERR_HIP, "emi_comm_create: stream");
    }
    *out = c;
    return EMI_OK;
}

int emi_comm_gather(emi_comm_t c, const void* dsend, void* drecv, size_t bytes, int root, void* hip_stream) {
    if (!c || root < 0 || root >= c->world || (bytes && !dsend)) return cfail(c, EMI_ERR_ARG, "emi_comm_gather: bad argument");
    if (c->rank == root && bytes && !drecv) return cfail(c, EMI_ERR_ARG, "emi_comm_gather: the root needs a receive buffer");
    Rccl* R = rccl();
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (hipSetDevice(c->device) != hipSuccess) return cfail(c, EMI_ERR_HIP, "emi_comm_gather: hipSetDevice");
    if (bytes == 0) return EMI_OK;
    if (c->rank == root &&
        hipMemcpyAsync((char*)drecv + (size_t)root * bytes, dsend, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return cfail(c, EMI_ERR_HIP, "emi_comm_gather: local copy");
    ncclResult_t st = R->GroupStart();
    if (st == ncclSuccess && c->rank == root) {
        for (int r = 0; r < c->world && st == ncclSuccess; ++r)
            if (r != root) st = R->Recv((char*)drecv + (size_t)r * bytes, bytes, ncclInt8, r, c->comm, s);
    } else if (st == ncclSuccess) {
        st = R->Send(dsend, bytes, ncclInt8, root, c->comm, s);
    }
    const ncclResult_t st2 = R->GroupEnd();
    if (st == ncclSuccess) st = st2;
    if (st != ncclSuccess) return cfail(c, EMI_ERR_COMM, "RCCL gather: %s", R->GetErrorString(st));
    if (!hip_stream && hipStreamSynchronize(s) != hipSuccess) return cfail(c, EMI_ERR_HIP, "emi_comm_gather: synchronize");
    return EMI_OK;
}

int emi_comm_destroy(emi_comm_t c) {
    if (!c) return EMI_ERR_ARG;
    Rccl* R = rccl();
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (R && c->comm) R->CommDestroy(c->comm);
    delete c;
    return EMI_OK;
}

const char* emi_comm_last_error(emi_comm_t c) { return c ? c->err.c_str() : g_err.c_str(); }

}  // extern "C"
