// gw_comm.cpp -- the one exchange step of the sharded scan: SNP rows shard over the GPUs of a node
// by contiguous ranges, one process per GPU, and the per-SNP records are gathered to one rank over RCCL / xGMI.
// This is the C-ABI side of it (include/gwas_scan.h "multi-GPU"), so that a C++ host (libgwaspp_amd/host,
// gwas_hip --world/--rank) shards without any Python: librccl is loaded at run time (dlopen), a build or a
// single-GPU user never needs it.  The gather is a grouped ncclSend / ncclRecv -- every peer's slice travels over its
// own xGMI link straight to the root (a ring all-gather would serialise through single links, and nobody but
// the root wants the records).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only; no link-time dependency

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/gwas_scan.h"

extern "C" int gw_set_error_message(int code, const char *msg);  // gw_capi.cpp: the calling thread's gw_last_error()

namespace {

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return gw_set_error_message(code, buf);
}

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        // a process that already holds an RCCL (torch ships its own librccl.so) keeps using that one: same SONAME.
        // GW_RCCL_LIBRARY names the one file to load instead (deployments with RCCL elsewhere; the loader's own test)
        std::vector<std::string> names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (const char *only = std::getenv("GW_RCCL_LIBRARY")) names.assign(1, only);
        std::string tried;
        for (const std::string &name : names) {
            r.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
            const char *e = dlerror();  // read once: the call clears the message
            tried += (tried.empty() ? "" : "; ") + name + ": " + (e ? e : "?");
        }
        if (!r.lib) {
            r.why = "librccl not loadable (" + tried + ")";
            return;
        }
#define GW_SYM(field, sym)                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, #sym));            \
    if (!r.field && r.why.empty()) r.why = "librccl.so lacks " #sym;
        GW_SYM(GetUniqueId, ncclGetUniqueId)
        GW_SYM(CommInitRank, ncclCommInitRank)
        GW_SYM(CommDestroy, ncclCommDestroy)
        GW_SYM(GroupStart, ncclGroupStart)
        GW_SYM(GroupEnd, ncclGroupEnd)
        GW_SYM(Send, ncclSend)
        GW_SYM(Recv, ncclRecv)
        GW_SYM(AllGather, ncclAllGather)
        GW_SYM(Broadcast, ncclBroadcast)
        GW_SYM(GetErrorString, ncclGetErrorString)
#undef GW_SYM
    });
    return r;
}

#define GW_NCCL(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r__ = (expr);                                                                 \
        if (r__ != ncclSuccess) return fail(GW_E_HIP, "%s: %s", #expr, rccl().GetErrorString(r__)); \
    } while (0)
#define GW_HIPC(expr)                                                                       \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            (void)hipGetLastError();                                                        \
            return fail(e__ == hipErrorOutOfMemory ? GW_E_NOMEM : GW_E_HIP, "%s: %s", #expr, \
                        hipGetErrorString(e__));                                            \
        }                                                                                   \
    } while (0)

}  // namespace

struct gw_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, world = 1;
    hipStream_t stream = nullptr;  // used when the caller passes no stream
};

extern "C" {

int gw_shard_range(uint64_t n_snps_total, int rank, int world, uint64_t *begin, uint64_t *end) {
    if (!begin || !end) return fail(GW_E_INVALID, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(GW_E_INVALID, "rank %d outside a world of %d", rank, world);
    const uint64_t base = n_snps_total / (uint64_t)world, extra = n_snps_total % (uint64_t)world;
    const uint64_t r = (uint64_t)rank;
    *begin = r * base + (r < extra ? r : extra);
    *end = *begin + base + (r < extra ? 1u : 0u);
    return GW_OK;
}

int gw_comm_new_id(void *id) {
    static_assert(GW_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!id) return fail(GW_E_INVALID, "id is NULL");
    Rccl &r = rccl();
    if (!r.why.empty()) return fail(GW_E_HIP, "%s", r.why.c_str());
    ncclUniqueId u;
    GW_NCCL(r.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return GW_OK;
}

int gw_comm_create(int device, int rank, int world, const void *id, gw_comm **out) {
    if (!out || !id) return fail(GW_E_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(GW_E_INVALID, "rank %d outside a world of %d", rank, world);
    Rccl &r = rccl();
    if (!r.why.empty()) return fail(GW_E_HIP, "%s", r.why.c_str());
    GW_HIPC(hipSetDevice(device));
    (void)hipGetLastError();  // RCCL reads the runtime's sticky "last error": a stale one would fail its init
    gw_comm *c = new (std::nothrow) gw_comm();
    if (!c) return fail(GW_E_NOMEM, "out of host memory");
    c->device = device;
    c->rank = rank;
    c->world = world;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclResult_t rc = r.CommInitRank(&c->comm, world, u, rank);
    if (rc != ncclSuccess) {
        delete c;
        return fail(GW_E_HIP, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device, r.GetErrorString(rc));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        (void)r.CommDestroy(c->comm);
        delete c;
        return fail(GW_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = c;
    return GW_OK;
}

int gw_comm_destroy(gw_comm *c) {
    if (!c) return GW_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
    return GW_OK;
}

int gw_comm_gather(gw_comm *c, const void *d_send, size_t bytes, void *d_recv, int root, void *hip_stream) {
    if (!c || (!d_send && bytes)) return fail(GW_E_INVALID, "NULL argument");
    if (root < 0 || root >= c->world) return fail(GW_E_INVALID, "root %d outside a world of %d", root, c->world);
    if (c->rank == root && !d_recv && bytes) return fail(GW_E_INVALID, "the root needs a receive buffer");
    if (bytes == 0) return GW_OK;
    Rccl &r = rccl();
    GW_HIPC(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (c->rank == root) {
        char *dst = static_cast<char *>(d_recv);
        // the root's own slice never touches a link
        GW_HIPC(hipMemcpyAsync(dst + (size_t)root * bytes, d_send, bytes, hipMemcpyDeviceToDevice, s));
        if (c->world > 1) {
            GW_NCCL(r.GroupStart());
            for (int p = 0; p < c->world; ++p)
                if (p != root) {
                    ncclResult_t rc = r.Recv(dst + (size_t)p * bytes, bytes, ncclUint8, p, c->comm, s);
                    if (rc != ncclSuccess) {
                        (void)r.GroupEnd();
                        return fail(GW_E_HIP, "ncclRecv from rank %d: %s", p, r.GetErrorString(rc));
                    }
                }
            GW_NCCL(r.GroupEnd());
        }
    } else {
        GW_NCCL(r.GroupStart());
        ncclResult_t rc = r.Send(d_send, bytes, ncclUint8, root, c->comm, s);
        if (rc != ncclSuccess) {
            (void)r.GroupEnd();
            return fail(GW_E_HIP, "ncclSend to rank %d: %s", root, r.GetErrorString(rc));
        }
        GW_NCCL(r.GroupEnd());
    }
    return GW_OK;
}

int gw_comm_allgather(gw_comm *c, const void *d_send, size_t bytes, void *d_recv, void *hip_stream) {
    if (!c || ((!d_send || !d_recv) && bytes)) return fail(GW_E_INVALID, "NULL argument");
    if (bytes == 0) return GW_OK;
    GW_HIPC(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (c->world == 1) {  // nothing to exchange: the own slice is the whole result
        if (d_recv != d_send) GW_HIPC(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, s));
        return GW_OK;
    }
    GW_NCCL(rccl().AllGather(d_send, d_recv, bytes, ncclUint8, c->comm, s));
    return GW_OK;
}

int gw_comm_broadcast(gw_comm *c, void *d_buf, size_t bytes, int root, void *hip_stream) {
    if (!c || (!d_buf && bytes)) return fail(GW_E_INVALID, "NULL argument");
    if (root < 0 || root >= c->world) return fail(GW_E_INVALID, "root %d outside a world of %d", root, c->world);
    if (bytes == 0 || c->world == 1) return GW_OK;
    GW_HIPC(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    GW_NCCL(rccl().Broadcast(d_buf, d_buf, bytes, ncclUint8, root, c->comm, s));
    return GW_OK;
}

int gw_comm_synchronize(gw_comm *c, void *hip_stream) {
    if (!c) return fail(GW_E_INVALID, "NULL comm");
    GW_HIPC(hipSetDevice(c->device));
    GW_HIPC(hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : c->stream));
    return GW_OK;
}

// ---- device buffers for hosts without a HIP toolchain of their own (the C++ mirror is plain g++)
int gw_device_malloc(int device, size_t bytes, void **d_ptr) {
    if (!d_ptr) return fail(GW_E_INVALID, "NULL argument");
    *d_ptr = nullptr;
    GW_HIPC(hipSetDevice(device));
    if (bytes) GW_HIPC(hipMalloc(d_ptr, bytes));
    return GW_OK;
}

int gw_device_free(int device, void *d_ptr) {
    if (!d_ptr) return GW_OK;
    GW_HIPC(hipSetDevice(device));
    GW_HIPC(hipFree(d_ptr));
    return GW_OK;
}

int gw_device_memset(int device, void *d_ptr, int value, size_t bytes, void *hip_stream) {
    if (!d_ptr && bytes) return fail(GW_E_INVALID, "NULL argument");
    GW_HIPC(hipSetDevice(device));
    if (bytes) GW_HIPC(hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)hip_stream));
    return GW_OK;
}

int gw_device_download(int device, void *host_dst, const void *d_src, size_t bytes, void *hip_stream) {
    if ((!host_dst || !d_src) && bytes) return fail(GW_E_INVALID, "NULL argument");
    GW_HIPC(hipSetDevice(device));
    hipStream_t s = (hipStream_t)hip_stream;
    GW_HIPC(hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, s));
    GW_HIPC(hipStreamSynchronize(s));
    return GW_OK;
}

}  // extern "C"
