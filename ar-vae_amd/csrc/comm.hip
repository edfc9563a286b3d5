// Data-parallel collectives of the training step, owned by the library: RCCL on the caller's launch stream.
//
// The reference is single-process; what this file adds is the exchange a row-sharded step needs:
// an all-gather of the regularised latent / label columns before the all-pairs term (utils/trainer.py:369-403 averages over
// ALL pairs of the batch) and one SUM all-reduce of the flat gradient arena before Adam (utils/trainer.py:140-147,170-174).
// Issued from here they are ordinary stream work: no Work objects, no watchdog thread polling events, capturable into the
// same HIP graph as the kernels around them (ar-vae_amd/graphed.py).
//
// RCCL is resolved at run time (dlopen / dlsym): a process that already holds an RCCL (PyTorch-ROCm ships its own
// librccl.so.1 next to its HIP runtime) gets THAT copy, so one RCCL and one HIP runtime serve the process; single-GPU
// users never load it.  The header <rccl/rccl.h> is used for its types only; a host without RCCL's development files (a
// single-GPU or cross-compile box) builds the same library from the handful of declarations below -- RCCL's ABI for the entry
// points bound here (NCCL 2.x: opaque ncclComm_t, 128-byte unique id, the enum values of nccl.h).
#include <dlfcn.h>
#include <string.h>

#include <chrono>
#include <future>
#include <memory>
#include <mutex>
#include <thread>

#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7,
               ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
}
#endif

#include "common.h"

namespace arvae {
namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    char why[256] = "";
};

std::once_flag g_rccl_once;
Rccl g_rccl;

template <typename F> bool bind(void *h, const char *name, F &fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

void load_rccl() {
    Rccl &r = g_rccl;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) {                   // the copy the process already holds (PyTorch's), if any
        r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (r.handle != nullptr) break;
    }
    for (int i = 0; r.handle == nullptr && i < 2; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (r.handle == nullptr) {
        snprintf(r.why, sizeof(r.why), "librccl.so.1 not found: %s", dlerror());
        return;
    }
    int missing = 0;
    missing += !bind(r.handle, "ncclGetVersion", r.GetVersion);
    missing += !bind(r.handle, "ncclGetUniqueId", r.GetUniqueId);
    missing += !bind(r.handle, "ncclCommInitRank", r.CommInitRank);
    missing += !bind(r.handle, "ncclCommDestroy", r.CommDestroy);
    missing += !bind(r.handle, "ncclCommAbort", r.CommAbort);
    missing += !bind(r.handle, "ncclCommGetAsyncError", r.CommGetAsyncError);
    missing += !bind(r.handle, "ncclGetErrorString", r.GetErrorString);
    missing += !bind(r.handle, "ncclAllReduce", r.AllReduce);
    missing += !bind(r.handle, "ncclAllGather", r.AllGather);
    missing += !bind(r.handle, "ncclBroadcast", r.Broadcast);
    missing += !bind(r.handle, "ncclGroupStart", r.GroupStart);
    missing += !bind(r.handle, "ncclGroupEnd", r.GroupEnd);
    if (missing != 0) {
        snprintf(r.why, sizeof(r.why), "librccl.so.1 lacks an entry point this library binds");
        r.handle = nullptr;
    }
}

const Rccl *rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.handle != nullptr ? &g_rccl : nullptr;
}

struct Comm {
    uint32_t magic;
    ncclComm_t comm;
    int rank, world;
};
constexpr uint32_t COMM_MAGIC = 0x41524343u;       // "ARCC"

Comm *as_comm(arvae_comm_t c) {
    Comm *p = reinterpret_cast<Comm *>(c);
    return (p != nullptr && p->magic == COMM_MAGIC) ? p : nullptr;
}

int nccl_fail(const Rccl *r, const char *what, ncclResult_t e) {
    return fail(ARVAE_E_COMM, "%s: %s", what, r->GetErrorString(e));
}

bool dtype_of(int32_t dtype, ncclDataType_t &t) {
    switch (dtype) {
        case ARVAE_COMM_F32: t = ncclFloat32; return true;
        case ARVAE_COMM_F64: t = ncclFloat64; return true;
        case ARVAE_COMM_I64: t = ncclInt64; return true;
        case ARVAE_COMM_U8: t = ncclUint8; return true;
    }
    return false;
}

}  // namespace
}  // namespace arvae

using namespace arvae;

extern "C" int arvae_comm_available(void) {
    const Rccl *r = rccl();
    if (r == nullptr) return fail(ARVAE_E_COMM, "%s", g_rccl.why);
    int v = 0;
    if (r->GetVersion(&v) != ncclSuccess) return fail(ARVAE_E_COMM, "ncclGetVersion failed");
    return v;
}

extern "C" int arvae_comm_unique_id(void *id_out) {
    const Rccl *r = rccl();
    if (r == nullptr) return fail(ARVAE_E_COMM, "%s", g_rccl.why);
    if (id_out == nullptr) return fail(ARVAE_E_INVALID, "arvae_comm_unique_id: null id");
    static_assert(sizeof(ncclUniqueId) == ARVAE_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    const ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId", e);
    memcpy(id_out, &id, sizeof(id));
    return ARVAE_OK;
}

// ncclCommInitRank is a collective: it returns when EVERY rank of the job is in it.  A rank that died after the unique id went
// out would leave the others in it forever, so the call runs on a helper thread (bound to the caller's device) and this thread
// waits for it with a deadline.  After a timeout the helper is still inside RCCL and cannot be cancelled: the call returns
// ARVAE_E_COMM and the PROCESS must exit (the launcher ends the job); nothing of the half-built communicator is touched again.
extern "C" int arvae_comm_init(const void *id_bytes, int32_t rank, int32_t world, int32_t timeout_ms, arvae_comm_t *out) {
    const Rccl *r = rccl();
    if (r == nullptr) return fail(ARVAE_E_COMM, "%s", g_rccl.why);
    if (id_bytes == nullptr || out == nullptr || world < 1 || rank < 0 || rank >= world || timeout_ms < 0)
        return fail(ARVAE_E_INVALID, "arvae_comm_init: bad arguments (rank %d of %d)", rank, world);
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return fail(ARVAE_E_NODEVICE, "arvae_comm_init: no current HIP device");
    struct Result { ncclComm_t comm = nullptr; ncclResult_t rc = ncclSuccess; hipError_t hip = hipSuccess; };
    auto result = std::make_shared<Result>();                       // outlives this frame if the helper never returns
    auto done = std::make_shared<std::promise<void>>();
    std::future<void> ready = done->get_future();
    std::thread([=] {
        result->hip = hipSetDevice(device);
        if (result->hip == hipSuccess) result->rc = r->CommInitRank(&result->comm, world, id, rank);
        done->set_value();
    }).detach();
    if (timeout_ms == 0) ready.wait();                               // 0: no deadline (the caller has its own)
    else if (ready.wait_for(std::chrono::milliseconds(timeout_ms)) != std::future_status::ready)
        return fail(ARVAE_E_COMM, "ncclCommInitRank: rank %d of %d waited %d ms for the other ranks; the job cannot continue "
                                  "(this process must exit)", rank, world, timeout_ms);
    if (result->hip != hipSuccess) return fail(ARVAE_E_LAUNCH, "arvae_comm_init: hipSetDevice(%d) failed on the helper thread", device);
    if (result->rc != ncclSuccess) return nccl_fail(r, "ncclCommInitRank", result->rc);
    Comm *p = new Comm{COMM_MAGIC, result->comm, rank, world};
    *out = reinterpret_cast<arvae_comm_t>(p);
    return ARVAE_OK;
}

extern "C" int arvae_comm_destroy(arvae_comm_t comm) {
    Comm *p = as_comm(comm);
    if (p == nullptr) return fail(ARVAE_E_INVALID, "arvae_comm_destroy: not a communicator");
    const Rccl *r = rccl();
    const ncclResult_t e = r->CommDestroy(p->comm);
    p->magic = 0;
    delete p;
    return e == ncclSuccess ? ARVAE_OK : nccl_fail(r, "ncclCommDestroy", e);
}

extern "C" int arvae_comm_abort(arvae_comm_t comm) {
    Comm *p = as_comm(comm);
    if (p == nullptr) return fail(ARVAE_E_INVALID, "arvae_comm_abort: not a communicator");
    const Rccl *r = rccl();
    const ncclResult_t e = r->CommAbort(p->comm);
    p->magic = 0;
    delete p;
    return e == ncclSuccess ? ARVAE_OK : nccl_fail(r, "ncclCommAbort", e);
}

extern "C" int arvae_comm_rank(arvae_comm_t comm) {
    Comm *p = as_comm(comm);
    return p != nullptr ? p->rank : fail(ARVAE_E_INVALID, "not a communicator");
}

extern "C" int arvae_comm_world(arvae_comm_t comm) {
    Comm *p = as_comm(comm);
    return p != nullptr ? p->world : fail(ARVAE_E_INVALID, "not a communicator");
}

extern "C" int arvae_comm_async_error(arvae_comm_t comm) {
    Comm *p = as_comm(comm);
    if (p == nullptr) return fail(ARVAE_E_INVALID, "arvae_comm_async_error: not a communicator");
    const Rccl *r = rccl();
    ncclResult_t state = ncclSuccess;
    const ncclResult_t e = r->CommGetAsyncError(p->comm, &state);
    if (e != ncclSuccess) return nccl_fail(r, "ncclCommGetAsyncError", e);
    if (state != ncclSuccess && state != ncclInProgress) return nccl_fail(r, "communicator", state);
    return ARVAE_OK;
}

extern "C" int arvae_comm_all_gather(arvae_comm_t comm, const void *send, void *recv, int64_t count, int32_t dtype,
                                     arvae_stream_t stream) {
    Comm *p = as_comm(comm);
    ncclDataType_t t;
    if (p == nullptr || !dtype_of(dtype, t) || count < 0 || (count > 0 && (send == nullptr || recv == nullptr)))
        return fail(ARVAE_E_INVALID, "arvae_comm_all_gather: bad arguments");
    if (count == 0) return ARVAE_OK;
    const Rccl *r = rccl();
    const ncclResult_t e = r->AllGather(send, recv, (size_t)count, t, p->comm, as_stream(stream));
    return e == ncclSuccess ? ARVAE_OK : nccl_fail(r, "ncclAllGather", e);
}

extern "C" int arvae_comm_all_reduce(arvae_comm_t comm, void *buf, int64_t count, int32_t dtype, int32_t op,
                                     arvae_stream_t stream) {
    Comm *p = as_comm(comm);
    ncclDataType_t t;
    if (p == nullptr || !dtype_of(dtype, t) || count < 0 || (count > 0 && buf == nullptr) ||
        (op != ARVAE_COMM_SUM && op != ARVAE_COMM_MAX && op != ARVAE_COMM_MIN))
        return fail(ARVAE_E_INVALID, "arvae_comm_all_reduce: bad arguments");
    if (count == 0) return ARVAE_OK;
    const Rccl *r = rccl();
    const ncclRedOp_t o = op == ARVAE_COMM_SUM ? ncclSum : (op == ARVAE_COMM_MAX ? ncclMax : ncclMin);
    const ncclResult_t e = r->AllReduce(buf, buf, (size_t)count, t, o, p->comm, as_stream(stream));
    return e == ncclSuccess ? ARVAE_OK : nccl_fail(r, "ncclAllReduce", e);
}

extern "C" int arvae_comm_broadcast(arvae_comm_t comm, void *buf, int64_t count, int32_t dtype, int32_t root,
                                    arvae_stream_t stream) {
    Comm *p = as_comm(comm);
    ncclDataType_t t;
    if (p == nullptr || !dtype_of(dtype, t) || count < 0 || (count > 0 && buf == nullptr) || root < 0 || root >= p->world)
        return fail(ARVAE_E_INVALID, "arvae_comm_broadcast: bad arguments");
    if (count == 0) return ARVAE_OK;
    const Rccl *r = rccl();
    const ncclResult_t e = r->Broadcast(buf, buf, (size_t)count, t, root, p->comm, as_stream(stream));
    return e == ncclSuccess ? ARVAE_OK : nccl_fail(r, "ncclBroadcast", e);
}

// Several collectives as ONE RCCL launch (ncclGroupStart / ncclGroupEnd): the z and label all-gathers of a step.
extern "C" int arvae_comm_group_begin(void) {
    const Rccl *r = rccl();
    if (r == nullptr) return fail(ARVAE_E_COMM, "%s", g_rccl.why);
    const ncclResult_t e = r->GroupStart();
    return e == ncclSuccess ? ARVAE_OK : nccl_fail(r, "ncclGroupStart", e);
}

extern "C" int arvae_comm_group_end(void) {
    const Rccl *r = rccl();
    if (r == nullptr) return fail(ARVAE_E_COMM, "%s", g_rccl.why);
    const ncclResult_t e = r->GroupEnd();
    return e == ncclSuccess ? ARVAE_OK : nccl_fail(r, "ncclGroupEnd", e);
}
