// relp_rccl.cpp -- RCCL as the collectives of the native multi-GPU loop (relp_shard_run).
//
// The library is not linked against RCCL: the five entry points are looked up at run time in the
// librccl.so.1 the process already holds (PyTorch's torch.distributed loads one) or, failing that, in the
// one the loader finds.  One communicator per engine, one rank per GPU, everything on the engine's stream.
#include "relp_engine_internal.hpp"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace relp {
namespace {

// the part of RCCL's C API used here (rccl.h: ncclUniqueId is 128 opaque bytes passed by value;
// ncclUint8 = 1, ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0)
struct UniqueId { char internal[RELP_RCCL_ID_BYTES]; };
constexpr int kUint8 = 1, kFloat64 = 8, kSum = 0;
using Comm = void*;

struct Api {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Api& api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (a.lib) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (a.lib) break;
            a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!a.lib) return;
        auto sym = [&](const char* n) { return dlsym(a.lib, n); };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce;
    });
    return a;
}

int rccl_allgather(void* ctx, const void* send, void* recv, int64_t bytes, void* stream) {
    return api().AllGather(send, recv, (size_t)bytes, kUint8, (Comm)ctx, (hipStream_t)stream);
}

int rccl_allreduce_sum(void* ctx, double* buf, int64_t count, void* stream) {
    return api().AllReduce(buf, buf, (size_t)count, kFloat64, kSum, (Comm)ctx, (hipStream_t)stream);
}

}  // namespace

relp_status_t Engine::rccl_attach(const uint8_t* id) {
    if (lu_) return fail(RELP_E_UNSUPPORTED, "the LU engine is not sharded");
    Api& a = api();
    if (!a.ok) return fail(RELP_E_UNSUPPORTED, "no RCCL library (librccl.so.1) could be loaded");
    rccl_release();
    if (cfg_.device >= 0) HIP_TRY(hipSetDevice(cfg_.device));
    UniqueId uid;
    std::memcpy(uid.internal, id, sizeof(uid.internal));
    Comm comm = nullptr;
    const int rc = a.CommInitRank(&comm, std::max(cfg_.shard_count, 1), uid, cfg_.shard_rank);
    if (rc != 0 || !comm)
        return fail(RELP_E_HIP, std::string("ncclCommInitRank: ") + (a.GetErrorString ? a.GetErrorString(rc) : "failed"));
    rccl_comm_ = comm;
    const relp_status_t st = shard_set_collectives(rccl_allgather, rccl_allreduce_sum, comm);
    if (st) rccl_release();
    return st;
}

void Engine::rccl_release() {
    if (!rccl_comm_) return;
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (coll_ctx_ == rccl_comm_) { coll_allgather_ = nullptr; coll_allreduce_ = nullptr; coll_ctx_ = nullptr; }
    (void)api().CommDestroy((Comm)rccl_comm_);
    rccl_comm_ = nullptr;
}

}  // namespace relp

extern "C" relp_status_t relp_rccl_unique_id(uint8_t* id) {
    if (!id) return RELP_E_ARG;
    relp::Api& a = relp::api();
    if (!a.ok) return RELP_E_UNSUPPORTED;
    relp::UniqueId uid;
    if (a.GetUniqueId(&uid) != 0) return RELP_E_HIP;
    std::memcpy(id, uid.internal, sizeof(uid.internal));
    return RELP_OK;
}
