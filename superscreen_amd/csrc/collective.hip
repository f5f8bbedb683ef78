// RCCL entry points of the C ABI (include/superscreen_hip.h, section 7): the one collective of the
// solver path -- the in-place sum all-reduce of the concatenated inter-film coupling vector
// (solver/solve.py:491-515 sums the same fields inside one process) -- and the communicator helpers a
// caller without its own RCCL binding needs.
//
// librccl.so is opened lazily with dlopen: libsuperscreen_hip.so has no load-time dependency on RCCL
// (single-GPU users never touch it), and inside a PyTorch process the name resolves to the RCCL that
// torch already loaded (same soname), so there is one RCCL per process.
//
// The four entry points used are declared HERE, by their public, stable NCCL 2 ABI (opaque communicator pointer,
// 128-byte unique id passed by value, int-sized enums: ncclSuccess = 0, ncclSum = 0, ncclFloat32 = 7,
// ncclFloat64 = 8): the library builds on ROCm installations without the RCCL development headers.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.hpp"

namespace ssa {
namespace {

struct ncclUniqueId {
    char internal[SSA_RCCL_UNIQUE_ID_BYTES];
};
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclRedOp_t ncclSum = 0;
constexpr ncclDataType_t ncclFloat32 = 7, ncclFloat64 = 8;

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                               hipStream_t) = nullptr;
    bool ok = false;
};

const RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *name : names) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
        api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(api.handle, "ncclCommInitRank"));
        api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(dlsym(api.handle, "ncclAllReduce"));
        api.ok = api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_reduce;
    });
    return api;
}

static_assert(sizeof(ncclUniqueId) == SSA_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");

}  // namespace
}  // namespace ssa

using namespace ssa;

extern "C" int ssa_rccl_unique_id(void *id_out) {
    if (!id_out) return SSA_ERR_INVALID_ARGUMENT;
    const RcclApi &api = rccl();
    if (!api.ok) return SSA_ERR_RCCL;
    ncclUniqueId id;
    if (api.get_unique_id(&id) != ncclSuccess) return SSA_ERR_RCCL;
    memcpy(id_out, &id, sizeof(id));
    return SSA_OK;
}

extern "C" int ssa_rccl_comm_create(void **comm_out, int nranks, int rank, const void *id) {
    if (!comm_out || !id || nranks < 1 || rank < 0 || rank >= nranks) return SSA_ERR_INVALID_ARGUMENT;
    const RcclApi &api = rccl();
    if (!api.ok) return SSA_ERR_RCCL;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    if (api.comm_init_rank(&comm, nranks, uid, rank) != ncclSuccess) return SSA_ERR_RCCL;
    *comm_out = comm;
    return SSA_OK;
}

extern "C" int ssa_rccl_comm_destroy(void *comm) {
    if (!comm) return SSA_ERR_INVALID_ARGUMENT;
    const RcclApi &api = rccl();
    if (!api.ok) return SSA_ERR_RCCL;
    return api.comm_destroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? SSA_OK : SSA_ERR_RCCL;
}

extern "C" int ssa_coupling_allreduce(void *buf, int64_t count, int dtype, void *rccl_comm, void *stream) {
    if (!buf || count < 0 || !rccl_comm) return SSA_ERR_INVALID_ARGUMENT;
    if (dtype != SSA_F32 && dtype != SSA_F64) return SSA_ERR_INVALID_ARGUMENT;
    if (count == 0) return SSA_OK;
    const RcclApi &api = rccl();
    if (!api.ok) return SSA_ERR_RCCL;
    const ncclDataType_t dt = (dtype == SSA_F64) ? ncclFloat64 : ncclFloat32;
    const ncclResult_t rc = api.all_reduce(buf, buf, static_cast<size_t>(count), dt, ncclSum,
                                           static_cast<ncclComm_t>(rccl_comm), as_stream(stream));
    return rc == ncclSuccess ? SSA_OK : SSA_ERR_RCCL;
}
