// Data-parallel exchange through RCCL behind the C ABI (include/rgda_hip.h: rgda_comm_*).  The reference is single-GPU and
// has no counterpart (SURVEY.md 8b lists "RCCL wrappers for (e)" in the boundary's minimum op set): a host that is not
// Python / torch gets the gradient all-reduce, the bf16 payload's all-to-all + all-gather and the 48 KB prototype-statistics
// all-reduce from these entry points; regda_amd/ddp.py keeps using torch.distributed (whose "nccl" backend is this same
// library) and offers `RcclComm` as the torch-free route.
// librccl.so is NOT a link-time dependency: it is resolved on the first rgda_comm_* call, preferring a copy the process
// already holds (torch ships its own) -- the kernels of this library load and run on a box without RCCL.
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {
typedef struct { char internal[RGDA_COMM_ID_BYTES]; } nccl_uid;     // ncclUniqueId: 128 opaque bytes, passed BY VALUE
typedef void* nccl_comm;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(nccl_comm*, int, nccl_uid, int);
typedef int (*fn_destroy)(nccl_comm);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, nccl_comm, hipStream_t);
typedef int (*fn_all_to_all)(const void*, void*, size_t, int, nccl_comm, hipStream_t);
typedef int (*fn_get_version)(int*);

// The ncclDataType_t / ncclRedOp_t values below are written out by hand (rccl.h is not a build dependency); they have been
// stable since NCCL 2.10 (where ncclBfloat16 = 9 arrived) and ncclAllToAll exists in RCCL from 2.7 on: a library that
// reports an older version -- or none -- is refused instead of being called with enum values it may read differently.
constexpr int kMinNcclVersion = 21000;     // ncclGetVersion's encoding from 2.9 on: major * 10000 + minor * 100 + patch

struct Rccl {
    void* h = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_all_to_all all_to_all = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static const Rccl R = [] {
        Rccl r;
        // a copy already mapped into the process first (RTLD_NOLOAD): two RCCL instances in one process would each
        // bootstrap their own topology and compete for the same xGMI channels
        // (0) whatever RCCL the process ALREADY holds, under any file name or soname (torch bundles its own copy and
        // loads it by path): the global symbol scope answers for it; the handle that owns the symbol is recovered with
        // dladdr so that every entry point below comes from that same copy
        if (void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
            Dl_info info;
            if (dladdr(sym, &info) && info.dli_fname) r.h = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);
        }
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names)
            if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.h) return r;
        fn_get_version get_version = (fn_get_version)dlsym(r.h, "ncclGetVersion");
        int version = 0;
        if (!get_version || get_version(&version) != 0 || version < kMinNcclVersion) return r;      // r.ok stays false
        r.get_uid = (fn_get_uid)dlsym(r.h, "ncclGetUniqueId");
        r.init_rank = (fn_init_rank)dlsym(r.h, "ncclCommInitRank");
        r.destroy = (fn_destroy)dlsym(r.h, "ncclCommDestroy");
        r.all_reduce = (fn_all_reduce)dlsym(r.h, "ncclAllReduce");
        r.all_gather = (fn_all_gather)dlsym(r.h, "ncclAllGather");
        r.all_to_all = (fn_all_to_all)dlsym(r.h, "ncclAllToAll");
        r.ok = r.get_uid && r.init_rank && r.destroy && r.all_reduce && r.all_gather && r.all_to_all;
        return r;
    }();
    return R;
}

// rgda dtype code -> ncclDataType_t (rccl.h: ncclInt64 = 4, ncclFloat32 = 7, ncclFloat64 = 8, ncclBfloat16 = 9)
int nccl_dtype(int dtype) {
    switch (dtype) {
        case RGDA_COMM_F32: return 7;
        case RGDA_COMM_BF16: return 9;
        case RGDA_COMM_I64: return 4;
        case RGDA_COMM_F64: return 8;
        default: return -1;
    }
}
}  // namespace

extern "C" int rgda_comm_unique_id(void* id) {
    if (!id) return RGDA_ERR_ARG;
    const Rccl& r = rccl();
    if (!r.ok) return RGDA_ERR_UNSUPPORTED;
    nccl_uid u;
    if (r.get_uid(&u) != 0) return RGDA_ERR_LAUNCH;
    memcpy(id, &u, RGDA_COMM_ID_BYTES);
    return RGDA_OK;
}

extern "C" int rgda_comm_init(const void* id, int rank, int world, rgda_comm_t* comm) {
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return RGDA_ERR_ARG;
    const Rccl& r = rccl();
    if (!r.ok) return RGDA_ERR_UNSUPPORTED;
    nccl_uid u;
    memcpy(&u, id, RGDA_COMM_ID_BYTES);
    nccl_comm c = nullptr;
    if (r.init_rank(&c, world, u, rank) != 0 || !c) return RGDA_ERR_LAUNCH;     // (on the calling thread's current device)
    *comm = c;
    return RGDA_OK;
}

extern "C" int rgda_comm_destroy(rgda_comm_t comm) {
    if (!comm) return RGDA_ERR_ARG;
    const Rccl& r = rccl();
    if (!r.ok) return RGDA_ERR_UNSUPPORTED;
    return r.destroy(comm) == 0 ? RGDA_OK : RGDA_ERR_LAUNCH;
}

extern "C" int rgda_comm_all_reduce(rgda_comm_t comm, void* buf, int64_t n, int dtype, rgda_stream_t stream) {
    if (!comm || !buf || n < 0 || nccl_dtype(dtype) < 0) return RGDA_ERR_ARG;
    if (n == 0) return RGDA_OK;
    const Rccl& r = rccl();
    if (!r.ok) return RGDA_ERR_UNSUPPORTED;
    return r.all_reduce(buf, buf, (size_t)n, nccl_dtype(dtype), /* ncclSum */ 0, comm, to_stream(stream)) == 0 ? RGDA_OK : RGDA_ERR_LAUNCH;
}

extern "C" int rgda_comm_all_gather(rgda_comm_t comm, const void* send, void* recv, int64_t n_per_rank, int dtype,
                                    rgda_stream_t stream) {
    if (!comm || !send || !recv || n_per_rank < 0 || nccl_dtype(dtype) < 0) return RGDA_ERR_ARG;
    if (n_per_rank == 0) return RGDA_OK;
    const Rccl& r = rccl();
    if (!r.ok) return RGDA_ERR_UNSUPPORTED;
    return r.all_gather(send, recv, (size_t)n_per_rank, nccl_dtype(dtype), comm, to_stream(stream)) == 0 ? RGDA_OK : RGDA_ERR_LAUNCH;
}

extern "C" int rgda_comm_all_to_all(rgda_comm_t comm, const void* send, void* recv, int64_t n_per_rank, int dtype,
                                    rgda_stream_t stream) {
    if (!comm || !send || !recv || send == recv || n_per_rank < 0 || nccl_dtype(dtype) < 0) return RGDA_ERR_ARG;
    if (n_per_rank == 0) return RGDA_OK;
    const Rccl& r = rccl();
    if (!r.ok) return RGDA_ERR_UNSUPPORTED;
    return r.all_to_all(send, recv, (size_t)n_per_rank, nccl_dtype(dtype), comm, to_stream(stream)) == 0 ? RGDA_OK : RGDA_ERR_LAUNCH;
}
