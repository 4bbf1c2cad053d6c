// The ONE collective of the path (north_star: "a single RCCL all-gather over xGMI to reassemble action logits"): every
// rank steps its own shard of the environments and the (B_local, 7) action records are all-gathered into the (world * B_local, 7) record
// of the whole rollout batch -- enqueued by the library itself on the step's stream, right behind the (hipGraph-replayed) step, instead of a
// Python torch.distributed call per step.  RCCL is resolved at run time (dlopen) and only when a communicator is asked for: a single-GPU
// user of libhcm.so has no load-time dependency on it.
// No reference counterpart: the reference evaluates one environment in one process (hierarchical_trainer.py:1088-1107).
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "model.h"

namespace {

constexpr int kUniqueIdBytes = 128;                 // NCCL_UNIQUE_ID_BYTES (rccl.h)
struct UniqueId { char internal[kUniqueIdBytes]; };
typedef void* Comm;
enum { kNcclSuccess = 0, kNcclFloat32 = 7 };        // ncclResult_t / ncclDataType_t values of rccl.h

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommAbort)(Comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

void rccl_load(Rccl& r) {
    // the copy the process already maps (a torch process: torch/lib/librccl.so) wins by SONAME; otherwise ROCm's
    // HCM_RCCL_LIB=<path>: an explicit library instead (a site's own RCCL build; tests/stub_rccl's shared-memory stand-in, which gives
    // hcm_act_gather a real peer on a one-GPU box) -- named explicitly, it must load: no silent fall-through to the system copy
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (const char* own = std::getenv("HCM_RCCL_LIB"); own && *own) {
        r.lib = dlopen(own, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { r.err = std::string("HCM_RCCL_LIB=") + own + " not loadable: " + (dlerror() ? dlerror() : "?"); return; }
    }
    for (const char* n : names) {
        if (r.lib) break;
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.lib) { r.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
    r.GetUniqueId = (int (*)(UniqueId*))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(Comm*, int, UniqueId, int))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (int (*)(Comm))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (int (*)(const void*, void*, size_t, int, Comm, hipStream_t))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
    r.CommAbort = (int (*)(Comm))dlsym(r.lib, "ncclCommAbort");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) { r.err = "librccl lacks an expected symbol"; r.lib = nullptr; }
}

// resolved once per process, also when two handles on two threads ask for their communicators at the same moment
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(r); });
    return r;
}

int fail(hcm_ctx* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}
std::string nccl_msg(const Rccl& r, const char* what, int rc) {
    return std::string(what) + " failed: " + (r.GetErrorString ? r.GetErrorString(rc) : "?") + " (" + std::to_string(rc) + ")";
}

}  // namespace

namespace hcm {
void comm_destroy(hcm_ctx* h) {
    if (h && h->comm) {
        Rccl& r = rccl();
        if (r.CommDestroy) (void)r.CommDestroy((Comm)h->comm);
        h->comm = nullptr;
    }
}
}  // namespace hcm

extern "C" {

int hcm_comm_unique_id(void* out128) {
    if (!out128) return HCM_ERR_ARG;
    Rccl& r = rccl();
    if (!r.lib) return HCM_ERR_STATE;
    UniqueId id;
    if (r.GetUniqueId(&id) != kNcclSuccess) return HCM_ERR_HIP;
    std::memcpy(out128, id.internal, kUniqueIdBytes);
    return HCM_OK;
}

int hcm_comm_init(hcm_handle h, const void* unique_id128, int rank, int world) {
    if (!h) return HCM_ERR_ARG;
    if (!unique_id128 || world < 1 || rank < 0 || rank >= world) return fail(h, HCM_ERR_ARG, "hcm_comm_init: bad argument");
    if (!h->finalized) return fail(h, HCM_ERR_STATE, "hcm_comm_init before hcm_finalize");
    if (h->comm) return fail(h, HCM_ERR_STATE, "hcm_comm_init: the handle already has a communicator");
    Rccl& r = rccl();
    if (!r.lib) return fail(h, HCM_ERR_STATE, "hcm_comm_init: " + r.err);
    UniqueId id;
    std::memcpy(id.internal, unique_id128, kUniqueIdBytes);
    Comm c = nullptr;
    const int rc = r.CommInitRank(&c, world, id, rank);        // collective: every rank of the job calls it (blocks until all have)
    if (rc != kNcclSuccess || !c) return fail(h, HCM_ERR_HIP, nccl_msg(r, "ncclCommInitRank", rc));
    h->comm = c;
    h->comm_world = world;
    h->comm_rank = rank;
    return HCM_OK;
}

int hcm_comm_abort(hcm_handle h) {
    if (!h) return HCM_ERR_ARG;
    if (!h->comm) return HCM_OK;
    Rccl& r = rccl();
    // ncclCommAbort frees the communicator and fails outstanding / future operations on it on THIS rank; peers blocked in a collective with
    // this rank are released by their own abort (or by this process exiting)
    if (r.CommAbort) (void)r.CommAbort((Comm)h->comm); else if (r.CommDestroy) (void)r.CommDestroy((Comm)h->comm);
    h->comm = nullptr;
    h->comm_world = 0;
    h->comm_rank = 0;
    return HCM_OK;
}

int hcm_gather_poison(hcm_handle h, int B, float* record, float* gathered, void* stream) {
    if (!h) return HCM_ERR_ARG;
    h->gather_joined = 0;
    if (!h->comm) return fail(h, HCM_ERR_STATE, "hcm_gather_poison: no communicator (hcm_comm_init)");
    // (no engine buffer is involved: any B >= 1 is accepted, also one beyond max_batch -- the peers' count is what has to be matched)
    if (!record || !gathered || B < 1) return fail(h, HCM_ERR_ARG, "hcm_gather_poison: bad buffer / batch");
    (void)hipMemsetAsync(record, 0xFF, (size_t)B * 7 * sizeof(float), (hipStream_t)stream);
    Rccl& r = rccl();
    h->gather_joined = 1;
    const int nrc = r.AllGather(record, gathered, (size_t)B * 7, kNcclFloat32, (Comm)h->comm, (hipStream_t)stream);
    if (nrc != kNcclSuccess) return fail(h, HCM_ERR_HIP, nccl_msg(r, "ncclAllGather", nrc));
    return HCM_OK;
}

int hcm_act_gather(hcm_handle h, const void* rgb, int rgb_dtype, const float* depth, const void* ids, int ids_dtype, const int32_t* lengths,
                   int B, int L, const float* hi_h_in, const float* lo_h_in, const float* mask, float* record, float* hi_h_out, float* lo_h_out,
                   int flags, float* gathered, void* stream) {
    if (!h) return HCM_ERR_ARG;
    h->gather_joined = 0;       // hcm_query(HCM_GATHER_JOINED): the caller's way to tell "refused in front of the collective" from "failed inside the step"
    if (!h->comm) return fail(h, HCM_ERR_STATE, "hcm_act_gather: no communicator (hcm_comm_init)");
    if (!gathered) return fail(h, HCM_ERR_ARG, "hcm_act_gather: null gather buffer");
    // Pure argument errors return WITHOUT joining the collective: every rank passes the same B (the collective's element count must agree across
    // ranks, B * 7 floats each), so a B the handle cannot run is refused identically on every rank and nobody enters ncclAllGather; joining with a
    // count the peers do not share is undefined in NCCL (round-4 advisor).  Only failures BEHIND this point take the poisoned-record path.
    if (!record || B < 1 || B > h->cfg.max_batch)
        return fail(h, HCM_ERR_ARG, "hcm_act_gather: bad record buffer, or batch outside [1, max_batch] (no collective was entered; all ranks must pass the same B)");
    const int rc = hcm_act_ex(h, rgb, rgb_dtype, depth, ids, ids_dtype, lengths, B, L, hi_h_in, lo_h_in, mask, record, hi_h_out, lo_h_out, flags, stream);
    const std::string step_err = rc != HCM_OK ? h->err : std::string();
    // A rank whose own step failed STILL takes part in the collective -- with a poisoned (all-NaN) record -- and reports its error afterwards:
    // the other ranks are already inside (or about to enter) ncclAllGather for this step, and the private communicator has no watchdog that
    // would release them (round-3 advisor).  They see NaN rows for this rank's environments; hcm_comm_abort tears the communicator down.
    if (rc != HCM_OK) (void)hipMemsetAsync(record, 0xFF, (size_t)B * 7 * sizeof(float), (hipStream_t)stream);
    // stream order makes the record complete before the collective reads it; every rank contributes B rows of 7 floats
    Rccl& r = rccl();
    h->gather_joined = 1;
    const int nrc = r.AllGather(record, gathered, (size_t)B * 7, kNcclFloat32, (Comm)h->comm, (hipStream_t)stream);
    if (rc != HCM_OK) return fail(h, rc, step_err + " [this rank still contributed a NaN record to the step's all-gather]");
    if (nrc != kNcclSuccess) return fail(h, HCM_ERR_HIP, nccl_msg(r, "ncclAllGather", nrc));
    return HCM_OK;
}

}  // extern "C"
