// TEST INFRASTRUCTURE, not product: a stand-in for the six RCCL entry points robo-vln_amd/csrc/comm.cpp resolves (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclCommAbort, ncclAllGather, ncclGetErrorString), so that hcm_act_gather's multi-rank paths -- a normal step, a rank whose step
// failed, hcm_comm_abort releasing a blocked rank -- run with a REAL peer on a box that has one GPU (two processes, one device each on cuda:0).
// The ranks meet in a POSIX shared-memory segment named by the unique id; an all-gather is, in stream order on the caller's stream:
//   device -> shared-memory copy of this rank's block, a host function that publishes the step and waits for every rank's (or for an abort), a
//   shared-memory -> device copy of the whole record.
// Same contract as the real collective where the tests look: stream-ordered, blocks until every rank has joined, ncclCommAbort releases only the
// calling rank's own pending operation.  libhcm loads it through HCM_RCCL_LIB (tests/test_comm_peer_gpu.py); nothing in the product links it.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>

namespace {
constexpr int kMaxRanks = 8, kIdBytes = 128;
constexpr size_t kSlotBytes = 1 << 16;                      // per rank, per parity: 64 KB (the records are B * 7 floats)
struct Shared {
    std::atomic<int> joined;                                // ncclCommInitRank rendezvous
    std::atomic<uint64_t> seq[kMaxRanks];                   // number of all-gathers rank r has published
    std::atomic<int> gone[kMaxRanks];                       // rank r aborted / destroyed its communicator
    alignas(64) unsigned char slot[2][kMaxRanks][kSlotBytes];
};
struct Comm {
    Shared* sh = nullptr; int rank = 0, world = 0; uint64_t step = 0; std::atomic<int> aborted{0}; char name[64];
    unsigned char* stage = nullptr;                        // pinned host staging (hipHostMalloc): shared memory itself is not registered
};
struct Op { Comm* c; uint64_t step; size_t bytes; };
struct Id { char b[kIdBytes]; };

void publish_and_wait(void* p) {
    Op* op = (Op*)p; Comm* c = op->c; Shared* sh = c->sh;
    std::memcpy(sh->slot[op->step & 1][c->rank], c->stage, op->bytes);
    sh->seq[c->rank].store(op->step + 1, std::memory_order_release);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < c->world; ++r)
        while (sh->seq[r].load(std::memory_order_acquire) < op->step + 1) {
            if (c->aborted.load()) goto out;                // this rank's own abort releases it (as ncclCommAbort does); a peer's does not
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { std::fprintf(stderr, "stub_rccl: rank %d gave up waiting for rank %d\n", c->rank, r); goto out; }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
out:
    for (int r = 0; r < c->world; ++r) std::memcpy(c->stage + (size_t)r * op->bytes, sh->slot[op->step & 1][r], op->bytes);
    delete op;
}
}  // namespace

extern "C" {
int ncclGetUniqueId(Id* id) {
    std::memset(id->b, 0, kIdBytes);
    std::snprintf(id->b, kIdBytes, "/hcm_stub_rccl_%d_%lld", (int)getpid(), (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return 0;
}
int ncclCommInitRank(void** out, int world, Id id, int rank) {
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return 4;
    id.b[kIdBytes - 1] = 0;
    int fd = shm_open(id.b, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return 2; }        // (a fresh segment is zero-filled: every atomic starts at 0)
    void* m = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return 2;
    Comm* c = new Comm;
    c->sh = (Shared*)m; c->rank = rank; c->world = world;
    std::strncpy(c->name, id.b, sizeof(c->name) - 1); c->name[sizeof(c->name) - 1] = 0;
    if (hipHostMalloc((void**)&c->stage, kSlotBytes * kMaxRanks, hipHostMallocDefault) != hipSuccess) { delete c; return 1; }
    c->sh->joined.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->joined.load() < world) {                                  // collective, like the real call
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return 6;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    *out = c;
    return 0;
}
static int release(void* comm, bool abort) {
    Comm* c = (Comm*)comm;
    if (!c) return 4;
    c->aborted.store(1);
    c->sh->gone[c->rank].store(abort ? 2 : 1);
    (void)hipDeviceSynchronize();                                            // pending host functions of this rank have returned
    if (c->rank == 0) shm_unlink(c->name);
    (void)hipHostFree(c->stage);
    munmap(c->sh, sizeof(Shared));
    delete c;
    return 0;
}
int ncclCommDestroy(void* comm) { return release(comm, false); }
int ncclCommAbort(void* comm) { return release(comm, true); }
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c || dtype != 7) return 4;                                          // ncclFloat32 only (what comm.cpp sends)
    const size_t bytes = count * 4;
    if (bytes > kSlotBytes) return 4;
    if (hipMemcpyAsync(c->stage, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
    Op* op = new Op{c, c->step++, bytes};
    if (hipLaunchHostFunc(stream, publish_and_wait, op) != hipSuccess) { delete op; return 1; }
    if (hipMemcpyAsync(recv, c->stage, bytes * c->world, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    return 0;
}
const char* ncclGetErrorString(int rc) { return rc == 0 ? "success" : rc == 4 ? "invalid argument" : rc == 6 ? "rendezvous timed out" : "stub_rccl error"; }
}
