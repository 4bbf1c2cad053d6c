// Does hipGraphLaunch scale across host threads?  Three linear graphs of N tiny kernels on three streams: host time to enqueue them from ONE thread
// back to back vs from three threads at once, against one forked graph holding all 3N nodes (the step graph's shape).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_launch_mt tools/native/graph_launch_mt.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(float* p, int n) { if (threadIdx.x == 0 && n < 0) p[0] = 1.f; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 80, REP = 50;
    float* d; CK(hipMalloc(&d, 1024));
    hipStream_t s[3];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipGraphExec_t ge[3];
    for (int i = 0; i < 3; ++i) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s[i], hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < N; ++k) tiny<<<1, 64, 0, s[i]>>>(d, k);
        CK(hipStreamEndCapture(s[i], &g));
        CK(hipGraphInstantiate(&ge[i], g, nullptr, nullptr, 0));
    }
    // one forked graph: s[0] origin, s[1], s[2] forked
    hipGraphExec_t gf;
    {
        hipEvent_t ef, ej[2];
        CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
        for (auto& e : ej) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        hipGraph_t g;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(ef, s[0]));
        CK(hipStreamWaitEvent(s[1], ef, 0)); CK(hipStreamWaitEvent(s[2], ef, 0));
        for (int i = 1; i < 3; ++i) for (int k = 0; k < N; ++k) tiny<<<1, 64, 0, s[i]>>>(d, k);
        for (int k = 0; k < N; ++k) tiny<<<1, 64, 0, s[0]>>>(d, k);
        for (int i = 1; i < 3; ++i) { CK(hipEventRecord(ej[i - 1], s[i])); CK(hipStreamWaitEvent(s[0], ej[i - 1], 0)); }
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&gf, g, nullptr, nullptr, 0));
    }
    auto sync = [&]() { for (auto& x : s) (void)hipStreamSynchronize(x); };
    for (int w = 0; w < 3; ++w) { for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge[i], s[i]); (void)hipGraphLaunch(gf, s[0]); sync(); }
    double t_seq = 0, t_seq_wall = 0, t_mt = 0, t_mt_wall = 0, t_fork = 0, t_fork_wall = 0, t_one = 0;
    for (int r = 0; r < REP; ++r) {
        double t0 = now();
        (void)hipGraphLaunch(ge[0], s[0]);
        t_one += now() - t0; sync();
        t0 = now();
        for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge[i], s[i]);
        t_seq += now() - t0; sync(); t_seq_wall += now() - t0;
        t0 = now();
        (void)hipGraphLaunch(gf, s[0]);
        t_fork += now() - t0; sync(); t_fork_wall += now() - t0;
    }
    // persistent worker threads, spin on a generation counter
    std::atomic<int> gen{0}, done{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int i = 1; i < 3; ++i)
        th.emplace_back([&, i]() {
            (void)hipSetDevice(0);
            int seen = 0;
            while (true) {
                while (gen.load(std::memory_order_acquire) == seen) { if (stop.load()) return; }
                ++seen;
                (void)hipGraphLaunch(ge[i], s[i]);
                done.fetch_add(1, std::memory_order_release);
            }
        });
    for (int r = 0; r < REP + 3; ++r) {
        double t0 = now();
        done.store(0);
        gen.fetch_add(1, std::memory_order_release);
        (void)hipGraphLaunch(ge[0], s[0]);
        while (done.load(std::memory_order_acquire) < 2) {}
        double t1 = now();
        sync();
        if (r >= 3) { t_mt += t1 - t0; t_mt_wall += now() - t0; }
    }
    stop.store(true);
    for (auto& t : th) t.join();
    printf("N = %d kernels per chain\n", N);
    printf("one graph (N nodes), one thread:              host %.1f us (%.2f us/node)\n", t_one / REP, t_one / REP / N);
    printf("three graphs back to back, one thread:        host %.1f us, to completion %.1f us\n", t_seq / REP, t_seq_wall / REP);
    printf("three graphs from three threads:              host %.1f us, to completion %.1f us\n", t_mt / REP, t_mt_wall / REP);
    printf("one forked graph (3N nodes), one thread:      host %.1f us, to completion %.1f us\n", t_fork / REP, t_fork_wall / REP);
    // eager launches for scale
    double t_e = 0;
    for (int r = 0; r < REP; ++r) { double t0 = now(); for (int k = 0; k < N; ++k) tiny<<<1, 64, 0, s[0]>>>(d, k); t_e += now() - t0; sync(); }
    printf("N eager launches, one thread:                 host %.1f us (%.2f us/launch)\n", t_e / REP, t_e / REP / N);
    return 0;
}
