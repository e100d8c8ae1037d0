// launchcost.hip -- what the host pays per HIP call on this runtime (us per call, the stream kept shallow):
// eager kernel launches with small / 600-byte argument blocks, event record, event synchronize on a completed event,
// hipGraphLaunch of a 5-kernel chain, a 64 KB memcpy into pinned memory.  Decides how a batch is issued.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

struct Big { char b[600]; };
__global__ void k_small(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(Big a, int *p) { if (p && threadIdx.x == 9999) *p = a.b[3]; }
__global__ void k_work(int *p, int n) { for (int i = 0; i < n; ++i) if (threadIdx.x == 9999) p[i] = i; __builtin_amdgcn_s_sleep(100); }

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *d;
    CK(hipMalloc(&d, 4096));
    Big big;
    memset(&big, 1, sizeof(big));
    const int reps = 2000;
    auto run = [&](const char *name, auto f, int per = 1) {
        for (int i = 0; i < 50; ++i) f();
        hipStreamSynchronize(s);
        double t = 0;
        for (int i = 0; i < reps; ++i) {
            const double a = now();
            f();
            t += now() - a;
            if ((i & 15) == 15) hipStreamSynchronize(s); // keep the queue shallow: the cost of an un-throttled call
        }
        printf("%-44s %7.2f us per call\n", name, t / reps / per);
    };
    run("launch, 8-byte args", [&] { hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, d); });
    run("launch, 600-byte args", [&] { hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s, big, d); });
    run("5 launches back to back (per launch)", [&] { for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s, big, d); }, 5);
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    run("hipEventRecord (no timing)", [&] { hipEventRecord(ev, s); });
    hipStreamSynchronize(s);
    run("hipEventSynchronize (completed)", [&] { hipEventSynchronize(ev); });
    run("hipEventQuery (completed)", [&] { (void)hipEventQuery(ev); });
    run("hipStreamQuery", [&] { (void)hipStreamQuery(s); });
    hipEvent_t evt;
    CK(hipEventCreate(&evt));
    run("hipEventRecord (timing)", [&] { hipEventRecord(evt, s); });
    // a graph of five dependent kernels
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s, big, d);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    run("hipGraphLaunch, 5 kernels (per graph)", [&] { hipGraphLaunch(ge, s); });
    // device-side: how long do 5 dependent kernels take eager vs graph (empty kernels)
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms;
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int r = 0; r < 200; ++r) for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s, big, d);
    hipEventRecord(b, s); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    printf("%-44s %7.2f us per 5-kernel chain (device)\n", "eager chain", ms * 1000 / 200);
    hipEventRecord(a, s);
    for (int r = 0; r < 200; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(b, s); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    printf("%-44s %7.2f us per 5-kernel chain (device)\n", "graph chain", ms * 1000 / 200);
    void *pin;
    CK(hipHostMalloc(&pin, 1 << 17, hipHostMallocMapped));
    std::vector<char> src(1 << 17, 3);
    run("memcpy 77 KB into pinned memory", [&] { memcpy(pin, src.data(), 77000); });
    // two streams alternating (the bench's contexts)
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    run("launch alternating two streams", [&] { static int k = 0; hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, (k++ & 1) ? s : s2, big, d); });
    hipStreamSynchronize(s2);
    return 0;
}
