// tools/probe_check.hip -- what do the events of hipExtLaunchKernelGGL measure?  A kernel that spins for a known time
// (s_memrealtime, 100 MHz) is launched behind a long one on the same stream; its event pair is compared with the
// spin time, with a hipEventRecord bracket around the same launch, and with the same pair while a second stream keeps
// the device busy.  (hipcc --offload-arch=gfx950 -O2 tools/probe_check.hip -o tools/probe_check)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = __builtin_amdgcn_s_memrealtime() - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    hipEvent_t a, b, c, d;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c)); CK(hipEventCreate(&d));
    unsigned long long *out;
    CK(hipMalloc(&out, 8));
    for (int busy = 0; busy < 2; ++busy)
        for (unsigned long long us : {10ull, 50ull, 300ull}) {
            double ext = 0, br = 0;
            const int reps = 20;
            for (int r = 0; r < reps; ++r) {
                if (busy) hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, s2, 100ull * 2000ull, nullptr); // 2 ms on every CU
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, 100ull * 300ull, nullptr);             // 300 us ahead on the stream
                hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, a, b, 0, 100ull * us, out);
                CK(hipEventRecord(c, s1));
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s1, 100ull * us, out);
                CK(hipEventRecord(d, s1));
                CK(hipDeviceSynchronize());
                float m1 = 0, m2 = 0;
                CK(hipEventElapsedTime(&m1, a, b));
                CK(hipEventElapsedTime(&m2, c, d));
                ext += m1 * 1e3;
                br += m2 * 1e3;
            }
            std::printf("{\"spin_us\": %llu, \"other_stream_busy\": %d, \"ext_launch_pair_us\": %.2f, \"event_record_bracket_us\": %.2f}\n", us, busy,
                        ext / reps, br / reps);
        }
    return 0;
}
