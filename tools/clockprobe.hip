// clockprobe: what shader clock does the device hold under a light, launch-bound load?
// Measures (a) cycles/wall time of a spin kernel, (b) wall time of an empty kernel and of
// a 2-level dependent gather, back to back on one stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long *out, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)v; }
}
__global__ void empty_k(int *p) { if (p == nullptr && threadIdx.x == 1234) p[0] = 1; }
__global__ void gather2(const int *idx, const double *tab, double *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = tab[idx[i]];
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 64);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a, s);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 200000);
        hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("spin: %llu shader cycles, %llu realtime ticks (100 MHz) -> %.0f MHz, wall %.3f ms\n", h[0], h[1], double(h[0]) / (double(h[1]) / 100.0), ms);
    }
    const int n = 1000, N = 1 << 20;
    int *idx; double *tab, *out; hipMalloc(&idx, n * 4); hipMalloc(&tab, N * 8); hipMalloc(&out, n * 8);
    std::vector<int> hi(n); for (int i = 0; i < n; ++i) hi[i] = (i * 7919) % N;
    hipMemcpy(idx, hi.data(), n * 4, hipMemcpyHostToDevice); hipMemset(tab, 0, N * 8);
    for (int which = 0; which < 2; ++which) {
        const int K = 2000;
        hipStreamSynchronize(s);
        hipEventRecord(a, s);
        for (int k = 0; k < K; ++k) {
            if (which == 0) hipLaunchKernelGGL(empty_k, dim3(4), dim3(256), 0, s, (int *)out);
            else hipLaunchKernelGGL(gather2, dim3(4), dim3(256), 0, s, idx, tab, out, n);
        }
        hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s: %.2f us per launch (back to back, %d launches)\n", which ? "2-level gather (4 blocks)" : "empty kernel", ms * 1e3 / K, K);
    }
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 200000);
    hipStreamSynchronize(s);
    unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("spin after load: %.0f MHz\n", double(h[0]) / (double(h[1]) / 100.0));
    return 0;
}
