// How fast do a kernel's stores reach host-mapped memory over PCIe, by store width per lane?  (What bounds the last kernel
// of the closed loop: k_pack_frames / k_ev_apply write ~0.55 MB of delivery records per tick into the host's pinned block.)
//   hipcc --offload-arch=gfx950 -O3 tools/pcie_store_width.hip -o tools/pcie_store_width && tools/pcie_store_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename T, bool ATOMIC>
__global__ void __launch_bounds__(256) k_store(T *dst, size_t n, T v)
{
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) {
        if (ATOMIC) {
            if constexpr (sizeof(T) <= 8) __hip_atomic_store(&dst[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else dst[i] = v;
        } else {
            dst[i] = v;
        }
    }
}

template <typename T, bool ATOMIC> static int run(const char *what, void *host, size_t bytes, int grid, hipStream_t s, T v)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const size_t n = bytes / sizeof(T);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_store<T, ATOMIC>), dim3(grid), dim3(256), 0, s, reinterpret_cast<T *>(host), n, v);
    CK(hipStreamSynchronize(s));
    float best = 1e9f;
    for (int it = 0; it < 10; ++it) {
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL((k_store<T, ATOMIC>), dim3(grid), dim3(256), 0, s, reinterpret_cast<T *>(host), n, v);
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    std::printf("{\"store\": \"%s\", \"bytes\": %zu, \"workgroups\": %d, \"us\": %.1f, \"GBps\": %.1f}\n", what, bytes, grid, best * 1e3, bytes / (best * 1e-3) / 1e9);
    return 0;
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (size_t bytes : {size_t(576) << 10, size_t(8) << 20}) {
        void *host = nullptr;
        CK(hipHostMalloc(&host, bytes, hipHostMallocMapped));
        for (int grid : {64, 512}) {
            if (run<uint32_t, true>("4 B per lane, system-scope atomic store", host, bytes, grid, s, 7u)) return 1;
            if (run<uint32_t, false>("4 B per lane, plain", host, bytes, grid, s, 7u)) return 1;
            if (run<double, true>("8 B per lane, system-scope atomic store", host, bytes, grid, s, 7.0)) return 1;
            if (run<double, false>("8 B per lane, plain", host, bytes, grid, s, 7.0)) return 1;
            if (run<uint4, false>("16 B per lane, plain", host, bytes, grid, s, make_uint4(1, 2, 3, 4))) return 1;
        }
        CK(hipHostFree(host));
    }
    return 0;
}
