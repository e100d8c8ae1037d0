// fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports on gfx950 for the access patterns of the engine's stages,
// against byte counts known by construction (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate
// on a known byte count in your own access pattern").  Every kernel reads a 1 GiB buffer once (four times the Infinity
// Cache), in the pattern named; run under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- tools/fetch_calib
// and compare FETCH_SIZE (KiB) with the bytes printed here: tools/pmc_traffic.py takes the per-pattern factors
// (true bytes / reported bytes) from profiles/fetch_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr size_t kBytes = size_t(1) << 30;

template <typename T> __device__ void sink(T v, uint32_t *out) { if (v == T(0x7fffff01)) *out = 1; }

// 16 B per lane, consecutive: the sweep's stream of pre-filter records (rm_filter.hip: nd.rxf)
__global__ void calib_stream16(const uint4 *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) acc ^= p[i].x ^ p[i].w;
    sink(acc, out);
}
// 8 B per lane, consecutive: rssi columns (rm_reorder.hip)
__global__ void calib_stream8(const uint2 *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) acc ^= p[i].x ^ p[i].y;
    sink(acc, out);
}
// 4 B per lane, consecutive: candidate entries, node indices (rm_exact.hip, rm_reorder.hip)
__global__ void calib_stream4(const uint32_t *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) acc ^= p[i];
    sink(acc, out);
}
// gather of aligned 32-byte records at scattered indices, every record exactly once (a permutation by an odd multiplier
// modulo a power of two): the exact stage's RxCompact gather
__global__ void calib_gather32(const uint4 *p, size_t n_rec, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n_rec; i += size_t(gridDim.x) * blockDim.x) {
        const size_t r = (i * 2654435761ull) & (n_rec - 1);
        acc ^= p[2 * r].x ^ p[2 * r + 1].w;
    }
    sink(acc, out);
}
// gather of aligned 64-byte records: rm_tx_record, RxRecord
__global__ void calib_gather64(const uint4 *p, size_t n_rec, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n_rec; i += size_t(gridDim.x) * blockDim.x) {
        const size_t r = (i * 2654435761ull) & (n_rec - 1);
        acc ^= p[4 * r].x ^ p[4 * r + 1].y ^ p[4 * r + 2].z ^ p[4 * r + 3].w;
    }
    sink(acc, out);
}
// short runs: 64 consecutive 8-byte values (one wave, 512 B) at scattered 512-byte-aligned places -- a frame's segment of
// link records (rm_reorder.hip reads a frame's <= 64 records one per lane)
__global__ void calib_runs8(const uint2 *p, size_t n_runs, uint32_t *out)
{
    uint32_t acc = 0;
    const int lane = threadIdx.x & 63;
    for (size_t w = (blockIdx.x * size_t(blockDim.x) + threadIdx.x) >> 6; w < n_runs; w += (size_t(gridDim.x) * blockDim.x) >> 6) {
        const size_t r = (w * 2654435761ull) & (n_runs - 1);
        acc ^= p[r * 64 + lane].x;
    }
    sink(acc, out);
}

int main()
{
    void *buf;
    uint32_t *out;
    if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, kBytes);
    hipDeviceSynchronize();
    const dim3 grid(4096), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_stream16, grid, block, 0, 0, static_cast<const uint4 *>(buf), kBytes / 16, out);
        hipLaunchKernelGGL(calib_stream8, grid, block, 0, 0, static_cast<const uint2 *>(buf), kBytes / 8, out);
        hipLaunchKernelGGL(calib_stream4, grid, block, 0, 0, static_cast<const uint32_t *>(buf), kBytes / 4, out);
        hipLaunchKernelGGL(calib_gather32, grid, block, 0, 0, static_cast<const uint4 *>(buf), kBytes / 32, out);
        hipLaunchKernelGGL(calib_gather64, grid, block, 0, 0, static_cast<const uint4 *>(buf), kBytes / 64, out);
        hipLaunchKernelGGL(calib_runs8, grid, block, 0, 0, static_cast<const uint2 *>(buf), kBytes / 512, out);
        hipDeviceSynchronize();
    }
    printf("every kernel reads %zu bytes (= %zu KiB) once\n", kBytes, kBytes / 1024);
    return 0;
}
