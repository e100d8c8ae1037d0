// mul48_probe.hip -- does a 48-bit modular product built from 24-bit halves equal the 64-bit one on the
// device?  It does not with hipcc 7.2 / gfx950 (-O3): the masks feeding the widened 24 x 24 product are
// dropped when it is lowered to v_mad_u64_u32.  (The variant was tried in the java.util.Random walk and
// gave wrong draws; the engine does not use it.)
//   hipcc --offload-arch=gfx950 -O3 tools/mul48_probe.hip -o tools/mul48_probe && ./tools/mul48_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __host__ inline uint64_t mul48(uint64_t a, uint64_t b)
{
    const uint32_t a0 = uint32_t(a) & 0xFFFFFFu, a1 = uint32_t(a >> 24) & 0xFFFFFFu;
    const uint32_t b0 = uint32_t(b) & 0xFFFFFFu, b1 = uint32_t(b >> 24) & 0xFFFFFFu;
    const uint64_t low = uint64_t(a0) * uint64_t(b0);
    const uint32_t cross = (a1 * b0 + a0 * b1) & 0xFFFFFFu;
    return (low + (uint64_t(cross) << 24)) & ((1ull << 48) - 1);
}
__global__ void k(const uint64_t *a, const uint64_t *b, uint64_t *bad, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t m = (1ull << 48) - 1;
    if (mul48(a[i], b[i]) != ((a[i] * b[i]) & m)) atomicAdd((unsigned long long *)bad, 1ull);
}
int main()
{
    const int n = 1 << 22;
    uint64_t *ha = new uint64_t[n], *hb = new uint64_t[n];
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; ha[i] = s & ((1ull << 48) - 1);
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; hb[i] = (i & 1) ? 0x5DEECE66Dull : (s & ((1ull << 48) - 1));
    }
    uint64_t *da, *db, *dbad, bad = 0;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dbad, 8);
    hipMemcpy(da, ha, n * 8, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 8, hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dbad, n);
    hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
    printf("mismatches on the device: %llu of %d\n", (unsigned long long)bad, n);
    return bad ? 1 : 0;
}
