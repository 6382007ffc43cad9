// FETCH_SIZE calibration: three read shapes with KNOWN byte counts, to be run under
//   rocprofv3 --pmc FETCH_SIZE -- tools/ubench_fetch
// so that the counter (KiB) can be compared with the bytes each kernel must fetch from HBM:
//   k_stream16   every lane reads consecutive 16-byte words (wide coalesced loads: the shape the guide's "x2 on gfx950" is about)
//   k_gather64   every lane reads ONE random 64-byte segment (an affine point: the access of the MSM accumulate kernels)
//   k_gather32   every lane reads ONE random 32-byte segment (a packed field element)
// Tables are 4 GiB (far beyond the 256 MB of L2 + MALL reach per pass with random indices), indices come from a hash of the lane id.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o tools/ubench_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32;
typedef uint64_t u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void k_stream16(const uint4* __restrict__ t, u64 n16, u32* __restrict__ out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i < n16; i += (u64)gridDim.x * blockDim.x) { const uint4 v = t[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
template <int SEG16> __global__ void k_gather(const uint4* __restrict__ t, u64 nseg, u64 reads, u32* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= reads) return;
    const uint4* p = t + (mix(i * 0x9E3779B97F4A7C15ULL + 7) % nseg) * SEG16;
    uint4 acc = p[0];
#pragma unroll
    for (int j = 1; j < SEG16; j++) { const uint4 v = p[j]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}
int main() {
    const u64 bytes = 4ull << 30;
    void* t; u32* out;
    CHECK(hipMalloc(&t, bytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(t, 0x5a, bytes)); CHECK(hipMemset(out, 0, 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
    const u64 reads = 1ull << 26;   // 64 M lanes
    CHECK(hipEventRecord(e0)); k_stream16<<<256 * 16, 256>>>((const uint4*)t, bytes / 16, out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_stream16  must fetch %.1f MiB  (%.2f ms, %.0f GB/s)\n", bytes / 1048576.0, ms, bytes / ms / 1e6);
    CHECK(hipEventRecord(e0)); k_gather<4><<<(u32)(reads / 256), 256>>>((const uint4*)t, bytes / 64, reads, out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_gather<4> (64-B segments) must fetch %.1f MiB  (%.2f ms, %.0f GB/s)\n", reads * 64 / 1048576.0, ms, reads * 64 / ms / 1e6);
    CHECK(hipEventRecord(e0)); k_gather<2><<<(u32)(reads / 256), 256>>>((const uint4*)t, bytes / 32, reads, out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_gather<2> (32-B segments) must fetch %.1f MiB of payload, %.1f MiB in 64-B sectors  (%.2f ms)\n", reads * 32 / 1048576.0, reads * 64 / 1048576.0, ms);
    return 0;
}
