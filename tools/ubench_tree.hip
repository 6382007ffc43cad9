// Does a LONE wave pay for code size?  The tree kernels (bucket reduce, marginals) run one Jacobian addition (~50 KB of straight-line
// code) per level in a wave that has its SIMD to itself and measure ~15 us per level, against ~9 us the product rate predicts.
// k_jadd<K>: every lane runs `iters` rounds of K textually distinct (unrolled) additions: the loop body is K x ~50 KB.
// Build: hipcc --offload-arch=gfx950 -O3 -I ark_bulletproofs_amd/csrc tools/ubench_tree.hip -o tools/ubench_tree
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "msm.cuh"
using namespace arkbp;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class C, int K> __global__ void __launch_bounds__(256) k_jadd(const u32* a, u32* out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Jac acc = jac_dbl<C>(jac_from_aff<C>(aff_load_dev(a + 16 * i)));
    Jac q = jac_dbl<C>(acc);
#pragma unroll 1
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int j = 0; j < K; j++) { acc = jac_add<C>(acc, q); asm volatile("" ::: "memory"); }
    }
    fe_pack(out + 8 * i, fe_canon<typename C::Fq>(acc.X));
}

// ---- what does one level of the LDS tree (block_sum_jac) cost next to the addition it contains? ----
template <class C, int MODE> __global__ void __launch_bounds__(256) k_tree(const u32* a, u32* out, int reps) {
    __shared__ u32 sh[256 * 27];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const Jac P = jac_dbl<C>(jac_from_aff<C>(aff_load_dev(a + 16 * i)));
    u32 sink = 0;
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
        Jac acc = P;
        acc.X.l[0] ^= (u32)r & 1u;   // (keeps the rounds distinct for the optimiser; the values need not be curve points for timing)
        if (MODE == 0) acc = block_sum_jac<C>(acc, sh);
        else {
            // six shuffle levels in every wave, then the four wave sums through LDS and two more shuffle levels in wave 0
            const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
#pragma unroll 1
            for (int o = 32; o >= 1; o >>= 1) { const Jac other = jac_shfl_down(acc, o); if ((int)lane < o) acc = jac_add<C>(acc, other); }
            if (lane == 0) { for (int q = 0; q < 9; q++) { sh[q * 4 + wv] = acc.X.l[q]; sh[(9 + q) * 4 + wv] = acc.Y.l[q]; sh[(18 + q) * 4 + wv] = acc.Z.l[q]; } }
            __syncthreads();
            if (wv == 0) {
                if (lane < 4) { for (int q = 0; q < 9; q++) { acc.X.l[q] = sh[q * 4 + lane]; acc.Y.l[q] = sh[(9 + q) * 4 + lane]; acc.Z.l[q] = sh[(18 + q) * 4 + lane]; } }
#pragma unroll 1
                for (int o = 2; o >= 1; o >>= 1) { const Jac other = jac_shfl_down(acc, o); if ((int)lane < o) acc = jac_add<C>(acc, other); }
            }
            __syncthreads();
        }
        sink += acc.X.l[1];
    }
    out[i] = sink;
}
template <class F> double time_kernel(F launch, int reps = 5) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps / 1e3;
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t lanes = (size_t)cus * 8 * 256;
    void* buf; CHECK(hipMalloc(&buf, lanes * 64 + lanes * 32));
    { std::vector<u32> h(lanes * 16); u32 x = 12345; for (size_t i = 0; i < h.size(); i++) { x = x * 1664525u + 1013904223u; h[i] = (i % 8 == 7) ? (x >> 8) : x; } CHECK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    u32* in = (u32*)buf; u32* outp = in + lanes * 16;
    const int total = 64;   // additions per lane in every configuration
#define RUN(K) \
    for (int cfg = 0; cfg < 4; cfg++) { \
        const int threads = cfg == 0 ? 64 : 256, blocks = cfg <= 1 ? cus : cfg == 2 ? 2 * cus : 4 * cus; \
        const double s = time_kernel([&] { k_jadd<Secq, K><<<blocks, threads>>>(in, outp, total / K); }); \
        printf("K=%d (loop body ~%3d KB)  %4d blocks x %3d lanes  %8.1f us  %6.2f us per addition\n", K, K * 50, blocks, threads, s * 1e6, s * 1e6 / total); \
    }
    RUN(1) RUN(2) RUN(4) RUN(8)
    for (int blocks : {cus / 2, cus, cus + 20, 2 * cus}) {
        const int reps = 8;
        double s0 = time_kernel([&] { k_tree<Secq, 0><<<blocks, 256>>>(in, outp, reps); });
        double s1 = time_kernel([&] { k_tree<Secq, 1><<<blocks, 256>>>(in, outp, reps); });
        printf("tree of 256 lanes, %4d blocks:  LDS tree %7.1f us per tree (%5.2f us per level)   shuffle tree %7.1f us per tree (%5.2f us per level)\n", blocks, s0 * 1e6 / reps,
               s0 * 1e6 / reps / 8, s1 * 1e6 / reps, s1 * 1e6 / reps / 8);
    }
    return 0;
}
