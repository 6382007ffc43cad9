// Instruction-rate microbenchmark for the integer paths a 256-bit modmul can be built from on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -I ark_bulletproofs_amd/csrc tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "msm.cuh"
using namespace arkbp;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

// each kernel: 8 independent chains, ITER iterations of 64 instrs (8 per chain)
#define DEF_KERNEL32(name, ASM)                                                                   \
    __global__ void name(u32* out, int iters, u32 seed) {                                         \
        u32 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        u32 b = seed * 77 + 1;                                                                    \
        for (int i = 0; i < iters; i++) {                                                         \
            REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)             \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
    }
#define A_ADD(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define A_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define A_MULHI(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n"
#define A_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %" #k "\n"
#define A_MULHI24(k) "v_mul_hi_u32_u24 %" #k ", %" #k ", %8\n"
#define A_ADDC(k) "v_addc_co_u32 %" #k ", vcc, %" #k ", %8, vcc\n"
#define A_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define A_MADU32(k) "v_mad_u32_u16 %" #k ", %" #k ", %8, %" #k "\n"
DEF_KERNEL32(k_add, A_ADD)
DEF_KERNEL32(k_mullo, A_MULLO)
DEF_KERNEL32(k_mulhi, A_MULHI)
DEF_KERNEL32(k_mad24, A_MAD24)
DEF_KERNEL32(k_mulhi24, A_MULHI24)
DEF_KERNEL32(k_addc, A_ADDC)

#define DEF_KERNEL64(name, ASM)                                                                   \
    __global__ void name(u64* out, int iters, u32 seed) {                                         \
        u64 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        u32 b = seed * 77 + 1, c = seed * 1234567 + 3;                                            \
        for (int i = 0; i < iters; i++) {                                                         \
            REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)             \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) \
        }                                                                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
    }
#define A_MAD64(k) "v_mad_u64_u32 %" #k ", vcc, %8, %9, %" #k "\n"
#define A_LSHLADD64(k) "v_lshl_add_u64 %" #k ", %" #k ", 0, %" #k "\n"
DEF_KERNEL64(k_mad64, A_MAD64)
DEF_KERNEL64(k_lshladd64, A_LSHLADD64)

__global__ void k_fma64(double* out, int iters, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    double b = 1.0000001, c = 1e-9;
#define A_FMA64(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
    for (int i = 0; i < iters; i++) {
        REP8(asm volatile(A_FMA64(0) A_FMA64(1) A_FMA64(2) A_FMA64(3) A_FMA64(4) A_FMA64(5) A_FMA64(6) A_FMA64(7)
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <class P> __global__ void k_femul(const u32* a, u32* out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fe x = fe_unpack(a + 8 * i), y = fe_unpack(a + 8 * i + 8);
    for (int k = 0; k < iters; k++) { x = fe_mul<P>(x, y); y = fe_mul<P>(y, x); }
    fe_pack(out + 8 * i, fe_canon<P>(fe_norm(fe_add(x, y))));
}
template <class P> __global__ void k_fesqr(const u32* a, u32* out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fe x = fe_unpack(a + 8 * i), y = fe_unpack(a + 8 * i + 8);
    for (int k = 0; k < iters; k++) { x = fe_sqr<P>(x); y = fe_sqr<P>(y); }
    fe_pack(out + 8 * i, fe_canon<P>(fe_norm(fe_add(x, y))));
}
template <class C> __global__ void k_madd(const u32* a, u32* out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Aff q = aff_load_dev(a + 16 * i);
    Jac acc = jac_dbl<C>(jac_from_aff<C>(q));
    for (int k = 0; k < iters; k++) acc = jac_madd<C>(acc, q);
    fe_pack(out + 8 * i, fe_canon<typename C::Fq>(acc.X));
}
template <class C> __global__ void k_dbl(const u32* a, u32* out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Jac acc = jac_from_aff<C>(aff_load_dev(a + 16 * i));
    for (int k = 0; k < iters; k++) acc = jac_dbl<C>(acc);
    fe_pack(out + 8 * i, fe_canon<typename C::Fq>(acc.X));
}

// ---- A/B arm (a) of the accumulate study: batched-affine addition (Montgomery's trick) ----------------------------------------
// pts: 2*npairs affine points (resident layout, 64 B each); out[i] = pts[2i] + pts[2i+1] (affine); pref: npairs x 8 words scratch.
// Lane t owns the pairs t, t + T, t + 2T, ... (T = lanes of the grid: coalesced at every step), M of them per inversion:
// pass 1 multiplies the denominators x2 - x1 into a running product (stored per pair), one inversion, pass 2 walks back and
// forms lambda, x3, y3.  6 modular products per addition + one inversion per M, against 11 for jac_madd — but every addition
// moves 2 x 64 B in, 32 + 32 B of prefix scratch, 64 B out, and reads the x coordinates twice.
template <class C> __global__ void __launch_bounds__(256) k_baff(const u32* __restrict__ pts, u32* __restrict__ pref, u32* __restrict__ out, u32 npairs, u32 M) {
    typedef typename C::Fq F;
    const u32 T = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 >= npairs) return;
    Fe acc = fe_one<F>();
    u32 cnt = 0;
    for (u32 p = t0; cnt < M && p < npairs; cnt++, p += T) {
        u32 w[8];
        load_words8(w, pts + (size_t)(2 * p) * 16); const Fe x1 = fe_unpack(w);
        load_words8(w, pts + (size_t)(2 * p + 1) * 16); const Fe x2 = fe_unpack(w);
        acc = fe_mul<F>(acc, fe_wred<F>(fe_sub<F, 2>(x2, x1)));
        fe_pack(w, fe_canon<F>(acc));
        store_words8(pref + (size_t)p * 8, w);
    }
    Fe inv = fe_inv<F>(acc);
    for (int j = (int)cnt - 1; j >= 0; j--) {
        const u32 p = t0 + (u32)j * T;
        const Aff P = load_aff_dev(pts + (size_t)(2 * p) * 16), Q = load_aff_dev(pts + (size_t)(2 * p + 1) * 16);
        Fe prev = fe_one<F>();
        if (j > 0) { u32 w[8]; load_words8(w, pref + (size_t)(p - T) * 8); prev = fe_unpack(w); }
        const Fe d = fe_wred<F>(fe_sub<F, 2>(Q.x, P.x));
        const Fe dinv = fe_mul<F>(inv, prev);
        inv = fe_mul<F>(inv, d);
        const Fe lam = fe_mul<F>(fe_wred<F>(fe_sub<F, 2>(Q.y, P.y)), dinv);
        const Fe x3 = fe_wred<F>(fe_sub<F, 4>(fe_sqr<F>(lam), fe_norm(fe_add(P.x, Q.x))));
        const Fe y3 = fe_wred<F>(fe_sub<F, 2>(fe_mul<F>(lam, fe_wred<F>(fe_sub<F, 2>(P.x, x3))), P.y));
        Aff o; o.x = fe_canon<F>(x3); o.y = fe_canon<F>(y3);
        u32 w[16];
        aff_store_dev(w, o);
        store_words8(out + (size_t)p * 16, w);
        store_words8(out + (size_t)p * 16 + 8, w + 8);
    }
}
// the same additions as mixed Jacobian adds through memory (what one level of a Jacobian tree costs): out = 96 B Jacobian
template <class C> __global__ void __launch_bounds__(256) k_madd_mem(const u32* __restrict__ pts, u32* __restrict__ out, u32 npairs) {
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const Aff P = load_aff_dev(pts + (size_t)(2 * p) * 16), Q = load_aff_dev(pts + (size_t)(2 * p + 1) * 16);
    const Jac r = jac_madd<C>(jac_madd<C>(jac_inf<C>(), P), Q);
    store_jac_ws<C>(out + (size_t)p * 24, r);
}

template <class F> double time_kernel(F launch, int reps = 5) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    const int blocks = cus * 8, threads = 256;  // 8 waves/SIMD
    void* buf; CHECK(hipMalloc(&buf, (size_t)blocks * threads * 64 + 64));
    CHECK(hipMemset(buf, 1, (size_t)blocks * threads * 64 + 64));
    { std::vector<u32> h((size_t)blocks * threads * 16 + 16); u32 x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 4; } CHECK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    const int iters = 2000;
    const double ninstr = (double)iters * 64 * blocks * threads / 64;  // wave-instructions
#define RUN(name, T, arg)                                                                              \
    { double s = time_kernel([&] { name<<<blocks, threads>>>((T*)buf, iters, arg); });                 \
      printf("%-14s %8.3f ms  %7.2f wave-instr/clk/CU (at 2.4GHz)  %6.2f T lane-ops/s\n", #name, s * 1e3, \
             ninstr / s / cus / 2.4e9, ninstr * 64 / s / 1e12); }
    RUN(k_add, u32, 12345u)
    RUN(k_addc, u32, 12345u)
    RUN(k_mullo, u32, 12345u)
    RUN(k_mulhi, u32, 12345u)
    RUN(k_mad24, u32, 12345u)
    RUN(k_mulhi24, u32, 12345u)
    RUN(k_mad64, u64, 12345u)
    RUN(k_lshladd64, u64, 12345u)
    RUN(k_fma64, double, 1.5)
    u32* in = (u32*)buf; u32* outp = (u32*)buf + (size_t)blocks * threads * 8;
    for (int occ = 1; occ <= 8; occ *= 2) {
        int b2 = cus * occ, it = 200;
        double s = time_kernel([&] { k_femul<SecqFq><<<b2, threads>>>(in, outp, it); });
        printf("fe_mul(29-bit)  waves/SIMD=%d  %8.3f ms  %8.2f G modmul/s\n", occ, s * 1e3, 2.0 * it * b2 * threads / s / 1e9);
    }
    for (int occ = 1; occ <= 8; occ *= 2) {
        int b2 = cus * occ, it = 200;
        double s = time_kernel([&] { k_fesqr<SecqFq><<<b2, threads>>>(in, outp, it); });
        printf("fe_sqr(29-bit)  waves/SIMD=%d  %8.3f ms  %8.2f G modsqr/s\n", occ, s * 1e3, 2.0 * it * b2 * threads / s / 1e9);
    }
    for (int occ = 1; occ <= 8; occ *= 2) {
        int b2 = cus * occ, it = 64;
        double s = time_kernel([&] { k_madd<Secq><<<b2, threads>>>(in, outp, it); });
        printf("jac_madd secq   waves/SIMD=%d  %8.3f ms  %8.2f G madd/s\n", occ, s * 1e3, 1.0 * it * b2 * threads / s / 1e9);
        s = time_kernel([&] { k_dbl<Secq><<<b2, threads>>>(in, outp, it); });
        printf("jac_dbl  secq   waves/SIMD=%d  %8.3f ms  %8.2f G dbl/s\n", occ, s * 1e3, 1.0 * it * b2 * threads / s / 1e9);
    }
    { int b2 = cus * 4, it = 64;
      double s = time_kernel([&] { k_madd<Zorro><<<b2, threads>>>(in, outp, it); });
      printf("jac_madd zorro  waves/SIMD=4  %8.3f ms  %8.2f G madd/s\n", s * 1e3, 1.0 * it * b2 * threads / s / 1e9);
      s = time_kernel([&] { k_dbl<Zorro><<<b2, threads>>>(in, outp, it); });
      printf("jac_dbl  zorro  waves/SIMD=4  %8.3f ms  %8.2f G dbl/s\n", s * 1e3, 1.0 * it * b2 * threads / s / 1e9); }
    // ---- batched-affine additions through memory vs mixed adds (accumulate A/B, arm a) ----
    {
        const u32 npairs = 1u << 22;   // 4 M additions: 512 MB of input points
        u32 *pts, *pref, *outb;
        CHECK(hipMalloc(&pts, (size_t)npairs * 2 * 64)); CHECK(hipMalloc(&pref, (size_t)npairs * 32)); CHECK(hipMalloc(&outb, (size_t)npairs * 96));
        // distinct valid-looking field elements are enough for timing (no exceptional cases): x in [1, 2^250)
        { std::vector<u32> h((size_t)npairs * 2 * 16); u32 x = 777; for (size_t i = 0; i < h.size(); i++) { x = x * 1664525u + 1013904223u; h[i] = (i % 8 == 7) ? (x >> 8) : x; } CHECK(hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
        double s = time_kernel([&] { k_madd_mem<Secq><<<npairs / 256, 256>>>(pts, outb, npairs); });
        printf("madd through memory (2 x 64 B in, 96 B out)        %8.3f ms  %7.2f G adds/s  %6.0f GB/s\n", s * 1e3, npairs / s / 1e9, npairs * 224.0 / s / 1e9);
        for (u32 M : {8u, 16u, 32u, 64u, 128u, 256u}) {
            const u32 lanes = npairs / M;
            s = time_kernel([&] { k_baff<Secq><<<(lanes + 255) / 256, 256>>>(pts, pref, outb, npairs, M); });
            printf("batched-affine add, M = %3u per inversion (%7u lanes) %8.3f ms  %7.2f G adds/s  %6.0f GB/s\n", M, lanes, s * 1e3, npairs / s / 1e9, npairs * 320.0 / s / 1e9);
        }
        CHECK(hipFree(pts)); CHECK(hipFree(pref)); CHECK(hipFree(outb));
    }
    return 0;
}
