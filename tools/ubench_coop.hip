// Cooperating lanes for the latency-bound group arithmetic (SURVEY.md §7 "both layouts must be benchmarked").
//   arm 0  one lane per point operation (ec.cuh: jac_add / jac_madd / jac_dbl)            — what the kernels use
//   arm 1  four lanes (a DPP quad) per point operation (ecq.cuh: qjac_add / qjac_madd / qjac_dbl): one modular product per lane and
//          dependency level, operands exchanged by DPP quad_perm
//   arm 2  limb-per-lane Montgomery product (ecq.cuh: fe_mul_limblane): 9 lanes of a 16-lane row hold one field element
// Every arm is checked against arm 0 on the same inputs before it is timed.  Times are per operation in a chain of dependent
// operations (the shape of a reduction tree level or of a scalar-multiplication ladder), for 1 wave per CU, 1 wave per SIMD and
// 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -I ark_bulletproofs_amd/csrc tools/ubench_coop.hip -o tools/ubench_coop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "msm.cuh"
#include "ecq.cuh"
using namespace arkbp;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class C> __device__ __forceinline__ void start_points(const u32* a, int i, Jac& acc, Jac& q, Aff& qa) {
    qa = aff_load_dev(a + 16 * i);
    acc = jac_dbl<C>(jac_from_aff<C>(qa));
    q = jac_dbl<C>(acc);
}
template <class C> __device__ __forceinline__ void emit(u32* out, int i, const Jac& acc) {
    typedef typename C::Fq F;
    fe_pack(out + 24 * i, fe_canon<F>(acc.X)); fe_pack(out + 24 * i + 8, fe_canon<F>(acc.Y)); fe_pack(out + 24 * i + 16, fe_canon<F>(acc.Z));
}
// OP: 0 add, 1 madd, 2 dbl.  One lane per point.
template <class C, int OP> __global__ void __launch_bounds__(256) k_single(const u32* a, u32* out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Jac acc, q; Aff qa;
    start_points<C>(a, i, acc, q, qa);
#pragma unroll 1
    for (int k = 0; k < iters; k++) {
        if (OP == 0) acc = jac_add<C>(acc, q); else if (OP == 1) acc = jac_madd<C>(acc, qa); else acc = jac_dbl<C>(acc);
    }
    emit<C>(out, i, acc);
}
// four lanes per point: point index = lane / 4; all four lanes of a quad write the (same) result, lane 0's copy is compared
template <class C, int OP> __global__ void __launch_bounds__(256) k_quad(const u32* a, u32* out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 2;
    const u32 ql = threadIdx.x & 3u;
    Jac acc, q; Aff qa;
    start_points<C>(a, i, acc, q, qa);
#pragma unroll 1
    for (int k = 0; k < iters; k++) {
        if (OP == 0) acc = qjac_add<C>(acc, q, ql); else if (OP == 1) acc = qjac_madd<C>(acc, qa, ql); else acc = qjac_dbl<C>(acc, ql);
    }
    if (ql == 0) emit<C>(out, i, acc);
}
// chain of dependent products x <- x * y: one lane per element / limb-per-lane (element index = lane / 16)
template <class F> __global__ void __launch_bounds__(256) k_mul_single(const u32* a, u32* out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fe x = fe_unpack(a + 16 * i), y = fe_unpack(a + 16 * i + 8);
#pragma unroll 1
    for (int k = 0; k < iters; k++) x = fe_mul<F>(x, y);
    fe_pack(out + 8 * i, fe_canon<F>(x));
}
template <class F> __global__ void __launch_bounds__(256) k_mul_limblane(const u32* a, u32* out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 4;
    const u32 j = threadIdx.x & 15u;
    const Fe x0 = fe_unpack(a + 16 * i), y0 = fe_unpack(a + 16 * i + 8);
    u32 xj = 0, yj = 0;
#pragma unroll
    for (int q = 0; q < 9; q++) { xj = j == (u32)q ? x0.l[q] : xj; yj = j == (u32)q ? y0.l[q] : yj; }
#pragma unroll 1
    for (int k = 0; k < iters; k++) xj = fe_mul_limblane<F>(xj, yj, j);
    // collect the row's limbs in its lane 0
    Fe r;
    r.l[0] = row_bcast_u32<0>(xj); r.l[1] = row_bcast_u32<1>(xj); r.l[2] = row_bcast_u32<2>(xj); r.l[3] = row_bcast_u32<3>(xj); r.l[4] = row_bcast_u32<4>(xj);
    r.l[5] = row_bcast_u32<5>(xj); r.l[6] = row_bcast_u32<6>(xj); r.l[7] = row_bcast_u32<7>(xj); r.l[8] = row_bcast_u32<8>(xj);
    if (j == 0) fe_pack(out + 8 * i, fe_canon<F>(fe_norm(r)));
}

template <class Fn> double time_kernel(Fn launch, int reps = 5) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps / 1e3;
}
static size_t mismatches(const std::vector<u32>& x, const std::vector<u32>& y, size_t items, size_t words) {
    size_t bad = 0;
    for (size_t i = 0; i < items; i++) { bool same = true; for (size_t w = 0; w < words; w++) same = same && x[i * words + w] == y[i * words + w]; bad += same ? 0 : 1; }
    return bad;
}
template <class C> void run_curve(const char* name, const u32* in, u32* outA, u32* outB, int cus) {
    const char* opn[3] = {"Jacobian add", "mixed add", "doubling"};
    const int total = 64;
    printf("== %s ==\n", name);
    // correctness: 4096 points, 7 chained operations each
    {
        const int pts = 4096;
        std::vector<u32> ha((size_t)pts * 24), hb((size_t)pts * 24);
#define CHECK_OP(OP) { \
            k_single<C, OP><<<pts / 256, 256>>>(in, outA, 7); k_quad<C, OP><<<pts * 4 / 256, 256>>>(in, outB, 7); CHECK(hipDeviceSynchronize()); \
            CHECK(hipMemcpy(ha.data(), outA, ha.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hb.data(), outB, hb.size() * 4, hipMemcpyDeviceToHost)); \
            printf("check %-13s quad vs single lane: %zu of %d results differ\n", opn[OP], mismatches(ha, hb, pts, 24), pts); }
        CHECK_OP(0) CHECK_OP(1) CHECK_OP(2)
        std::vector<u32> ma((size_t)pts * 8), mb((size_t)pts * 8);
        k_mul_single<typename C::Fq><<<pts / 256, 256>>>(in, outA, 5); k_mul_limblane<typename C::Fq><<<pts * 16 / 256, 256>>>(in, outB, 5); CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(ma.data(), outA, ma.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(mb.data(), outB, mb.size() * 4, hipMemcpyDeviceToHost));
        printf("check modular product limb-per-lane vs single lane: %zu of %d results differ\n", mismatches(ma, mb, pts, 8), pts);
    }
    for (int cfg = 0; cfg < 3; cfg++) {
        const int threads = cfg == 0 ? 64 : 256, blocks = cfg <= 1 ? cus : 2 * cus;
        const char* cn = cfg == 0 ? "1 wave per CU  " : cfg == 1 ? "1 wave per SIMD" : "2 waves per SIMD";
#define TIME_OP(OP) { \
            const double s0 = time_kernel([&] { k_single<C, OP><<<blocks, threads>>>(in, outA, total); }); \
            const double s1 = time_kernel([&] { k_quad<C, OP><<<blocks, threads>>>(in, outB, total); }); \
            printf("%s  %-13s  single lane %6.2f us/op (%7.2f G op/s chip-wide at this fill)   quad %6.2f us/op (%7.2f G op/s)   latency x%.2f  throughput x%.2f\n", cn, opn[OP], \
                   s0 * 1e6 / total, (double)blocks * threads * total / s0 / 1e9, s1 * 1e6 / total, (double)blocks * threads / 4 * total / s1 / 1e9, s0 / s1, (s0 / s1) / 4.0); }
        TIME_OP(0) TIME_OP(1) TIME_OP(2)
        const double m0 = time_kernel([&] { k_mul_single<typename C::Fq><<<blocks, threads>>>(in, outA, 256); });
        const double m1 = time_kernel([&] { k_mul_limblane<typename C::Fq><<<blocks, threads>>>(in, outB, 256); });
        printf("%s  modular product single lane %6.3f us (%7.1f G/s)   limb-per-lane (16 lanes) %6.3f us (%7.1f G/s)   latency x%.2f  throughput x%.3f\n", cn, m0 * 1e6 / 256,
               (double)blocks * threads * 256 / m0 / 1e9, m1 * 1e6 / 256, (double)blocks * threads / 16 * 256 / m1 / 1e9, m0 / m1, (m0 / m1) / 16.0);
    }
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t lanes = (size_t)cus * 2 * 256;
    void* buf; CHECK(hipMalloc(&buf, lanes * 64 + 2 * lanes * 96));
    { std::vector<u32> h(lanes * 16); u32 x = 12345; for (size_t i = 0; i < h.size(); i++) { x = x * 1664525u + 1013904223u; h[i] = (i % 8 == 7) ? (x >> 8) : x; } CHECK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    u32* in = (u32*)buf; u32* outA = in + lanes * 16; u32* outB = outA + lanes * 24;
    printf("%s, %d CUs\n", prop.name, cus);
    run_curve<Secq>("secq256k1 (a = 0)", in, outA, outB, cus);
    run_curve<Zorro>("zorro (a = 6)", in, outA, outB, cus);
    return 0;
}
