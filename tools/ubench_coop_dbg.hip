// Diagnostic for ecq.cuh on the GPU: runs qjac_add step by step in every lane of a quad next to the lane-private formulas and reports,
// per exchange level, how many lanes received a value that differs from what they compute themselves.  Two exchange primitives:
// DPP quad_perm (what ecq.cuh uses) and ds_bpermute (__shfl).
// Build: hipcc --offload-arch=gfx950 -O3 -I ark_bulletproofs_amd/csrc tools/ubench_coop_dbg.hip -o tools/ubench_coop_dbg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "msm.cuh"
using namespace arkbp;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int K, int MODE> __device__ __forceinline__ Fe bc(const Fe& a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        if (MODE == 0) r.l[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)a.l[i], K * 0x55, 0xF, 0xF, false);
        else if (MODE == 1) r.l[i] = (u32)__shfl((int)a.l[i], (int)((threadIdx.x & 60u) | K), 64);
        else r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)a.l[i], K * 0x55, 0xF, 0xF, true);
    }
    return r;
}
template <class F> __device__ __forceinline__ bool same(const Fe& a, const Fe& b) { return fe_eq_exact(fe_canon<F>(a), fe_canon<F>(b)); }

template <class C, int MODE> __global__ void __launch_bounds__(256) k_dbg(const u32* a, u32* bad /* 16 counters */, u32* out) {
    typedef typename C::Fq F;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 2;
    const u32 q = threadIdx.x & 3u;
    const Aff qa = aff_load_dev(a + 16 * i);
    const Jac p = jac_dbl<C>(jac_from_aff<C>(qa)), q_ = jac_dbl<C>(p);
    // lane-private values
    const Fe eA = fe_sqr<F>(p.Z), eB = fe_sqr<F>(q_.Z), eD = fe_mul<F>(p.Y, q_.Z), eE = fe_mul<F>(q_.Y, p.Z);
    const Fe eU1 = fe_mul<F>(p.X, eB), eU2 = fe_mul<F>(q_.X, eA), eS1 = fe_mul<F>(eD, eB), eS2 = fe_mul<F>(eE, eA);
    auto chk = [&](int slot, const Fe& got, const Fe& exp) { if (!same<F>(got, exp)) atomicAdd(&bad[slot], 1u); };
    Fe m = fe_mul<F>(quad_pick(q, p.Z, q_.Z, p.Y, q_.Y), quad_pick(q, p.Z, q_.Z, q_.Z, p.Z));
    chk(0, m, q == 0 ? eA : q == 1 ? eB : q == 2 ? eD : eE);                 // own product of level 1
    const Fe A = bc<0, MODE>(m), B = bc<1, MODE>(m);
    chk(1, A, eA); chk(2, B, eB);
    m = fe_mul<F>(quad_pick(q, p.X, q_.X, m, m), quad_pick(q, B, A, B, A));
    chk(3, m, q == 0 ? eU1 : q == 1 ? eU2 : q == 2 ? eS1 : eS2);
    const Fe U1 = bc<0, MODE>(m), U2 = bc<1, MODE>(m), S1 = bc<2, MODE>(m), S2 = bc<3, MODE>(m);
    chk(4, U1, eU1); chk(5, U2, eU2); chk(6, S1, eS1); chk(7, S2, eS2);
    const Fe H = fe_sub<F, 2>(U2, U1), r = fe_sub<F, 2>(S2, S1);
    const Fe eH = fe_sub<F, 2>(eU2, eU1), er = fe_sub<F, 2>(eS2, eS1);
    m = fe_mul<F>(quad_pick(q, H, p.Z, r, H), quad_pick(q, H, q_.Z, r, H));
    const Fe HH = bc<0, MODE>(m), Cc = bc<1, MODE>(m), rr = bc<2, MODE>(m);
    chk(8, HH, fe_sqr<F>(eH)); chk(9, Cc, fe_mul<F>(p.Z, q_.Z)); chk(10, rr, fe_sqr<F>(er));
    // the whole operation through ecq.cuh and through ec.cuh
    const Jac w = jac_add<C>(p, q_), g = qjac_add<C>(p, q_, q);
    chk(11, g.X, w.X); chk(12, g.Y, w.Y); chk(13, g.Z, w.Z);
    const Jac w2 = jac_madd<C>(p, qa), g2 = qjac_madd<C>(p, qa, q);
    chk(14, g2.X, w2.X); if (!same<F>(g2.Y, w2.Y) || !same<F>(g2.Z, w2.Z)) atomicAdd(&bad[15], 1u);
    if (t < 8) { fe_pack(out + 16 * t, fe_canon<F>(g.Y)); fe_pack(out + 16 * t + 8, fe_canon<F>(w.Y)); }
}
template <class C, int MODE> void run(const char* name, const u32* in, u32* bad, u32* out) {
    CHECK(hipMemset(bad, 0, 64));
    k_dbg<C, MODE><<<16, 256>>>(in, bad, out);
    CHECK(hipDeviceSynchronize());
    u32 h[16], ho[128];
    CHECK(hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ho, out, 512, hipMemcpyDeviceToHost));
    printf("%s  lanes (of 4096) with a wrong value: own1 %u | A %u B %u | own2 %u | U1 %u U2 %u S1 %u S2 %u | HH %u C %u rr %u | qjac_add X %u Y %u Z %u | qjac_madd X %u YZ %u\n", name,
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
    printf("    lane 0..3 qjac_add Y: %08x %08x %08x %08x   expected %08x\n", ho[0], ho[16], ho[32], ho[48], ho[8]);
}
int main() {
    const size_t pts = 1024;
    void* buf; CHECK(hipMalloc(&buf, pts * 64 + 64 + 1024));
    { std::vector<u32> h(pts * 16); u32 x = 12345; for (size_t i = 0; i < h.size(); i++) { x = x * 1664525u + 1013904223u; h[i] = (i % 8 == 7) ? (x >> 8) : x; } CHECK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    u32* in = (u32*)buf; u32* bad = in + pts * 16; u32* out = bad + 16;
    run<Secq, 0>("secq  DPP update_dpp(old=0, bound_ctrl=0)", in, bad, out);
    run<Secq, 2>("secq  DPP mov_dpp(bound_ctrl=1)         ", in, bad, out);
    run<Secq, 1>("secq  ds_bpermute (__shfl)              ", in, bad, out);
    run<Zorro, 0>("zorro DPP update_dpp                    ", in, bad, out);
    run<Zorro, 1>("zorro ds_bpermute                       ", in, bad, out);
    return 0;
}
