// Quad-cooperative group arithmetic for gfx950: FOUR adjacent lanes (a DPP "quad") share one point operation.
//
// Why: the latency-bound stretches of the path — the bucket-reduction and marginal trees of a mid-size MSM, the fold rounds
// with fewer lanes than the chip has (src/inner_product_proof.rs:143-155,219-224 at n <= 2^15) — are serial chains of group
// operations in waves that have their SIMD to themselves.  A wave issues one VALU instruction per ~4 cycles whatever its active
// lane count, so a Jacobian addition costs a lone lane 16 modular products back to back (7.7 us measured).  The 16 products have
// dependency depth 5: spread over the lanes of a quad, one product per lane and level, the same addition is 5 products deep.
// Operands travel between the lanes of a quad by DPP quad_perm moves (full-rate VALU, no LDS); every lane of the quad enters with
// the same (replicated) operands and leaves with the same result, so control flow stays quad-uniform (exceptional cases included).
//
// The north_star's "limb-per-lane" layout (9 lanes of a 16-lane DPP row hold the 9 limbs of one element) is the other cooperative
// shape; fe_mul_limblane below implements its Montgomery product for the A/B in tools/ubench_coop.hip.
#pragma once
#include "ec.cuh"

namespace arkbp {

#if defined(__HIPCC__)
#define ARKBP_QD __device__ __forceinline__
// value of x in lane (quad base + k), for every lane of the quad; k compile-time
// The result is pinned to a VGPR of its own (empty asm): LLVM's DPP combine otherwise folds the move into the consuming VALU
// instruction, and the folded in-place form it produced here — `v_subrev_u32_dpp v60, v60, v63 quad_perm:[1,1,1,1]`, destination =
// DPP source — delivered every lane its OWN value on gfx950 (tools/ubench_coop_dbg.hip, profiles/r03_ubench_coop_dbg.txt: the last
// exchange of qjac_add / qjac_madd came out wrong in the even lanes of every quad, with DPP and with ds_bpermute checks alike;
// plain v_mov_b32_dpp exchanges are right at every level).
template <int K> __device__ __forceinline__ u32 quad_bcast(u32 x) {
    u32 r = (u32)__builtin_amdgcn_update_dpp(0, (int)x, K * 0x55, 0xF, 0xF, false);
    asm volatile("" : "+v"(r));
    return r;
}
template <int K> __device__ __forceinline__ Fe quad_bcast_fe(const Fe& a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = quad_bcast<K>(a.l[i]);
    return r;
}
#else
#define ARKBP_QD inline
// CPU stand-in for the DPP exchange (tests/test_fp29_host.py): the four lanes of a quad run one after the other, several rounds;
// an exchange site hands out what the lanes stored there in the previous round.  A site's inputs depend only on earlier sites, so
// after r rounds the first r sites carry the final values — the schedule is checked without a GPU.
struct QuadSim {
    Fe slot[32][4];
    int site = 0, lane = 0;
};
inline QuadSim& quad_sim() { static thread_local QuadSim s; return s; }
template <int K> inline Fe quad_bcast_fe(const Fe& a) {
    QuadSim& s = quad_sim();
    const int c = s.site++;
    s.slot[c][s.lane] = a;
    return s.slot[c][K];
}
#endif
// this lane's choice among four replicated values by its position in the quad (q = lane & 3)
ARKBP_QD Fe quad_pick(u32 q, const Fe& a0, const Fe& a1, const Fe& a2, const Fe& a3) {
    Fe r;
    const bool hi = (q & 2u) != 0, odd = (q & 1u) != 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u32 lo = odd ? a1.l[i] : a0.l[i];
        const u32 up = odd ? a3.l[i] : a2.l[i];
        r.l[i] = hi ? up : lo;
    }
    return r;
}
ARKBP_QD Fe quad_pick2(u32 q, const Fe& a0, const Fe& a1) {   // lanes 0 / 1 (lanes 2, 3 take a0 / a1 as well)
    Fe r;
    const bool odd = (q & 1u) != 0;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = odd ? a1.l[i] : a0.l[i];
    return r;
}

// Jacobian + Jacobian, replicated in / replicated out.  Levels (products per level in lanes 0..3):
//   1: A = Z1^2 | B = Z2^2 | D = Y1*Z2 | E = Y2*Z1         2: U1 = X1*B | U2 = X2*A | S1 = D*B | S2 = E*A
//   3: HH = H^2 | C = Z1*Z2 | rr = r^2 | -                 4: HHH = H*HH | V = U1*HH | Z3 = C*H | -
//   5: T1 = r*(V - X3) | T2 = S1*HHH | - | -
template <class C> ARKBP_QD Jac qjac_add(const Jac& p, const Jac& q_, u32 q) {
    typedef typename C::Fq F;
    if (jac_is_inf(q_)) return p;
    if (jac_is_inf(p)) return q_;
    // level 1
    Fe m = fe_mul<F>(quad_pick(q, p.Z, q_.Z, p.Y, q_.Y), quad_pick(q, p.Z, q_.Z, q_.Z, p.Z));
    const Fe A = quad_bcast_fe<0>(m), B = quad_bcast_fe<1>(m);
    // level 2: lanes 2, 3 multiply their own product (D, E) by B, A
    m = fe_mul<F>(quad_pick(q, p.X, q_.X, m, m), quad_pick(q, B, A, B, A));
    const Fe U1 = quad_bcast_fe<0>(m), U2 = quad_bcast_fe<1>(m), S1 = quad_bcast_fe<2>(m), S2 = quad_bcast_fe<3>(m);
    const Fe H = fe_sub<F, 2>(U2, U1);    // V <= 3.2
    const Fe r = fe_sub<F, 2>(S2, S1);
    if (fe_is_zero_mod<F>(H)) {           // same x (quad-uniform: H is replicated)
        if (fe_is_zero_mod<F>(r)) return jac_dbl<C>(p);
        return jac_inf<C>();
    }
    // level 3
    m = fe_mul<F>(quad_pick(q, H, p.Z, r, H), quad_pick(q, H, q_.Z, r, H));
    const Fe HH = quad_bcast_fe<0>(m), Cc = quad_bcast_fe<1>(m), rr = quad_bcast_fe<2>(m);
    // level 4
    m = fe_mul<F>(quad_pick(q, H, U1, Cc, H), quad_pick(q, HH, HH, H, HH));
    const Fe HHH = quad_bcast_fe<0>(m), V = quad_bcast_fe<1>(m);
    Jac o;
    o.Z = quad_bcast_fe<2>(m);
    o.X = fe_wred<F>(fe_sub<F, 4>(rr, fe_add(HHH, fe_dbl(V))));
    // level 5
    m = fe_mul<F>(quad_pick2(q, r, S1), quad_pick2(q, fe_sub<F, 4>(V, o.X), HHH));
    o.Y = fe_wred<F>(fe_sub<F, 2>(quad_bcast_fe<0>(m), quad_bcast_fe<1>(m)));
    return o;
}

// Jacobian + affine (q.y may be a lazy negation), replicated in / out.
//   1: A = Z1^2 | E = y2*Z1     2: U2 = x2*A | S2 = E*A     3: HH = H^2 | rr = r^2 | Z3 = Z1*H     4: HHH = H*HH | V = X1*HH
//   5: T1 = r*(V - X3) | T2 = Y1*HHH
template <class C> ARKBP_QD Jac qjac_madd(const Jac& p, const Aff& a, u32 q) {
    typedef typename C::Fq F;
    if (aff_is_inf(a)) return p;
    if (jac_is_inf(p)) {
        Jac r;
        r.X = a.x;
        r.Y = fe_wred<F>(a.y);
        r.Z = fe_one<F>();
        return r;
    }
    Fe m = fe_mul<F>(quad_pick2(q, p.Z, a.y), p.Z);
    const Fe A = quad_bcast_fe<0>(m);
    m = fe_mul<F>(quad_pick2(q, a.x, m), A);
    const Fe U2 = quad_bcast_fe<0>(m), S2 = quad_bcast_fe<1>(m);
    const Fe H = fe_sub<F, 4>(U2, p.X);
    const Fe r = fe_sub<F, 4>(S2, p.Y);
    if (fe_is_zero_mod<F>(H)) {
        if (fe_is_zero_mod<F>(r)) return jac_dbl<C>(p);
        return jac_inf<C>();
    }
    m = fe_mul<F>(quad_pick(q, H, r, p.Z, H), quad_pick(q, H, r, H, H));
    const Fe HH = quad_bcast_fe<0>(m), rr = quad_bcast_fe<1>(m);
    Jac o;
    o.Z = quad_bcast_fe<2>(m);
    m = fe_mul<F>(quad_pick2(q, H, p.X), HH);
    const Fe HHH = quad_bcast_fe<0>(m), V = quad_bcast_fe<1>(m);
    o.X = fe_wred<F>(fe_sub<F, 4>(rr, fe_add(HHH, fe_dbl(V))));
    m = fe_mul<F>(quad_pick2(q, r, p.Y), quad_pick2(q, fe_sub<F, 4>(V, o.X), HHH));
    o.Y = fe_wred<F>(fe_sub<F, 2>(quad_bcast_fe<0>(m), quad_bcast_fe<1>(m)));
    return o;
}

// doubling, replicated in / out.
//   a = 0:  1: A = X^2 | B = Y^2 | Z3' = Y*Z      2: Cc = B^2 | D' = X*B       3: F = E^2 (E = 3A)      4: Y3' = E*(D - X3)
//   a != 0: 1: XX | YY | ZZ | Z3' = Y*Z           2: YYYY | S' = X*YY | Z4 = ZZ^2   3: F = M^2 (M = 3XX + a*Z4)   4: Y3' = M*(S - X3)
template <class C> ARKBP_QD Jac qjac_dbl(const Jac& p, u32 q) {
    typedef typename C::Fq F;
    if (jac_is_inf(p)) return p;
    Jac o;
    if (C::A_ZERO) {
        Fe m = fe_mul<F>(quad_pick(q, p.X, p.Y, p.Y, p.X), quad_pick(q, p.X, p.Y, p.Z, p.X));
        const Fe A = quad_bcast_fe<0>(m), B = quad_bcast_fe<1>(m);
        o.Z = fe_norm(fe_dbl(quad_bcast_fe<2>(m)));                 // 2*Y*Z, V <= 2.1
        m = fe_mul<F>(quad_pick2(q, B, p.X), B);
        const Fe Cc = quad_bcast_fe<0>(m);
        const Fe D = fe_times<4>(quad_bcast_fe<1>(m));
        const Fe E = fe_times<3>(A);
        const Fe Fs = fe_mul<F>(E, E);                              // (every lane: the product is needed replicated, no exchange)
        o.X = fe_wred<F>(fe_sub<F, 16>(Fs, fe_dbl(D)));
        const Fe Y3 = fe_mul<F>(E, fe_sub<F, 4>(D, o.X));
        o.Y = fe_wred<F>(fe_sub<F, 16>(Y3, fe_times<8>(Cc)));
    } else {
        Fe m = fe_mul<F>(quad_pick(q, p.X, p.Y, p.Z, p.Y), quad_pick(q, p.X, p.Y, p.Z, p.Z));
        const Fe XX = quad_bcast_fe<0>(m), YY = quad_bcast_fe<1>(m), ZZ = quad_bcast_fe<2>(m);
        o.Z = fe_norm(fe_dbl(quad_bcast_fe<3>(m)));
        m = fe_mul<F>(quad_pick(q, YY, p.X, ZZ, ZZ), quad_pick(q, YY, YY, ZZ, ZZ));
        const Fe YYYY = quad_bcast_fe<0>(m);
        const Fe S = fe_times<4>(quad_bcast_fe<1>(m));
        const Fe Z4 = quad_bcast_fe<2>(m);
        const Fe aZ4 = fe_times<(C::A_ZERO ? 1 : C::A_SMALL)>(Z4);
        const Fe M = fe_norm(fe_add(fe_times<3>(XX), aZ4));
        const Fe Fs = fe_mul<F>(M, M);
        o.X = fe_wred<F>(fe_sub<F, 16>(Fs, fe_dbl(S)));
        const Fe Y3 = fe_mul<F>(M, fe_sub<F, 4>(S, o.X));
        o.Y = fe_wred<F>(fe_sub<F, 16>(Y3, fe_times<8>(YYYY)));
    }
    return o;
}

#if defined(__HIPCC__)
// ---- limb-per-lane Montgomery product (the north_star's layout) ------------------------------------------------------------
// Lane j (j = 0..8) of a 16-lane row holds limb j of a and of b; lanes 9..15 of the row hold zeros.  Operand scanning: step i adds
// a_i * b_j into column accumulator j (a_i broadcast through the row: ds_swizzle — gfx9 DPP has no row_share), lane 0's column
// yields the Montgomery multiplier m_i (second broadcast), every lane adds m_i * p_j, then the accumulators move one lane down
// (DPP row_shl) and lane 0 takes the carry of the column that just became zero.  9 steps of 2 multiply-adds per lane instead of
// 162 in one lane; two carry steps bring the result back to limbs < 2^29 + 2^7 (L = 1 within the product contract; limb 8 keeps the
// excess as in fe_mul).  Measured against fe_mul in tools/ubench_coop.hip.
template <int I> __device__ __forceinline__ u32 row_bcast_u32(u32 x) {   // value of lane I of this lane's row of 16
    return (u32)__builtin_amdgcn_ds_swizzle((int)x, 0x10 | (I << 5));
}
__device__ __forceinline__ u32 row_from_above_u32(u32 x) {   // lane j <- lane j + 1 within the row; 0 for the row's last lane
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xF, 0xF, true);
}
__device__ __forceinline__ u32 row_from_below_u32(u32 x) {   // lane j <- lane j - 1 within the row; 0 for the row's first lane
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
}
template <class P, int I> __device__ __forceinline__ void limblane_step(u64& acc, u32 a_j, u32 b_j, u32 p_j, u32 j) {
    const u32 a_i = row_bcast_u32<I>(a_j);
    acc += (u64)a_i * b_j;
    const u32 acc0 = row_bcast_u32<0>((u32)acc);
    const u32 m = (acc0 * P::NINV29) & M29;
    acc += (u64)m * p_j;
    const u64 up = ((u64)row_from_above_u32((u32)(acc >> 32)) << 32) | row_from_above_u32((u32)acc);
    acc = j == 0 ? up + (acc >> 29) : up;   // lane 0 held column 0 (now divisible by 2^29): its quotient joins column 1
}
template <class P> __device__ __forceinline__ u32 fe_mul_limblane(u32 a_j, u32 b_j, u32 j /* lane & 15 */) {
    u32 p_j = 0;
#pragma unroll
    for (int t = 0; t < 9; t++) p_j = j == (u32)t ? P::P29[t] : p_j;
    u64 acc = 0;
    limblane_step<P, 0>(acc, a_j, b_j, p_j, j); limblane_step<P, 1>(acc, a_j, b_j, p_j, j); limblane_step<P, 2>(acc, a_j, b_j, p_j, j);
    limblane_step<P, 3>(acc, a_j, b_j, p_j, j); limblane_step<P, 4>(acc, a_j, b_j, p_j, j); limblane_step<P, 5>(acc, a_j, b_j, p_j, j);
    limblane_step<P, 6>(acc, a_j, b_j, p_j, j); limblane_step<P, 7>(acc, a_j, b_j, p_j, j); limblane_step<P, 8>(acc, a_j, b_j, p_j, j);
    // two carry steps: (column & M29) + (lower neighbour's column >> 29); limb 8 keeps its own excess
#pragma unroll
    for (int s = 0; s < 2; s++) {
        const u64 c = acc >> 29;
        const u64 cin = ((u64)row_from_below_u32((u32)(c >> 32)) << 32) | row_from_below_u32((u32)c);
        acc = (j == 8 ? acc : (acc & M29)) + cin;
    }
    return (u32)acc;
}

#endif  // __HIPCC__ (the limb-per-lane product exists on the device only)

}  // namespace arkbp
