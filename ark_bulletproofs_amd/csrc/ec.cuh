// Short-Weierstrass group arithmetic for gfx950 on top of fp29.cuh (Jacobian X:Y:Z, a = 0 for
// secq256k1, a = 6 for zorro — /root/reference/src/curve/zorro/g1.rs:24-38).  These are the group
// operations behind every `G::Group::msm(..)` / `into_affine()` call site of the reference
// (src/inner_product_proof.rs:104,124,143-155,187,202,219-224; src/r1cs/prover.rs:516-649;
// src/r1cs/verifier.rs:574,685).  Results are group elements: after `into_affine` they are
// bit-identical to ark-ec's whatever formulas either side uses.
//
// Invariants: coordinates of a stored Jac are L = 1, V <= 2 (see fp29.cuh); the identity has Z == 0
// exactly.  Aff coordinates are canonical (V < 1); the identity is x = y = 0 (not on either curve).
// All exceptional cases are handled (P+P, P+(-P), identity operands): batch verification legitimately
// feeds identity points and duplicate bases (src/r1cs/verifier.rs:672-674).
#pragma once
#include "fp29.cuh"

namespace arkbp {

struct Aff {
    Fe x, y;
};
struct Jac {
    Fe X, Y, Z;
};

ARKBP_HD bool aff_is_inf(const Aff& p) { return fe_is_zero_exact(p.x) && fe_is_zero_exact(p.y); }
ARKBP_HD bool jac_is_inf(const Jac& p) { return fe_is_zero_exact(p.Z); }
template <class C> ARKBP_HD Jac jac_inf() {
    Jac r;
    r.X = fe_one<typename C::Fq>();
    r.Y = fe_one<typename C::Fq>();
    r.Z = fe_zero<typename C::Fq>();
    return r;
}
template <class C> ARKBP_HD Jac jac_from_aff(const Aff& p) {
    if (aff_is_inf(p)) return jac_inf<C>();
    Jac r;
    r.X = p.x;
    r.Y = p.y;
    r.Z = fe_one<typename C::Fq>();
    return r;
}
template <class C> ARKBP_HD Aff aff_neg(const Aff& p) {
    typedef typename C::Fq F;
    Aff r;
    r.x = p.x;
    r.y = aff_is_inf(p) ? p.y : fe_canon<F>(fe_neg<F, 2>(p.y));
    return r;
}
// conditional negation without the canonicalisation (fine as an operand of jac_madd: V(y) <= 3)
template <class C> ARKBP_HD Aff aff_cneg_lazy(const Aff& p, bool neg) {
    typedef typename C::Fq F;
    Aff r;
    r.x = p.x;
    Fe ny = fe_neg<F, 2>(p.y);
    const bool flip = neg && !aff_is_inf(p);
#pragma unroll
    for (int i = 0; i < 9; i++) r.y.l[i] = flip ? ny.l[i] : p.y.l[i];
    return r;
}

// k*a for a small k, carried back to L = 1
template <int K> ARKBP_HD Fe fe_times(const Fe& a) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] * (u32)K;  // K <= 8: no 32-bit overflow for L = 1
    return fe_norm(r);
}

// (no identity branch: Z3 = 2*Y*Z is exactly zero when Z is, so the identity doubles to an identity with whatever X, Y — only
// Z == 0 marks it)
template <class C> ARKBP_HD Jac jac_dbl(const Jac& p) {
    typedef typename C::Fq F;
    Jac o;
    if (C::A_ZERO) {
        // dbl-2009-l shape: A = X^2, B = Y^2, C = B^2, D = 4XB, E = 3A, X3 = E^2 - 2D, Y3 = E(D - X3) - 8C, Z3 = 2YZ
        Fe A = fe_sqr<F>(p.X), B = fe_sqr<F>(p.Y), Cc = fe_sqr<F>(B);
        Fe D = fe_times<4>(fe_mul<F>(p.X, B));             // V <= 4.3
        Fe E = fe_times<3>(A);                             // V <= 3.4
        Fe X3 = fe_wred<F>(fe_sub<F, 16>(fe_sqr<F>(E), fe_dbl(D)));
        Fe Y3 = fe_mul<F>(E, fe_sub<F, 4>(D, X3));         // V(D - X3) <= 8.3
        o.X = X3;
        o.Y = fe_wred<F>(fe_sub<F, 16>(Y3, fe_times<8>(Cc)));
        o.Z = fe_mul<F>(fe_dbl(p.Y), p.Z);
    } else {
        // dbl-2007-bl shape with M = 3XX + a*ZZ^2, S = 4*X*YY
        Fe XX = fe_sqr<F>(p.X), YY = fe_sqr<F>(p.Y), YYYY = fe_sqr<F>(YY), ZZ = fe_sqr<F>(p.Z);
        Fe S = fe_times<4>(fe_mul<F>(p.X, YY));
        Fe Z4 = fe_sqr<F>(ZZ);
        static_assert(C::A_ZERO || (C::A_SMALL >= 1 && C::A_SMALL <= 8), "curve coefficient a must be a small integer");
        Fe aZ4 = fe_times<(C::A_ZERO ? 1 : C::A_SMALL)>(Z4);
        Fe M = fe_norm(fe_add(fe_times<3>(XX), aZ4));      // V <= 3.4 + 6.3
        Fe X3 = fe_wred<F>(fe_sub<F, 16>(fe_sqr<F>(M), fe_dbl(S)));
        Fe Y3 = fe_mul<F>(M, fe_sub<F, 4>(S, X3));
        o.X = X3;
        o.Y = fe_wred<F>(fe_sub<F, 16>(Y3, fe_times<8>(YYYY)));
        o.Z = fe_mul<F>(fe_dbl(p.Y), p.Z);
    }
    return o;
}

// Jacobian + affine.  q.y may be a lazy negation (V <= 3, L = 1).
template <class C> ARKBP_HD Jac jac_madd(const Jac& p, const Aff& q) {
    typedef typename C::Fq F;
    if (aff_is_inf(q)) return p;
    if (jac_is_inf(p)) {
        Jac r;
        r.X = q.x;
        r.Y = fe_wred<F>(q.y);
        r.Z = fe_one<F>();
        return r;
    }
    Fe Z1Z1 = fe_sqr<F>(p.Z);
    Fe U2 = fe_mul<F>(q.x, Z1Z1);
    Fe S2 = fe_mul<F>(q.y, fe_mul<F>(p.Z, Z1Z1));
    Fe H = fe_sub<F, 4>(U2, p.X);   // V <= 5.1
    Fe r = fe_sub<F, 4>(S2, p.Y);
    if (fe_is_zero_mod<F>(H)) {     // same x: doubling or cancellation (rare, duplicates only)
        if (fe_is_zero_mod<F>(r)) return jac_dbl<C>(p);
        return jac_inf<C>();
    }
    Fe HH = fe_sqr<F>(H);           // <= 1.8
    Fe HHH = fe_mul<F>(H, HH);      // <= 1.3
    Fe V = fe_mul<F>(p.X, HH);      // <= 1.2
    Jac o;
    o.X = fe_wred<F>(fe_sub<F, 4>(fe_sqr<F>(r), fe_add(HHH, fe_dbl(V))));
    // Y3 = r (V - X3) - Y1 H^3 as ONE fused product pair: a single Montgomery reduction for both products (243 limb products instead
    // of 324) and no weak reduction after (V < (5.1 * 5.2 + 4 * 1.3) / 32 + 1 < 2)
    o.Y = fe_mul2<F>(r, fe_sub<F, 4>(V, o.X), fe_neg<F, 4>(p.Y), HHH);
    o.Z = fe_mul<F>(p.Z, H);
    return o;
}

// The hot form of the mixed addition, for loops that add many points: the incomplete formulas as ONE basic block, and `rare` set
// when an operand is the identity or the points may be equal / opposite (the low-limb filter of fe_is_zero_mod on H: no false
// negatives) — the result is then meaningless and the caller redoes the addition with jac_madd.  With the exceptional returns
// inlined in the middle of jac_madd the register allocator spends ~400 moves per addition on them (hipcc -S of an accumulate-shaped
// loop: 3 340 VALU instructions per iteration against 2 680 for this form + a cold block); callers keep p alive (nine more
// registers) and read q again in the cold block.
template <class C> ARKBP_HD Jac jac_madd_fast(const Jac& p, const Aff& q, bool& rare) {
    typedef typename C::Fq F;
    Fe Z1Z1 = fe_sqr<F>(p.Z);
    Fe U2 = fe_mul<F>(q.x, Z1Z1);
    Fe S2 = fe_mul<F>(q.y, fe_mul<F>(p.Z, Z1Z1));
    Fe H = fe_sub<F, 4>(U2, p.X);   // V <= 5.1
    Fe r = fe_sub<F, 4>(S2, p.Y);
    rare = aff_is_inf(q) | jac_is_inf(p) | fe_maybe_zero_mod<F>(H);
    Fe HH = fe_sqr<F>(H);           // <= 1.8
    Fe HHH = fe_mul<F>(H, HH);      // <= 1.3
    Fe V = fe_mul<F>(p.X, HH);      // <= 1.2
    Jac o;
    o.X = fe_wred<F>(fe_sub<F, 4>(fe_sqr<F>(r), fe_add(HHH, fe_dbl(V))));
    // Y3 = r (V - X3) - Y1 H^3 as ONE fused product pair: a single Montgomery reduction for both products (243 limb products instead
    // of 324) and no weak reduction after (V < (5.1 * 5.2 + 4 * 1.3) / 32 + 1 < 2)
    o.Y = fe_mul2<F>(r, fe_sub<F, 4>(V, o.X), fe_neg<F, 4>(p.Y), HHH);
    o.Z = fe_mul<F>(p.Z, H);
    return o;
}

// Jacobian + Jacobian
template <class C> ARKBP_HD Jac jac_add(const Jac& p, const Jac& q) {
    typedef typename C::Fq F;
    if (jac_is_inf(q)) return p;
    if (jac_is_inf(p)) return q;
    Fe Z1Z1 = fe_sqr<F>(p.Z), Z2Z2 = fe_sqr<F>(q.Z);
    Fe U1 = fe_mul<F>(p.X, Z2Z2), U2 = fe_mul<F>(q.X, Z1Z1);
    Fe S1 = fe_mul<F>(p.Y, fe_mul<F>(q.Z, Z2Z2)), S2 = fe_mul<F>(q.Y, fe_mul<F>(p.Z, Z1Z1));
    Fe H = fe_sub<F, 2>(U2, U1);    // V <= 3.2
    Fe r = fe_sub<F, 2>(S2, S1);
    if (fe_is_zero_mod<F>(H)) {
        if (fe_is_zero_mod<F>(r)) return jac_dbl<C>(p);
        return jac_inf<C>();
    }
    Fe HH = fe_sqr<F>(H);
    Fe HHH = fe_mul<F>(H, HH);
    Fe V = fe_mul<F>(U1, HH);
    Jac o;
    o.X = fe_wred<F>(fe_sub<F, 4>(fe_sqr<F>(r), fe_add(HHH, fe_dbl(V))));
    o.Y = fe_mul2<F>(r, fe_sub<F, 4>(V, o.X), fe_neg<F, 2>(S1), HHH);   // (one reduction for both products, see jac_madd)
    o.Z = fe_mul<F>(fe_mul<F>(p.Z, q.Z), H);
    return o;
}

// (X/Z^2, Y/Z^3) given zinv = 1/Z; canonical output
template <class C> ARKBP_HD Aff jac_to_aff_with_zinv(const Jac& p, const Fe& zinv) {
    typedef typename C::Fq F;
    Aff r;
    if (jac_is_inf(p)) {
        r.x = fe_zero<F>();
        r.y = fe_zero<F>();
        return r;
    }
    Fe zi2 = fe_sqr<F>(zinv);
    r.x = fe_canon<F>(fe_mul<F>(p.X, zi2));
    r.y = fe_canon<F>(fe_mul<F>(p.Y, fe_mul<F>(zi2, zinv)));
    return r;
}
template <class C> ARKBP_HD Aff jac_to_aff(const Jac& p) {
    return jac_to_aff_with_zinv<C>(p, fe_inv<typename C::Fq>(p.Z));
}

// Tonelli-Shanks square root in Fq (p - 1 = 2^S * t; ark-ff's SqrtPrecomputation::TonelliShanks shape): false for a
// non-residue.  Either root may come out; callers order (y, -y) canonically like ark-ec's get_ys_from_x_unchecked.
template <class F> ARKBP_DEV_NOINLINE bool fe_sqrt(Fe& out, const Fe& a_in) {
    const Fe a = fe_wred<F>(a_in);
    if (fe_is_zero_mod<F>(a)) { out = fe_zero<F>(); return true; }
    // w = a^((t-1)/2), two exponent bits per step (a, a^2, a^3 at hand): 256 squarings + ~110 products instead of ~190 — the
    // exponents of these fields are long runs of ones (digit 3 = ONE product for two bits).  The exponent is a compile-time constant,
    // so the digit tests are scalar branches.
    const Fe a2 = fe_sqr<F>(a), a3 = fe_mul<F>(a2, a);
    Fe w = fe_one<F>();
#pragma unroll 1
    for (int wd = 7; wd >= 0; wd--) {
        const u32 e = F::TS_TM1H[wd];
#pragma unroll 1
        for (int bit = 30; bit >= 0; bit -= 2) {
            w = fe_sqr<F>(fe_sqr<F>(w));
            const u32 d = (e >> bit) & 3u;
            if (d) {
                Fe m;
#pragma unroll
                for (int i = 0; i < 9; i++) m.l[i] = d == 1u ? a.l[i] : d == 2u ? a2.l[i] : a3.l[i];
                w = fe_mul<F>(w, m);
            }
        }
    }
    Fe x = fe_mul<F>(w, a);        // a^((t+1)/2)
    Fe b = fe_mul<F>(x, w);        // a^t
    Fe z = fe_const<F, F::TS_Z29>();
    const Fe one = fe_one<F>();
    int v = F::TS_S;
#pragma unroll 1
    while (!fe_eq_mod<F>(b, one)) {
        int k = 0;
        Fe t = b;
#pragma unroll 1
        while (!fe_eq_mod<F>(t, one)) {
            t = fe_sqr<F>(t);
            if (++k == v) return false;   // a is a non-residue
        }
        Fe wz = z;
        for (int j = 0; j < v - k - 1; j++) wz = fe_sqr<F>(wz);
        z = fe_sqr<F>(wz);
        b = fe_mul<F>(b, z);
        x = fe_mul<F>(x, wz);
        v = k;
    }
    if (!fe_eq_mod<F>(fe_sqr<F>(x), a)) return false;
    out = x;
    return true;
}

// canonical-integer comparison a > b for canonical a, b (ark `Ord for Fp`)
ARKBP_HD bool fe_canon_gt(const Fe& a, const Fe& b) {
    bool gt = false, decided = false;
#pragma unroll
    for (int i = 8; i >= 0; i--) {
        if (!decided && a.l[i] != b.l[i]) { gt = a.l[i] > b.l[i]; decided = true; }
    }
    return gt;
}

// SW point from x and the sign flag of ark-serialize's compressed encoding (greatest = flag 0x80: y > -y)
template <class C> ARKBP_HD bool aff_from_x(Aff& out, const Fe& x_canon_rform, bool greatest) {
    typedef typename C::Fq F;
    Fe rhs = fe_mul<F>(fe_sqr<F>(x_canon_rform), x_canon_rform);
    if (!C::A_ZERO) rhs = fe_norm(fe_add(rhs, fe_times<(C::A_ZERO ? 1 : C::A_SMALL)>(x_canon_rform)));
    rhs = fe_norm(fe_add(rhs, fe_load_ark<F>(C::B)));
    Fe y;
    if (!fe_sqrt<F>(y, rhs)) return false;
    // order by the canonical INTEGER value (the residue itself, not its Montgomery representative)
    const Fe yc = fe_canon<F>(fe_mul<F>(y, fe_const<F, F::CANON29>()));
    const Fe ny = fe_canon<F>(fe_neg<F, 4>(fe_wred<F>(y)));
    const Fe nyc = fe_canon<F>(fe_mul<F>(ny, fe_const<F, F::CANON29>()));
    const bool y_is_larger = fe_canon_gt(yc, nyc);
    out.x = x_canon_rform;
    out.y = (greatest == y_is_larger) ? fe_canon<F>(y) : ny;
    return true;
}

// ---- 64-byte affine points in memory ------------------------------------------------------------
// ark layout at the C ABI: x || y as 8+8 u32 words, Montgomery w.r.t. 2^256; identity = all zero.
template <class C> ARKBP_HD Aff aff_load_ark(const u32* w) {
    typedef typename C::Fq F;
    Aff r;
    r.x = fe_canon<F>(fe_load_ark<F>(w));
    r.y = fe_canon<F>(fe_load_ark<F>(w + 8));
    return r;  // all-zero words -> exact zero limbs -> identity
}
template <class C> ARKBP_HD void aff_store_ark(u32* w, const Aff& p) {
    typedef typename C::Fq F;
    fe_store_ark<F>(w, p.x);
    fe_store_ark<F>(w + 8, p.y);
}
// device-resident layout: packed R' form
ARKBP_HD Aff aff_load_dev(const u32* w) {
    Aff r;
    r.x = fe_unpack(w);
    r.y = fe_unpack(w + 8);
    return r;
}
ARKBP_HD void aff_store_dev(u32* w, const Aff& p) {  // p canonical
    fe_pack(w, p.x);
    fe_pack(w + 8, p.y);
}

}  // namespace arkbp
