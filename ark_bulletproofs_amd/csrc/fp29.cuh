// 256-bit prime-field arithmetic for gfx950 (CDNA4) in a reduced radix: 9 limbs of 29 bits.
//
// Why this shape (measured on MI355X, tools/ubench.hip, profiles/r01_ubench.txt): v_mad_u64_u32
// issues at the same half rate as v_mul_lo_u32 / v_mad_u32_u24, so it is the widest multiplier per
// issue slot; what makes a 32-bit-limb Montgomery product slow is carry plumbing (≈4.5 VALU per limb
// product).  With 29-bit limbs a column of 18 products fits a 64-bit accumulator, so a product is ONE
// v_mad_u64_u32 and carries are extracted once per column; additions need no carry chain at all.
//
// Representation: x stands for the residue x * R'^-1 mod p with R' = 2^261 (Montgomery form w.r.t.
// R').  Buffers that cross the C ABI use ark-ff's layout (4 x u64, Montgomery w.r.t. R = 2^256 — what
// the reference's Fp256<MontBackend<_,4>> holds at e.g. src/inner_product_proof.rs:140-141);
// `fe_load_ark` / `fe_store_ark` convert at the boundary with one Montgomery product each.
// Device-resident tables keep the R' form, packed into 32 bytes (value < p).
//
// Pseudo-Mersenne fields (P::PM: both curves' SCALAR fields, 2^256 - 2^32 - 977 and 2^255 - 19): R' = 1, plain residues; a product
// is the 81-product schoolbook into eighteen 29-bit limbs, then the high nine limbs are folded onto the low nine with
// 2^261 mod p (a 38-bit constant: 9 + 9 small multiply-adds), the ~38-bit overflow is folded once more and the last few bits above
// the modulus width fold with 2^BITS mod p: ~100 v_mad_u64_u32 instead of ~160 for the Montgomery form, same contracts below (the
// result is < 2^BITS + 2^78: V < 1.000001).  The constants TO29 / FROM29 / R2_29 / ONE29 of such a field are generated for R' = 1
// (tools/gen_params.py; fe_pm_product below), so everything built on fe_mul / fe_sqr / fe_mul2 and the load / store helpers is representation-blind.
//
// Contracts (checked on the CPU by tests/test_fp29_host.py through csrc/fp29_selftest.cpp when
// ARKBP_CHECK_BOUNDS is defined).  L = largest limb / 2^29, V = value / p:
//   fe_mul / fe_sqr   in: L(a)*L(b) <= 6, V(a)*V(b) <= 900   out: L = 1, V < V(a)V(b)/32 + 1
//   fe_mul2           a*b + c*d, one reduction: L(a)L(b) + L(c)L(d) <= 6, V(a)V(b) + V(c)V(d) <= 900; out as fe_mul
//   fe_add            limb-wise, no carry                    out: L = La + Lb, V = Va + Vb
//   fe_sub<K>         in: L(a) <= 2, L(b) < 4, V(b) < K      out: L = 1, V = Va + K
//   fe_norm           carry pass                              out: L = 1
//   fe_wred           in: L = 1, V < 32                       out: L = 1, V <= 2
//   fe_canon          in: L = 1, V < 32                       out: canonical (0 <= x < p)
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ARKBP_HD __host__ __device__ __forceinline__
#define ARKBP_DEV_NOINLINE __device__ __noinline__
#else
#define ARKBP_HD inline
#define ARKBP_DEV_NOINLINE inline
#endif
#include "arkbp_params.h"

#ifdef ARKBP_CHECK_BOUNDS
#include <cstdio>
#include <cstdlib>
#define ARKBP_ASSERT(c, msg) do { if (!(c)) { fprintf(stderr, "fp29 bound violated: %s (%s:%d)\n", msg, __FILE__, __LINE__); abort(); } } while (0)
#else
#define ARKBP_ASSERT(c, msg)
#endif

namespace arkbp {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

static constexpr u32 M29 = (1u << 29) - 1;

struct Fe {
    u32 l[9];
};

template <class P> ARKBP_HD Fe fe_zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}
template <class P> ARKBP_HD Fe fe_one() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = P::ONE29[i];
    return r;
}
template <class P, const u32 (&C)[9]> ARKBP_HD Fe fe_const() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = C[i];
    return r;
}
// exact all-limbs-zero test (the identity's Z is stored as exact zero)
ARKBP_HD bool fe_is_zero_exact(const Fe& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.l[i];
    return o == 0;
}
ARKBP_HD bool fe_eq_exact(const Fe& a, const Fe& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// carry pass: limbs 0..7 < 2^29 afterwards; limb 8 keeps the excess
ARKBP_HD Fe fe_norm(const Fe& a) {
    Fe r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 t = a.l[i] + c;
        r.l[i] = t & M29;
        c = t >> 29;
    }
    r.l[8] = a.l[8] + c;
    return r;
}

ARKBP_HD Fe fe_add(const Fe& a, const Fe& b) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
ARKBP_HD Fe fe_dbl(const Fe& a) { return fe_add(a, a); }

// a - b + K*p with K in {2,4,8,16}; borrow-free: the K*p limbs are pre-spread to be >= 2^31 - 4
template <class P, int K> ARKBP_HD Fe fe_sub(const Fe& a, const Fe& b) {
    static_assert(K == 2 || K == 4 || K == 8 || K == 16, "K");
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u32 off = K == 2 ? P::SUBK2[i] : K == 4 ? P::SUBK4[i] : K == 8 ? P::SUBK8[i] : P::SUBK16[i];
        ARKBP_ASSERT(i == 8 || a.l[i] <= (2u << 29), "fe_sub: L(a) > 2");
        ARKBP_ASSERT(b.l[i] <= off, "fe_sub: limb of b exceeds the K*p offset");
        r.l[i] = a.l[i] + off - b.l[i];
    }
    return fe_norm(r);
}
template <class P, int K> ARKBP_HD Fe fe_neg(const Fe& a) { return fe_sub<P, K>(fe_zero<P>(), a); }

// Pseudo-Mersenne product: col(k, acc) adds the limb products of column k (weight 2^(29k), k = 0..16) to acc.  The HIGH columns
// 9..16 come first and are carried into nine 29-bit limbs H (the part of the product above 2^261, short of the carry out of column
// 8, which stays below); the low columns then take H * (2^261 mod p) along in the same accumulator chain — no second pass over the
// low limbs.  Result: value mod p, < 2^BITS + 2^78, limbs 0..7 < 2^29.
template <class P, class Col> ARKBP_HD Fe fe_pm_product(Col&& col) {
    static_assert(P::PM, "pseudo-Mersenne fields only");
    constexpr u32 c0 = P::PM_C0, c1 = P::PM_C1;       // 2^261 = c0 + c1 * 2^29 (mod p)
    u32 H[9];
    u64 acc = 0;
#pragma unroll
    for (int k = 9; k < 17; k++) {
        col(k, acc);
        H[k - 9] = (u32)acc & M29;
        acc >>= 29;
    }
    ARKBP_ASSERT(acc < (1ull << 32), "fe_pm_product: product exceeds 18 limbs");
    H[8] = (u32)acc;
    Fe r;
    acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        col(k, acc);
        acc += (u64)H[k] * c0;
        if (c1 != 0 && k > 0) acc += (u64)H[k - 1] * c1;
        r.l[k] = (u32)acc & M29;
        acc >>= 29;
    }
    if (c1 != 0) acc += (u64)H[8] * c1;
    const u64 o1 = acc;                                // what stands above 2^261 now: < 2^40
    ARKBP_ASSERT(o1 < (1ull << 41), "fe_pm_product: overflow too large");
    // last fold: everything above the modulus width — o1 at 2^261 and the top bits of limb 8 (bit BITS = bit BITS - 232 of that limb) —
    // comes back with 2^BITS mod p = d0 + d1 * 2^29, then ONE full carry pass: limbs 0..7 end strictly below 2^29 (callers multiply
    // limbs by small constants in 32 bits), limb 8 at most one above 2^(BITS - 232)
    constexpr int sh = P::BITS - 232;
    const u64 o = (o1 << (261 - P::BITS)) + (r.l[8] >> sh);
    r.l[8] &= (1u << sh) - 1u;
    acc = (u64)r.l[0] + o * P::PM_D0;
    r.l[0] = (u32)acc & M29; acc >>= 29;
    acc += (u64)r.l[1] + o * P::PM_D1;
    r.l[1] = (u32)acc & M29; acc >>= 29;
    u32 cy = (u32)acc;
#pragma unroll
    for (int j = 2; j < 8; j++) { const u32 t = r.l[j] + cy; r.l[j] = t & M29; cy = t >> 29; }
    r.l[8] += cy;
    return r;
}

// Montgomery product a*b/2^261 mod p, column-wise with one 64-bit accumulator.  (Pseudo-Mersenne fields: the plain product a*b mod p.)
template <class P> ARKBP_HD Fe fe_mul(const Fe& a, const Fe& b) {
#ifdef ARKBP_CHECK_BOUNDS
    { u64 la = 0, lb = 0; for (int i = 0; i < 9; i++) { if (a.l[i] > la) la = a.l[i]; if (b.l[i] > lb) lb = b.l[i]; }
      ARKBP_ASSERT((unsigned __int128)la * lb <= ((unsigned __int128)6 << 58) + ((unsigned __int128)1 << 40), "fe_mul: L(a)*L(b) > 6"); }
#endif
    if constexpr (P::PM) {
        return fe_pm_product<P>([&](int k, u64& acc) {
#pragma unroll
            for (int i = (k < 9 ? 0 : k - 8); i <= (k < 9 ? k : 8); i++) acc += (u64)a.l[i] * b.l[k - i];
        });
    }
    u32 m[9];
    Fe t;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * P::P29[k - i];
        m[k] = ((u32)acc * P::NINV29) & M29;
        acc += (u64)m[k] * P::P29[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (u64)m[i] * P::P29[k - i];
        t.l[k - 9] = (u32)acc & M29;
        acc >>= 29;
    }
    ARKBP_ASSERT(acc < (1ull << 29), "fe_mul: result exceeds 2^261");
    t.l[8] = (u32)acc;
    return t;
}

// a*b + c*d with ONE Montgomery reduction: (a*b + c*d) / 2^261 mod p.  243 limb products instead of 324 for two products and an
// addition.  in: L(a)*L(b) + L(c)*L(d) <= 6 (a column holds 9 + 9 operand products and 9 reduction products in 64 bits),
// V(a)V(b) + V(c)V(d) <= 900; out: L = 1, V < (V(a)V(b) + V(c)V(d)) / 32 + 1.
template <class P> ARKBP_HD Fe fe_mul2(const Fe& a, const Fe& b, const Fe& c, const Fe& d) {
#ifdef ARKBP_CHECK_BOUNDS
    { u64 la = 0, lb = 0, lc = 0, ld = 0;
      for (int i = 0; i < 9; i++) { if (a.l[i] > la) la = a.l[i]; if (b.l[i] > lb) lb = b.l[i]; if (c.l[i] > lc) lc = c.l[i]; if (d.l[i] > ld) ld = d.l[i]; }
      ARKBP_ASSERT((unsigned __int128)la * lb + (unsigned __int128)lc * ld <= ((unsigned __int128)6 << 58) + ((unsigned __int128)1 << 41), "fe_mul2: L(a)*L(b) + L(c)*L(d) > 6"); }
#endif
    if constexpr (P::PM) {
        return fe_pm_product<P>([&](int k, u64& acc) {
#pragma unroll
            for (int i = (k < 9 ? 0 : k - 8); i <= (k < 9 ? k : 8); i++) { acc += (u64)a.l[i] * b.l[k - i]; acc += (u64)c.l[i] * d.l[k - i]; }
        });
    }
    u32 m[9];
    Fe t;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (u64)a.l[i] * b.l[k - i]; acc += (u64)c.l[i] * d.l[k - i]; }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * P::P29[k - i];
        m[k] = ((u32)acc * P::NINV29) & M29;
        acc += (u64)m[k] * P::P29[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) { acc += (u64)a.l[i] * b.l[k - i]; acc += (u64)c.l[i] * d.l[k - i]; }
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (u64)m[i] * P::P29[k - i];
        t.l[k - 9] = (u32)acc & M29;
        acc >>= 29;
    }
    ARKBP_ASSERT(acc < (1ull << 29), "fe_mul2: result exceeds 2^261");
    t.l[8] = (u32)acc;
    return t;
}

// squaring: the 36 off-diagonal products are taken once against a doubled operand
template <class P> ARKBP_HD Fe fe_sqr(const Fe& a) {
    u32 m[9], a2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
#ifdef ARKBP_CHECK_BOUNDS
    for (int i = 0; i < 9; i++) ARKBP_ASSERT(a.l[i] <= (2u << 29) + 16, "fe_sqr: L(a) > 2");
#endif
    if constexpr (P::PM) {
        return fe_pm_product<P>([&](int k, u64& acc) {
#pragma unroll
            for (int i = (k < 9 ? 0 : k - 8); 2 * i < k; i++) acc += (u64)a2[i] * a.l[k - i];
            if ((k & 1) == 0) acc += (u64)a.l[k / 2] * a.l[k / 2];
        });
    }
    Fe t;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (u64)a2[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (u64)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * P::P29[k - i];
        m[k] = ((u32)acc * P::NINV29) & M29;
        acc += (u64)m[k] * P::P29[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) acc += (u64)a2[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (u64)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (u64)m[i] * P::P29[k - i];
        t.l[k - 9] = (u32)acc & M29;
        acc >>= 29;
    }
    t.l[8] = (u32)acc;
    return t;
}

// weak reduction: subtract floor-estimate(a / 2^WR_BITS) * p; needs L = 1 and a < 2^261.  Result <= 2p.
template <class P> ARKBP_HD Fe fe_wred(const Fe& a) {
    if constexpr (P::PM) {
        // pseudo-Mersenne: the bits above the modulus width come back multiplied by 2^BITS mod p (a 33-bit constant at most), one
        // carry pass: ~30 cheap instructions instead of nine signed 64-bit multiply-subtracts.  Result < 2^BITS + 2^40 <= 2p.
        constexpr int shp = P::BITS - 232;
        ARKBP_ASSERT(a.l[8] < (1u << 29) + 8u, "fe_wred: a >= 2^261");
        const u32 o = a.l[8] >> shp;                   // < 2^(29 - shp) + 1
        Fe r;
        u32 t = a.l[0] + o * P::PM_D0;                 // < 2^29 + 2^7 * 2^10
        r.l[0] = t & M29;
        t = a.l[1] + o * P::PM_D1 + (t >> 29);
        r.l[1] = t & M29;
        u32 cy = t >> 29;
#pragma unroll
        for (int j = 2; j < 8; j++) { t = a.l[j] + cy; r.l[j] = t & M29; cy = t >> 29; }
        r.l[8] = (a.l[8] & ((1u << shp) - 1u)) + cy;
        return r;
    }
    constexpr int sh = P::WR_BITS - 232;
    u32 q = a.l[8] >> sh;
    if (P::WR_SIGN > 0) q = q ? q - 1 : 0;  // p = 2^B + delta: q*p could exceed a, q-1 cannot
    Fe r;
    i64 acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc += (i64)a.l[i] - (i64)((u64)q * P::P29[i]);
        r.l[i] = (u32)acc & M29;
        acc >>= 29;  // arithmetic
    }
    ARKBP_ASSERT(acc == 0, "fe_wred: went negative or overflowed");
    return r;
}

// full reduction to the canonical representative; input L = 1, V < 32
template <class P> ARKBP_HD Fe fe_canon(const Fe& a) {
    Fe w = fe_wred<P>(a);  // < 2p (+ tiny), so at most two subtractions of p
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        Fe d;
        i32 c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            i32 t = (i32)w.l[i] - (i32)P::P29[i] + c;
            d.l[i] = (u32)t & M29;
            c = t >> 29;
        }
        const bool neg = c < 0;  // w < p
#pragma unroll
        for (int i = 0; i < 9; i++) w.l[i] = neg ? w.l[i] : d.l[i];
    }
    return w;
}

// a == 0 (mod p)?   Needs L = 1, V < 32.  Cheap filter on the low limb first: a = k*p forces
// a_0 = k*p_0 mod 2^29 with k within one of the top-limb estimate; the exact test runs only then.
template <class P> ARKBP_HD bool fe_maybe_zero_mod(const Fe& a) {   // the filter alone: false means a != 0 (mod p) for certain
    constexpr int sh = P::WR_BITS - 232;
    const u32 q = a.l[8] >> sh;
    bool maybe = false;
#pragma unroll
    for (int d = -1; d <= 1; d++) {
        const u32 k = q + (u32)d;
        maybe |= (a.l[0] == ((k * P::P29[0]) & M29));
    }
    return maybe;
}
template <class P> ARKBP_HD bool fe_is_zero_mod(const Fe& a) {
    if (!fe_maybe_zero_mod<P>(a)) return false;
    return fe_is_zero_exact(fe_canon<P>(a));
}
template <class P> ARKBP_HD bool fe_eq_mod(const Fe& a, const Fe& b) {  // L = 1 both, V(b) < 16
    return fe_is_zero_mod<P>(fe_sub<P, 16>(a, b));
}

// ---- packing: 9 x 29-bit limbs <-> 8 x 32-bit words (value must be < 2^256) ----------------------
ARKBP_HD void fe_pack(u32 w[8], const Fe& a) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i0 = (32 * j) / 29, o = 32 * j - 29 * i0;
        u64 x = (u64)a.l[i0] | ((u64)a.l[i0 + 1] << 29);
        if (i0 + 2 < 9 && 58 - o < 32) x |= (u64)a.l[i0 + 2] << 58;
        w[j] = (u32)(x >> o);
    }
}
ARKBP_HD Fe fe_unpack(const u32 w[8]) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int j0 = (29 * i) / 32, o = 29 * i - 32 * j0;
        u64 x = (u64)w[j0];
        if (j0 + 1 < 8) x |= (u64)w[j0 + 1] << 32;
        r.l[i] = (u32)(x >> o) & M29;
    }
    return r;
}

// ark-ff Montgomery words (x*2^256 mod p) -> R' form, and back (canonical output words)
template <class P> ARKBP_HD Fe fe_load_ark(const u32 w[8]) { return fe_mul<P>(fe_unpack(w), fe_const<P, P::TO29>()); }
template <class P> ARKBP_HD void fe_store_ark(u32 w[8], const Fe& a) { fe_pack(w, fe_canon<P>(fe_mul<P>(a, fe_const<P, P::FROM29>()))); }
// canonical integer words <-> R' form
template <class P> ARKBP_HD Fe fe_load_canon(const u32 w[8]) {
    if constexpr (P::PM) return fe_unpack(w);          // plain residues: the canonical integer IS the representation (any 256-bit value: V < 2.01)
    else return fe_mul<P>(fe_unpack(w), fe_const<P, P::R2_29>());
}
template <class P> ARKBP_HD void fe_store_canon(u32 w[8], const Fe& a) {
    if constexpr (P::PM) fe_pack(w, fe_canon<P>(a));
    else fe_pack(w, fe_canon<P>(fe_mul<P>(a, fe_const<P, P::CANON29>())));
}
// device-resident packed R' form (value < p)
template <class P> ARKBP_HD Fe fe_load_dev(const u32 w[8]) { return fe_unpack(w); }
template <class P> ARKBP_HD void fe_store_dev(u32 w[8], const Fe& a) { fe_pack(w, fe_canon<P>(a)); }

// a^(p-2) (0 -> 0); input/outputs L = 1, V <= 2.  Off the per-element path: kernels batch inversions.
template <class P> ARKBP_HD Fe fe_inv(const Fe& a) {
    // a^(p-2), two exponent bits per step (a, a^2, a^3 at hand): p - 2 is mostly ones for these moduli, and a digit 3 costs ONE
    // product for two bits: 256 squarings + ~125 products instead of ~250
    const Fe a2 = fe_sqr<P>(a), a3 = fe_mul<P>(a2, a);
    Fe r = fe_one<P>();
#pragma unroll 1
    for (int w = 7; w >= 0; w--) {
        const u32 e = P::PM2[w];
#pragma unroll 1
        for (int i = 30; i >= 0; i -= 2) {
            r = fe_sqr<P>(fe_sqr<P>(r));
            const u32 d = (e >> i) & 3u;
            if (d) {
                Fe m;
#pragma unroll
                for (int j = 0; j < 9; j++) m.l[j] = d == 1u ? a.l[j] : d == 2u ? a2.l[j] : a3.l[j];
                r = fe_mul<P>(r, m);
            }
        }
    }
    return r;
}

}  // namespace arkbp
