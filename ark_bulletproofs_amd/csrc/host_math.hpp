// Host-side (CPU) field and curve arithmetic of the PRODUCT: 4 x 64-bit limbs, ark-ff's own memory
// layout (Montgomery, R = 2^256), constants from arkbp_params.h.  It serves the O(1)-sized serial work
// that stays on the host by design — Fiat-Shamir scalars (src/transcript.rs:95-101), challenge
// inverses (src/inner_product_proof.rs:137,214), the window-combine tail of an MSM (256 doublings of
// ONE point), Pedersen commitments of single scalars (src/generators.rs:39-44), point serialisation —
// never per-element vector work, which runs in the HIP kernels.  Independent of oracle/.
#pragma once
#include <cstdint>
#include <cstring>
#include "arkbp_params.h"

namespace arkbp {
namespace host {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef unsigned __int128 u128;

struct F4 {
    u64 v[4];
    bool operator==(const F4& o) const { return ((v[0] ^ o.v[0]) | (v[1] ^ o.v[1]) | (v[2] ^ o.v[2]) | (v[3] ^ o.v[3])) == 0; }
    bool operator!=(const F4& o) const { return !(*this == o); }
    bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
};

template <class P> struct Fld {
    static inline bool geq_p(const u64* a) {
        for (int i = 3; i >= 0; i--) {
            if (a[i] > P::P64[i]) return true;
            if (a[i] < P::P64[i]) return false;
        }
        return true;
    }
    static inline void sub_p(u64* a) {
        u64 br = 0;
        for (int i = 0; i < 4; i++) {
            u128 t = (u128)a[i] - P::P64[i] - br;
            a[i] = (u64)t;
            br = (u64)(t >> 64) & 1;
        }
    }
    static inline F4 zero() { return F4{{0, 0, 0, 0}}; }
    static inline F4 one() { return F4{{P::R1_64[0], P::R1_64[1], P::R1_64[2], P::R1_64[3]}}; }
    static inline F4 add(const F4& a, const F4& b) {
        F4 r;
        u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (u64)c; c >>= 64; }
        if (c || geq_p(r.v)) sub_p(r.v);
        return r;
    }
    static inline F4 sub(const F4& a, const F4& b) {
        F4 r;
        u64 br = 0;
        for (int i = 0; i < 4; i++) { u128 t = (u128)a.v[i] - b.v[i] - br; r.v[i] = (u64)t; br = (u64)(t >> 64) & 1; }
        if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.v[i] + P::P64[i]; r.v[i] = (u64)c; c >>= 64; } }
        return r;
    }
    static inline F4 neg(const F4& a) { return a.is_zero() ? a : sub(zero(), a); }
    static inline F4 dbl(const F4& a) { return add(a, a); }
    // coarsely integrated operand scanning (CIOS), fully unrolled: row i of the product and its Montgomery step share one pass
    // over a 5-word accumulator (a * b + t + carry never exceeds 128 bits)
    static inline F4 mul(const F4& a, const F4& b) {
        u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#pragma GCC unroll 4
        for (int i = 0; i < 4; i++) {
            const u64 ai = a.v[i];
            u128 c = (u128)ai * b.v[0] + t0; t0 = (u64)c; c >>= 64;
            c += (u128)ai * b.v[1] + t1; t1 = (u64)c; c >>= 64;
            c += (u128)ai * b.v[2] + t2; t2 = (u64)c; c >>= 64;
            c += (u128)ai * b.v[3] + t3; t3 = (u64)c; c >>= 64;
            c += t4; t4 = (u64)c;
            const u64 t5 = (u64)(c >> 64);
            const u64 m = t0 * P::NINV64;
            c = (u128)m * P::P64[0] + t0; c >>= 64;
            c += (u128)m * P::P64[1] + t1; t0 = (u64)c; c >>= 64;
            c += (u128)m * P::P64[2] + t2; t1 = (u64)c; c >>= 64;
            c += (u128)m * P::P64[3] + t3; t2 = (u64)c; c >>= 64;
            c += t4; t3 = (u64)c; t4 = t5 + (u64)(c >> 64);
        }
        F4 r = {{t0, t1, t2, t3}};
        if (t4 || geq_p(r.v)) sub_p(r.v);
        return r;
    }
    static inline F4 sqr(const F4& a) { return mul(a, a); }
    static inline F4 from_u64(u64 x) { F4 a = {{x, 0, 0, 0}}, r2 = {{P::R2_64[0], P::R2_64[1], P::R2_64[2], P::R2_64[3]}}; return mul(a, r2); }
    static inline F4 from_canon(const u64 c[4]) { F4 a; memcpy(a.v, c, 32); F4 r2 = {{P::R2_64[0], P::R2_64[1], P::R2_64[2], P::R2_64[3]}}; return mul(a, r2); }
    // Montgomery form -> canonical integer: the reduction alone (a * 1 has no product to form): 20 multiplications instead of 36.
    // Serialising a proof's 256 commitments for the transcript is 512 of these (a fifth of the verifier's host time per proof).
    static inline void to_canon(u64 c[4], const F4& a) {
        u64 t0 = a.v[0], t1 = a.v[1], t2 = a.v[2], t3 = a.v[3];
#pragma GCC unroll 4
        for (int i = 0; i < 4; i++) {
            const u64 m = t0 * P::NINV64;
            u128 cc = (u128)m * P::P64[0] + t0; cc >>= 64;
            cc += (u128)m * P::P64[1] + t1; t0 = (u64)cc; cc >>= 64;
            cc += (u128)m * P::P64[2] + t2; t1 = (u64)cc; cc >>= 64;
            cc += (u128)m * P::P64[3] + t3; t2 = (u64)cc; cc >>= 64;
            t3 = (u64)cc;
        }
        u64 r[4] = {t0, t1, t2, t3};
        if (geq_p(r)) sub_p(r);   // (not reached for a < p: (a + M p) / 2^256 < p)
        memcpy(c, r, 32);
    }
    static inline F4 pow(const F4& a, const u64 e[4]) {   // fixed 4-bit windows: 256 squarings + <= 64 products + 14 for the table
        F4 tab[16];
        tab[0] = one(); tab[1] = a;
        for (int i = 2; i < 16; i++) tab[i] = mul(tab[i - 1], a);
        F4 r = one();
        bool started = false;
        for (int i = 63; i >= 0; i--) {
            const unsigned d = (unsigned)(e[i >> 4] >> ((i & 15) * 4)) & 15u;
            if (started) { r = sqr(r); r = sqr(r); r = sqr(r); r = sqr(r); }
            if (d) { r = started ? mul(r, tab[d]) : tab[d]; started = true; }
        }
        return r;
    }
    static inline F4 inv(const F4& a) { return pow(a, P::PM2_64); }  // 0 -> 0
    static inline int cmp_canon(const F4& a, const F4& b) {
        u64 x[4], y[4]; to_canon(x, a); to_canon(y, b);
        for (int i = 3; i >= 0; i--) { if (x[i] < y[i]) return -1; if (x[i] > y[i]) return 1; }
        return 0;
    }
    static inline bool sqrt(F4& out, const F4& a) {  // Tonelli-Shanks; either root
        if (a.is_zero()) { out = a; return true; }
        F4 z = {{P::TS_Z_64[0], P::TS_Z_64[1], P::TS_Z_64[2], P::TS_Z_64[3]}};
        F4 w = pow(a, P::TS_TM1H_64), x = mul(w, a), b = mul(x, w), o = one();
        int v = P::TS_S;
        while (b != o) {
            int k = 0;
            F4 t = b;
            while (t != o) { t = sqr(t); if (++k == v) return false; }
            F4 wz = z;
            for (int j = 0; j < v - k - 1; j++) wz = sqr(wz);
            z = sqr(wz); b = mul(b, z); x = mul(x, wz); v = k;
        }
        if (sqr(x) != a) return false;
        out = x;
        return true;
    }
    static inline void to_bytes(u8 out[32], const F4& a) { u64 c[4]; to_canon(c, a); memcpy(out, c, 32); }
    static inline bool from_bytes(F4& o, const u8 in[32]) { u64 c[4]; memcpy(c, in, 32); if (geq_p(c)) return false; o = from_canon(c); return true; }
};

struct A4 {  // affine, identity = (0,0)
    F4 x, y;
    bool is_inf() const { return x.is_zero() && y.is_zero(); }
    bool operator==(const A4& o) const { return x == o.x && y == o.y; }
};
struct J4 {
    F4 X, Y, Z;
};

template <class C> struct Grp {
    typedef Fld<typename C::Fq> F;
    typedef Fld<typename C::Fr> S;
    static inline A4 aff_inf() { return A4{F::zero(), F::zero()}; }
    static inline J4 inf() { return J4{F::one(), F::one(), F::zero()}; }
    static inline bool is_inf(const J4& p) { return p.Z.is_zero(); }
    static inline J4 from_aff(const A4& p) { return p.is_inf() ? inf() : J4{p.x, p.y, F::one()}; }
    static inline A4 generator() {
        A4 g; memcpy(g.x.v, C::GX64, 32); memcpy(g.y.v, C::GY64, 32); return g;
    }
    static inline A4 neg(const A4& p) { return A4{p.x, F::neg(p.y)}; }
    static inline J4 dbl(const J4& p) {
        if (is_inf(p)) return p;
        F4 XX = F::sqr(p.X), YY = F::sqr(p.Y), YYYY = F::sqr(YY);
        F4 S = F::mul(p.X, YY); S = F::dbl(F::dbl(S));
        F4 M = F::add(F::dbl(XX), XX);
        if (!C::A_ZERO) { F4 a; memcpy(a.v, C::A64, 32); F4 zz = F::sqr(p.Z); M = F::add(M, F::mul(a, F::sqr(zz))); }
        J4 o;
        o.X = F::sub(F::sqr(M), F::dbl(S));
        F4 y8 = F::dbl(F::dbl(F::dbl(YYYY)));
        o.Y = F::sub(F::mul(M, F::sub(S, o.X)), y8);
        o.Z = F::dbl(F::mul(p.Y, p.Z));
        return o;
    }
    static inline J4 add(const J4& p, const J4& q) {
        if (is_inf(p)) return q;
        if (is_inf(q)) return p;
        F4 z1 = F::sqr(p.Z), z2 = F::sqr(q.Z);
        F4 u1 = F::mul(p.X, z2), u2 = F::mul(q.X, z1);
        F4 s1 = F::mul(p.Y, F::mul(q.Z, z2)), s2 = F::mul(q.Y, F::mul(p.Z, z1));
        if (u1 == u2) return s1 == s2 ? dbl(p) : inf();
        F4 h = F::sub(u2, u1), r = F::sub(s2, s1), hh = F::sqr(h), hhh = F::mul(h, hh), v = F::mul(u1, hh);
        J4 o;
        o.X = F::sub(F::sub(F::sqr(r), hhh), F::dbl(v));
        o.Y = F::sub(F::mul(r, F::sub(v, o.X)), F::mul(s1, hhh));
        o.Z = F::mul(F::mul(p.Z, q.Z), h);
        return o;
    }
    static inline J4 madd(const J4& p, const A4& q) { return add(p, from_aff(q)); }
    static inline A4 to_aff(const J4& p) {
        if (is_inf(p)) return aff_inf();
        F4 zi = F::inv(p.Z), zi2 = F::sqr(zi);
        return A4{F::mul(p.X, zi2), F::mul(p.Y, F::mul(zi2, zi))};
    }
    // scalar given as a canonical 256-bit integer
    static inline J4 mul_canon(const A4& p, const u64 k[4]) {
        J4 r = inf(), pj = from_aff(p);
        for (int i = 255; i >= 0; i--) { r = dbl(r); if ((k[i >> 6] >> (i & 63)) & 1) r = add(r, pj); }
        return r;
    }
    static inline J4 mul(const A4& p, const F4& s) { u64 k[4]; S::to_canon(k, s); return mul_canon(p, k); }
    static inline bool on_curve(const A4& p) {
        if (p.is_inf()) return true;
        F4 a, b; memcpy(a.v, C::A64, 32); memcpy(b.v, C::B64, 32);
        return F::sqr(p.y) == F::add(F::add(F::mul(F::sqr(p.x), p.x), F::mul(a, p.x)), b);
    }
    // ark-serialize 0.4 SW point encodings: flag 0x80 = y > -y, 0x40 = identity
    // ark-serialize's sign flag: y > -y, i.e. canonical y > (p - 1) / 2 = p >> 1 (p odd); one Montgomery product instead of three
    static inline bool canon_gt_half(const u64 cy[4]) {
        for (int i = 3; i >= 0; i--) {
            const u64 h = (C::Fq::P64[i] >> 1) | (i < 3 ? (C::Fq::P64[i + 1] << 63) : 0);
            if (cy[i] != h) return cy[i] > h;
        }
        return false;
    }
    static inline u8 y_flag(const A4& p) {
        if (p.is_inf()) return 0x40;
        u64 cy[4]; F::to_canon(cy, p.y);
        return canon_gt_half(cy) ? 0x80 : 0x00;
    }
    static inline void ser_uncompressed(u8 out[65], const A4& p) {
        u64 cx[4], cy[4];
        F::to_canon(cx, p.x); F::to_canon(cy, p.y);
        memcpy(out, cx, 32); memcpy(out + 32, cy, 32);
        out[64] = p.is_inf() ? 0x40 : (canon_gt_half(cy) ? 0x80 : 0x00);
    }
    static inline void ser_compressed(u8 out[33], const A4& p) { F::to_bytes(out, p.x); out[32] = y_flag(p); }
    static inline bool from_x(A4& o, const F4& x, bool greatest) {
        F4 a, b; memcpy(a.v, C::A64, 32); memcpy(b.v, C::B64, 32);
        F4 rhs = F::add(F::add(F::mul(F::sqr(x), x), F::mul(a, x)), b), y;
        if (!F::sqrt(y, rhs)) return false;
        F4 ny = F::neg(y);
        bool y_smaller = F::cmp_canon(y, ny) < 0;
        o.x = x;
        o.y = (greatest == y_smaller) ? ny : y;
        return true;
    }
    static inline bool deser_compressed(A4& o, const u8 in[33]) {
        u8 fl = in[32];
        if ((fl & 0x3f) || (fl & 0xc0) == 0xc0) return false;
        F4 x;
        if (!F::from_bytes(x, in)) return false;
        if (fl & 0x40) { if (!x.is_zero()) return false; o = aff_inf(); return true; }
        return from_x(o, x, (fl & 0x80) != 0);
    }
};

}  // namespace host
}  // namespace arkbp
