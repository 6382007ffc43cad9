// GLV split of scalars for curves with the j = 0 endomorphism (secq256k1) — host + device code (tests/test_fp29_host.py runs it on
// the CPU against Python integers; msm.cuh uses it in the MSM partition kernel).
#pragma once
#include "fp29.cuh"

namespace arkbp {

// ---- GLV split of an MSM's scalars (curves with the j = 0 endomorphism phi(x, y) = (beta * x, y) = [lambda](x, y): secq256k1) ----
// s * P = k1 * P + k2 * phi(P) with |k1|, |k2| < 2^128: an n-term MSM over 256-bit scalars becomes a 2n-term MSM over 128-bit ones.
// The (term, window) work of the accumulate is unchanged; what halves is everything that scales with the number of WINDOWS — the
// buckets (reduction + aggregation trees) and above all the serial Horner tail on the host (128 doublings instead of 256).
// Rounded lattice coordinates c1 = (k * G1 + 2^383) >> 384, c2 = (k * G2 + 2^383) >> 384 (G1, G2 = floor(2^384 * b2 / r), floor(2^384 *
// -b1 / r)); k1 = k - c1 * a1 - c2 * a2, k2 = -c1 * b1 - c2 * b2, evaluated mod 2^160 in two's complement.  |k1| <= (a1 + a2) / 2 + eps
// < 1.09 * 2^127, |k2| <= (-b1 + b2) / 2 + eps < 1.28 * 2^127 (tools/gen_params.py asserts the basis; tests check 2 * 10^5 scalars
// against big integers); a magnitude that does not fit 128 bits returns false and the caller takes the ordinary schedule.
// out: mag1[4] | mag2[4] | signs (bit 0: k1 < 0, bit 1: k2 < 0) | 3 unused words.
static constexpr int MSM_GLV_WORDS = 12;
template <class C> ARKBP_HD void glv_mulhi(const u32 k[8], const u32 (&G)[9], u32 c[5]) {
    u32 acc[17];
#pragma unroll
    for (int i = 0; i < 17; i++) acc[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 carry = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const u64 t = (u64)k[i] * G[j] + acc[i + j] + carry;
            acc[i + j] = (u32)t;
            carry = t >> 32;
        }
        acc[i + 9] = (u32)carry;
    }
    u64 t = (u64)acc[11] + 0x80000000ull;   // + 2^383: round to nearest
    u32 cy = (u32)(t >> 32);
#pragma unroll
    for (int i = 0; i < 5; i++) { t = (u64)acc[12 + i] + cy; c[i] = (u32)t; cy = (u32)(t >> 32); }
}
// acc (5 words, mod 2^160) -= a * b for 5-word a, b (only the columns below 2^160)
ARKBP_HD void glv_submul5(u32 acc[5], const u32 a[5], const u32 (&b)[5]) {
    u32 prod[5];
#pragma unroll
    for (int i = 0; i < 5; i++) prod[i] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        u64 carry = 0;
#pragma unroll
        for (int j = 0; j + i < 5; j++) {
            const u64 t = (u64)a[i] * b[j] + prod[i + j] + carry;
            prod[i + j] = (u32)t;
            carry = t >> 32;
        }
    }
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const u64 t = (u64)acc[i] - prod[i] - borrow;
        acc[i] = (u32)t;
        borrow = (u32)(t >> 63);
    }
}
ARKBP_HD bool glv_sign_magnitude(u32 v[5], u32& neg) {
    neg = v[4] >> 31;
    if (neg) {
        u32 cy = 1;
#pragma unroll
        for (int i = 0; i < 5; i++) { const u64 t = (u64)(~v[i]) + cy; v[i] = (u32)t; cy = (u32)(t >> 32); }
    }
    return v[4] == 0;
}
template <class C> ARKBP_HD bool glv_split(const u32 k[8], u32 out[MSM_GLV_WORDS]) {
    u32 c1[5], c2[5];
    glv_mulhi<C>(k, C::GLV_G1W, c1);
    glv_mulhi<C>(k, C::GLV_G2W, c2);
    u32 k1[5], k2[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { k1[i] = k[i]; k2[i] = 0; }
    glv_submul5(k1, c1, C::GLV_A1W);
    glv_submul5(k1, c2, C::GLV_A2W);      // k1 = k - c1*a1 - c2*a2
    glv_submul5(k2, c2, C::GLV_B2W);      // k2 = c1*(-b1) - c2*b2 = -(c2*b2) - (-(c1*nb1)) : two steps
    {
        u32 t[5];
#pragma unroll
        for (int i = 0; i < 5; i++) t[i] = 0;
        glv_submul5(t, c1, C::GLV_NB1W);  // t = -(c1 * nb1)
        u32 borrow = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) { const u64 d = (u64)k2[i] - t[i] - borrow; k2[i] = (u32)d; borrow = (u32)(d >> 63); }
    }
    u32 n1, n2;
    const bool ok1 = glv_sign_magnitude(k1, n1), ok2 = glv_sign_magnitude(k2, n2);
#pragma unroll
    for (int i = 0; i < 4; i++) { out[i] = k1[i]; out[4 + i] = k2[i]; }
    out[8] = n1 | (n2 << 1);
    out[9] = out[10] = out[11] = 0;
    return ok1 && ok2;
}

}  // namespace arkbp
