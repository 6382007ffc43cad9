// Verifier front end on the device (gfx950): what batch_verify does per proof BEFORE the O(N) scalar work — the wire codec, the
// merlin / STROBE-128 transcript replay with its ChaCha20 -> Fr::rand challenge derivation, and the O(k + m) challenge arithmetic —
// as three kernels over a whole batch of same-shaped single-phase proofs.  Reference: src/r1cs/proof.rs:83-91 (from_bytes),
// src/r1cs/verifier.rs:279-287 (commit), :403-541 (verification_scalars), :604-691 (batch_verify), src/inner_product_proof.rs:
// 244-314, src/transcript.rs:45-102.  The host (r1cs_host.inc: batch_verify_device) only stages bytes and launches.
//
//   k_vfe_points   one lane per point: compressed proof points -> (x, y) by a device square root, commitments from ark layout;
//                  every point is written twice: as its 65-byte uncompressed serialization for the sponge ("items", laid out
//                  word-major across proofs so that the sponge's lane-per-proof reads coalesce) and in the resident layout as a
//                  base of the mega-check MSM.  Anything the reference rejects raises a status bit instead.
//   k_vfe_sponge   one lane per proof, 64 proofs per wavefront: Keccak-f[1600] with the 25-word state in registers during a
//                  permutation and in LDS between (so that message pieces can be XORed at run-time byte offsets), driven by the
//                  data-independent schedule of vfe_sched.hpp; each squeezed 32-byte seed goes through ChaCha20 and ark-ff's
//                  Fp::rand (raw limbs, masked, accepted iff < r: they ARE the Montgomery form) into a resident-form challenge.
//   k_vfe_consts   one lane per proof: the proof's one field inversion (Montgomery's trick over y and the u_i), power tables
//                  of z and y^-1, the parameter block k_vfy_tables / k_vfy_batch read, the alpha-scaled tail scalars, the B /
//                  B_blinding contributions;  k_vfe_wv: one lane per (proof, commitment): wV[j] = -sum c z^(q+1).
// Bounds: the sponge is a serial chain per proof (~6 K VALU instructions per permutation, ~150 permutations for m = 256): latency-
// bound at one wave per 64 proofs, it overlaps with the VALU-bound k_vfy_batch of other batches; k_vfe_points is VALU-bound in the
// square roots (~370 products per proof point); the rest is small.
#include <hip/hip_runtime.h>
#include "fe_io.cuh"
#include "vfe.hpp"
#include "vfe_sched.hpp"

namespace arkbp {
namespace vfe {

typedef uint8_t u8;

// ---- bytes <-> words ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_le_words8(u32 w[8], const u8* __restrict__ p) {
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = (u32)p[4 * j] | ((u32)p[4 * j + 1] << 8) | ((u32)p[4 * j + 2] << 16) | ((u32)p[4 * j + 3] << 24);
}
template <class F> __device__ __forceinline__ bool words_lt_p(const u32 w[8]) {   // the canonical range check of ark-serialize's Fp deserializer
    bool lt = false, decided = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (!decided && w[i] != F::P[i]) { lt = w[i] < F::P[i]; decided = true; }
    }
    return lt;
}
__device__ __forceinline__ void store_item(u64* __restrict__ msg, u32 P, u32 item, u32 p, const u32 x[8], const u32 y[8], u32 flag) {
    u64* o = msg + (size_t)item * ITEM_WORDS * P + p;
#pragma unroll
    for (int j = 0; j < 4; j++) o[(size_t)j * P] = (u64)x[2 * j] | ((u64)x[2 * j + 1] << 32);
#pragma unroll
    for (int j = 0; j < 4; j++) o[(size_t)(4 + j) * P] = (u64)y[2 * j] | ((u64)y[2 * j + 1] << 32);
    o[(size_t)8 * P] = flag;
}
// ark-serialize's sign flag of an affine point (SWFlags::from_y_coordinate): 0x80 iff y > -y as canonical integers
template <class F> __device__ __forceinline__ u32 y_sign_flag(const Fe& y_rform_canon) {
    const Fe yc = fe_canon<F>(fe_mul<F>(y_rform_canon, fe_const<F, F::CANON29>()));
    const Fe ny = fe_canon<F>(fe_neg<F, 4>(fe_wred<F>(y_rform_canon)));
    const Fe nyc = fe_canon<F>(fe_mul<F>(ny, fe_const<F, F::CANON29>()));
    return fe_canon_gt(yc, nyc) ? 0x80u : 0u;
}

// ---- k_vfe_points ------------------------------------------------------------------------------------------------------
// Thread order: the (11 + 2k) * P proof points first (point-major: consecutive lanes = consecutive proofs, so the waves that take
// square roots are full and their item stores coalesce), then the m * P commitments, then 5 * P scalars.
template <class C> __global__ void __launch_bounds__(256)
k_vfe_points(Shape sh, const u8* __restrict__ proofs, const u32* __restrict__ V, u64* __restrict__ msg, u32* __restrict__ tail_pts, u32* __restrict__ status) {
    typedef typename C::Fq F;
    typedef typename C::Fr S;
    const u32 P = sh.P, k = sh.k, m = sh.m;
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 n_pp = (u64)(11 + 2 * k) * P, n_v = (u64)m * P, n_s = (u64)5 * P;
    u32 bad = 0;
    if (gid < n_pp) {
        const u32 j = (u32)(gid / P), p = (u32)(gid % P);
        const u8* pr = proofs + (size_t)p * sh.plen;
        const u32 off = j < 11 ? 33u * j : j < 11 + k ? 467u + 33u * (j - 11) : 475u + 33u * k + 33u * (j - 11 - k);
        const u32 slot = j < 6 ? j : j < 11 ? 6u + m + (j - 6) : 11u + m + (j - 11);   // A_I1..S2 | V | T_1..T_6 | L | R  (verifier.rs:378-393)
        if (j == 0) {   // framing: both vectors announce k points (proof.rs: Vec<G> = u64 length || items)
            u32 l[8];
#pragma unroll
            for (int i = 0; i < 8; i++) l[i] = pr[459 + i];
            u32 r[8];
#pragma unroll
            for (int i = 0; i < 8; i++) r[i] = pr[467 + 33 * k + i];
            bool okl = l[0] == k, okr = r[0] == k;
#pragma unroll
            for (int i = 1; i < 8; i++) { okl = okl && l[i] == 0; okr = okr && r[i] == 0; }
            if (!okl || !okr) bad |= ST_FRAMING;
        }
        u32 xw[8], yw[8];
        load_le_words8(xw, pr + off);
        const u32 fl = pr[off + 32];
        u32 o[16];
        u32 flag_out = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) o[i] = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) yw[i] = 0;
        bool x_zero = true;
#pragma unroll
        for (int i = 0; i < 8; i++) x_zero = x_zero && xw[i] == 0;
        if ((fl & 0x3f) || (fl & 0xc0) == 0xc0 || !words_lt_p<F>(xw) || ((fl & 0x40) && !x_zero)) {
            bad |= ST_FORMAT;
        } else if (fl & 0x40) {
            // the identity: fine on the wire; validate_and_append_point rejects it for all but A_I2, A_O2, S2 (verifier.rs:420-470)
            if (j < 3 || j > 5) bad |= ST_IDENTITY;
            flag_out = 0x40;
        } else {
            Aff a;
            if (!aff_from_x<C>(a, fe_canon<F>(fe_load_canon<F>(xw)), (fl & 0x80) != 0)) bad |= ST_FORMAT;
            else {
                aff_store_dev(o, a);
                u32 t[8];
                fe_store_canon<F>(t, a.y);
#pragma unroll
                for (int i = 0; i < 8; i++) yw[i] = t[i];
                flag_out = y_sign_flag<F>(a.y);
            }
        }
        store_item(msg, P, sh.nV + j, p, xw, yw, flag_out);
        u32* tp = tail_pts + ((size_t)p * sh.tail + slot) * 16;
        store_words8(tp, o);
        store_words8(tp + 8, o + 8);
    } else if (gid < n_pp + n_v) {
        const u64 g = gid - n_pp;
        const u32 j = (u32)(g / P), p = (u32)(g % P);   // (point-major like the proof points: 64-byte segments per lane in V and in the tail, coalesced item words)
        u32 w[16];
        load_words8(w, V + ((size_t)p * m + j) * 16);
        load_words8(w + 8, V + ((size_t)p * m + j) * 16 + 8);
        const Aff a = aff_load_ark<C>(w);
        u32 o[16];
        aff_store_dev(o, a);
        u32* tp = tail_pts + ((size_t)p * sh.tail + 6 + j) * 16;
        store_words8(tp, o);
        store_words8(tp + 8, o + 8);
        if (sh.nV) {
            u32 xw[8], yw[8];
            fe_store_canon<F>(xw, a.x);
            fe_store_canon<F>(yw, a.y);
            const bool inf = fe_is_zero_exact(a.x) && fe_is_zero_exact(a.y);
            store_item(msg, P, j, p, xw, yw, inf ? 0x40u : y_sign_flag<F>(a.y));
        }
    } else if (gid < n_pp + n_v + n_s) {
        const u64 g = gid - n_pp - n_v;
        const u32 j = (u32)(g / P), p = (u32)(g % P);
        const u8* pr = proofs + (size_t)p * sh.plen;
        const u32 off = j < 3 ? 363u + 32u * j : 475u + 66u * k + 32u * (j - 3);   // t_x, t_x_blinding, e_blinding | a, b
        u32 w[8], z[8];
        load_le_words8(w, pr + off);
#pragma unroll
        for (int i = 0; i < 8; i++) z[i] = 0;
        if (!words_lt_p<S>(w)) bad |= ST_FORMAT;
        if (j < 3) store_item(msg, P, sh.nV + 11 + 2 * k + j, p, w, z, 0);   // (only the first 32 bytes of a scalar item are absorbed)
    }
    if (bad) atomicOr(status, bad);
}

// ---- Keccak-f[1600], one state per lane ---------------------------------------------------------------------------------
__device__ __forceinline__ u64 rotl64(u64 v, int c) { return c == 0 ? v : (v << c) | (v >> (64 - c)); }
__constant__ const u64 KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
    0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
__device__ __forceinline__ void keccak_f1600(u64 (&a)[25]) {
    // rho offsets by lane index x + 5y
    constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        u64 c[5], d[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
#pragma unroll
        for (int y = 0; y < 5; y++) {
#pragma unroll
            for (int x = 0; x < 5; x++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y] ^ d[x], RHO[x + 5 * y]);
        }
#pragma unroll
        for (int y = 0; y < 5; y++) {
#pragma unroll
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        }
        a[0] ^= KECCAK_RC[round];
    }
}

// ---- ChaCha20 (rand_chacha 0.3: 64-bit block counter, stream 0) and ark-ff's Fp::rand ---------------------------------------
__device__ __forceinline__ u32 rotl32(u32 v, int c) { return (v << c) | (v >> (32 - c)); }
__device__ __forceinline__ void chacha20_block(const u32 key[8], u32 ctr, u32 out[16]) {
    const u32 in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7], ctr, 0u, 0u, 0u};
    u32 x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = in[i];
#define ARKBP_QR(a, b, c, d)                                                                                                                  \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);                                                \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
#pragma unroll 1
    for (int i = 0; i < 10; i++) {
        ARKBP_QR(0, 4, 8, 12) ARKBP_QR(1, 5, 9, 13) ARKBP_QR(2, 6, 10, 14) ARKBP_QR(3, 7, 11, 15)
        ARKBP_QR(0, 5, 10, 15) ARKBP_QR(1, 6, 11, 12) ARKBP_QR(2, 7, 8, 13) ARKBP_QR(3, 4, 9, 14)
    }
#undef ARKBP_QR
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}
// TranscriptProtocol::challenge_scalar (src/transcript.rs:95-101): ChaChaRng::from_seed(seed) then Fr::rand: four next_u64 (eight
// sequential 32-bit words) per attempt, top limb masked to the modulus width, accepted iff < r; the limbs are the Montgomery form
template <class F> __device__ __forceinline__ Fe challenge_from_seed(const u32 seed[8]) {
    u32 blk[16], w[8];
#pragma unroll 1
    for (u32 attempt = 0;; attempt++) {
        if (!(attempt & 1u)) chacha20_block(seed, attempt >> 1, blk);
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = (attempt & 1u) ? blk[8 + j] : blk[j];
        if (F::BITS < 256) w[7] &= 0xffffffffu >> (256 - F::BITS);
        if (words_lt_p<F>(w)) break;
    }
    return fe_load_ark<F>(w);
}

// ---- k_vfe_sponge --------------------------------------------------------------------------------------------------------
template <class C> __global__ void __launch_bounds__(64)
k_vfe_sponge(const u32* __restrict__ sched, const u64* __restrict__ state0, u32 state_stride, const u64* __restrict__ msg, u32 P, u32 nchal,
             u32* __restrict__ chal, u8* __restrict__ seeds) {
    typedef typename C::Fr F;
    __shared__ u64 st[25 * 64];
    const u32 lane = threadIdx.x, p_raw = blockIdx.x * 64u + lane;
    const bool live = p_raw < P;
    const u32 p = live ? p_raw : P - 1;   // idle lanes of the last wave replay the last proof and store nothing
#pragma unroll
    for (int i = 0; i < 25; i++) st[i * 64 + lane] = state0[(size_t)p * state_stride + i];
    const u32 nblocks = __builtin_amdgcn_readfirstlane(sched[0]);
    u32 at = 1;
#pragma unroll 1
    for (u32 b = 0; b < nblocks; b++) {
        const u32 hdr = __builtin_amdgcn_readfirstlane(sched[at]);
        const u32 npieces = hdr & 0xffffu, sq = hdr >> 16;
        const u32* cimg = sched + at + 1;
        const u32* pcs = cimg + 42;
#pragma unroll 1
        for (u32 pi = 0; pi < npieces; pi++) {
            const u32 item = __builtin_amdgcn_readfirstlane(pcs[2 * pi]), w1 = __builtin_amdgcn_readfirstlane(pcs[2 * pi + 1]);
            const u32 so = w1 & 0xffu, len = (w1 >> 8) & 0xffu, dst = w1 >> 16;
            const u64* src = msg + (size_t)item * ITEM_WORDS * P + p;
#pragma unroll 1
            for (u32 d0 = dst & ~7u; d0 < dst + len; d0 += 8) {   // the 64-bit state word at byte offset d0
                const u32 lo = d0 > dst ? d0 : dst, hi = d0 + 8 < dst + len ? d0 + 8 : dst + len, nb = hi - lo;
                const u32 s = so + (lo - dst), sw = s >> 3, sb = s & 7u;
                u64 v = src[(size_t)sw * P] >> (8 * sb);
                if (sb + nb > 8) v |= src[(size_t)(sw + 1) * P] << (8 * (8 - sb));
                if (nb < 8) v &= (1ull << (8 * nb)) - 1ull;
                st[(d0 >> 3) * 64 + lane] ^= v << (8 * (lo - d0));
            }
        }
        u64 a[25];
#pragma unroll
        for (int i = 0; i < 25; i++) a[i] = st[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < 21; i++) a[i] ^= (u64)cimg[2 * i] | ((u64)cimg[2 * i + 1] << 32);
        keccak_f1600(a);
        if (sq != NO_SQUEEZE) {
            // STROBE PRF: the first 32 bytes of the state are the output and are zeroed
            u32 seed[8];
#pragma unroll
            for (int j = 0; j < 4; j++) { seed[2 * j] = (u32)a[j]; seed[2 * j + 1] = (u32)(a[j] >> 32); a[j] = 0; }
            const Fe c = challenge_from_seed<F>(seed);
            if (live) {
                store_fe_dev<F>(chal + ((size_t)p * nchal + sq) * 8, c);
                if (seeds) {
                    u32* so32 = reinterpret_cast<u32*>(seeds + ((size_t)p * nchal + sq) * 32);
#pragma unroll
                    for (int j = 0; j < 8; j++) so32[j] = seed[j];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 25; i++) st[i * 64 + lane] = a[i];
        at += 1 + 42 + 2 * npieces;
    }
}

// ---- k_vfe_consts ------------------------------------------------------------------------------------------------------------
// One lane per proof.  chal: y z u x w u_1..u_k r (resident).  Writes the parameter block [ztab 32 | y^-1 tab 32 | allinv x a b u alpha
// alpha*r*x^2 r*x | u_sq k | 0 ..], the alpha-scaled tail scalars except the commitments' (k_vfe_wv), and alpha*sB, alpha*sBb.
template <class C> __global__ void __launch_bounds__(64)
k_vfe_consts(Shape sh, const u8* __restrict__ proofs, const u32* __restrict__ chal, const u32* __restrict__ alphas, u32* __restrict__ pb_out,
             u32* __restrict__ tail_sc, u32* __restrict__ ws, u32* __restrict__ sB_out) {
    typedef typename C::Fr F;
    const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= sh.P) return;
    const u32 k = sh.k, m = sh.m, nchal = 6 + k;
    const u32* ch = chal + (size_t)p * nchal * 8;
    const u8* pr = proofs + (size_t)p * sh.plen;
    auto scalar_at = [&](u32 off) { u32 w[8]; load_le_words8(w, pr + off); return fe_load_canon<F>(w); };
    const Fe t_x = scalar_at(363), t_xb = scalar_at(395), e_b = scalar_at(427), a = scalar_at(475 + 66 * k), b = scalar_at(507 + 66 * k);
    const Fe y = load_fe_dev<F>(ch), z = load_fe_dev<F>(ch + 8), u = load_fe_dev<F>(ch + 16), x = load_fe_dev<F>(ch + 24), w = load_fe_dev<F>(ch + 32);
    const Fe r = load_fe_dev<F>(ch + (size_t)(5 + k) * 8);
    const Fe alpha = load_fe_dev<F>(alphas + (size_t)p * 8);
    u32* pref = ws + (size_t)p * 32 * 8;
    u32* pb = pb_out + (size_t)p * PB_WORDS;
    u32* ts = tail_sc + (size_t)p * sh.tail * 8;
    // batch_inversion leaves zeros untouched and allinv multiplies the non-zero inverses (inner_product_proof.rs:281-288); the one
    // inversion also serves y (verifier.rs:473; 0 stays 0 as in verify_prepare_t)
    Fe acc = fe_one<F>();
    for (u32 i = 0; i < k; i++) {
        store_fe_dev<F>(pref + (size_t)i * 8, acc);
        const Fe ui = load_fe_dev<F>(ch + (size_t)(5 + i) * 8);
        if (!fe_is_zero_exact(ui)) acc = fe_mul<F>(acc, ui);
    }
    Fe inv, y_inv;
    if (fe_is_zero_exact(y)) { inv = fe_inv<F>(acc); y_inv = y; }
    else { const Fe both = fe_inv<F>(fe_mul<F>(acc, y)); inv = fe_mul<F>(both, y); y_inv = fe_mul<F>(both, acc); }
    const Fe allinv = inv;
    for (u32 i = k; i-- > 0;) {
        const Fe ui = load_fe_dev<F>(ch + (size_t)(5 + i) * 8);
        Fe chi = fe_zero<F>();
        if (!fe_is_zero_exact(ui)) { chi = fe_mul<F>(inv, load_fe_dev<F>(pref + (size_t)i * 8)); inv = fe_mul<F>(inv, ui); }
        const Fe usq = fe_sqr<F>(ui), uisq = fe_sqr<F>(chi);
        store_fe_dev<F>(pb + 576 + (size_t)i * 8, usq);
        store_fe_canon<F>(ts + (size_t)(11 + m + i) * 8, fe_mul<F>(alpha, usq));          // L_j: u_j^2   (verifier.rs:536, batch weight alpha :649-664)
        store_fe_canon<F>(ts + (size_t)(11 + m + k + i) * 8, fe_mul<F>(alpha, uisq));     // R_j: u_j^-2
    }
    { const Fe zero = fe_zero<F>(); for (u32 i = k; i < 32; i++) store_fe_dev<F>(pb + 576 + (size_t)i * 8, zero); }   // (u_sq[31] | pad)
    {
        Fe c = z, ci = y_inv;
        for (u32 j = 0; j < 32; j++) {
            store_fe_dev<F>(pb + (size_t)j * 8, c);
            store_fe_dev<F>(pb + 256 + (size_t)j * 8, ci);
            c = fe_sqr<F>(c); ci = fe_sqr<F>(ci);
        }
    }
    const Fe xx = fe_sqr<F>(x), xxx = fe_mul<F>(xx, x), rxx = fe_mul<F>(r, xx), rx = fe_mul<F>(r, x);
    u32* cst = pb + 512;
    store_fe_dev<F>(cst, allinv); store_fe_dev<F>(cst + 8, x); store_fe_dev<F>(cst + 16, a); store_fe_dev<F>(cst + 24, b); store_fe_dev<F>(cst + 32, u);
    store_fe_dev<F>(cst + 40, alpha); store_fe_dev<F>(cst + 48, fe_mul<F>(alpha, rxx)); store_fe_dev<F>(cst + 56, rx);
    // tails (verifier.rs:521-541): A_I1 x, A_O1 x^2, S1 x^3, A_I2 u x, A_O2 u x^2, S2 u x^3 | V_j wV_j r x^2 | T_1 r x, T_3 r x^3, T_4 r x^4, T_5 r x^5, T_6 r x^6
    const Fe ax = fe_mul<F>(alpha, x), axx = fe_mul<F>(alpha, xx), axxx = fe_mul<F>(alpha, xxx);
    store_fe_canon<F>(ts, ax); store_fe_canon<F>(ts + 8, axx); store_fe_canon<F>(ts + 16, axxx);
    store_fe_canon<F>(ts + 24, fe_mul<F>(u, ax)); store_fe_canon<F>(ts + 32, fe_mul<F>(u, axx)); store_fe_canon<F>(ts + 40, fe_mul<F>(u, axxx));
    const Fe arxx = fe_mul<F>(alpha, rxx);
    u32* tt = ts + (size_t)(6 + m) * 8;
    store_fe_canon<F>(tt, fe_mul<F>(alpha, rx));
    store_fe_canon<F>(tt + 8, fe_mul<F>(arxx, x)); store_fe_canon<F>(tt + 16, fe_mul<F>(arxx, xx)); store_fe_canon<F>(tt + 24, fe_mul<F>(arxx, xxx));
    store_fe_canon<F>(tt + 32, fe_mul<F>(fe_mul<F>(arxx, xx), xx));
    // B: w (t_x - a b) - r t_x (+ r x^2 (wc + delta), formed by k_vfy_batch);  B_blinding: -e_blinding - r t_x_blinding   (verifier.rs:526-531)
    const Fe sB = fe_sub<F, 2>(fe_mul<F>(w, fe_sub<F, 2>(t_x, fe_mul<F>(a, b))), fe_mul<F>(r, t_x));
    const Fe sBb = fe_neg<F, 4>(fe_add(e_b, fe_mul<F>(r, t_xb)));
    store_fe_dev<F>(sB_out + (size_t)p * 16, fe_mul<F>(alpha, sB));
    store_fe_dev<F>(sB_out + (size_t)p * 16 + 8, fe_mul<F>(alpha, sBb));
}

// One lane per (proof, commitment): wV[j] = -sum over the constraints q naming V_j of c * z^(q+1) (verifier.rs:334-338); its tail
// scalar is alpha * r * x^2 * wV[j] (:533).  voff / vq / vc: the terms by commitment.
template <class C> __global__ void __launch_bounds__(256)
k_vfe_wv(Shape sh, const u32* __restrict__ pb_in, const u32* __restrict__ voff, const u32* __restrict__ vq, const u32* __restrict__ vc, u32* __restrict__ tail_sc) {
    typedef typename C::Fr F;
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (u64)sh.P * sh.m) return;
    const u32 p = (u32)(gid / sh.m), j = (u32)(gid % sh.m);
    const u32* pb = pb_in + (size_t)p * PB_WORDS;
    Fe acc = fe_zero<F>();
    for (u32 e = voff[j], e1 = voff[j + 1]; e < e1; e++)
        acc = fe_addr<F>(acc, fe_mul<F>(pow_table<F>(pb, vq[e] + 1u), load_fe_dev<F>(vc + (size_t)e * 8)));
    const Fe wv = fe_neg<F, 4>(acc);
    store_fe_canon<F>(tail_sc + ((size_t)p * sh.tail + 6 + j) * 8, fe_mul<F>(wv, load_fe_dev<F>(pb + 512 + 48)));
}

// sums[0] = sum_p in[2p], sums[1] = sum_p in[2p + 1] (ark words out)
template <class C> __global__ void __launch_bounds__(256)
k_vfe_sum2(const u32* __restrict__ in, u32 P, u32* __restrict__ sums) {
    typedef typename C::Fr F;
    __shared__ u32 sh[9 * 256];
    Fe s0 = fe_zero<F>(), s1 = fe_zero<F>();
    for (u32 p = threadIdx.x; p < P; p += 256) {
        s0 = fe_addr<F>(s0, load_fe_dev<F>(in + (size_t)p * 16));
        s1 = fe_addr<F>(s1, load_fe_dev<F>(in + (size_t)p * 16 + 8));
    }
    s0 = block_sum_fe<F>(s0, sh);
    __syncthreads();
    s1 = block_sum_fe<F>(s1, sh);
    if (threadIdx.x == 0) {
        u32 w[8];
        fe_store_ark<F>(w, s0); store_words8(sums, w);
        fe_store_ark<F>(w, s1); store_words8(sums + 8, w);
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------
template <class C> static int t_points(hipStream_t st, const Shape& sh, const u8* d_proofs, const u32* d_V, u64* d_msg, u32* d_tail_pts, u32* d_status) {
    const u64 total = (u64)(11 + 2 * sh.k + sh.m + 5) * sh.P;
    hipLaunchKernelGGL(k_vfe_points<C>, dim3((u32)((total + 255) / 256)), dim3(256), 0, st, sh, d_proofs, d_V, d_msg, d_tail_pts, d_status);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
template <class C> static int t_sponge(hipStream_t st, const Shape& sh, const u32* d_sched, const u64* d_state0, u32 stride, const u64* d_msg, u32* d_chal, u8* d_seeds) {
    hipLaunchKernelGGL(k_vfe_sponge<C>, dim3((sh.P + 63) / 64), dim3(64), 0, st, d_sched, d_state0, stride, d_msg, sh.P, 6 + sh.k, d_chal, d_seeds);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
template <class C> static int t_prepare(hipStream_t st, const Shape& sh, const u8* d_proofs, const u32* d_chal, const u32* d_alpha, const u32* d_voff, const u32* d_vq,
                                        const u32* d_vc, u32* d_pb, u32* d_tail_sc, u32* d_ws, u32* d_sums) {
    u32* d_sB = d_ws + (size_t)sh.P * 32 * 8;   // (scratch: P x 32 prefix products, then P x 2 head scalars)
    hipLaunchKernelGGL(k_vfe_consts<C>, dim3((sh.P + 63) / 64), dim3(64), 0, st, sh, d_proofs, d_chal, d_alpha, d_pb, d_tail_sc, d_ws, d_sB);
    if (sh.m) hipLaunchKernelGGL(k_vfe_wv<C>, dim3((u32)(((u64)sh.P * sh.m + 255) / 256)), dim3(256), 0, st, sh, d_pb, d_voff, d_vq, d_vc, d_tail_sc);
    hipLaunchKernelGGL(k_vfe_sum2<C>, dim3(1), dim3(256), 0, st, d_sB, sh.P, d_sums);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_points(int curve, hipStream_t st, const Shape& sh, const uint8_t* d_proofs, const uint32_t* d_V, uint64_t* d_msg, uint32_t* d_tail_pts, uint32_t* d_status) {
    return curve == 0 ? t_points<Secq>(st, sh, d_proofs, d_V, d_msg, d_tail_pts, d_status) : t_points<Zorro>(st, sh, d_proofs, d_V, d_msg, d_tail_pts, d_status);
}
int launch_sponge(int curve, hipStream_t st, const Shape& sh, const uint32_t* d_sched, const uint64_t* d_state0, uint32_t state_stride, const uint64_t* d_msg, uint32_t* d_chal,
                  uint8_t* d_seeds_or_null) {
    return curve == 0 ? t_sponge<Secq>(st, sh, d_sched, d_state0, state_stride, d_msg, d_chal, d_seeds_or_null)
                      : t_sponge<Zorro>(st, sh, d_sched, d_state0, state_stride, d_msg, d_chal, d_seeds_or_null);
}
int launch_prepare(int curve, hipStream_t st, const Shape& sh, const uint8_t* d_proofs, const uint32_t* d_chal, const uint32_t* d_alpha, const uint32_t* d_voff,
                   const uint32_t* d_vq, const uint32_t* d_vc, uint32_t* d_pb, uint32_t* d_tail_sc, uint32_t* d_ws, uint32_t* d_sums) {
    return curve == 0 ? t_prepare<Secq>(st, sh, d_proofs, d_chal, d_alpha, d_voff, d_vq, d_vc, d_pb, d_tail_sc, d_ws, d_sums)
                      : t_prepare<Zorro>(st, sh, d_proofs, d_chal, d_alpha, d_voff, d_vq, d_vc, d_pb, d_tail_sc, d_ws, d_sums);
}

}  // namespace vfe
}  // namespace arkbp
