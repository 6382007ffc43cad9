// Inner-product-argument round kernels for gfx950 — the device side of
// `InnerProductProof::create` (/root/reference/src/inner_product_proof.rs:37-239).
//
// All four vectors stay resident in HBM for the whole recursion (the reference re-collects fresh Vecs
// for every msm call, :86-122,:174-200); per round the host sees only L, R (2 points down) and
// returns u, u^-1 (2 scalars up) — the Fiat-Shamir barrier of :132-137.
//
//   k_ipa_scalars   one lane per i < n: the 2n MSM scalars of L and R (:92-102,:112-122,:180-185,
//                   :195-200; round 1 folds G_factors/H_factors in) written as canonical integers,
//                   plus the two inner products c_L = <a_L,b_R>, c_R = <a_R,b_L> (:83-84,:171-172)
//                   as per-workgroup partial sums (LDS tree)
//   k_ipa_ip_finish one workgroup: sums the partials, appends c_L / c_R as scalar 2n
//   k_ipa_fold_ab   a_L <- a_L*u + u^-1*a_R, b_L <- b_L*u^-1 + u*b_R  (:140-141,:217-218)
//   k_ipa_fold_pts  G_L[i] <- s1*G_L[i] + s2*G_R[i], H_L[i] <- t1*H_L[i] + t2*H_R[i] (:143-155,:219-224)
//                   as a joint (Shamir) double-and-add per lane, normalised back to affine in-lane.
//                   The reference does this with a 2-term Pippenger msm + one inversion PER ELEMENT and
//                   it is ~95% of its prove time.
// Storage: scalars and points are in the engine's resident layout (packed radix-2^29 Montgomery form).
#pragma once
#include "msm.cuh"

namespace arkbp {

// ark-layout scalars (Montgomery R = 2^256) -> resident layout, and back
template <class F> __global__ void k_scalars_import(const u32* __restrict__ in, u32* __restrict__ out, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
    load_words8(w, in + (size_t)i * 8);
    store_fe_dev<F>(out + (size_t)i * 8, fe_load_ark<F>(w));
}
template <class F> __global__ void k_scalars_export(const u32* __restrict__ in, u32* __restrict__ out, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
    fe_store_ark<F>(w, load_fe_dev<F>(in + (size_t)i * 8));
    store_words8(out + (size_t)i * 8, w);
}

// several vectors of ark-layout scalars in ONE launch (the witness of a small statement: a_L, a_R, a_O, s_L, s_R; its coefficient
// table and the power tables of y and z): `in` holds them back to back (8 words per scalar; pinned host memory read through the
// bus: no copy in front), dst[k] receives the resident form of the cnt[k] scalars of vector k.  Ten stream operations (five copies,
// five launches of ~4 us each plus their gaps) are ~100 us of a 500 us proof.
static constexpr int IMPORT_MAX = 5;
struct ImportList {
    u32* dst[IMPORT_MAX];
    u32 cnt[IMPORT_MAX];
    int nseg;
};
template <class F> __global__ void k_scalars_import_multi(const u32* __restrict__ in, ImportList l) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 i = t;
    u32* out = nullptr;
#pragma unroll
    for (int k = 0; k < IMPORT_MAX; k++) {
        if (k < l.nseg && !out) { if (i < l.cnt[k]) out = l.dst[k]; else i -= l.cnt[k]; }
    }
    if (!out) return;
    u32 w[8];
    load_words8(w, in + (size_t)t * 8);
    store_fe_dev<F>(out + (size_t)i * 8, fe_load_ark<F>(w));
}

// several buffers zeroed by one launch (the witness-derived vectors at the end of a small proof: eleven memsets are eleven stream
// operations).  n16[k]: 16-byte units of buffer k.
static constexpr int ZERO_MAX = 12;
struct ZeroList {
    uint4* p[ZERO_MAX];
    u32 n16[ZERO_MAX];
    int count;
};
__global__ void __launch_bounds__(256) k_zero_many(ZeroList l) {
    const u32 T = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < ZERO_MAX; k++) {
        if (k >= l.count) break;
        uint4* q = l.p[k];
        const u32 n = l.n16[k];
        for (u32 i = t0; i < n; i += T) q[i] = make_uint4(0, 0, 0, 0);
    }
}

struct Words8 {
    u32 w[8];
};
// n = half length.  sL/sR: (2n+1) x 8 words each (canonical integers).  partials: gridDim.x x 2 x 8 words.
template <class C> __global__ void __launch_bounds__(256)
k_ipa_scalars(const u32* __restrict__ a, const u32* __restrict__ b, const u32* __restrict__ Gf, const u32* __restrict__ Hf, int first, u32 n,
              u32* __restrict__ sL, u32* __restrict__ sR, u32* __restrict__ partials, int pending, Words8 gGw, Words8 gHw,
              const u32* __restrict__ rho_pow /* pending == 2: H_true[i] = K * rho^i * Hhat[i], K = gHw, table of rho^(2^k) */) {
    typedef typename C::Fr F;
    __shared__ u32 sh[9 * 256];
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    Fe pl = fe_zero<F>(), pr = fe_zero<F>();
    if (i < n) {
        Fe aL = load_fe_dev<F>(a + (size_t)i * 8), aR = load_fe_dev<F>(a + (size_t)(n + i) * 8);
        Fe bL = load_fe_dev<F>(b + (size_t)i * 8), bR = load_fe_dev<F>(b + (size_t)(n + i) * 8);
        pl = fe_mul<F>(aL, bR);
        pr = fe_mul<F>(aR, bL);
        Fe xl = aL, yl = bR, xr = aR, yr = bL;
        if (first) {
            xl = fe_mul<F>(aL, load_fe_dev<F>(Gf + (size_t)(n + i) * 8));
            yl = fe_mul<F>(bR, load_fe_dev<F>(Hf + (size_t)i * 8));
            xr = fe_mul<F>(aR, load_fe_dev<F>(Gf + (size_t)i * 8));
            yr = fe_mul<F>(bL, load_fe_dev<F>(Hf + (size_t)(n + i) * 8));
        }
        if (pending) {  // the resident G/H hold Ghat = G / gamma_G, Hhat = H / gamma_H (see k_ipa_fold_uniform)
            const Fe gG = fe_load_ark<F>(gGw.w), gH = fe_load_ark<F>(gHw.w);
            xl = fe_mul<F>(xl, gG); xr = fe_mul<F>(xr, gG);
            if (pending == 2) {   // geometric pending factor: yl pairs with H_L[i] (index i), yr with H_R[i] (index n + i)
                const Fe ri = fe_mul<F>(gH, pow_table<F>(rho_pow, i));
                yl = fe_mul<F>(yl, ri);
                yr = fe_mul<F>(yr, fe_mul<F>(ri, pow_table<F>(rho_pow, n)));
            } else {
                yl = fe_mul<F>(yl, gH); yr = fe_mul<F>(yr, gH);
            }
        }
        store_fe_canon<F>(sL + (size_t)i * 8, xl);
        store_fe_canon<F>(sL + (size_t)(n + i) * 8, yl);
        store_fe_canon<F>(sR + (size_t)i * 8, xr);
        store_fe_canon<F>(sR + (size_t)(n + i) * 8, yr);
    }
    pl = block_sum_fe<F>(fe_wred<F>(pl), sh);
    pr = block_sum_fe<F>(fe_wred<F>(pr), sh);
    if (threadIdx.x == 0) {
        store_fe_dev<F>(partials + (size_t)blockIdx.x * 16, pl);
        store_fe_dev<F>(partials + (size_t)blockIdx.x * 16 + 8, pr);
    }
}
// The MSM scalars of the round AFTER a deferred first fold (see k_ipa_fold_tab2): a and b are folded, G and H are still the
// generator tables, Ghat'[j] = G[n+j] + tG1 * G[j] and Hhat'[j] = H[n+j] + tH1 * H[j] exist only as formulas.  With half length m
// (= n/2) and x_l, y_l, x_r, y_r as in k_ipa_scalars (pending factors included):
//   L = sum x_l[i] * Ghat'[m+i] + sum y_l[i] * Hhat'[i]  = <x_l, G[n+m..)> + <x_l*tG1, G[m..)> + <y_l, H[n..)> + <y_l*tH1, H[0..)>
//   R = sum x_r[i] * Ghat'[i]   + sum y_r[i] * Hhat'[m+i]= <x_r, G[n..)>   + <x_r*tG1, G[0..)> + <y_r, H[n+m..)> + <y_r*tH1, H[m..)>
// sL / sR: 4m scalars in that order (canonical integers), then the inner products (k_ipa_ip_finish).
template <class C> __global__ void __launch_bounds__(256)
k_ipa_scalars_deferred(const u32* __restrict__ a, const u32* __restrict__ b, u32 m, u32* __restrict__ sL, u32* __restrict__ sR, u32* __restrict__ partials,
                       int pending, Words8 gGw, Words8 gHw, const u32* __restrict__ rho_pow, Words8 tG1w, Words8 tH1w) {
    typedef typename C::Fr F;
    __shared__ u32 sh[9 * 256];
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    Fe pl = fe_zero<F>(), pr = fe_zero<F>();
    if (i < m) {
        Fe aL = load_fe_dev<F>(a + (size_t)i * 8), aR = load_fe_dev<F>(a + (size_t)(m + i) * 8);
        Fe bL = load_fe_dev<F>(b + (size_t)i * 8), bR = load_fe_dev<F>(b + (size_t)(m + i) * 8);
        pl = fe_mul<F>(aL, bR);
        pr = fe_mul<F>(aR, bL);
        Fe xl = aL, yl = bR, xr = aR, yr = bL;
        if (pending) {
            const Fe gG = fe_load_ark<F>(gGw.w), gH = fe_load_ark<F>(gHw.w);
            xl = fe_mul<F>(xl, gG); xr = fe_mul<F>(xr, gG);
            if (pending == 2) {
                const Fe ri = fe_mul<F>(gH, pow_table<F>(rho_pow, i));
                yl = fe_mul<F>(yl, ri);
                yr = fe_mul<F>(yr, fe_mul<F>(ri, pow_table<F>(rho_pow, m)));
            } else {
                yl = fe_mul<F>(yl, gH); yr = fe_mul<F>(yr, gH);
            }
        }
        const Fe tG1 = fe_load_ark<F>(tG1w.w), tH1 = fe_load_ark<F>(tH1w.w);
        store_fe_canon<F>(sL + (size_t)i * 8, xl);
        store_fe_canon<F>(sL + (size_t)(m + i) * 8, fe_mul<F>(xl, tG1));
        store_fe_canon<F>(sL + (size_t)(2 * m + i) * 8, yl);
        store_fe_canon<F>(sL + (size_t)(3 * m + i) * 8, fe_mul<F>(yl, tH1));
        store_fe_canon<F>(sR + (size_t)i * 8, xr);
        store_fe_canon<F>(sR + (size_t)(m + i) * 8, fe_mul<F>(xr, tG1));
        store_fe_canon<F>(sR + (size_t)(2 * m + i) * 8, yr);
        store_fe_canon<F>(sR + (size_t)(3 * m + i) * 8, fe_mul<F>(yr, tH1));
    }
    pl = block_sum_fe<F>(fe_wred<F>(pl), sh);
    pr = block_sum_fe<F>(fe_wred<F>(pr), sh);
    if (threadIdx.x == 0) {
        store_fe_dev<F>(partials + (size_t)blockIdx.x * 16, pl);
        store_fe_dev<F>(partials + (size_t)blockIdx.x * 16 + 8, pr);
    }
}
template <class C> __global__ void __launch_bounds__(256)
k_ipa_ip_finish(const u32* __restrict__ partials, u32 nparts, u32* __restrict__ outL, u32* __restrict__ outR, Words8 qw, int with_qw) {
    typedef typename C::Fr F;
    __shared__ u32 sh[9 * 256];
    Fe pl = fe_zero<F>(), pr = fe_zero<F>();
    for (u32 j = threadIdx.x; j < nparts; j += 256) {
        pl = fe_addr<F>(pl, load_fe_dev<F>(partials + (size_t)j * 16));
        pr = fe_addr<F>(pr, load_fe_dev<F>(partials + (size_t)j * 16 + 8));
    }
    pl = block_sum_fe<F>(pl, sh);
    pr = block_sum_fe<F>(pr, sh);
    if (threadIdx.x == 0) {
        store_fe_canon<F>(outL, pl);
        store_fe_canon<F>(outR, pr);
        if (with_qw) {   // Q = qw * B: the scalars of B for a fixed-base round (slot after the inner products)
            const Fe q = fe_load_ark<F>(qw.w);
            store_fe_canon<F>(outL + 8, fe_mul<F>(pl, q));
            store_fe_canon<F>(outR + 8, fe_mul<F>(pr, q));
        }
    }
}

// u, u_inv: ark Montgomery words
template <class C> __global__ void __launch_bounds__(256)
k_ipa_fold_ab(u32* __restrict__ a, u32* __restrict__ b, u32 n, Words8 uw, Words8 uiw) {
    typedef typename C::Fr F;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe u = fe_load_ark<F>(uw.w), ui = fe_load_ark<F>(uiw.w);
    Fe aL = load_fe_dev<F>(a + (size_t)i * 8), aR = load_fe_dev<F>(a + (size_t)(n + i) * 8);
    Fe bL = load_fe_dev<F>(b + (size_t)i * 8), bR = load_fe_dev<F>(b + (size_t)(n + i) * 8);
    store_fe_dev<F>(a + (size_t)i * 8, fe_norm(fe_add(fe_mul<F>(aL, u), fe_mul<F>(ui, aR))));
    store_fe_dev<F>(b + (size_t)i * 8, fe_norm(fe_add(fe_mul<F>(bL, ui), fe_mul<F>(u, bR))));
}

// joint double-and-add: s1*P1 + s2*P2 with lane-private canonical 256-bit s1, s2.  Lanes disagree on which of
// {P1, P2, P1+P2} to add at each step, so the addend is SELECTED (predicated moves over a 3-entry Jacobian table) and one
// general addition runs for the whole wave; branching per lane would serialise three different add shapes.
template <class C> __device__ __forceinline__ Jac shamir2(const Aff& P1, const Aff& P2, const u32 s1[8], const u32 s2[8]) {
    const Jac T1 = jac_from_aff<C>(P1), T2 = jac_from_aff<C>(P2);
    const Jac T3 = jac_madd<C>(T1, P2);
    Jac acc = jac_inf<C>();
#pragma unroll 1
    for (int wd = 7; wd >= 0; wd--) {
        const u32 e1 = s1[wd], e2 = s2[wd];
#pragma unroll 1
        for (int bit = 31; bit >= 0; bit--) {
            acc = jac_dbl<C>(acc);
            const u32 b1 = (e1 >> bit) & 1, b2 = (e2 >> bit) & 1;
            Jac t;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                t.X.l[i] = b2 ? (b1 ? T3.X.l[i] : T2.X.l[i]) : T1.X.l[i];
                t.Y.l[i] = b2 ? (b1 ? T3.Y.l[i] : T2.Y.l[i]) : T1.Y.l[i];
                t.Z.l[i] = b2 ? (b1 ? T3.Z.l[i] : T2.Z.l[i]) : T1.Z.l[i];
            }
            const Jac r = jac_add<C>(acc, t);
            const bool take = (b1 | b2) != 0;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                acc.X.l[i] = take ? r.X.l[i] : acc.X.l[i];
                acc.Y.l[i] = take ? r.Y.l[i] : acc.Y.l[i];
                acc.Z.l[i] = take ? r.Z.l[i] : acc.Z.l[i];
            }
        }
    }
    return acc;
}

// ---- frozen-generator tail rounds --------------------------------------------------------------------------------------
// Below ~2^11 elements a fold launch is one lane's serial ladder (>= 1 ms) however few points it folds.  From length n0 on the
// engine therefore stops folding G and H: it keeps the vectors G0, H0 of that moment and a coefficient per element,
//   G_true[j] = sum over t = j (mod len) of cG[t] * G0[t]      (len = current vector length; same for H),
// so a fold is cG[t] *= (t mod len < len/2 ? u^-1 : u) (and the mirror for cH), and L, R of a round are MSMs over ALL of G0, H0
// (2*n0 + 1 terms, every base used by exactly one of L, R; the other gets a zero scalar, which costs nothing in the bucket sort).
// a and b fold as before.  (src/inner_product_proof.rs:166-224 computes the same L, R, a', b'; G', H' are never output.)
template <class C> __global__ void __launch_bounds__(256)
k_ipa_freeze_init(u32* __restrict__ cG, u32* __restrict__ cH, u32 n0, int pending, Words8 gGw, Words8 gHw, const u32* __restrict__ rho_pow) {
    typedef typename C::Fr F;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n0) return;
    Fe g = fe_one<F>(), h = fe_one<F>();
    if (pending) {
        g = fe_load_ark<F>(gGw.w);
        h = fe_load_ark<F>(gHw.w);
        if (pending == 2) h = fe_mul<F>(h, pow_table<F>(rho_pow, t));
    }
    store_fe_dev<F>(cG + (size_t)t * 8, g);
    store_fe_dev<F>(cH + (size_t)t * 8, h);
}
// n = half of the current length.  sL / sR: (2*n0 + 1) x 8 words (canonical); partials as in k_ipa_scalars.
template <class C> __global__ void __launch_bounds__(256)
k_ipa_frozen_scalars(const u32* __restrict__ a, const u32* __restrict__ b, const u32* __restrict__ cG, const u32* __restrict__ cH, u32 n, u32 n0,
                     u32* __restrict__ sL, u32* __restrict__ sR, u32* __restrict__ partials) {
    typedef typename C::Fr F;
    __shared__ u32 sh[9 * 256];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    Fe pl = fe_zero<F>(), pr = fe_zero<F>();
    if (t < n) {   // <a_L, b_R>, <a_R, b_L>
        pl = fe_mul<F>(load_fe_dev<F>(a + (size_t)t * 8), load_fe_dev<F>(b + (size_t)(n + t) * 8));
        pr = fe_mul<F>(load_fe_dev<F>(a + (size_t)(n + t) * 8), load_fe_dev<F>(b + (size_t)t * 8));
    }
    if (t < n0) {
        const u32 r = t & (2 * n - 1);
        const bool lo = r < n;
        const u32 j = lo ? r : r - n;
        const Fe g = load_fe_dev<F>(cG + (size_t)t * 8), h = load_fe_dev<F>(cH + (size_t)t * 8);
        const Fe zero = fe_zero<F>();
        // G part: L pairs a_L[j] with G_R[j] (r >= n), R pairs a_R[j] with G_L[j] (r < n)
        const Fe xg = fe_mul<F>(load_fe_dev<F>(a + (size_t)(lo ? n + j : j) * 8), g);
        store_fe_canon<F>(sL + (size_t)t * 8, lo ? zero : xg);
        store_fe_canon<F>(sR + (size_t)t * 8, lo ? xg : zero);
        // H part: L pairs b_R[j] with H_L[j] (r < n), R pairs b_L[j] with H_R[j] (r >= n)
        const Fe xh = fe_mul<F>(load_fe_dev<F>(b + (size_t)(lo ? n + j : j) * 8), h);
        store_fe_canon<F>(sL + (size_t)(n0 + t) * 8, lo ? xh : zero);
        store_fe_canon<F>(sR + (size_t)(n0 + t) * 8, lo ? zero : xh);
    }
    pl = block_sum_fe<F>(fe_wred<F>(pl), sh);
    pr = block_sum_fe<F>(fe_wred<F>(pr), sh);
    if (threadIdx.x == 0) {
        store_fe_dev<F>(partials + (size_t)blockIdx.x * 16, pl);
        store_fe_dev<F>(partials + (size_t)blockIdx.x * 16 + 8, pr);
    }
}
template <class C> __global__ void __launch_bounds__(256)
k_ipa_frozen_fold(u32* __restrict__ cG, u32* __restrict__ cH, u32 n, u32 n0, Words8 uw, Words8 uiw) {
    typedef typename C::Fr F;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n0) return;
    const Fe u = fe_load_ark<F>(uw.w), ui = fe_load_ark<F>(uiw.w);
    const bool lo = (t & (2 * n - 1)) < n;
    store_fe_dev<F>(cG + (size_t)t * 8, fe_mul<F>(load_fe_dev<F>(cG + (size_t)t * 8), lo ? ui : u));
    store_fe_dev<F>(cH + (size_t)t * 8, fe_mul<F>(load_fe_dev<F>(cH + (size_t)t * 8), lo ? u : ui));
}

// Fold epilogue.  jac_ws == nullptr: the lane inverts its own Z (one a^(p-2) ladder, ~450 products) and writes the affine point.
// Large rounds pass a workspace instead: lane t leaves its Jacobian result there and k_ipa_fold_finish converts the whole
// round with ONE inversion per m points (Montgomery's trick along a strided run of points per lane).
template <class C> __device__ __forceinline__ void fold_emit(u32* __restrict__ V, u32 i, u32 t, const Jac& acc, u32* __restrict__ jac_ws) {
    if (jac_ws) { store_jac_ws<C>(jac_ws + (size_t)t * 24, acc); return; }
    const Aff o = jac_to_aff<C>(acc);
    u32 w[16];
    aff_store_dev(w, o);
    store_words8(V + (size_t)i * 16, w);
    store_words8(V + (size_t)i * 16 + 8, w + 8);
}
// lane t0 owns the points t0, t0 + T, t0 + 2T, ... (T = all lanes of the grid; coalesced at every step)
template <class C> __global__ void __launch_bounds__(256)
k_ipa_fold_finish(const u32* __restrict__ jac_ws, u32* __restrict__ pref_ws, u32* __restrict__ G, u32* __restrict__ H, u32 n, int which, u32 total, u32 m) {
    typedef typename C::Fq F;
    const u32 T = gridDim.x * blockDim.x;
    const u32 t0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (t0 >= total) return;
    Fe acc = fe_one<F>();
    u32 cnt = 0;
    for (u32 p = t0; cnt < m && p < total; cnt++, p += T) {
        u32 w[8];
        load_words8(w, jac_ws + (size_t)p * 24 + 16);
        const Fe z = fe_unpack(w);
        if (!fe_is_zero_exact(z)) acc = fe_mul<F>(acc, z);
        fe_pack(w, fe_canon<F>(acc));
        store_words8(pref_ws + (size_t)p * 8, w);
    }
    Fe inv = fe_inv<F>(acc);
    for (int j = (int)cnt - 1; j >= 0; j--) {
        const u32 p = t0 + (u32)j * T;
        const Jac P = load_jac_ws(jac_ws + (size_t)p * 24);
        Aff o;
        if (jac_is_inf(P)) { o.x = fe_zero<F>(); o.y = fe_zero<F>(); }
        else {
            Fe prev = fe_one<F>();
            if (j > 0) { u32 w[8]; load_words8(w, pref_ws + (size_t)(p - T) * 8); prev = fe_unpack(w); }
            const Fe zinv = fe_mul<F>(inv, prev);
            inv = fe_mul<F>(inv, P.Z);
            o = jac_to_aff_with_zinv<C>(P, zinv);
        }
        const bool isH = which == 2 || (which == 3 && p >= n);
        const u32 i = (which == 3 && isH) ? p - n : p;
        u32* V = isH ? H : G;
        u32 w[16];
        aff_store_dev(w, o);
        store_words8(V + (size_t)i * 16, w);
        store_words8(V + (size_t)i * 16 + 8, w + 8);
    }
}

// One lane per output point: lanes [0,n) fold G, lanes [n,2n) fold H.  first != 0: per-element factors.
// u, u_inv ark Montgomery words.  In place: lane i reads elements i and n+i of its vector, writes i.
template <class C> __global__ void __launch_bounds__(256)
k_ipa_fold_pts(u32* __restrict__ G, u32* __restrict__ H, const u32* __restrict__ Gf, const u32* __restrict__ Hf, int first, u32 n, Words8 uw,
               Words8 uiw, int which /* 1: G only, 2: H only, 3: both */, u32* __restrict__ jac_ws) {
    typedef typename C::Fr F;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (which == 3 ? 2 * n : n)) return;
    const bool isH = which == 2 || (which == 3 && t >= n);
    const u32 i = (which == 3 && isH) ? t - n : t;
    u32* V = isH ? H : G;
    const u32* Vf = isH ? Hf : Gf;
    // G: (u^-1 * Gf[i], u * Gf[n+i])      H: (u * Hf[i], u^-1 * Hf[n+i])
    Fe c1 = fe_load_ark<F>(isH ? uw.w : uiw.w), c2 = fe_load_ark<F>(isH ? uiw.w : uw.w);
    if (first) {
        c1 = fe_mul<F>(c1, load_fe_dev<F>(Vf + (size_t)i * 8));
        c2 = fe_mul<F>(c2, load_fe_dev<F>(Vf + (size_t)(n + i) * 8));
    }
    u32 s1[8], s2[8];
    fe_store_canon<F>(s1, c1);
    fe_store_canon<F>(s2, c2);
    const Aff P1 = load_aff_dev(V + (size_t)i * 16), P2 = load_aff_dev(V + (size_t)(n + i) * 16);
    const Jac r = shamir2<C>(P1, P2, s1, s2);
    fold_emit<C>(V, i, t, r, jac_ws);
}

// Uniform-scalar rounds (every round after the first; src/inner_product_proof.rs:219-224).  With the same (u^-1, u) for all
// i,  u^-1*G_L[i] + u*G_R[i] = u * (G_R[i] + u^-2 * G_L[i]):  the lane computes only Ghat'[i] = G_R[i] + t*G_L[i] (ONE scalar
// multiplication, t = u^-2 for G, u^2 for H) and the common factor is kept as a pending scalar gamma that k_ipa_scalars
// multiplies into the next rounds' MSM scalars — G and H themselves are never output by the prover.  t is wave-uniform, so its
// non-adjacent form is computed once on the host: bit i of plus/minus = digit +1/-1 at 2^i (257 digits).
struct Naf {
    u32 plus[9], minus[9];
};
// acc + T in the ladders and table folds: the one-block form, redone with the complete formulas in the rare exceptional cases
// (ec.cuh jac_madd_fast); T stays in registers there
template <class C> __device__ __forceinline__ Jac ladder_madd(const Jac& acc, const Aff& T) {
    bool rare;
    Jac nxt = jac_madd_fast<C>(acc, T, rare);
    if (__builtin_expect(rare, 0)) nxt = jac_madd<C>(acc, T);
    return nxt;
}
// QUAD: four lanes (a DPP quad) share one point's ladder (ecq.cuh) — for the rounds whose points do not fill the chip; lane 0 of
// the quad emits.
template <class C, bool QUAD = false> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
k_ipa_fold_uniform(u32* __restrict__ G, u32* __restrict__ H, u32 n, Naf tG, Naf tH, int which /* 1: G only, 2: H only, 3: both */,
                   u32* __restrict__ jac_ws) {
    const u32 tl = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 t = QUAD ? tl >> 2 : tl, ql = threadIdx.x & 3u;
    if (t >= (which == 3 ? 2 * n : n)) return;
    const bool isH = which == 2 || (which == 3 && t >= n);  // waves are homogeneous for n >= 64
    const u32 i = (which == 3 && isH) ? t - n : t;
    u32* V = isH ? H : G;
    const Aff P1 = load_aff_dev(V + (size_t)i * 16), P2 = load_aff_dev(V + (size_t)(n + i) * 16);
    const Aff N1 = aff_cneg_lazy<C>(P1, true);
    Jac acc = jac_inf<C>();
#pragma unroll 1
    for (int wd = 8; wd >= 0; wd--) {
        const u32 ep = isH ? tH.plus[wd] : tG.plus[wd], em = isH ? tH.minus[wd] : tG.minus[wd];
#pragma unroll 1
        for (int bit = (wd == 8 ? 0 : 31); bit >= 0; bit--) {
            acc = QUAD ? qjac_dbl<C>(acc, ql) : jac_dbl<C>(acc);
            if (((ep | em) >> bit) & 1) {   // ONE addition site: the digit's sign picks y (wave-uniform)
                Aff T1 = P1; T1.y = ((em >> bit) & 1) ? N1.y : P1.y;
                acc = QUAD ? qjac_madd<C>(acc, T1, ql) : ladder_madd<C>(acc, T1);
            }
        }
    }
    acc = QUAD ? qjac_madd<C>(acc, P2, ql) : jac_madd<C>(acc, P2);
    if (!QUAD || ql == 0) fold_emit<C>(V, i, t, acc, jac_ws);
}

// GLV variant of the uniform fold for curves with the j = 0 endomorphism phi(x, y) = (beta*x, y) = [lambda](x, y) (secq256k1):
// t = t1 + t2*lambda with |t1|, |t2| < 2^129 (host-side lattice decomposition, signs folded into the digit masks), so
// t*P1 = t1*P1 + t2*phi(P1) needs 129 doublings instead of 256.  Same contract as k_ipa_fold_uniform otherwise.
struct Naf2 {
    u32 p1[5], m1[5], p2[5], m2[5];   // bit i: digit +1 / -1 at 2^i of t1 (p1/m1) and t2 (p2/m2), 130 digits
};
template <class C, bool QUAD = false> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
k_ipa_fold_glv(u32* __restrict__ G, u32* __restrict__ H, u32 n, Naf2 tG, Naf2 tH, int which /* 1: G only, 2: H only, 3: both */,
               u32* __restrict__ jac_ws) {
    typedef typename C::Fq F;
    const u32 tl = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 t = QUAD ? tl >> 2 : tl, ql = threadIdx.x & 3u;
    if (t >= (which == 3 ? 2 * n : n)) return;
    const bool isH = which == 2 || (which == 3 && t >= n);
    const u32 i = (which == 3 && isH) ? t - n : t;
    u32* V = isH ? H : G;
    const Aff P1 = load_aff_dev(V + (size_t)i * 16), P2 = load_aff_dev(V + (size_t)(n + i) * 16);
    Aff Q1;   // phi(P1); the identity (0,0) maps to itself
    Q1.x = fe_canon<F>(fe_mul<F>(P1.x, fe_const<F, C::BETA29>()));
    Q1.y = P1.y;
    const Aff N1 = aff_cneg_lazy<C>(P1, true);   // phi keeps y: -phi(P1) = (beta*x, -y)
    Jac acc = jac_inf<C>();
#pragma unroll 1
    for (int wd = 4; wd >= 0; wd--) {
        const u32 a = isH ? tH.p1[wd] : tG.p1[wd], b = isH ? tH.m1[wd] : tG.m1[wd];
        const u32 c = isH ? tH.p2[wd] : tG.p2[wd], d = isH ? tH.m2[wd] : tG.m2[wd];
#pragma unroll 1
        for (int bit = (wd == 4 ? 1 : 31); bit >= 0; bit--) {
            acc = QUAD ? qjac_dbl<C>(acc, ql) : jac_dbl<C>(acc);
            if (QUAD) {
                if (((a | b) >> bit) & 1) { Aff T1 = P1; T1.y = ((b >> bit) & 1) ? N1.y : P1.y; acc = qjac_madd<C>(acc, T1, ql); }   // masks are wave-uniform
                if (((c | d) >> bit) & 1) { Aff T2 = Q1; T2.y = ((d >> bit) & 1) ? N1.y : P1.y; acc = qjac_madd<C>(acc, T2, ql); }
            } else {
#pragma unroll 1
                for (int h = 0; h < 2; h++) {   // ONE addition site for both halves (the masks are wave-uniform)
                    const u32 nz = h ? (c | d) : (a | b), ng = h ? d : b;
                    if ((nz >> bit) & 1) {
                        Aff T;
#pragma unroll
                        for (int l = 0; l < 9; l++) { T.x.l[l] = h ? Q1.x.l[l] : P1.x.l[l]; T.y.l[l] = ((ng >> bit) & 1) ? N1.y.l[l] : P1.y.l[l]; }
                        acc = ladder_madd<C>(acc, T);
                    }
                }
            }
        }
    }
    acc = QUAD ? qjac_madd<C>(acc, P2, ql) : jac_madd<C>(acc, P2);
    if (!QUAD || ql == 0) fold_emit<C>(V, i, t, acc, jac_ws);
}

// ---- fixed-base tables for the FIRST fold round ------------------------------------------------------------------------------
// Round 1 folds the generator tables themselves: Ghat'[i] = G[n+i] + t*G[i] with ONE multiplier t for all i (uniform rounds, see
// k_ipa_fold_uniform) and bases that are the same for every proof — BulletproofGens is fixed, the reference clones it per proof
// (src/r1cs/prover.rs:796-797).  So t*G[i] can be a fixed-base multiplication: the table holds e * 2^(w*j) * G[i] for every
// window j and digit magnitude e in [1, 2^(w-1)], affine, laid out [j][e-1][i].  Because t is wave-uniform, every lane of a wave
// needs the SAME (j, e) row: a window's look-up is one coalesced 64 B x 64 read, and the lane does one mixed add per non-zero
// digit and NO doublings — 34 mixed adds (w = 8, GLV halves) instead of 130 doublings + ~88 mixed adds.  Half of all fold work of a
// proof is in round 1.  The tables are large (2^19 bases, w = 8: 73 GB per vector) and live in HBM for the life of the ctx — the
// MI355X has 288 GB of it and this path touches HBM at < 1 % of its bandwidth otherwise.
struct FtabDigits {
    unsigned short e1[44], e2[44];   // per window: |digit| of t1 (and of t2, the GLV half that goes with phi(P)); 0 = skip
    unsigned long long neg1, neg2;   // bit j: the digit of window j is negative
    u32 nwin;
};
// one window of the table for bases [0, n): state[i] = 2^(w*j) * P_i on entry (Jacobian; first != 0: taken from the generator table),
// 2^(w*(j+1)) * P_i on exit; tmp[(e-1)*n + i] = e * state_in[i] (Jacobian) for e = 1..E, E = 2^(w-1)
template <class C> __global__ void __launch_bounds__(256)
k_ftab_window(const u32* __restrict__ gens, u32* __restrict__ state, u32* __restrict__ tmp, u32 n, u32 E, int first) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Jac B;
    if (first) { const Aff p = load_aff_dev(gens + (size_t)i * 16); B = jac_madd<C>(jac_inf<C>(), p); }
    else B = load_jac_ws(state + (size_t)i * 24);
    Jac acc = B;
    store_jac_ws<C>(tmp + (size_t)i * 24, acc);
#pragma unroll 1
    for (u32 e = 2; e <= E; e++) {
        acc = jac_add<C>(acc, B);
        store_jac_ws<C>(tmp + ((size_t)(e - 1) * n + i) * 24, acc);
    }
    store_jac_ws<C>(state + (size_t)i * 24, jac_dbl<C>(acc));   // 2 * E * B = 2^w * B
}
// tmp (E x n Jacobian) -> out (E x n affine, resident layout) with one inversion per lane (Montgomery's trick along e)
template <class C> __global__ void __launch_bounds__(256)
k_ftab_normalize(const u32* __restrict__ tmp, u32* __restrict__ pref, u32* __restrict__ out, u32 n, u32 E) {
    typedef typename C::Fq F;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe acc = fe_one<F>();
#pragma unroll 1
    for (u32 e = 0; e < E; e++) {
        u32 w[8];
        load_words8(w, tmp + ((size_t)e * n + i) * 24 + 16);
        const Fe z = fe_unpack(w);
        if (!fe_is_zero_exact(z)) acc = fe_mul<F>(acc, z);
        fe_pack(w, fe_canon<F>(acc));
        store_words8(pref + ((size_t)e * n + i) * 8, w);
    }
    Fe inv = fe_inv<F>(acc);
#pragma unroll 1
    for (int e = (int)E - 1; e >= 0; e--) {
        const Jac P = load_jac_ws(tmp + ((size_t)e * n + i) * 24);
        Aff o;
        if (jac_is_inf(P)) { o.x = fe_zero<F>(); o.y = fe_zero<F>(); }
        else {
            Fe prev = fe_one<F>();
            if (e > 0) { u32 w[8]; load_words8(w, pref + ((size_t)(e - 1) * n + i) * 8); prev = fe_unpack(w); }
            const Fe zinv = fe_mul<F>(inv, prev);
            inv = fe_mul<F>(inv, P.Z);
            o = jac_to_aff_with_zinv<C>(P, zinv);
        }
        u32 w[16];
        aff_store_dev(w, o);
        store_words8(out + ((size_t)e * n + i) * 16, w);
        store_words8(out + ((size_t)e * n + i) * 16 + 8, w + 8);
    }
}
// ---- integrity check of the precomputed tables (bp_gens_tables_check) ---------------------------------------------------------
// The tables are up to ~150 GB of HBM that only ever enter a proof a few rows at a time (the rows a round's digits select), so a
// wrong entry would surface as one rare bad proof.  The check walks EVERY entry through the chain rule the builder used, with
// different formulas (mixed addition / doubling of the stored AFFINE neighbours, compared projectively: no inversion), anchored
// at the generator itself:   T[0][0][i] = G[i],  T[j][e][i] = T[j][e-1][i] + T[j][0][i],  T[j+1][0][i] = 2 * T[j][E-1][i].
// Jacobian J equals affine A ?  (X == x*Z^2, Y == y*Z^3; identity <-> identity)
template <class C> __device__ __forceinline__ bool jac_equals_aff(const Jac& J, const Aff& A) {
    typedef typename C::Fq F;
    const bool ji = jac_is_inf(J), ai = aff_is_inf(A);
    if (ji || ai) return ji && ai;
    const Fe zz = fe_sqr<F>(J.Z);
    if (!fe_eq_mod<F>(fe_wred<F>(J.X), fe_mul<F>(A.x, zz))) return false;
    return fe_eq_mod<F>(fe_wred<F>(J.Y), fe_mul<F>(A.y, fe_mul<F>(zz, J.Z)));
}
template <class C> __global__ void __launch_bounds__(256)
k_ftab_check(const u32* __restrict__ gens, const u32* __restrict__ T, u32 n, u32 E, u32 nwin, unsigned long long* __restrict__ bad,
             u32 g_first = 0, u32 g_stride = 1 /* column i of the table stands for generator g_first + i * g_stride (a rank's slice) */) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 wrong = 0;
    Aff first = load_aff_dev(gens + ((size_t)g_first + (size_t)i * g_stride) * 16);      // what T[j][0][i] must equal (j = 0), then only its Jacobian successor
    Jac expect = jac_from_aff<C>(first);
#pragma unroll 1
    for (u32 j = 0; j < nwin; j++) {
        const u32* row = T + ((size_t)j * E) * n * 16;
        const Aff base = load_aff_dev(row + (size_t)i * 16);
        if (!jac_equals_aff<C>(expect, base)) wrong++;
        Aff prev = base;
#pragma unroll 1
        for (u32 e = 2; e <= E; e++) {
            const Aff cur = load_aff_dev(row + ((size_t)(e - 1) * n + i) * 16);
            if (!jac_equals_aff<C>(jac_madd<C>(jac_from_aff<C>(prev), base), cur)) wrong++;
            prev = cur;
        }
        expect = jac_dbl<C>(jac_from_aff<C>(prev));       // 2 * E * base = 2^w * base
    }
    if (wrong) atomicAdd(bad, (unsigned long long)wrong);
}
// fixed-base MSM rows: row[0][i] = P_i, row[r+1][i] = 16 * row[r][i]
template <class C> __global__ void __launch_bounds__(256)
k_fb_rows_check(const u32* __restrict__ gens, const u32* __restrict__ T, u32 n, size_t row_points, u32 R, unsigned long long* __restrict__ bad) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 wrong = 0;
    Jac expect = jac_from_aff<C>(load_aff_dev(gens + (size_t)i * 16));
#pragma unroll 1
    for (u32 r = 0; r < R; r++) {
        const Aff cur = load_aff_dev(T + ((size_t)r * row_points + i) * 16);
        if (!jac_equals_aff<C>(expect, cur)) wrong++;
        expect = jac_dbl<C>(jac_dbl<C>(jac_dbl<C>(jac_dbl<C>(jac_from_aff<C>(cur)))));
    }
    if (wrong) atomicAdd(bad, (unsigned long long)wrong);
}

// the first-round uniform fold from the tables: lane t -> V[i] = V[n+i] + t_V * Base(g_first + i*g_stride), V = G (lanes [0, n)) or H.
// G, H: the working vectors (their upper halves are read, their lower halves written — or the Jacobian workspace, see fold_emit).
// one window of a fixed-base multiplication: acc += (+-) e1 * 2^(w*j) * Base[gi] (+ the GLV half: (+-) e2 * 2^(w*j) * phi(Base[gi])),
// digits wave-uniform.  ONE addition site for both halves; the table entry is read again in the rare exceptional case.
template <class C> __device__ __forceinline__ Jac ftab_step(Jac acc, const u32* __restrict__ T, u32 n_tab, u32 E, u32 gi, u32 j, u32 e1, u32 e2, bool neg1, bool neg2) {
    typedef typename C::Fq F;
#pragma unroll 1
    for (int h = 0; h < (C::HAS_GLV ? 2 : 1); h++) {
        const u32 e = h ? e2 : e1;
        if (!e) continue;
        const u32* src = T + (((size_t)j * E + (e - 1)) * n_tab + gi) * 16;
        const bool neg = h ? neg2 : neg1;
        Aff P = load_aff_dev(src);
        if constexpr (C::HAS_GLV) { if (h && !aff_is_inf(P)) P.x = fe_mul<F>(P.x, fe_const<F, C::BETA29>()); }   // phi(x, y) = (beta * x, y)
        bool rare;
        Jac nxt = jac_madd_fast<C>(acc, aff_cneg_lazy<C>(P, neg), rare);
        if (__builtin_expect(rare, 0)) {
            Aff Q = load_aff_dev(src);
            if constexpr (C::HAS_GLV) { if (h && !aff_is_inf(Q)) Q.x = fe_mul<F>(Q.x, fe_const<F, C::BETA29>()); }
            nxt = jac_madd<C>(acc, aff_cneg_lazy<C>(Q, neg));
        }
        acc = nxt;
    }
    return acc;
}
template <class C> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
k_ipa_fold_tab(const u32* __restrict__ TG, const u32* __restrict__ TH, u32 n_tab, u32 E, u32* __restrict__ G, u32* __restrict__ H, u32 n, FtabDigits dG,
               FtabDigits dH, int which, u32 g_first, u32 g_stride, u32* __restrict__ jac_ws,
               const u32* __restrict__ Gin /* nullable: read the upper halves from here (the resident generator tables) instead of G / H */,
               const u32* __restrict__ Hin) {
    typedef typename C::Fq F;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (which == 3 ? 2 * n : n)) return;
    const bool isH = which == 2 || (which == 3 && t >= n);   // waves are homogeneous for n >= 64
    const u32 i = (which == 3 && isH) ? t - n : t;
    u32* V = isH ? H : G;
    const u32* T = isH ? TH : TG;
    const u32 gi = g_first + i * g_stride;
    Jac acc = jac_inf<C>();
    const u32 nwin = isH ? dH.nwin : dG.nwin;
#pragma unroll 1
    for (u32 j = 0; j < nwin; j++) {
        const u32 e1 = isH ? dH.e1[j] : dG.e1[j], e2 = isH ? dH.e2[j] : dG.e2[j];
        acc = ftab_step<C>(acc, T, n_tab, E, gi, j, e1, e2, (((isH ? dH.neg1 : dG.neg1) >> j) & 1ull) != 0, (((isH ? dH.neg2 : dG.neg2) >> j) & 1ull) != 0);
    }
    const u32* Vin = isH ? (Hin ? Hin : H) : (Gin ? Gin : G);
    acc = jac_madd<C>(acc, load_aff_dev(Vin + (size_t)(n + i) * 16));
    fold_emit<C>(V, i, t, acc, jac_ws);
}

// ---- TWO fold rounds from the tables ------------------------------------------------------------------------------------------
// The first fold round need not be materialised at all: with the second round's multiplier t2 known,
//   Ghat''[i] = Ghat'[m+i] + t2 * Ghat'[i] = G[n+m+i] + t1 * G[m+i] + t2 * G[n+i] + (t1*t2) * G[i]        (n = 2m; same for H)
// is three fixed-base multiplications of GENERATORS — table look-ups, no doublings — instead of one table round (n points) plus
// one ladder round (m points, 130 doublings + ~88 mixed adds each).  The second round's L and R are MSMs over the generator tables
// with split scalars (k_ipa_scalars_deferred), i.e. fixed-base MSMs over the precomputed rows.  Needs tables for bases [0, n+m).
// d[0] / d[1] / d[2]: digits of t1, t2, t1*t2 (per vector), multiplying the bases m+i, n+i, i.
struct FtabDigits3 {
    FtabDigits d[3];
};
// lanes [0, m): G, lanes [m, 2m): H.  Gin / Hin: the resident generator tables (bases n+m+i are read from them); G / H: the working
// vectors that receive the m folded points each (or the Jacobian workspace, see fold_emit).
// ONE loop body for the three multipliers, and the digit words are selected field by field: a whole-struct select
// (isH ? dH.d[k] : dG.d[k]) makes a private copy of the kernel arguments — 1.2 KB of scratch per lane, 1.3 GB of scratch traffic
// per proof, measured as 0.66 GB of HBM writes in the first version of this kernel.
template <class C> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
k_ipa_fold_tab2(const u32* __restrict__ TG, const u32* __restrict__ TH, u32 n_tab, u32 E, u32* __restrict__ G, u32* __restrict__ H, u32 m, FtabDigits3 dG,
                FtabDigits3 dH, u32* __restrict__ jac_ws, const u32* __restrict__ Gin, const u32* __restrict__ Hin,
                u32 g_first /* table base of local element j: g_first + j * g_stride (an index-cyclic slice; 0, 1 otherwise) */, u32 g_stride) {
    typedef typename C::Fq F;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * m) return;
    const bool isH = t >= m;                        // waves are homogeneous for m >= 64
    const u32 i = isH ? t - m : t;
    const u32 n = 2 * m;
    const u32* T = isH ? TH : TG;
    Jac acc = jac_inf<C>();
#pragma unroll 1
    for (u32 mu = 0; mu < 3; mu++) {
        const u32 gi = g_first + (mu == 0 ? m + i : mu == 1 ? n + i : i) * g_stride;        // t1 * Base[m+i] + t2 * Base[n+i] + t1*t2 * Base[i]
        const u32 nwin = isH ? dH.d[mu].nwin : dG.d[mu].nwin;
        const unsigned long long neg1 = isH ? dH.d[mu].neg1 : dG.d[mu].neg1, neg2 = isH ? dH.d[mu].neg2 : dG.d[mu].neg2;
#pragma unroll 1
        for (u32 j = 0; j < nwin; j++) {
            const u32 e1 = isH ? dH.d[mu].e1[j] : dG.d[mu].e1[j], e2 = isH ? dH.d[mu].e2[j] : dG.d[mu].e2[j];
            acc = ftab_step<C>(acc, T, n_tab, E, gi, j, e1, e2, ((neg1 >> j) & 1ull) != 0, ((neg2 >> j) & 1ull) != 0);
        }
    }
    acc = jac_madd<C>(acc, load_aff_dev((isH ? Hin : Gin) + (size_t)(n + m + i) * 16));
    fold_emit<C>(isH ? H : G, i, t, acc, jac_ws);
}

}  // namespace arkbp
