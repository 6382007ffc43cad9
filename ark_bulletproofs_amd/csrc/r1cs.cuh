// R1CS prover / verifier vector kernels for gfx950 — the O(N) field work of
// `Prover::prove_and_return_transcript` (/root/reference/src/r1cs/prover.rs:671-703, 744-756, 781-789,
// with util.rs:75-102) and of `Verifier::verification_scalars` (src/r1cs/verifier.rs:465-514,
// inner_product_proof.rs:279-311).  Vectors are HBM-resident in the engine's scalar layout (32 B,
// radix-2^29 Montgomery form, packed); nothing of length N returns to the host.
//
//   k_r1cs_poly_t     one pass over a_L,a_R,a_O,s_L,s_R,wL,wR,wO: forms l1,l2,l3 / r0,r1,r3 in registers
//                     (prover.rs:685-698) and reduces the six coefficient inner products t1..t6
//                     (util.rs:75-93) — the reference materialises 6 vectors and makes 8 passes
//   k_r1cs_sum6       finishes the per-workgroup partials
//   k_r1cs_poly_eval  l_vec = l(x), r_vec = r(x) with the power-of-two padding (prover.rs:746-756) and the
//                     IPA factor vectors G_factors / H_factors (prover.rs:781-789), written once
//   k_vfy_scalars     g_scalars / h_scalars of the verifier's mega-check (verifier.rs:492-514) with s[i]
//                     evaluated in closed form from the squared challenges (inner_product_proof.rs:302-311)
#pragma once
#include "ipa.cuh"

namespace arkbp {

template <class F> __device__ __forceinline__ Fe pow_from_table(const u32* __restrict__ tab, u32 e) { return pow_table<F>(tab, e); }

// ypow: 64 x 8 words: y^(2^k) for k < 32, then y^-(2^k) for k < 32.
// partials: gridDim.x x 6 x 8 words.
template <class C> __global__ void __launch_bounds__(256)
k_r1cs_poly_t(const u32* __restrict__ aL, const u32* __restrict__ aR, const u32* __restrict__ aO, const u32* __restrict__ sL,
              const u32* __restrict__ sR, const u32* __restrict__ wL, const u32* __restrict__ wR, const u32* __restrict__ wO,
              const u32* __restrict__ ypow, u32 n, u32* __restrict__ partials, u32* __restrict__ final_ark = nullptr /* one workgroup: the six sums
              in ark words, as k_r1cs_sum would leave them (a small statement saves that launch) */) {
    typedef typename C::Fr F;
    __shared__ u32 sh[9 * 256];
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    Fe t[6];
#pragma unroll
    for (int k = 0; k < 6; k++) t[k] = fe_zero<F>();
    if (i < n) {
        const Fe yi = pow_from_table<F>(ypow, i), yni = pow_from_table<F>(ypow + 32 * 8, i);
        const size_t o = (size_t)i * 8;
        const Fe l1 = fe_wred<F>(fe_norm(fe_add(load_fe_dev<F>(aL + o), fe_mul<F>(yni, load_fe_dev<F>(wR + o)))));
        const Fe l2 = load_fe_dev<F>(aO + o), l3 = load_fe_dev<F>(sL + o);
        const Fe r0 = fe_wred<F>(fe_sub<F, 4>(load_fe_dev<F>(wO + o), yi));
        const Fe r1 = fe_wred<F>(fe_norm(fe_add(fe_mul<F>(yi, load_fe_dev<F>(aR + o)), load_fe_dev<F>(wL + o))));
        const Fe r3 = fe_mul<F>(yi, load_fe_dev<F>(sR + o));
        t[0] = fe_mul<F>(l1, r0);                                                   // t1 = <l1,r0>
        t[1] = fe_addr<F>(fe_mul<F>(l1, r1), fe_mul<F>(l2, r0));                    // t2 = <l1,r1> + <l2,r0>
        t[2] = fe_addr<F>(fe_mul<F>(l2, r1), fe_mul<F>(l3, r0));                    // t3 = <l2,r1> + <l3,r0>
        t[3] = fe_addr<F>(fe_mul<F>(l1, r3), fe_mul<F>(l3, r1));                    // t4 = <l1,r3> + <l3,r1>
        t[4] = fe_mul<F>(l2, r3);                                                   // t5 = <l2,r3>
        t[5] = fe_mul<F>(l3, r3);                                                   // t6 = <l3,r3>
    }
#pragma unroll   // (a rolled loop indexes t[] dynamically and sends the six accumulators through scratch: 440 MB of writes at N = 2^20)
    for (int k = 0; k < 6; k++) {
        Fe s = block_sum_fe<F>(fe_wred<F>(t[k]), sh);
        if (threadIdx.x == 0) {
            if (final_ark) { u32 w[8]; fe_store_ark<F>(w, s); store_words8(final_ark + (size_t)k * 8, w); }
            else store_fe_dev<F>(partials + ((size_t)blockIdx.x * 6 + k) * 8, s);
        }
    }
}
// out: cnt x 8 words, ark Montgomery layout (read by the host)
template <class C> __global__ void __launch_bounds__(256) k_r1cs_sum(const u32* __restrict__ partials, u32 nparts, u32 cnt, u32* __restrict__ out) {
    typedef typename C::Fr F;
    __shared__ u32 sh[9 * 256];
    for (u32 k = 0; k < cnt; k++) {
        Fe s = fe_zero<F>();
        for (u32 j = threadIdx.x; j < nparts; j += 256) s = fe_addr<F>(s, load_fe_dev<F>(partials + ((size_t)j * cnt + k) * 8));
        s = block_sum_fe<F>(s, sh);
        if (threadIdx.x == 0) {
            u32 w[8];
            fe_store_ark<F>(w, s);
            store_words8(out + (size_t)k * 8, w);
        }
    }
}

// x, u: ark Montgomery words.  Writes l_vec, r_vec, Gf, Hf for i < N (padded length).
template <class C> __global__ void __launch_bounds__(256)
k_r1cs_poly_eval(const u32* __restrict__ aL, const u32* __restrict__ aR, const u32* __restrict__ aO, const u32* __restrict__ sL,
                 const u32* __restrict__ sR, const u32* __restrict__ wL, const u32* __restrict__ wR, const u32* __restrict__ wO,
                 const u32* __restrict__ ypow, u32 n, u32 n1, u32 N, Words8 xw, Words8 uw, u32* __restrict__ lvec, u32* __restrict__ rvec,
                 u32* __restrict__ Gf, u32* __restrict__ Hf) {
    typedef typename C::Fr F;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const Fe x = fe_load_ark<F>(xw.w);
    const Fe yi = pow_from_table<F>(ypow, i), yni = pow_from_table<F>(ypow + 32 * 8, i);
    const size_t o = (size_t)i * 8;
    Fe lv, rv;
    if (i < n) {
        const Fe l1 = fe_wred<F>(fe_norm(fe_add(load_fe_dev<F>(aL + o), fe_mul<F>(yni, load_fe_dev<F>(wR + o)))));
        const Fe l2 = load_fe_dev<F>(aO + o), l3 = load_fe_dev<F>(sL + o);
        const Fe r0 = fe_wred<F>(fe_sub<F, 4>(load_fe_dev<F>(wO + o), yi));
        const Fe r1 = fe_wred<F>(fe_norm(fe_add(fe_mul<F>(yi, load_fe_dev<F>(aR + o)), load_fe_dev<F>(wL + o))));
        const Fe r3 = fe_mul<F>(yi, load_fe_dev<F>(sR + o));
        // l(x) = x*(l1 + x*(l2 + x*l3));  r(x) = r0 + x*(r1 + x*(x*r3))     (util.rs:95-102, l0 = r2 = 0)
        Fe acc = fe_addr<F>(fe_mul<F>(x, l3), l2);
        acc = fe_addr<F>(fe_mul<F>(acc, x), l1);
        lv = fe_mul<F>(acc, x);
        acc = fe_mul<F>(fe_mul<F>(x, r3), x);
        acc = fe_addr<F>(acc, r1);
        rv = fe_addr<F>(fe_mul<F>(acc, x), r0);
    } else {
        lv = fe_zero<F>();
        rv = fe_neg<F, 4>(fe_wred<F>(yi));  // padding: r_vec[i] = -y^i (prover.rs:753-756)
    }
    store_fe_dev<F>(lvec + o, lv);
    store_fe_dev<F>(rvec + o, rv);
    const Fe gf = i < n1 ? fe_one<F>() : fe_load_ark<F>(uw.w);
    store_fe_dev<F>(Gf + o, gf);
    store_fe_dev<F>(Hf + o, fe_mul<F>(yni, gf));
}

// Verifier scalars.  chal: u_sq[k] (resident words, creation order); consts: resident words
// [allinv, x, a, b, u, alpha]; ypow as above (only the inverse half is used).
// g[i] = u_or_1 * (x * y^-i * wR[i] - a * s[i]),  h[i] = u_or_1 * (y^-i * (x*wL[i] + wO[i] - b*s[N-1-i]) - 1)
// with wL/wR/wO = 0 for i >= n.  s[i] = allinv * prod_{j: bit j of i set} u_sq[k-1-j]  (ipp :302-311),
// s[N-1-i] = the same product over the clear bits.
// ACC = false (Verifier::verify, verifier.rs:492-514): g_out/h_out <- canonical integers, ready for the MSM.
// ACC = true  (batch_verify, verifier.rs:649-664): g_out/h_out (resident form) += alpha * g / h.
template <class C, bool ACC> __global__ void __launch_bounds__(256)
k_vfy_scalars(const u32* __restrict__ wL, const u32* __restrict__ wR, const u32* __restrict__ wO, const u32* __restrict__ ypow,
              const u32* __restrict__ chal, const u32* __restrict__ consts, u32 n, u32 n1, u32 N, u32 k, u32* __restrict__ g_out,
              u32* __restrict__ h_out) {
    typedef typename C::Fr F;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const Fe allinv = load_fe_dev<F>(consts), x = load_fe_dev<F>(consts + 8), a = load_fe_dev<F>(consts + 16), b = load_fe_dev<F>(consts + 24);
    Fe s_i = allinv, s_rev = allinv;
#pragma unroll 1
    for (u32 j = 0; j < k; j++) {
        const Fe usq = load_fe_dev<F>(chal + (size_t)(k - 1 - j) * 8);
        if ((i >> j) & 1) s_i = fe_mul<F>(s_i, usq); else s_rev = fe_mul<F>(s_rev, usq);
    }
    const Fe yni = pow_from_table<F>(ypow + 32 * 8, i);
    const size_t o = (size_t)i * 8;
    Fe g, h;
    if (i < n) {
        const Fe ywR = fe_mul<F>(yni, load_fe_dev<F>(wR + o));
        g = fe_sub<F, 2>(fe_mul<F>(x, ywR), fe_mul<F>(a, s_i));
        Fe t = fe_addr<F>(fe_mul<F>(x, load_fe_dev<F>(wL + o)), load_fe_dev<F>(wO + o));
        t = fe_sub<F, 2>(t, fe_mul<F>(b, s_rev));
        h = fe_sub<F, 2>(fe_mul<F>(yni, t), fe_one<F>());
    } else {
        g = fe_neg<F, 2>(fe_mul<F>(a, s_i));
        h = fe_sub<F, 2>(fe_mul<F>(yni, fe_neg<F, 2>(fe_mul<F>(b, s_rev))), fe_one<F>());
    }
    if (i >= n1) {
        const Fe u = load_fe_dev<F>(consts + 32);
        g = fe_mul<F>(g, u);
        h = fe_mul<F>(h, u);
    }
    if (ACC) {
        const Fe alpha = load_fe_dev<F>(consts + 40);
        store_fe_dev<F>(g_out + o, fe_norm(fe_add(load_fe_dev<F>(g_out + o), fe_mul<F>(alpha, g))));
        store_fe_dev<F>(h_out + o, fe_norm(fe_add(load_fe_dev<F>(h_out + o), fe_mul<F>(alpha, h))));
    } else {
        store_fe_canon<F>(g_out + o, g);
        store_fe_canon<F>(h_out + o, h);
    }
}
// ---- batch verification of proofs that share one circuit template (same constraint matrices) -------------------------
// The reference's batch_verify runs verification_scalars per proof on the CPU (src/r1cs/verifier.rs:617-627): flattened
// constraints (O(nnz)), the s vector, g/h scalars (O(N)) — then alpha-scales and scatter-adds (:649-664).  Here the
// constraint matrices W_L, W_R, W_O live on the GPU once, in CSC form (column = multiplier index; entry = (constraint q,
// coefficient id)), and ONE launch handles a whole set of proofs: lane (i, chunk) walks the proofs of its chunk and keeps
//   sum_p alpha_p * g_p[i],  sum_p alpha_p * h_p[i],  sum_p (alpha_p r_p x_p^2) * y_p^-i wR_p[i] wL_p[i]   (the delta terms)
// in registers.  Per-proof inputs are a 3.3 KB parameter block (power tables of z and y^-1, challenges, a, b, alpha).
struct VfyTemplateDev {
    const u32* coefs;        // distinct coefficients, resident words
    const u32* m_off;        // n+1: per multiplier index i, the entries of W_L, W_R, W_O columns i merged and sorted by constraint
    const u32* m_ent;        // (vector << 30) | constraint index q     (vector 0 = W_L, 1 = W_R, 2 = W_O)
    const u32* m_c;          // coefficient id per merged entry; bit 31 / bit 30 flag the coefficients +1 / -1
    const u32* const_q;      // constant terms (Variable::One) of the constraints: wc = -sum z^(q+1) * coef  (verifier.rs:339-341)
    const u32* const_c;
    u32 n_const;
};
static constexpr u32 VFY_PB_WORDS = 832;  // ztab[32] | yinv_tab[32] | consts[8]: allinv,x,a,b,u,alpha,alpha*coefD,r*x | u_sq[31] | pad
static constexpr u32 VFY_PB_SCALARS = VFY_PB_WORDS / 8;

// Entries of the per-proof split tables are kept as their nine 29-bit limbs (12 words each, 16-byte aligned): k_vfy_batch reads ~20
// of them per (proof, element) and the packed form costs ~25 unpack instructions per read
static constexpr u32 VT_W = 12;
__device__ __forceinline__ void store_fe_limbs(u32* __restrict__ p, const Fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
    q[1] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
    p[8] = a.l[8];
}
__device__ __forceinline__ Fe load_fe_limbs(const u32* __restrict__ p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    Fe r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = p[8];
    return r;
}
// Per-proof split tables for k_vfy_batch.  With i = hi * 2^LOB + lo, and the proof's constants folded into the LOW halves so that the
// batch kernel gets each weighted quantity with ONE product:
//   -alpha * a * s[i]    (s[i]     = allinv * prod_{bit j of i set} u_sq[k-1-j])    = s_lo[lo] * s_hi[hi]    (inner_product_proof.rs:302-311 in closed form)
//   -alpha * y^-i * b * s[N-1-i]  (s[N-1-i] = allinv * prod_{bit j of i clear} u_sq[k-1-j])  = r_lo[lo] * r_hi[hi]   (y^-i splits the same way,
//                                            so its halves ride in r_lo / r_hi; the signs ride in the low halves: the batch kernel ADDS
//                                            both quantities inside fused product pairs)
//   alpha * x * y^-i                                                                 = yx_lo[lo] * y_hi[hi]
//   alpha * y^-i                                                                     = ya_lo[lo] * y_hi[hi]
//   z^e  (e = q + 1 <= Q)                                                            = z_lo[e & 255] * z_hi[e >> 8]
//   alpha * x * z^e  (the constant terms, whose weight alpha*r*x^2 is (r*x) times this)  = zx_lo[e & 255] * z_hi[e >> 8]
// Layout per proof (entries of VT_W words: the nine limbs of a product, i.e. < 1.04 p): [s_lo | s_hi | r_lo | r_hi | yx_lo | y_hi | ya_lo | z_lo (256) | z_hi (nzhi) | zx_lo (256)],
// stride vfy_tab_stride() = 3 * (2^LOB + 2^HIB) + 2^LOB + 512 + nzhi.  grid (ceil(max(2^LOB, 2^HIB, 256, nzhi) / 256), P).
// ... followed by the proof's constants x, alpha, u, r*x in the same limb form (VFY_TAB_CONSTS entries): k_vfy_batch reads them once
// per (proof, element) and the packed parameter block costs ~25 unpack instructions per read
static constexpr u32 VFY_TAB_CONSTS = 4;
__host__ __device__ inline size_t vfy_tab_stride(u32 nlo, u32 nhi, u32 nzhi) { return (size_t)3 * (nlo + nhi) + nlo + 256 + nzhi + 256 + VFY_TAB_CONSTS; }
template <class C> __global__ void __launch_bounds__(256)
k_vfy_tables(const u32* __restrict__ params, const u32* __restrict__ perm, u32 P, u32 k, u32 LOB, u32 nzhi, u32* __restrict__ tables) {
    typedef typename C::Fr F;
    const u32 tIdx = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
    if (p >= P) return;
    const u32 HIB = k - LOB, nlo = 1u << LOB, nhi = 1u << HIB;
    const u32* pb = params + (size_t)perm[p] * VFY_PB_WORDS;   // perm: the group's proofs as positions in the block's parameter array
    const u32* ztab = pb;
    const u32* ytab = pb + 256;
    const u32* cst = pb + 512;
    const u32* usq = pb + 576;
    u32* T = tables + (size_t)p * vfy_tab_stride(nlo, nhi, nzhi) * VT_W;
    u32* Z = T + ((size_t)3 * (nlo + nhi) + nlo) * VT_W;
    if (tIdx < 256) {
        const Fe zp = pow_table<F>(ztab, tIdx);
        store_fe_limbs(Z + (size_t)tIdx * VT_W, zp);
        store_fe_limbs(Z + (size_t)(256 + nzhi + tIdx) * VT_W, fe_mul<F>(fe_mul<F>(load_fe_dev<F>(cst + 40), load_fe_dev<F>(cst + 8)), zp));
    }
    if (tIdx < nzhi) store_fe_limbs(Z + (size_t)(256 + tIdx) * VT_W, pow_table<F>(ztab, tIdx << 8));
    if (tIdx < VFY_TAB_CONSTS) {   // x, alpha, u, r*x
        const u32 src = tIdx == 0 ? 8u : tIdx == 1 ? 40u : tIdx == 2 ? 32u : 56u;
        store_fe_limbs(Z + (size_t)(256 + nzhi + 256 + tIdx) * VT_W, load_fe_dev<F>(cst + src));
    }
    if (tIdx < nlo) {
        const Fe allinv = load_fe_dev<F>(cst), x = load_fe_dev<F>(cst + 8), a = load_fe_dev<F>(cst + 16), b = load_fe_dev<F>(cst + 24);
        const Fe alpha = load_fe_dev<F>(cst + 40);
        const Fe ya = fe_mul<F>(alpha, pow_table<F>(ytab, tIdx));
        Fe s = fe_neg<F, 2>(fe_mul<F>(allinv, fe_mul<F>(alpha, a))), r = fe_mul<F>(fe_neg<F, 2>(fe_mul<F>(allinv, b)), ya);
        for (u32 j = 0; j < LOB; j++) {
            const Fe q = load_fe_dev<F>(usq + (size_t)(k - 1 - j) * 8);
            if ((tIdx >> j) & 1) s = fe_mul<F>(s, q); else r = fe_mul<F>(r, q);
        }
        store_fe_limbs(T + (size_t)tIdx * VT_W, fe_wred<F>(s));
        store_fe_limbs(T + (size_t)(nlo + nhi + tIdx) * VT_W, r);
        store_fe_limbs(T + (size_t)(2 * (nlo + nhi) + tIdx) * VT_W, fe_mul<F>(ya, x));
        store_fe_limbs(T + (size_t)(3 * (nlo + nhi) + tIdx) * VT_W, ya);
    }
    if (tIdx < nhi) {
        const Fe yh = pow_table<F>(ytab, tIdx << LOB);
        Fe s = fe_one<F>(), r = yh;
        for (u32 j = 0; j < HIB; j++) {
            const Fe q = load_fe_dev<F>(usq + (size_t)(k - 1 - (LOB + j)) * 8);
            if ((tIdx >> j) & 1) s = fe_mul<F>(s, q); else r = fe_mul<F>(r, q);
        }
        store_fe_limbs(T + (size_t)(nlo + tIdx) * VT_W, s);
        store_fe_limbs(T + (size_t)(nlo + nhi + nlo + tIdx) * VT_W, r);
        store_fe_limbs(T + (size_t)(2 * (nlo + nhi) + nlo + tIdx) * VT_W, yh);
    }
}

// grid (ceil(N/256), nchunks).  g_part/h_part: [nchunks][N] resident words; d_part: [nchunks * gridDim.x] resident words.
// Instruction diet (the kernel is VALU-issue-bound: profiles/r02_sq_vfy_batch_after.txt, ~80 % of the SIMD cycles issue): sums are kept
// LAZY — limb-wise additions, one carry pass / weak reduction per group of terms instead of per term —, the proof's constants come
// in limb form from the tables, and both alpha*g and alpha*h end in ONE fused product pair (fe_mul2: a single Montgomery reduction):
//   alpha*g = (alpha*x*y^-i)*w_R + (-alpha*a*s[i]),   alpha*h = M + [(alpha*y^-i)*w_O + (-alpha*y^-i*b*s[N-1-i])] - alpha,
//   M = (alpha*x*y^-i)*w_L, which the delta term reuses: (r*x) * M * w_R.
// Latency diet (SQ counters: a fifth of the wave cycles waited on memory, and the waits sit in the dependent chain of the column loop —
// entry word -> table addresses -> table entries -> product): (i) the lane's column entries are the same for every proof of the
// chunk, so their first VFY_EMAX (entry, coefficient id) pairs are read ONCE into LDS (and the lane's first constant term into
// registers); (ii) the column loop is software-pipelined: the table entries (and the coefficient) of entry x + 1 are requested
// before the products of entry x, the operands of the constant term at the top of the proof.  The kernel then needs ~230 VGPRs = two
// waves per SIMD, which measured faster than three waves without the prefetches.  Staging the proof's z tables in LDS (34 KB per
// workgroup and proof, two barriers) was measured too and lost: 1.18 ms against 1.10 (profiles/r03_vfy_batch_ab.txt).
static constexpr u32 VFY_EMAX = 6;
template <class C> __global__ void __launch_bounds__(256)
k_vfy_batch(VfyTemplateDev t, const u32* __restrict__ params, const u32* __restrict__ perm, const u32* __restrict__ coef_tabs, u32 coef_stride,
            u32 P, u32 per_chunk, u32 n, u32 n1, u32 N, u32 k, u32* __restrict__ g_part,
            u32* __restrict__ h_part, u32* __restrict__ d_part, const u32* __restrict__ tables, u32 LOB, u32 nzhi) {
    typedef typename C::Fr F;
    __shared__ u32 sh[9 * 256];
    __shared__ u32 es[2 * VFY_EMAX * 256];
    (void)params; (void)perm;
    const u32 tid = threadIdx.x;
    const u32 i = blockIdx.x * blockDim.x + tid;
    const u32 chunk = blockIdx.y;
    const u32 p0 = chunk * per_chunk, p1 = min(P, p0 + per_chunk);
    u32 e0 = 0, ne = 0;
    if (i < n) { e0 = t.m_off[i]; ne = t.m_off[i + 1] - e0; }
    for (u32 x = 0; x < min(ne, VFY_EMAX); x++) { es[x * 256u + tid] = t.m_ent[e0 + x]; es[(VFY_EMAX + x) * 256u + tid] = t.m_c[e0 + x]; }   // (read back by this lane only)
    // lanes share the constant terms (e = i, i + N, ..): the first one of this lane
    const bool has_c0 = i < t.n_const;
    u32 c0_cid = 0x80000000u, c0_q1 = 1;
    if (has_c0) { c0_cid = t.const_c[i]; c0_q1 = t.const_q[i] + 1u; }   // (a constant per gate is common: the range-proof gadget has one per bit)
    // accumulators over the chunk's proofs: limb-wise sums of L = 1 values, carried and weakly reduced every fourth proof
    // (a term has V <= 6: four of them on top of a reduced value stay below the weak reduction's V < 32)
    Fe ag = fe_zero<F>(), ah = fe_zero<F>(), ad = fe_zero<F>();
    u32 pending = 0;
    if (i < N) {
        const u32 nlo = 1u << LOB, nhi = 1u << (k - LOB), lo = i & (nlo - 1u), hi = i >> LOB;
        const size_t stride = vfy_tab_stride(nlo, nhi, nzhi) * VT_W;
        for (u32 p = p0; p < p1; p++) {
            // coefficient values: the template's own table, or this proof's (instances of one gadget differ in public constants
            // and in the challenges their randomized constraints carry)
            const u32* coefs = coef_tabs ? coef_tabs + (size_t)p * coef_stride : t.coefs;
            const u32* T = tables + (size_t)p * stride;
            const u32* Z = T + ((size_t)3 * (nlo + nhi) + nlo) * VT_W;   // z^e = Z[e & 255] * Z[256 + (e >> 8)]
            const u32* K = Z + (size_t)(256 + nzhi + 256) * VT_W;        // x, alpha, u, r*x
            // next column entry in flight: its words, the two table entries of its power of z, its coefficient.  The loads are
            // UNCONDITIONAL (a unit coefficient reads entry 0 of the table and ignores it; the last entry of a column is requested
            // twice): a conditional load merges with the old value in a register copy right behind the load, and that copy waits
            // for the data in front of the products it was issued early to hide behind (seen in the ISA: `s_waitcnt vmcnt(1)`
            // directly after the loads; 66 % of the issue slots used).  Entries beyond the VFY_EMAX that live in LDS take the plain
            // loop at the end.
            const u32 ne_fast = min(ne, VFY_EMAX);
            u32 ent_n = 0, cid_n = 0x80000000u, cw_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            Fe zl_n = fe_zero<F>(), zh_n = fe_zero<F>();
            auto fetch = [&](u32 x) {   // x < ne_fast
                ent_n = es[x * 256u + tid];
                cid_n = es[(VFY_EMAX + x) * 256u + tid];
                const u32 q = (ent_n & 0x3fffffffu) + 1u;
                zl_n = load_fe_limbs(Z + (size_t)(q & 255u) * VT_W);
                zh_n = load_fe_limbs(Z + (size_t)(256u + (q >> 8)) * VT_W);
                load_words8(cw_n, coefs + (size_t)((cid_n & 0xc0000000u) ? 0u : cid_n) * 8);
            };
            if (ne_fast) fetch(0);
            // wc enters the B scalar with the same weight as delta: r * x^2 * (wc + delta) (verifier.rs:529); lanes share the terms;
            // wcp collects alpha * x * (their sum).  Computed first: its table reads share the round trip of the reads below, and
            // alpha * x * z^(q+1) = zx_lo[e & 255] * z_hi[e >> 8] is one product
            Fe wcp = fe_zero<F>();
            if (has_c0) {
                const Fe zq0 = fe_mul<F>(load_fe_limbs(Z + (size_t)(256u + nzhi + (c0_q1 & 255u)) * VT_W), load_fe_limbs(Z + (size_t)(256u + (c0_q1 >> 8)) * VT_W));
                if (c0_cid & 0x80000000u) wcp = zq0;
                else if (c0_cid & 0x40000000u) wcp = fe_wred<F>(fe_neg<F, 2>(zq0));
                else wcp = fe_mul<F>(zq0, load_fe_dev<F>(coefs + (size_t)c0_cid * 8));
                for (u32 e = i + N; e < t.n_const; e += N) {   // (more constant terms than lanes: rare)
                    const u32 cid = t.const_c[e];
                    const u32 q1 = t.const_q[e] + 1u;
                    const Fe zq = fe_mul<F>(load_fe_limbs(Z + (size_t)(256u + nzhi + (q1 & 255u)) * VT_W), load_fe_limbs(Z + (size_t)(256u + (q1 >> 8)) * VT_W));
                    Fe term;
                    if (cid & 0x80000000u) term = zq;
                    else if (cid & 0x40000000u) term = fe_wred<F>(fe_neg<F, 2>(zq));
                    else term = fe_mul<F>(zq, load_fe_dev<F>(coefs + (size_t)cid * 8));
                    wcp = fe_addr<F>(wcp, term);
                }
            }
            const Fe alpha = load_fe_limbs(K + VT_W);
            // the proof's constants ride in the low halves of the split tables (k_vfy_tables): one product each
            const Fe ns_lo = load_fe_limbs(T + (size_t)lo * VT_W), s_hi = load_fe_limbs(T + (size_t)(nlo + hi) * VT_W);                                   // product: -alpha * a * s[i]
            const Fe nr_lo = load_fe_limbs(T + (size_t)(nlo + nhi + lo) * VT_W), r_hi = load_fe_limbs(T + (size_t)(nlo + nhi + nlo + hi) * VT_W);         // product: -alpha * y^-i * b * s[N-1-i]
            Fe g, h, dl = fe_zero<F>();
            if (i < n) {
                const Fe yhi = load_fe_limbs(T + (size_t)(2 * (nlo + nhi) + nlo + hi) * VT_W);
                const Fe YA = fe_mul<F>(load_fe_limbs(T + (size_t)(3 * (nlo + nhi) + lo) * VT_W), yhi);                                                // alpha * y^-i
                const Fe YX = fe_mul<F>(load_fe_limbs(T + (size_t)(2 * (nlo + nhi) + lo) * VT_W), yhi);                                                // alpha * x * y^-i
                // columns i of W_L, W_R, W_O in one pass over their entries sorted by constraint index (equal indices reuse the
                // power); z^(q+1) is one product of two entries of the proof's split table.  The three sums are lazy: a carry pass
                // and weak reduction after every sixth entry (limbs of <= 7 summed L = 1 terms fit 32 bits).
                Fe wL = fe_zero<F>(), wR = fe_zero<F>(), wO = fe_zero<F>(), zp = fe_one<F>();
                u32 cur = 0, since = 0;   // exponent zp holds (0 = none yet; q + 1 >= 1 always)
                // one column entry: bit 31 of cid: the coefficient is +1, bit 30: it is -1 (most gadget constraints): no product.  exp_z for
                // constraint q is z^(q+1) (verifier.rs:323-345).  (A macro, not a lambda: with the operands passed by reference the field
                // elements went through scratch memory.)
#define ARKBP_VFY_CONSUME(ent, cid, zl, zh, cw)                                                                                              \
    {                                                                                                                                        \
        const u32 q1 = ((ent) & 0x3fffffffu) + 1u, vec = (ent) >> 30;                                                                         \
        if (q1 != cur) zp = fe_mul<F>(zl, zh);                                                                                               \
        cur = q1;                                                                                                                            \
        Fe term;                                                                                                                             \
        if ((cid) & 0x80000000u) term = zp;                                                                                                  \
        else if ((cid) & 0x40000000u) term = fe_neg<F, 2>(zp);                                                                               \
        else term = fe_mul<F>(zp, fe_unpack(cw));                                                                                            \
        if (vec == 0) wL = fe_add(wL, term); else if (vec == 1) wR = fe_add(wR, term); else wO = fe_add(wO, term);                           \
        if (++since == 6) { wL = fe_wred<F>(fe_norm(wL)); wR = fe_wred<F>(fe_norm(wR)); wO = fe_wred<F>(fe_norm(wO)); since = 0; }           \
    }
                for (u32 x = 0; x < ne_fast; x++) {
                    const u32 ent = ent_n, cid = cid_n;
                    const Fe zl = zl_n, zh = zh_n;
                    u32 cw[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) cw[j] = cw_n[j];
                    fetch(min(x + 1u, ne_fast - 1u));
                    ARKBP_VFY_CONSUME(ent, cid, zl, zh, cw)
                }
#pragma unroll 1
                for (u32 x = ne_fast; x < ne; x++) {   // (a column with more than VFY_EMAX entries: rare)
                    const u32 ent = t.m_ent[e0 + x], cid = t.m_c[e0 + x];
                    const u32 q = (ent & 0x3fffffffu) + 1u;
                    const Fe zl = load_fe_limbs(Z + (size_t)(q & 255u) * VT_W), zh = load_fe_limbs(Z + (size_t)(256u + (q >> 8)) * VT_W);
                    u32 cw[8];
                    load_words8(cw, coefs + (size_t)((cid & 0xc0000000u) ? 0u : cid) * 8);
                    ARKBP_VFY_CONSUME(ent, cid, zl, zh, cw)
                }
#undef ARKBP_VFY_CONSUME
                wL = fe_norm(wL); wR = fe_norm(wR); wO = fe_norm(wO);   // L = 1, V <= 2 + 6 * 3
                // alpha * g = alpha*x*y^-i*wR - alpha*a*s[i];  alpha * h = alpha*y^-i * (x*wL + wO - b*s[N-1-i]) - alpha;
                // alpha*r*x^2 * y^-i*wR*wL = (r*x) * (alpha*x*y^-i*wL) * wR   (verifier.rs:477-500, weighted by the batch's alpha)
                g = fe_mul2<F>(YX, wR, ns_lo, s_hi);
                const Fe M = fe_mul<F>(YX, wL);
                h = fe_sub<F, 2>(fe_add(fe_mul2<F>(YA, wO, nr_lo, r_hi), M), alpha);   // (V <= 2.3 + 1.7 + 2)
                dl = fe_mul<F>(M, wR);
            } else {
                g = fe_mul<F>(ns_lo, s_hi);
                h = fe_sub<F, 2>(fe_mul<F>(nr_lo, r_hi), alpha);
            }
            // u_or_1 = 1 for the phase-1 multipliers, u for the randomized-phase ones and on the padding (verifier.rs:486-489)
            if (i >= n1) {
                const Fe u = load_fe_limbs(K + 2 * VT_W);
                g = fe_mul<F>(g, u);
                h = fe_mul<F>(h, u);
            }
            ag = fe_add(ag, g);
            ah = fe_add(ah, h);
            // alpha*r*x^2 * (y^-i*wR*wL - wc terms) = (r*x) * (alpha*x*y^-i*wR*wL - alpha*x*wc): one product for both
            if (i < n || has_c0) {
                Fe tsum;
                if (i < n && has_c0) tsum = fe_sub<F, 4>(dl, wcp);
                else if (i < n) tsum = dl;
                else tsum = fe_neg<F, 4>(wcp);
                ad = fe_add(ad, fe_mul<F>(load_fe_limbs(K + 3 * VT_W), tsum));
            }
            if (++pending == 4) { ag = fe_wred<F>(fe_norm(ag)); ah = fe_wred<F>(fe_norm(ah)); ad = fe_wred<F>(fe_norm(ad)); pending = 0; }
        }
        ag = fe_wred<F>(fe_norm(ag)); ah = fe_wred<F>(fe_norm(ah)); ad = fe_wred<F>(fe_norm(ad));
        store_fe_dev<F>(g_part + ((size_t)chunk * N + i) * 8, ag);
        store_fe_dev<F>(h_part + ((size_t)chunk * N + i) * 8, ah);
    }
    const Fe dsum = block_sum_fe<F>(ad, sh);
    if (threadIdx.x == 0) store_fe_dev<F>(d_part + ((size_t)chunk * gridDim.x + blockIdx.x) * 8, dsum);
}
// acc_g[i] += sum_c g_part[c][i] (same for h); acc_* in resident form, i < N
template <class C> __global__ void __launch_bounds__(256)
k_vfy_batch_fold(const u32* __restrict__ g_part, const u32* __restrict__ h_part, u32 nchunks, u32 N, u32* __restrict__ acc_g, u32* __restrict__ acc_h) {
    typedef typename C::Fr F;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Fe g = load_fe_dev<F>(acc_g + (size_t)i * 8), h = load_fe_dev<F>(acc_h + (size_t)i * 8);
    for (u32 c = 0; c < nchunks; c++) {
        g = fe_addr<F>(g, load_fe_dev<F>(g_part + ((size_t)c * N + i) * 8));
        h = fe_addr<F>(h, load_fe_dev<F>(h_part + ((size_t)c * N + i) * 8));
    }
    store_fe_dev<F>(acc_g + (size_t)i * 8, g);
    store_fe_dev<F>(acc_h + (size_t)i * 8, h);
}

// InnerProductProof::verify scalars (src/inner_product_proof.rs:338-352): g[i] = a * s[i] * G_factors[i],
// h[i] = b * s[n-1-i] * H_factors[i] as canonical integers.  consts: resident words [allinv, a, b]; chal: u_sq[k].
template <class C> __global__ void __launch_bounds__(256)
k_ipa_vfy_scalars(const u32* __restrict__ Gf, const u32* __restrict__ Hf, const u32* __restrict__ chal, const u32* __restrict__ consts, u32 n, u32 k,
                  u32* __restrict__ g_out, u32* __restrict__ h_out) {
    typedef typename C::Fr F;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fe allinv = load_fe_dev<F>(consts), a = load_fe_dev<F>(consts + 8), b = load_fe_dev<F>(consts + 16);
    Fe s_i = allinv, s_rev = allinv;
#pragma unroll 1
    for (u32 j = 0; j < k; j++) {
        const Fe usq = load_fe_dev<F>(chal + (size_t)(k - 1 - j) * 8);
        if ((i >> j) & 1) s_i = fe_mul<F>(s_i, usq); else s_rev = fe_mul<F>(s_rev, usq);
    }
    const size_t o = (size_t)i * 8;
    store_fe_canon<F>(g_out + o, fe_mul<F>(fe_mul<F>(a, s_i), load_fe_dev<F>(Gf + o)));
    store_fe_canon<F>(h_out + o, fe_mul<F>(fe_mul<F>(b, s_rev), load_fe_dev<F>(Hf + o)));
}

// tail scalars of the batched mega-check (verifier.rs:652-683): s <- canonical(alpha_p * s) for every tail scalar of proof p.
// tails: ark words in, canonical integers out; alphas: resident form; toff: P+1 prefix offsets of the per-proof tail ranges.
template <class F> __global__ void __launch_bounds__(256)
k_vfy_tail_scale(u32* __restrict__ tails, const u32* __restrict__ alphas, const u32* __restrict__ toff, u32 P, u32 T) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= T) return;
    u32 lo = 0, hi = P;   // toff[lo] <= j < toff[hi]
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (toff[mid] <= j) lo = mid; else hi = mid; }
    u32 w[8];
    load_words8(w, tails + (size_t)j * 8);
    const Fe s = fe_load_ark<F>(w), a = load_fe_dev<F>(alphas + (size_t)lo * 8);
    fe_store_canon<F>(w, fe_mul<F>(s, a));
    store_words8(tails + (size_t)j * 8, w);
}

// ---- prover-side flattened_constraints (src/r1cs/prover.rs:354-397) for single-phase statements ----------------------------
// The recorded constraints are indexed once, at statement construction, as the merged CSC the verifier templates use (per
// multiplier index i: the entries of columns i of W_L, W_R, W_O in constraint order); after the challenge z is known one launch
// evaluates w_L, w_R, w_O = z^(q+1)-weighted column sums on the GPU instead of the host's pass over all terms + 3 uploads.
// Z: 256 + nzhi resident scalars, z^e = Z[e & 255] * Z[256 + (e >> 8)].
template <class C> __global__ void __launch_bounds__(256)
k_r1cs_ztables(const u32* __restrict__ ztab /* z^(2^j), 32 resident scalars */, u32 nzhi, u32* __restrict__ Z) {
    typedef typename C::Fr F;
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 256) store_fe_dev<F>(Z + (size_t)t * 8, pow_table<F>(ztab, t));
    if (t < nzhi) store_fe_dev<F>(Z + (size_t)(256 + t) * 8, pow_table<F>(ztab, t << 8));
}
template <class C> __global__ void __launch_bounds__(256)
k_r1cs_flatten(const u32* __restrict__ m_off, const u32* __restrict__ m_ent, const u32* __restrict__ m_c, const u32* __restrict__ coefs,
               const u32* __restrict__ Z, u32 n, u32* __restrict__ wL_out, u32* __restrict__ wR_out, u32* __restrict__ wO_out) {
    typedef typename C::Fr F;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe wL = fe_zero<F>(), wR = fe_zero<F>(), wO = fe_zero<F>(), zp = fe_one<F>();
    u32 cur = 0;
    for (u32 e = m_off[i], e1 = m_off[i + 1]; e < e1; e++) {
        const u32 ent = m_ent[e], q1 = (ent & 0x3fffffffu) + 1u, vec = ent >> 30;   // exp_z of constraint q is z^(q+1) (prover.rs:368-390)
        if (q1 != cur) zp = fe_mul<F>(load_fe_dev<F>(Z + (size_t)(q1 & 255u) * 8), load_fe_dev<F>(Z + (size_t)(256u + (q1 >> 8)) * 8));
        cur = q1;
        const u32 cid = m_c[e];
        Fe term;
        if (cid & 0x80000000u) term = zp;
        else if (cid & 0x40000000u) term = fe_wred<F>(fe_neg<F, 2>(zp));
        else term = fe_mul<F>(zp, load_fe_dev<F>(coefs + (size_t)cid * 8));
        if (vec == 0) wL = fe_addr<F>(wL, term); else if (vec == 1) wR = fe_addr<F>(wR, term); else wO = fe_addr<F>(wO, term);
    }
    store_fe_dev<F>(wL_out + (size_t)i * 8, wL);
    store_fe_dev<F>(wR_out + (size_t)i * 8, wR);
    store_fe_dev<F>(wO_out + (size_t)i * 8, wO);
}

// resident form -> canonical integers, in place
template <class F> __global__ void k_scalars_to_canon(u32* __restrict__ v, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_fe_canon<F>(v + (size_t)i * 8, load_fe_dev<F>(v + (size_t)i * 8));
}

}  // namespace arkbp
