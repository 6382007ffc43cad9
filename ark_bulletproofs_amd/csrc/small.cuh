// Direct window tables of the first generators — the small-statement path of the prover.
//
// The reference's own benchmark sweeps k-shuffles with 2 ... 2046 multipliers (benches/r1cs_secq256k1.rs:152-250).  At these sizes
// every MSM of `Prover::prove` (src/r1cs/prover.rs:516-649) and of `InnerProductProof::create`
// (src/inner_product_proof.rs:86-131, 174-213) is over a few thousand FIXED bases at most, and what the proof costs is latency:
// a 256-bit ladder is ~1 ms on one lane however few points there are, a bucket pipeline is five launches.  With
//   T[base][w][d - 1] = d * 16^w * base      (w < 64, 1 <= d <= 15; affine, resident layout; 60 KiB per base)
// a term's contribution is the sum of <= 64 table entries, so an MSM over n such bases is a plain SUM of <= 64 n affine points:
// no doublings, no buckets, no sort.  k_dt_accum gives every lane a few (term, window) pairs (mixed additions), then one
// quad-cooperative tree per workgroup; k_dt_finish adds the workgroups' partial points.  Several MSMs (the three commitments of a
// phase; L and R of a round) share one launch.
//
// The inner-product argument on top never folds G and H (ipa.cuh "frozen-generator tail", here from round 1): a round's L and R are
// MSMs over ALL the original generators with the per-element coefficients, which is exactly the shape above.  The results are group
// elements, so L, R (and everything else) are bit-identical to the folding schedule's.
#pragma once
#include "ipa.cuh"
#include "ecq.cuh"

namespace arkbp {

static constexpr u32 DT_WINDOWS = 64, DT_ENT = 15;
static constexpr u32 DT_PER_BASE = DT_WINDOWS * DT_ENT;            // entries (64 B each) per base
static constexpr size_t DT_BASE_BYTES = (size_t)DT_PER_BASE * 64;

// ---- building the tables ---------------------------------------------------------------------------------------------------------
// step 1, one lane per base: wb[base][w] = 16^w * base (a chain of 4 doublings per window), Jacobian workspace
template <class C> __global__ void __launch_bounds__(64)
k_dt_window_bases(const u32* __restrict__ bases /* nb x 16 words, resident affine */, u32 nb, u32* __restrict__ ws /* nb x 64 x 24 words */) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    Jac acc = jac_from_aff<C>(load_aff_dev(bases + (size_t)i * 16));
#pragma unroll 1
    for (u32 w = 0; w < DT_WINDOWS; w++) {
        store_jac_ws<C>(ws + ((size_t)i * DT_WINDOWS + w) * 24, acc);
#pragma unroll 1
        for (int j = 0; j < 4; j++) acc = jac_dbl<C>(acc);
    }
}
// step 2, one lane per entry: d * wb[base][w] (a 4-bit ladder), normalised in-lane
template <class C> __global__ void __launch_bounds__(256)
k_dt_entries(const u32* __restrict__ ws, u32 nb, u32* __restrict__ tab) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)nb * DT_PER_BASE) return;
    const u32 d = (u32)(t % DT_ENT) + 1u;
    const size_t bw = t / DT_ENT;   // base * 64 + w
    const Jac P = load_jac_ws(ws + bw * 24);
    Jac acc = jac_inf<C>();
#pragma unroll 1
    for (int bit = 3; bit >= 0; bit--) {
        acc = jac_dbl<C>(acc);
        if ((d >> bit) & 1) acc = jac_add<C>(acc, P);
    }
    const Aff o = jac_to_aff<C>(acc);
    u32 wd[16];
    aff_store_dev(wd, o);
    store_words8(tab + t * 16, wd);
    store_words8(tab + t * 16 + 8, wd + 8);
}

// ---- sums over the tables --------------------------------------------------------------------------------------------------------
// One MSM = up to DT_MAXSEG runs of (scalars, consecutive bases) + one immediate term whose scalar comes with the launch (a
// commitment's blinding factor: no copy).  Scalars: 8 words each, canonical integers or the resident form of C::Fr.
// A run may take every other block of `fold_n` elements of its vectors: term j stands for element
//   t = (j / fold_n) * 2 * fold_n + (j % fold_n) + (fold_hi ? fold_n : 0)
// — a frozen round of the inner-product argument pairs G with one half of every period of the current length and H with the other
// (ipa.cuh k_ipa_frozen_scalars); the elements of the other half have zero scalars and are not even visited.
static constexpr int DT_MAXSEG = 3, DT_MAXOUT = 6;
struct DtSeg {
    const u32* sc;
    u32 base0, count, resident;   // resident: 0 = canonical integers, 1 = the resident form of C::Fr, 2 = ark Montgomery words
    u32 fold_n, fold_hi;
};
struct DtJob {
    DtSeg seg[DT_MAXSEG];
    u32 nseg, terms;        // terms = sum of the counts (+ 1 with an immediate term, which is term 0)
    u32 has_imm, imm_base;
    u32 imm[8];             // canonical words
};
struct DtJobs {
    DtJob job[DT_MAXOUT];
};

// Jacobian point as ark Montgomery words (3 x 8; Z = 0 words for the identity): what the host's J4 reads
template <class C> __device__ __forceinline__ void store_jac_ark(u32* __restrict__ o, const Jac& a) {
    typedef typename C::Fq F;
    u32 wd[8];
    const bool inf = jac_is_inf(a);
    fe_store_ark<F>(wd, a.X); store_words8(o, wd);
    fe_store_ark<F>(wd, a.Y); store_words8(o + 8, wd);
    if (inf) { for (int i = 0; i < 8; i++) wd[i] = 0; } else fe_store_ark<F>(wd, a.Z);
    store_words8(o + 16, wd);
}
template <class C> __device__ __forceinline__ Jac load_jac_ark(const u32* __restrict__ p) {
    typedef typename C::Fq F;
    u32 w[8];
    Jac r;
    load_words8(w, p + 16);
    bool z = true;
#pragma unroll
    for (int q = 0; q < 8; q++) z = z && w[q] == 0;
    if (z) return jac_inf<C>();
    r.Z = fe_load_ark<F>(w);
    load_words8(w, p); r.X = fe_load_ark<F>(w);
    load_words8(w, p + 8); r.Y = fe_load_ark<F>(w);
    return r;
}

// Everything here is latency: a lone lane needs ~8 us per point addition.  So FOUR lanes share every addition (ecq.cuh: one modular
// product per lane and dependency level, ~2.5-3 us) — a quad walks its (term, window) pairs with quad-cooperative mixed additions,
// the next table entry in flight meanwhile, and the 64 quads of a workgroup meet in a 6-level tree of quad-cooperative additions.
// sh: 64 x 27 words.  Result valid in every lane of quad 0.
template <class C> __device__ __forceinline__ Jac dt_quad_tree(const Jac& acc, u32* __restrict__ sh) {
    const u32 q = threadIdx.x & 3u, quad = threadIdx.x >> 2;
    if (q == 0) lds_put_jac(sh, 64, quad, acc);
    __syncthreads();
#pragma unroll 1
    for (u32 half = 32; half >= 1; half >>= 1) {
        // quad j < half: slot j += slot j + half (no quad of the level reads a slot another one writes)
        if (quad < half) {
            const Jac r = qjac_add<C>(lds_get_jac(sh, 64, quad), lds_get_jac(sh, 64, quad + half), q);
            if (q == 0) lds_put_jac(sh, 64, quad, r);
        }
        __syncthreads();
    }
    return lds_get_jac(sh, 64, 0);
}
// grid (nblk, nout), 256 lanes = 64 quads.  A quad takes UNITS of four consecutive windows of one term (half a scalar word: one
// 4-byte load, or one canonicalisation, per four additions; the four table entries are requested together and arrive under the
// first addition).  The job's run descriptors are read ONCE into scalar registers — indexing the kernel-argument array per pair
// costs a chain of vector loads (~3 us) in front of every addition.  out: [nout][nblk] points (ark words); with nblk == 1 these
// are the results.
static constexpr u32 DT_UNIT = 4, DT_UNITS_PER_TERM = DT_WINDOWS / DT_UNIT;
static_assert(DT_UNIT == 4, "k_dt_accum writes its four additions out");
template <class C> __global__ void __launch_bounds__(256)
k_dt_accum(const u32* __restrict__ tab, DtJobs jobs, u32* __restrict__ out) {
    typedef typename C::Fr Fr;
    __shared__ u32 sh[64 * 27];
    const DtJob& jb = jobs.job[blockIdx.y];
    const u32 has_imm = jb.has_imm, imm_base = jb.imm_base, nseg = jb.nseg;
    const u32 units = jb.terms * DT_UNITS_PER_TERM;
    const u32* sc_[DT_MAXSEG];
    u32 base0_[DT_MAXSEG], count_[DT_MAXSEG], res_[DT_MAXSEG], fn_[DT_MAXSEG], fh_[DT_MAXSEG];
#pragma unroll
    for (int i = 0; i < DT_MAXSEG; i++) {
        sc_[i] = jb.seg[i].sc; base0_[i] = jb.seg[i].base0; count_[i] = jb.seg[i].count; res_[i] = jb.seg[i].resident; fn_[i] = jb.seg[i].fold_n; fh_[i] = jb.seg[i].fold_hi;
    }
    const u32 q = threadIdx.x & 3u, nquads = gridDim.x * 64u;
    Jac acc = jac_inf<C>();
#pragma unroll 1
    for (u32 un = blockIdx.x * 64u + (threadIdx.x >> 2); un < units; un += nquads) {
        u32 term = un / DT_UNITS_PER_TERM;
        const u32 w0 = (un % DT_UNITS_PER_TERM) * DT_UNIT;   // windows w0 .. w0 + 3: one half of word w0 / 8
        u32 word, base;
        if (has_imm && term == 0) {
            word = jb.imm[0];
#pragma unroll
            for (int j = 1; j < 8; j++) word = (w0 >> 3) == (u32)j ? jb.imm[j] : word;
            base = imm_base;
        } else {
            term -= has_imm;
            int s = 0;
            if (nseg > 1 && term >= count_[0]) { term -= count_[0]; s = 1; if (nseg > 2 && term >= count_[1]) { term -= count_[1]; s = 2; } }
            const u32* scp = s == 0 ? sc_[0] : s == 1 ? sc_[1] : sc_[2];
            const u32 b0 = s == 0 ? base0_[0] : s == 1 ? base0_[1] : base0_[2], res = s == 0 ? res_[0] : s == 1 ? res_[1] : res_[2];
            const u32 fn = s == 0 ? fn_[0] : s == 1 ? fn_[1] : fn_[2], fh = s == 0 ? fh_[0] : s == 1 ? fh_[1] : fh_[2];
            if (fn) term = (term / fn) * 2u * fn + (term % fn) + (fh ? fn : 0u);
            base = b0 + term;
            if (res) {
                u32 k[8];
                load_words8(k, scp + (size_t)term * 8);
                if (res == 1) fe_store_canon<Fr>(k, fe_unpack(k)); else fe_store_canon<Fr>(k, fe_load_ark<Fr>(k));
                word = k[0];
#pragma unroll
                for (int j = 1; j < 8; j++) word = (w0 >> 3) == (u32)j ? k[j] : word;
            } else {
                word = scp[(size_t)term * 8 + (w0 >> 3)];
            }
        }
        const u32 dig = (word >> (4u * (w0 & 7u))) & 0xffffu;   // four digits
        const u32* T = tab + ((size_t)base * DT_WINDOWS + w0) * DT_ENT * 16;
        // (written out: as a loop the four additions stay rolled and the entries go through scratch memory)
        const u32 d0 = dig & 15u, d1 = (dig >> 4) & 15u, d2 = (dig >> 8) & 15u, d3 = (dig >> 12) & 15u;
        const Aff p0 = load_aff_dev(T + ((size_t)0 * DT_ENT + (d0 ? d0 - 1u : 0u)) * 16);   // (digit 0: entry 1 is read and not used)
        const Aff p1 = load_aff_dev(T + ((size_t)1 * DT_ENT + (d1 ? d1 - 1u : 0u)) * 16);
        const Aff p2 = load_aff_dev(T + ((size_t)2 * DT_ENT + (d2 ? d2 - 1u : 0u)) * 16);
        const Aff p3 = load_aff_dev(T + ((size_t)3 * DT_ENT + (d3 ? d3 - 1u : 0u)) * 16);
        if (d0) acc = qjac_madd<C>(acc, p0, q);   // (quad-uniform: the four lanes hold the same digits)
        if (d1) acc = qjac_madd<C>(acc, p1, q);
        if (d2) acc = qjac_madd<C>(acc, p2, q);
        if (d3) acc = qjac_madd<C>(acc, p3, q);
    }
    acc = dt_quad_tree<C>(acc, sh);
    if (threadIdx.x == 0) store_jac_ark<C>(out + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 24, acc);
}
// grid (nout): sum of the nblk partial points of output blockIdx.x -> res[blockIdx.x]
template <class C> __global__ void __launch_bounds__(256)
k_dt_finish(const u32* __restrict__ part, u32 nblk, u32* __restrict__ res) {
    __shared__ u32 sh[64 * 27];
    const u32 q = threadIdx.x & 3u;
    Jac acc = jac_inf<C>();
#pragma unroll 1
    for (u32 j = threadIdx.x >> 2; j < nblk; j += 64) acc = qjac_add<C>(acc, load_jac_ark<C>(part + ((size_t)blockIdx.x * nblk + j) * 24), q);
    acc = dt_quad_tree<C>(acc, sh);
    if (threadIdx.x == 0) store_jac_ark<C>(res + (size_t)blockIdx.x * 24, acc);
}

// One launch per round of the inner-product argument over the direct tables, in front of the table sums: the fold the previous
// challenge asks for (a, b: src/inner_product_proof.rs:140-141 / 217-218, into the other buffer pair; the per-element coefficients
// of G and H in place), this round's MSM scalars and its two inner products c_L = <a_L, b_R>, c_R = <a_R, b_L> (:83-84, :171-172)
// — what k_ipa_fold_ab, k_ipa_frozen_fold, k_ipa_frozen_scalars and k_ipa_ip_finish do in four launches.  n = half of the
// current length (after the fold), n0 = number of generators.  sL / sR as in k_ipa_frozen_scalars, but only the half of the
// entries that the round's sums visit is written (DtSeg::fold_n); slots 2 n0 / 2 n0 + 1: c and c * qw.
// The workgroups meet through a ticket counter (zero between launches): the last one to arrive adds the partial inner products.
template <class C> __global__ void __launch_bounds__(256)
k_dt_round(const u32* __restrict__ a_in, const u32* __restrict__ b_in, u32* __restrict__ a_out, u32* __restrict__ b_out, u32* __restrict__ cG, u32* __restrict__ cH,
           u32 n, u32 n0, int do_fold, Words8 uw, Words8 uiw, u32* __restrict__ sL, u32* __restrict__ sR, u32* __restrict__ partials, u32* __restrict__ counter,
           Words8 qw) {
    typedef typename C::Fr F;
    __shared__ u32 sh[9 * 256];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    Fe u = fe_zero<F>(), ui = fe_zero<F>();
    if (do_fold) { u = fe_load_ark<F>(uw.w); ui = fe_load_ark<F>(uiw.w); }
    // element i of the current a / b (length 2n): folded on the fly from the vectors of twice the length
    auto A = [&](u32 i) -> Fe {
        const Fe lo = load_fe_dev<F>(a_in + (size_t)i * 8);
        if (!do_fold) return lo;
        return fe_norm(fe_add(fe_mul<F>(lo, u), fe_mul<F>(ui, load_fe_dev<F>(a_in + (size_t)(2 * n + i) * 8))));
    };
    auto B = [&](u32 i) -> Fe {
        const Fe lo = load_fe_dev<F>(b_in + (size_t)i * 8);
        if (!do_fold) return lo;
        return fe_norm(fe_add(fe_mul<F>(lo, ui), fe_mul<F>(u, load_fe_dev<F>(b_in + (size_t)(2 * n + i) * 8))));
    };
    Fe pl = fe_zero<F>(), pr = fe_zero<F>();
    if (t < n0) {
        Fe g = load_fe_dev<F>(cG + (size_t)t * 8), h = load_fe_dev<F>(cH + (size_t)t * 8);
        if (do_fold) {   // the coefficients follow the fold of length 4n -> 2n (k_ipa_frozen_fold)
            const bool lo_prev = (t & (4 * n - 1)) < 2 * n;
            g = fe_mul<F>(g, lo_prev ? ui : u);
            h = fe_mul<F>(h, lo_prev ? u : ui);
            store_fe_dev<F>(cG + (size_t)t * 8, g);
            store_fe_dev<F>(cH + (size_t)t * 8, h);
        }
        const u32 r = t & (2 * n - 1);
        const bool lo = r < n;
        const u32 idx = lo ? r + n : r - n;
        const Fe ai = A(idx), bi = B(idx);
        // G: L pairs a_L[j] with the upper half of a period, R pairs a_R[j] with the lower half; H the other way round
        store_fe_canon<F>((lo ? sR : sL) + (size_t)t * 8, fe_mul<F>(ai, g));
        store_fe_canon<F>((lo ? sL : sR) + (size_t)(n0 + t) * 8, fe_mul<F>(bi, h));
        if (t < 2 * n) {
            const Fe at = A(t), bt = B(t);
            if (do_fold) { store_fe_dev<F>(a_out + (size_t)t * 8, at); store_fe_dev<F>(b_out + (size_t)t * 8, bt); }
            if (t < n) { pl = fe_mul<F>(at, bi); pr = fe_mul<F>(ai, bt); }   // (t < n: idx = t + n)
        }
    }
    pl = block_sum_fe<F>(fe_wred<F>(pl), sh);
    pr = block_sum_fe<F>(fe_wred<F>(pr), sh);
    if (gridDim.x > 1) {
        if (threadIdx.x == 0) {
            store_fe_dev<F>(partials + (size_t)blockIdx.x * 16, pl);
            store_fe_dev<F>(partials + (size_t)blockIdx.x * 16 + 8, pr);
        }
        // publish, take a ticket; the last arriver acquires and reads every partial (one agent-scope release and one acquire per launch)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sh[0] = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const bool last = sh[0] == gridDim.x - 1u;
        __syncthreads();
        if (!last) return;
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        pl = fe_zero<F>(); pr = fe_zero<F>();
        for (u32 j = threadIdx.x; j < gridDim.x; j += 256) {
            pl = fe_addr<F>(pl, load_fe_dev<F>(partials + (size_t)j * 16));
            pr = fe_addr<F>(pr, load_fe_dev<F>(partials + (size_t)j * 16 + 8));
        }
        pl = block_sum_fe<F>(pl, sh);
        pr = block_sum_fe<F>(pr, sh);
        if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        u32* oL = sL + (size_t)2 * n0 * 8;
        u32* oR = sR + (size_t)2 * n0 * 8;
        store_fe_canon<F>(oL, pl);
        store_fe_canon<F>(oR, pr);
        const Fe qv = fe_load_ark<F>(qw.w);
        store_fe_canon<F>(oL + 8, fe_mul<F>(pl, qv));
        store_fe_canon<F>(oR + 8, fe_mul<F>(pr, qv));
    }
}

// Pedersen commitments of a statement's inputs (PedersenGens::commit, src/generators.rs:39-44; one per Prover::commit,
// src/r1cs/prover.rs:327-341) when there are few of them and the caller waits: ONE WAVE per commitment — its 16 quads walk the 128
// (base, window) pairs of v * B + blind * B_blinding, 8 each, and meet in a 4-level tree of cross-lane moves; the Jacobian result goes
// to the host, which normalises the whole batch with one inversion.  (k_pc_commit — one lane per commitment, 8-bit windows — is the
// throughput form for the 2^20 + 2 commitments of a wide statement: ~0.6 ms however few there are.)
// tab: direct window tables of [B, B_blinding]; v, blind: ark Montgomery words; out: m x 24 ark words (Z = 0: identity).
template <class C> __global__ void __launch_bounds__(256)
k_dt_commit(const u32* __restrict__ tab, const u32* __restrict__ v, const u32* __restrict__ blind, u32 m, u32* __restrict__ out) {
    typedef typename C::Fr Fr;
    const u32 lane = threadIdx.x & 63u, q = lane & 3u, quad = lane >> 2;
    const u32 i = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (i >= m) return;   // (wave-uniform)
    u32 kv[8], kb[8];
    load_words8(kv, v + (size_t)i * 8); fe_store_canon<Fr>(kv, fe_load_ark<Fr>(kv));
    load_words8(kb, blind + (size_t)i * 8); fe_store_canon<Fr>(kb, fe_load_ark<Fr>(kb));
    // this quad's windows: quad + 16 e, e < 4 — digit (quad & 7) of the words (quad >> 3) + 2 e — for both scalars.  Written out per
    // base (four entries requested together, four additions): a loop over them would index registers dynamically (scratch memory)
    auto word_of = [&](const u32 (&k)[8], u32 idx) { u32 r = k[0];
#pragma unroll
        for (int j = 1; j < 8; j++) r = idx == (u32)j ? k[j] : r;
        return r; };
    Jac acc = jac_inf<C>();
    const u32 wi = quad >> 3, sh4 = 4u * (quad & 7u);
#define ARKBP_DT_COMMIT_BASE(K, B)                                                                                                        \
    {                                                                                                                                     \
        const u32 d0 = (word_of(K, wi) >> sh4) & 15u, d1 = (word_of(K, wi + 2u) >> sh4) & 15u, d2 = (word_of(K, wi + 4u) >> sh4) & 15u,     \
                  d3 = (word_of(K, wi + 6u) >> sh4) & 15u;                                                                                \
        const u32* T = tab + ((size_t)(B) * DT_WINDOWS + quad) * DT_ENT * 16;                                                             \
        const Aff p0 = load_aff_dev(T + ((size_t)0 * 16 * DT_ENT + (d0 ? d0 - 1u : 0u)) * 16);                                            \
        const Aff p1 = load_aff_dev(T + ((size_t)1 * 16 * DT_ENT + (d1 ? d1 - 1u : 0u)) * 16);                                            \
        const Aff p2 = load_aff_dev(T + ((size_t)2 * 16 * DT_ENT + (d2 ? d2 - 1u : 0u)) * 16);                                            \
        const Aff p3 = load_aff_dev(T + ((size_t)3 * 16 * DT_ENT + (d3 ? d3 - 1u : 0u)) * 16);                                            \
        if (d0) acc = qjac_madd<C>(acc, p0, q);                                                                                           \
        if (d1) acc = qjac_madd<C>(acc, p1, q);                                                                                           \
        if (d2) acc = qjac_madd<C>(acc, p2, q);                                                                                           \
        if (d3) acc = qjac_madd<C>(acc, p3, q);                                                                                           \
    }
    ARKBP_DT_COMMIT_BASE(kv, 0u)
    ARKBP_DT_COMMIT_BASE(kb, 1u)
#undef ARKBP_DT_COMMIT_BASE
#pragma unroll 1
    for (int off = 32; off >= 4; off >>= 1) acc = qjac_add<C>(acc, jac_shfl_down(acc, off), q);   // (quad j += quad j + off / 4; lanes past the live range compute unused sums)
    if (lane == 0) store_jac_ark<C>(out + (size_t)i * 24, acc);
}

}  // namespace arkbp
