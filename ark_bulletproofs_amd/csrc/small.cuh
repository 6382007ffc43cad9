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
// grid (nblk, nout), 256 lanes = 64 quads.  Pair p = term * 64 + window: the 16 quads of a wave share a term's scalar (a broadcast
// load) and read 16 entries of that base's 60 KiB.  out: [nout][nblk] points (ark words); with nblk == 1 these are the results.
template <class C> __global__ void __launch_bounds__(256)
k_dt_accum(const u32* __restrict__ tab, DtJobs jobs, u32* __restrict__ out) {
    typedef typename C::Fr Fr;
    __shared__ u32 sh[64 * 27];
    const DtJob& jb = jobs.job[blockIdx.y];
    const u32 pairs = jb.terms * DT_WINDOWS;
    const u32 q = threadIdx.x & 3u, nquads = gridDim.x * 64u;
    // the table entry of pair p (false: digit 0, or p past the end)
    auto fetch = [&](u32 p, Aff& pt) -> bool {
        if (p >= pairs) return false;
        u32 term = p / DT_WINDOWS;
        const u32 w = p % DT_WINDOWS;
        u32 k[8];
        u32 base;
        if (jb.has_imm && term == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) k[j] = jb.imm[j];
            base = jb.imm_base;
        } else {
            term -= jb.has_imm;
            u32 s = 0;
            while (s + 1 < jb.nseg && term >= jb.seg[s].count) { term -= jb.seg[s].count; s++; }
            const DtSeg& sg = jb.seg[s];
            if (sg.fold_n) term = (term / sg.fold_n) * 2u * sg.fold_n + (term % sg.fold_n) + (sg.fold_hi ? sg.fold_n : 0u);
            load_words8(k, sg.sc + (size_t)term * 8);
            if (sg.resident == 1) fe_store_canon<Fr>(k, fe_unpack(k));
            else if (sg.resident == 2) fe_store_canon<Fr>(k, fe_load_ark<Fr>(k));
            base = sg.base0 + term;
        }
        const u32 d = (k[w >> 3] >> (4u * (w & 7u))) & 15u;
        if (!d) return false;
        pt = load_aff_dev(tab + (((size_t)base * DT_WINDOWS + w) * DT_ENT + (d - 1u)) * 16);
        return true;
    };
    Jac acc = jac_inf<C>();
    u32 p = blockIdx.x * 64u + (threadIdx.x >> 2);
    Aff cur = {}, nxt = {};
    bool have = fetch(p, cur);
#pragma unroll 1
    while (p < pairs) {
        p += nquads;
        const bool have_n = fetch(p, nxt);            // (in flight during the addition below)
        if (have) acc = qjac_madd<C>(acc, cur, q);   // (quad-uniform: the four lanes fetched the same entry)
        cur = nxt; have = have_n;
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
    auto fetch = [&](u32 j, Aff& pt) -> bool {   // pair j of this quad: base j / 4, window quad + 16 * (j % 4)
        if (j >= 8u) return false;
        const u32 b = j >> 2, w = quad + 16u * (j & 3u);
        const u32 word = b ? kb[w >> 3] : kv[w >> 3];
        const u32 d = (word >> (4u * (w & 7u))) & 15u;
        if (!d) return false;
        pt = load_aff_dev(tab + (((size_t)b * DT_WINDOWS + w) * DT_ENT + (d - 1u)) * 16);
        return true;
    };
    Jac acc = jac_inf<C>();
    Aff cur = {}, nxt = {};
    bool have = fetch(0, cur);
#pragma unroll 1
    for (u32 j = 0; j < 8u; j++) {
        const bool have_n = fetch(j + 1u, nxt);
        if (have) acc = qjac_madd<C>(acc, cur, q);
        cur = nxt; have = have_n;
    }
#pragma unroll 1
    for (int off = 32; off >= 4; off >>= 1) acc = qjac_add<C>(acc, jac_shfl_down(acc, off), q);   // (quad j += quad j + off / 4; lanes past the live range compute unused sums)
    if (lane == 0) store_jac_ark<C>(out + (size_t)i * 24, acc);
}

}  // namespace arkbp
