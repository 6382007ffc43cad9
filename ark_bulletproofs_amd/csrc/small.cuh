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
static constexpr int DT_MAXSEG = 3, DT_MAXOUT = 4;
struct DtSeg {
    const u32* sc;
    u32 base0, count, resident;
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

// grid (nblk, nout).  Pair p = term * 64 + window: the 64 lanes of a wave share a term's scalar (one broadcast load) and read 64
// entries of that base's 60 KiB.  out: [nout][nblk] points (ark words); with nblk == 1 these are the results.
template <class C> __global__ void __launch_bounds__(256)
k_dt_accum(const u32* __restrict__ tab, DtJobs jobs, u32* __restrict__ out) {
    typedef typename C::Fr Fr;
    __shared__ u32 sh[256 * 27];
    const DtJob& jb = jobs.job[blockIdx.y];
    const u32 pairs = jb.terms * DT_WINDOWS;
    Jac acc = jac_inf<C>();
#pragma unroll 1
    for (u32 p = blockIdx.x * 256u + threadIdx.x; p < pairs; p += gridDim.x * 256u) {
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
            load_words8(k, jb.seg[s].sc + (size_t)term * 8);
            if (jb.seg[s].resident) fe_store_canon<Fr>(k, fe_unpack(k));
            base = jb.seg[s].base0 + term;
        }
        const u32 d = (k[w >> 3] >> (4u * (w & 7u))) & 15u;
        if (d) acc = jac_madd<C>(acc, load_aff_dev(tab + (((size_t)base * DT_WINDOWS + w) * DT_ENT + (d - 1u)) * 16));
    }
    acc = block_sum_jac_quad<C>(acc, sh);
    if (threadIdx.x == 0) store_jac_ark<C>(out + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 24, acc);
}
// grid (nout): sum of the nblk partial points of output blockIdx.x -> res[blockIdx.x]
template <class C> __global__ void __launch_bounds__(256)
k_dt_finish(const u32* __restrict__ part, u32 nblk, u32* __restrict__ res) {
    __shared__ u32 sh[256 * 27];
    Jac acc = jac_inf<C>();
    for (u32 j = threadIdx.x; j < nblk; j += 256) acc = jac_add<C>(acc, load_jac_ark<C>(part + ((size_t)blockIdx.x * nblk + j) * 24));
    acc = block_sum_jac_quad<C>(acc, sh);
    if (threadIdx.x == 0) store_jac_ark<C>(res + (size_t)blockIdx.x * 24, acc);
}

}  // namespace arkbp
