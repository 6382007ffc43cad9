// Variable-base multi-scalar multiplication on gfx950 — the device replacement for every
// `G::Group::msm(&bases, &scalars)` call of the reference (ark-ec VariableBaseMSM; call sites
// src/inner_product_proof.rs:104,124,187,202,375; src/r1cs/prover.rs:516,532,546,607,622,635;
// src/r1cs/verifier.rs:574,685).  The result is a group element, so any correct schedule gives the
// affine coordinates ark produces; this one is laid out for 256 CUs x 64-lane waves:
//
//   1. k_msm_digits    signed c-bit digits of every scalar (ark Montgomery or canonical words in HBM),
//                      per-(window,|digit|) histogram: global atomics, wave-aggregated when lanes collide
//   2. k_msm_scan_*    exclusive scans: entry offsets + chunk offsets for every reduction level
//   3. k_msm_scatter   counting-sort scatter of (term index, sign) into bucket order
//   4. k_msm_accum     level 1: one lane per CH-entry chunk of a bucket, mixed Jacobian+affine adds of
//                      gathered 64-byte bases.  Chunking (not "one lane per bucket") keeps skewed inputs
//                      balanced: 0/1 witness vectors put every term of a window in ONE bucket.
//   5. k_msm_reduce    levels 2..K: CH-ary tree over the per-chunk partial sums until one point per bucket
//   6. k_msm_marginals one 256-lane workgroup per (window w, bit k): sum of the buckets whose value has bit
//                      k set, LDS tree.  sum_w sum_v v*B[w][v] = sum_{w,k} 2^(c*w+k) * T[w][k]
//   7. host            the <= 280-term Horner over T (256 doublings of ONE point: a serial chain that a
//                      2.4 GHz scalar core runs ~40x faster than a single GPU lane).
//
// Algorithmic bytes: 96 B per term (64 B affine base + 32 B scalar), SURVEY.md §8(d).
#pragma once
#include "ec.cuh"
#include "fe_io.cuh"
#include "ecq.cuh"
#include "glv.cuh"

namespace arkbp {

#ifndef ARKBP_MSM_CH
#define ARKBP_MSM_CH 16
#endif
static constexpr int MSM_CH = ARKBP_MSM_CH;  // entries per level-1 lane / fan-in of the reduction tree on the skew-tolerant path
static constexpr int MSM_CHL = MSM_CH == 8 ? 3 : MSM_CH == 16 ? 4 : MSM_CH == 32 ? 5 : 6;
static_assert((1 << MSM_CHL) == MSM_CH, "ARKBP_MSM_CH must be 8, 16, 32 or 64");
static constexpr int MSM_CHL_BINNED = MSM_CHL;   // measured: whole-bucket lanes (64) lose more to divergence and a thin grid than the tree levels cost
static constexpr int MSM_MAXLVL = 8;   // 16^8 = 2^32 >= any bucket population
static constexpr int MSM_MAXSEG = 6;   // (the deferred second IPA round reads G and H in two pieces each, plus B)

// A logical base vector made of up to MSM_MAXSEG device-resident segments (e.g. G_R || H_L || Q) — the
// reference materialises such concatenations into fresh Vecs (src/inner_product_proof.rs:86-91).
struct BaseSegs {
    const u32* ptr[MSM_MAXSEG];   // 16 words per point, packed R' form (aff_store_dev)
    u32 start[MSM_MAXSEG + 1];    // prefix counts; start[nseg] = n
    int nseg;
    // fixed-base mode (the bases are generator tables with precomputed rows, see k_msm_fb_partition): ptr[k] names row 0 of
    // segment k's table, row r (= 2^(4r) * base) lies row_words[k] * r words further; an entry's term field is (window << tbits) | term
    u32 fixed_c4;                 // window bits / 4 (0 = ordinary MSM)
    u32 tbits;
    u64 row_words[MSM_MAXSEG];
    // GLV mode (glv_split below): an entry's term field is 2 * term + half; half 1 stands for phi(base) = (beta * x, y)
    u32 glv;
    // element stride of a segment in points (0 = 1): an index-cyclic slice of a generator table read in place (term j of the
    // segment is base first + j * stride — the sharded prover's rank r owns the bases r, r + world, ..)
    u32 stride[MSM_MAXSEG];
};
// The segment's fields are SELECTED (compares against scalar registers), not indexed by the lane-varying segment number: indexing
// the kernel-argument struct compiles to four dependent vector loads in front of every gather of the accumulate loops.
__device__ __forceinline__ const u32* seg_base_ptr(const BaseSegs& s, u32 idx) {
    u32 w = 0;
    if (s.glv) idx >>= 1;
    if (s.fixed_c4) { w = idx >> s.tbits; idx &= (1u << s.tbits) - 1u; }
    const u32* ptr = s.ptr[0];
    u32 start = s.start[0], sd = s.stride[0];
    u64 rw = s.row_words[0];
#pragma unroll
    for (int j = 1; j < MSM_MAXSEG; j++) {
        const bool in = j < s.nseg && idx >= s.start[j];
        ptr = in ? s.ptr[j] : ptr; start = in ? s.start[j] : start; sd = in ? s.stride[j] : sd; rw = in ? s.row_words[j] : rw;
    }
    sd = sd ? sd : 1u;
    return ptr + (size_t)(idx - start) * sd * 16 + (size_t)(w * s.fixed_c4) * rw;
}

// The scalar vector of an MSM, likewise: up to MSM_MAXSEG device-resident runs (e.g. blinding || a_L || a_R of a commitment,
// src/r1cs/prover.rs:516-531) read in place instead of being concatenated first.
struct ScalSegs {
    const u32* ptr[MSM_MAXSEG];   // 8 words per scalar
    u32 start[MSM_MAXSEG + 1];
    int nseg;
};
__device__ __forceinline__ const u32* seg_scalar_ptr(const ScalSegs& s, u32 idx) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < MSM_MAXSEG; j++) k += (j < s.nseg && idx >= s.start[j]) ? 1 : 0;
    return s.ptr[k] + (size_t)(idx - s.start[k]) * 8;
}

struct MsmPlan {
    int c, W, NB;        // window bits, windows, buckets per window (2^(c-1))
    u32 B;               // W * NB
    u32 n;
    int w_lo, w_hi;      // windows this call accumulates (multi-GPU window sharding); [0, W) = all
};

// One-pass sort: bucket (w, v) owns the fixed slot range [base[w] + v*cap[w], +cap[w]) so the histogram pass can place
// entries directly; cap[w] is ~2x the expected population (windows with few possible digit values get wider slots).  A bucket
// that outgrows its slots raises `overflow` and the exact two-pass path (k_msm_scatter) is used for that MSM instead.
static constexpr int MSM_MAXW = 88;
struct SlotPlan {
    u32 base[MSM_MAXW];
    u32 cap[MSM_MAXW];
};

__device__ __forceinline__ Aff load_aff_dev(const u32* p) {
    u32 w[16];
    load_words8(w, p);
    load_words8(w + 8, p + 8);
    return aff_load_dev(w);
}
// the same in two steps: the 64 bytes as they lie in memory (a gather issued one loop iteration ahead must not be UNPACKED there —
// the unpack would wait for it in front of the addition it is meant to hide behind), then the limbs
struct Raw16 { uint4 a, b, c, d; };
__device__ __forceinline__ Raw16 load_raw16(const u32* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    Raw16 r;
    r.a = q[0]; r.b = q[1]; r.c = q[2]; r.d = q[3];
    return r;
}
__device__ __forceinline__ Aff aff_from_raw(const Raw16& r) {
    const u32 w[16] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w, r.c.x, r.c.y, r.c.z, r.c.w, r.d.x, r.d.y, r.d.z, r.d.w};
    return aff_load_dev(w);
}
// Jacobian points in workspace: 3 x 8 words, canonical R' form; Z == 0 words <=> identity
template <class C> __device__ __forceinline__ void store_jac_ws(u32* p, const Jac& j) {
    typedef typename C::Fq F;
    u32 w[8];
    fe_pack(w, fe_canon<F>(j.X)); store_words8(p, w);
    fe_pack(w, fe_canon<F>(j.Y)); store_words8(p + 8, w);
    fe_pack(w, fe_canon<F>(j.Z)); store_words8(p + 16, w);
}
__device__ __forceinline__ Jac load_jac_ws(const u32* p) {
    u32 w[8];
    Jac j;
    load_words8(w, p); j.X = fe_unpack(w);
    load_words8(w, p + 8); j.Y = fe_unpack(w);
    load_words8(w, p + 16); j.Z = fe_unpack(w);
    return j;
}

// signed digit w of a canonical scalar (8 words), carry in/out; |d| <= 2^(c-1)
__device__ __forceinline__ int msm_digit(const u32 k[8], int w, int c, u32& carry) {
    const int bit = w * c;
    const int wi = bit >> 5, bo = bit & 31;
    u64 x = 0;
    if (wi < 8) x = k[wi];
    if (wi + 1 < 8) x |= (u64)k[wi + 1] << 32;
    u32 d = ((u32)(x >> bo) & ((1u << c) - 1)) + carry;
    carry = d > (1u << (c - 1)) ? 1u : 0u;
    return (int)d - (int)(carry << c);
}

__device__ __forceinline__ void load_words12(u32 w[12], const u32* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w; w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w;
}
__device__ __forceinline__ void store_words12(u32* p, const u32 w[12]) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]); q[1] = make_uint4(w[4], w[5], w[6], w[7]); q[2] = make_uint4(w[8], w[9], w[10], w[11]);
}
// half h of a split scalar as an 8-word magnitude (upper words zero) and its sign
__device__ __forceinline__ void glv_half(const u32 g[12], int h, u32 k[8], bool& neg) {
#pragma unroll
    for (int i = 0; i < 4; i++) { k[i] = h ? g[4 + i] : g[i]; k[4 + i] = 0; }
    neg = ((g[8] >> h) & 1u) != 0;
}

// Wave-aggregated counting.  Skewed inputs (the short top window, 0/1 witness vectors, repeated scalars) put
// most lanes of a wave on the same counter; then one lane adds the whole group's count.  For spread keys the
// probe (one ballot) fails and every lane issues its own atomic.  Returns this lane's slot when `cursor`.
__device__ __forceinline__ u32 wave_count(u32* __restrict__ ctr, u32 key, bool valid) {
    const u32 lane = __lane_id();
    u64 act = __ballot(valid);
    u32 pos = 0;
    if (!act) return 0;
    const int lead0 = __ffsll((unsigned long long)act) - 1;
    const u32 k0 = __shfl(key, lead0);
    const u64 m0 = __ballot(valid && key == k0);
    if (__popcll(m0) < 4) {
        if (valid) pos = atomicAdd(&ctr[key], 1u);
        return pos;
    }
    while (act) {
        const int lead = __ffsll((unsigned long long)act) - 1;
        const u32 kk = __shfl(key, lead);
        const u64 m = __ballot(valid && key == kk);
        u32 base = 0;
        if ((int)lane == lead) base = atomicAdd(&ctr[kk], (u32)__popcll(m));
        base = __shfl(base, lead);
        if (valid && key == kk) pos = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
        act &= ~m;
    }
    return pos;
}

// 1. digits + histogram.  scalars_mont: 0 canonical integers, 1 ark Montgomery words, 2 resident layout.
// lds_hist != 0 (one workgroup, a histogram of lds_hist = W * NB counters that fits the dynamic LDS): the counting runs in LDS and
// the histogram is stored once — a returning device-scope atomic is ~2 us on this part and the loop below issues one per WINDOW in
// sequence: 159 us for the 25-term MSM of a single verification (86 windows), 13 us this way.
template <class C> __global__ void __launch_bounds__(256)
k_msm_digits(ScalSegs scalars, u32* __restrict__ canon, u32* __restrict__ hist, MsmPlan pl, int scalars_mont, SlotPlan sp,
             u32* __restrict__ slots, u32* __restrict__ overflow, u32 lds_hist) {
    typedef typename C::Fr Fr;
    extern __shared__ u32 lh[];
    if (lds_hist) {
        for (u32 j = threadIdx.x; j < lds_hist; j += blockDim.x) lh[j] = 0;
        __syncthreads();
    }
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < pl.n;
    u32 k[8];
#pragma unroll
    for (int j = 0; j < 8; j++) k[j] = 0;
    if (live) {
        load_words8(k, seg_scalar_ptr(scalars, i));
        if (scalars_mont == 1) {          // ark Montgomery words (R = 2^256)
            Fe s = fe_load_ark<Fr>(k);
            fe_store_canon<Fr>(k, s);
        } else if (scalars_mont == 2) {   // the engine's resident layout (packed radix-2^29 Montgomery form)
            Fe s = fe_unpack(k);
            fe_store_canon<Fr>(k, s);
        }
        store_words8(canon + (size_t)i * 8, k);
    }
    u32 carry = 0;
    for (int w = 0; w < pl.W; w++) {
        const int d = msm_digit(k, w, pl.c, carry);   // every window is decoded: the signed-digit carry chains through all of them
        const bool valid = live && d != 0 && w >= pl.w_lo && w < pl.w_hi;
        const u32 v = (u32)(d < 0 ? -d : d) - 1;
        u32 pos;
        if (lds_hist) pos = valid ? atomicAdd(&lh[(u32)w * pl.NB + v], 1u) : 0u;
        else pos = wave_count(hist, (u32)w * pl.NB + v, valid);
        if (valid) {
            if (pos < sp.cap[w]) slots[sp.base[w] + v * sp.cap[w] + pos] = (i << 1) | (d < 0 ? 1u : 0u);
            else *overflow = 1u;
        }
    }
    if (lds_hist) {   // (hist is all zero between MSMs: a plain store of the non-zero counters)
        __syncthreads();
        for (u32 j = threadIdx.x; j < lds_hist; j += blockDim.x) if (lh[j]) hist[j] = lh[j];
    }
}

// 1b. Two-level sort for large MSMs.  Returning device-scope atomics run at the memory side on this multi-XCD part
// (~6 G/s measured), and k_msm_digits issues one per (term, window).  Here a workgroup first counts its tile of terms per
// coarse BIN (the top bits of |digit|-1) in LDS, reserves room in each bin's region with ONE global atomic per (window, bin),
// and places its entries; k_msm_bin_sort then sorts every bin region by the remaining LB bucket bits inside LDS and emits
// the per-bucket population (hist) and start (boff) with plain stores.  Bin regions have fixed capacity (mean + 8 sigma);
// an overflow flags the MSM for the exact path.  The short top window (few distinct digits) keeps the slot scheme.
struct BinPlan {
    u32 LB;      // bucket bits resolved inside a bin
    u32 NBIN;    // bins per window = NB >> LB
    u32 cap;     // entries per bin region
    u32 wb;      // windows [0, wb) are binned; [wb, W) use SlotPlan
    u32 tpt;     // terms per lane in k_msm_bin_partition
    u32 top_nb;  // > 0: the slot window has this many possible buckets (<= 2048) and is counted per workgroup in LDS
    u32 glv;     // the scalars are split (glv_split): every term contributes two half-terms 2 * i, 2 * i + 1 of 128 bits
};
// entry in a bin region: (term << (LB+1)) | (fine bucket << 1) | sign; with bp.glv the term field is 2 * i + half and `canon` holds
// MSM_GLV_WORDS words per scalar (the split magnitudes and signs) instead of the 8 canonical words
template <class C> __global__ void __launch_bounds__(256)
k_msm_bin_partition(ScalSegs scalars, u32* __restrict__ canon, u32* __restrict__ hist, MsmPlan pl, int scalars_mont, BinPlan bp,
                    SlotPlan sp, u32* __restrict__ bin_cur, u32* __restrict__ ent, u32* __restrict__ overflow, int wa, int we, int first) {
    typedef typename C::Fr Fr;
    extern __shared__ u32 lds_cnt[];   // (we - wa) * NBIN bin counters, then top_nb bucket counters of the slot window
    const u32 nbinc = (u32)(we - wa) * bp.NBIN;
    const bool top = first && bp.top_nb;           // this launch also places the slot window, aggregated per workgroup
    const u32 ncnt = nbinc + (top ? bp.top_nb : 0u);
    u32* lds_top = lds_cnt + nbinc;
    const int wtop = pl.W - 1;
    const int nh = bp.glv ? 2 : 1;
    for (u32 x = threadIdx.x; x < ncnt; x += 256) lds_cnt[x] = 0;
    __syncthreads();
    const u32 tile0 = blockIdx.x * 256u * bp.tpt;
    const int wend = top ? pl.W : we;
    for (u32 t = 0; t < bp.tpt; t++) {
        const u32 i = tile0 + t * 256u + threadIdx.x;
        if (i >= pl.n) break;
        u32 k[8];
        u32 g[MSM_GLV_WORDS];
        if (first) {
            load_words8(k, seg_scalar_ptr(scalars, i));
            if (scalars_mont == 1) { Fe s = fe_load_ark<Fr>(k); fe_store_canon<Fr>(k, s); }
            else if (scalars_mont == 2) { Fe s = fe_unpack(k); fe_store_canon<Fr>(k, s); }
            if constexpr (C::HAS_GLV) {
                if (bp.glv) {
                    if (!glv_split<C>(k, g)) *overflow = 1u;
                    store_words12(canon + (size_t)i * MSM_GLV_WORDS, g);
                } else store_words8(canon + (size_t)i * 8, k);
            } else store_words8(canon + (size_t)i * 8, k);
        } else {
            if (bp.glv) load_words12(g, canon + (size_t)i * MSM_GLV_WORDS); else load_words8(k, canon + (size_t)i * 8);
        }
        for (int h = 0; h < nh; h++) {
            bool hneg = false;
            if (bp.glv) glv_half(g, h, k, hneg);
            u32 carry = 0;
            for (int w = 0; w < wend; w++) {
                const int d = msm_digit(k, w, pl.c, carry);
                if (d == 0 || w < pl.w_lo || w >= pl.w_hi) continue;
                const u32 v = (u32)(d < 0 ? -d : d) - 1;
                if (w >= (int)bp.wb) { if (top && v < bp.top_nb) atomicAdd(&lds_top[v], 1u); }
                else if (w >= wa && w < we) atomicAdd(&lds_cnt[(u32)(w - wa) * bp.NBIN + (v >> bp.LB)], 1u);
            }
        }
    }
    __syncthreads();
    for (u32 x = threadIdx.x; x < ncnt; x += 256) {
        const u32 cn = lds_cnt[x];
        if (cn) lds_cnt[x] = x < nbinc ? atomicAdd(&bin_cur[(u32)wa * bp.NBIN + x], cn) : atomicAdd(&hist[(u32)wtop * pl.NB + (x - nbinc)], cn);
    }
    __syncthreads();
    const u32 fmask = (1u << bp.LB) - 1u;
    for (u32 t = 0; t < bp.tpt; t++) {
        const u32 i = tile0 + t * 256u + threadIdx.x;
        const bool live = i < pl.n;   // no early exit: wave_count below is a wave-wide operation
        u32 k[8];
        u32 g[MSM_GLV_WORDS];
#pragma unroll
        for (int j = 0; j < 8; j++) k[j] = 0;
#pragma unroll
        for (int j = 0; j < MSM_GLV_WORDS; j++) g[j] = 0;
        if (live) { if (bp.glv) load_words12(g, canon + (size_t)i * MSM_GLV_WORDS); else load_words8(k, canon + (size_t)i * 8); }
        for (int h = 0; h < nh; h++) {
            bool hneg = false;
            if (bp.glv) glv_half(g, h, k, hneg);
            const u32 vi = bp.glv ? 2u * i + (u32)h : i;   // term field of the entry
            u32 carry = 0;
            for (int w = 0; w < pl.W; w++) {
                const int d = msm_digit(k, w, pl.c, carry);
                const bool valid = live && d != 0 && w >= pl.w_lo && w < pl.w_hi;
                const u32 v = (u32)(d < 0 ? -d : d) - 1;
                const u32 sign = ((d < 0) != hneg) ? 1u : 0u;
                if (w >= (int)bp.wb) {          // slot scheme (only in the launch that owns the first window group)
                    if (!first) continue;
                    u32 pos;
                    if (bp.top_nb) pos = valid ? atomicAdd(&lds_top[min(v, bp.top_nb - 1u)], 1u) : 0u;
                    else pos = wave_count(hist, (u32)w * pl.NB + v, valid);
                    if (valid) {
                        if (pos < sp.cap[w] && (!bp.top_nb || v < bp.top_nb)) ent[sp.base[w] + v * sp.cap[w] + pos] = (vi << 1) | sign;
                        else *overflow = 1u;
                    }
                } else if (valid && w >= wa && w < we) {
                    const u32 bin = v >> bp.LB;
                    const u32 pos = atomicAdd(&lds_cnt[(u32)(w - wa) * bp.NBIN + bin], 1u);
                    if (pos < bp.cap) ent[((size_t)w * bp.NBIN + bin) * bp.cap + pos] = (vi << (bp.LB + 1)) | ((v & fmask) << 1) | sign;
                    else *overflow = 1u;
                }
            }
        }
    }
}
// grid (NBIN, W).  Binned window: sort the bin region by fine bucket in LDS; write hist and boff for its 2^LB buckets and the
// entries back as (term << 1) | sign.  Slot window: boff only (hist was counted by k_msm_bin_partition's atomics).
__global__ void __launch_bounds__(256)
k_msm_bin_sort(u32* __restrict__ ent, u32* __restrict__ bin_cur, u32* __restrict__ hist, u32* __restrict__ boff, MsmPlan pl, BinPlan bp,
               SlotPlan sp) {
    extern __shared__ u32 lds[];
    const u32 w = blockIdx.y, bin = blockIdx.x, tid = threadIdx.x;
    if (w >= bp.wb) {
        for (u32 v = bin * 256u + tid; v < (u32)pl.NB; v += bp.NBIN * 256u) boff[w * pl.NB + v] = sp.base[w] + v * sp.cap[w];
        return;
    }
    const u32 NF = 1u << bp.LB;
    u32* cnt = lds;            // NF counters, then cursors
    u32* wsum = lds + NF;      // 4 wave sums
    u32* buf = lds + NF + 4;   // cap entries
    const size_t region = ((size_t)w * bp.NBIN + bin) * bp.cap;
    const u32 n = min(bin_cur[w * bp.NBIN + bin], bp.cap);
    for (u32 x = tid; x < NF; x += 256) cnt[x] = 0;
    __syncthreads();
    if (tid == 0) bin_cur[w * bp.NBIN + bin] = 0;   // consumed (every lane has read it): the cursors are all-zero between MSMs
    const u32 fmask = NF - 1u;
    for (u32 x = tid; x < n; x += 256) {
        const u32 e = ent[region + x];
        buf[x] = e;
        atomicAdd(&cnt[(e >> 1) & fmask], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt[0..NF): lane t owns the contiguous run [t*per, (t+1)*per)
    const u32 per = (NF + 255u) / 256u;
    u32 run = 0;
    for (u32 x = tid * per; x < min((tid + 1) * per, NF); x++) run += cnt[x];
    u32 incl = run;
    const u32 lane = tid & 63u, wv = tid >> 6;
    for (int o = 1; o < 64; o <<= 1) { const u32 t2 = __shfl_up(incl, o); if ((int)lane >= o) incl += t2; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    u32 excl = incl - run;
    for (u32 q = 0; q < wv; q++) excl += wsum[q];
    const u32 b0 = w * pl.NB + (bin << bp.LB);
    for (u32 x = tid * per; x < min((tid + 1) * per, NF); x++) {
        const u32 cn = cnt[x];
        hist[b0 + x] = cn;
        boff[b0 + x] = (u32)region + excl;
        cnt[x] = excl;        // becomes the placement cursor
        excl += cn;
    }
    __syncthreads();
    for (u32 x = tid; x < n; x += 256) {
        const u32 e = buf[x];
        const u32 pos = atomicAdd(&cnt[(e >> 1) & fmask], 1u);
        ent[region + pos] = ((e >> (bp.LB + 1)) << 1) | (e & 1u);
    }
}

// 1c. Fixed-base MSM over generator tables.  BulletproofGens never change, and 7/9 of a proof's MSM terms are over them (the
// commitments A_I, A_O, S and the first round's L, R; src/r1cs/prover.rs:516-559, src/inner_product_proof.rs:83-131).  With the
// rows 2^(c*w) * P_i precomputed (bp_gens_msm_tables), sum_i s_i P_i = sum_i sum_w d_iw * (2^(c*w) P_i): EVERY window's digits
// fall into ONE shared bucket set of 2^(c-1) buckets — W * n (digit, table-row) pairs sorted by |digit| alone.  The per-window
// costs (bucket aggregation, W * c doublings of the Horner tail) are paid once instead of W times, so the window can be wide
// (c = 20 at 2^21 terms: 13 mixed adds per term instead of 17-18) and the result needs no doublings at all.
// This kernel is k_msm_bin_partition for that layout: bin regions are per bin (not per window), an entry is
// (((window << tbits) | term) << (LB+1)) | (fine bucket << 1) | sign.
template <class C> __global__ void __launch_bounds__(256)
k_msm_fb_partition(ScalSegs scalars, u32* __restrict__ canon, MsmPlan pl, int scalars_mont, BinPlan bp, u32 tbits, u32* __restrict__ bin_cur,
                   u32* __restrict__ ent, u32* __restrict__ overflow) {
    typedef typename C::Fr Fr;
    extern __shared__ u32 lds_cnt[];   // NBIN bin counters
    for (u32 x = threadIdx.x; x < bp.NBIN; x += 256) lds_cnt[x] = 0;
    __syncthreads();
    const u32 tile0 = blockIdx.x * 256u * bp.tpt;
    for (u32 t = 0; t < bp.tpt; t++) {
        const u32 i = tile0 + t * 256u + threadIdx.x;
        if (i >= pl.n) break;
        u32 k[8];
        load_words8(k, seg_scalar_ptr(scalars, i));
        if (scalars_mont == 1) { Fe s = fe_load_ark<Fr>(k); fe_store_canon<Fr>(k, s); }
        else if (scalars_mont == 2) { Fe s = fe_unpack(k); fe_store_canon<Fr>(k, s); }
        store_words8(canon + (size_t)i * 8, k);
        u32 carry = 0;
        for (int w = 0; w < pl.W; w++) {
            const int d = msm_digit(k, w, pl.c, carry);
            if (d == 0) continue;
            const u32 v = (u32)(d < 0 ? -d : d) - 1;
            atomicAdd(&lds_cnt[v >> bp.LB], 1u);
        }
    }
    __syncthreads();
    for (u32 x = threadIdx.x; x < bp.NBIN; x += 256) {
        const u32 cn = lds_cnt[x];
        if (cn) lds_cnt[x] = atomicAdd(&bin_cur[x], cn);
    }
    __syncthreads();
    const u32 fmask = (1u << bp.LB) - 1u;
    for (u32 t = 0; t < bp.tpt; t++) {
        const u32 i = tile0 + t * 256u + threadIdx.x;
        if (i >= pl.n) break;
        u32 k[8];
        load_words8(k, canon + (size_t)i * 8);
        u32 carry = 0;
        for (int w = 0; w < pl.W; w++) {
            const int d = msm_digit(k, w, pl.c, carry);
            if (d == 0) continue;
            const u32 v = (u32)(d < 0 ? -d : d) - 1;
            const u32 bin = v >> bp.LB;
            const u32 pos = atomicAdd(&lds_cnt[bin], 1u);
            if (pos < bp.cap) ent[(size_t)bin * bp.cap + pos] = ((((u32)w << tbits) | i) << (bp.LB + 1)) | ((v & fmask) << 1) | (d < 0 ? 1u : 0u);
            else *overflow = 1u;
        }
    }
}
// rows of a fixed-base table: tmp[r * n + i] = 2^(4r) * P_i (Jacobian), r < R
template <class C> __global__ void __launch_bounds__(256)
k_msm_fb_rows(const u32* __restrict__ gens, u32* __restrict__ tmp, u32 n, u32 R) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Jac acc = jac_madd<C>(jac_inf<C>(), load_aff_dev(gens + (size_t)i * 16));
#pragma unroll 1
    for (u32 r = 0; r < R; r++) {
        store_jac_ws<C>(tmp + ((size_t)r * n + i) * 24, acc);
        acc = jac_dbl<C>(jac_dbl<C>(jac_dbl<C>(jac_dbl<C>(acc))));
    }
}
// 2. scans.  lvl_off[k] (k = 0..nl-1) has B+1 entries: level 0 counts entries, level k >= 1 counts
// ceil(cnt / CH^k) chunks.  totals[k] = lvl_off[k][B]; totals[NLMAX] = max bucket population.
// Three launches: per-tile sums, one-workgroup scan of the tile sums, per-tile exclusive scan.
static constexpr int MSM_NLMAX = MSM_MAXLVL + 1;
// population of a bucket at level k+1 given level k (level 0 = entries, level k >= 1 = partial sums).  "Special" buckets (the
// narrow top window: a handful of buckets holding ~n/2 entries each) go through the first spl levels like everyone else (fully
// parallel over chunks) and are then finished by k_msm_reduce_special in ONE launch (one workgroup per bucket, <= 32 partials
// per lane), instead of stretching the generic tree by lg16(n) levels for their sake.
__device__ __forceinline__ u32 msm_next_level(u32 v, int k, int chl, bool special, int spl /* level whose partials the special kernel sums */) {
    if (special && k >= spl) return v ? 1u : 0u;
    return (v + (1u << chl) - 1) >> chl;
}
static constexpr int MSM_SCAN_TILE = 2048;  // buckets per workgroup (256 lanes x 8)

__global__ void __launch_bounds__(256) k_msm_scan_tiles(const u32* __restrict__ hist, u32* __restrict__ tile_sums, u32 B, int nl, int chl, u32 b_gen, int spl) {
    __shared__ u32 sh[MSM_NLMAX + 1][4];
    const u32 base = blockIdx.x * MSM_SCAN_TILE + threadIdx.x * 8;
    u32 sum[MSM_NLMAX + 1];
    for (int k = 0; k <= MSM_NLMAX; k++) sum[k] = 0;
    for (u32 j = 0; j < 8; j++) {
        const u32 b = base + j;
        u32 v = b < B ? hist[b] : 0;
        if (b < b_gen) sum[MSM_NLMAX] = max(sum[MSM_NLMAX], v);   // the depth of the generic tree; special buckets have their own kernel
        for (int k = 0; k < nl; k++) { sum[k] += v; v = msm_next_level(v, k, chl, b >= b_gen, spl); }
    }
    for (int k = 0; k <= MSM_NLMAX; k++) {
        u32 v = sum[k];
        for (int o = 32; o >= 1; o >>= 1) { u32 t = __shfl_xor(v, o); v = (k == MSM_NLMAX) ? max(v, t) : v + t; }
        if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x <= MSM_NLMAX) {
        const int k = threadIdx.x;
        u32 v = (k == MSM_NLMAX) ? max(max(sh[k][0], sh[k][1]), max(sh[k][2], sh[k][3])) : sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3];
        tile_sums[(size_t)blockIdx.x * (MSM_NLMAX + 1) + k] = v;
    }
}
// one workgroup: exclusive scan of the tile sums in place, totals out
__global__ void __launch_bounds__(64) k_msm_scan_top(u32* __restrict__ tile_sums, u32 ntiles, u32* __restrict__ totals, u32* __restrict__ lvl_off, u32 B) {
    const int k = threadIdx.x;
    if (k > MSM_NLMAX) return;
    u32 run = 0;
    for (u32 t = 0; t < ntiles; t++) {
        u32* p = &tile_sums[(size_t)t * (MSM_NLMAX + 1) + k];
        const u32 v = *p;
        if (k == MSM_NLMAX) run = max(run, v); else { *p = run; run += v; }
    }
    totals[k] = run;
    if (k < MSM_NLMAX) lvl_off[(size_t)k * (B + 1) + B] = run;
}
__global__ void __launch_bounds__(256) k_msm_scan_apply(u32* __restrict__ hist, const u32* __restrict__ tile_sums, u32* __restrict__ lvl_off, u32 B, int nl, int chl, u32 b_gen, int spl) {
    __shared__ u32 sh[MSM_NLMAX][4];
    const u32 base = blockIdx.x * MSM_SCAN_TILE + threadIdx.x * 8;
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 cnt[8];
    u32 sum[MSM_NLMAX];
    for (int k = 0; k < MSM_NLMAX; k++) sum[k] = 0;
    for (u32 j = 0; j < 8; j++) {
        const u32 b = base + j;
        u32 v = b < B ? hist[b] : 0;
        if (b < B) hist[b] = 0;   // last reader: the histogram is all-zero between MSMs (no memset per MSM)
        cnt[j] = v;
        for (int k = 0; k < nl; k++) { sum[k] += v; v = msm_next_level(v, k, chl, b >= b_gen, spl); }
    }
    u32 excl[MSM_NLMAX];
    for (int k = 0; k < nl; k++) {   // inclusive scan across the wave, then exclusive
        u32 v = sum[k];
        for (int o = 1; o < 64; o <<= 1) { u32 t = __shfl_up(v, o); if ((int)lane >= o) v += t; }
        if (lane == 63) sh[k][wv] = v;
        excl[k] = v - sum[k];
    }
    __syncthreads();
    for (int k = 0; k < nl; k++) {
        u32 add = tile_sums[(size_t)blockIdx.x * (MSM_NLMAX + 1) + k];
        for (u32 w2 = 0; w2 < wv; w2++) add += sh[k][w2];
        excl[k] += add;
    }
    for (u32 j = 0; j < 8; j++) {
        const u32 b = base + j;
        if (b >= B) break;
        u32 v = cnt[j];
        for (int k = 0; k < nl; k++) { lvl_off[(size_t)k * (B + 1) + b] = excl[k]; excl[k] += v; v = msm_next_level(v, k, chl, b >= b_gen, spl); }
    }
}

// 3. scatter into bucket order.  cursor starts as a copy of lvl_off[0].
__global__ void __launch_bounds__(256) k_msm_scatter(const u32* __restrict__ canon, u32* __restrict__ cursor, u32* __restrict__ entries, MsmPlan pl) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < pl.n;
    u32 k[8];
#pragma unroll
    for (int j = 0; j < 8; j++) k[j] = 0;
    if (live) load_words8(k, canon + (size_t)i * 8);
    u32 carry = 0;
    for (int w = 0; w < pl.W; w++) {
        const int d = msm_digit(k, w, pl.c, carry);
        const bool valid = live && d != 0 && w >= pl.w_lo && w < pl.w_hi;
        const u32 pos = wave_count(cursor, (u32)w * pl.NB + (u32)(d < 0 ? -d : d) - 1, valid);
        if (valid) entries[pos] = (i << 1) | (d < 0 ? 1u : 0u);
    }
}

// largest b with off[b] <= j   (off has B+1 monotone entries, off[B] > j)
__device__ __forceinline__ u32 find_bucket(const u32* __restrict__ off, u32 B, u32 j) {
    u32 lo = 0, hi = B;  // invariant: off[lo] <= j < off[hi]
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (off[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

// 4. level 1: lane j sums the j-th CH-entry chunk (mixed adds of gathered affine bases)
template <class C> __global__ void __launch_bounds__(256)
k_msm_accum(BaseSegs segs, const u32* __restrict__ entries, const u32* __restrict__ off0, const u32* __restrict__ off1, u32* __restrict__ out,
            u32 B, u32 nchunks, int slotted, SlotPlan sp, u32 NB, const u32* __restrict__ boff, int chl) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nchunks) return;
    const u32 b = find_bucket(off1, B, j);
    u32 beg, end;
    // a bucket of cnt entries owns k = ceil(cnt / 2^chl) chunks; they share the entries EVENLY (ceil(cnt / k) each) instead of k - 1
    // full chunks and a remainder: the lanes of a wave then run nearly the same number of mixed adds (a wave lasts as long as its
    // longest lane: with ~52 entries per bucket the 16 + 16 + 16 + 4 split left a fifth of the lane-steps idle)
    const u32 r = j - off1[b], k = off1[b + 1] - off1[b];
    u32 s0, cnt;
    if (slotted == 2) {  // two-level sort: bucket b starts at boff[b]; its population is off0[b+1] - off0[b]
        s0 = boff[b]; cnt = off0[b + 1] - off0[b];
    } else if (slotted) {  // entries = slot array; the bucket's population is off0[b+1] - off0[b]
        const u32 w = b / NB, v = b - w * NB;
        s0 = sp.base[w] + v * sp.cap[w]; cnt = off0[b + 1] - off0[b];
    } else {
        s0 = off0[b]; cnt = off0[b + 1] - off0[b];
    }
    const u32 per = k > 1 ? (cnt + k - 1) / k : (1u << chl);
    beg = s0 + min(r * per, cnt);
    end = min(beg + per, s0 + cnt);
    Jac acc = jac_inf<C>();
    if (beg < end) {
        // the gather of entry e+1 (entry word, then a 64-byte base somewhere in a table far larger than L2) is issued before the
        // mixed add of entry e, so its latency hides behind ~3 k VALU instructions instead of stalling the lane.  The first entry is
        // lifted, not added; the additions run the one-block form and redo the rare exceptional ones (ec.cuh jac_madd_fast).
        // (the gather of e + 1 stays raw until the next iteration, and its entry word was read an iteration earlier still: neither
        // the unpack nor the address computation waits on memory in front of the addition)
        u32 ent = entries[beg];
        u32 ent_n = beg + 1 < end ? entries[beg + 1] : 0u;
        Raw16 raw = load_raw16(seg_base_ptr(segs, ent >> 1));
        for (u32 e = beg; e < end; e++) {
            Raw16 raw_n = raw;
            u32 ent_nn = 0;
            if (e + 1 < end) raw_n = load_raw16(seg_base_ptr(segs, ent_n >> 1));
            if (e + 2 < end) ent_nn = entries[e + 2];
            const Aff p = aff_from_raw(raw);
            if (e == beg) {
                acc = jac_from_aff<C>(aff_cneg_lazy<C>(p, ent & 1));
                acc.Y = fe_wred<typename C::Fq>(acc.Y);
            } else {
                bool rare;
                Jac nxt = jac_madd_fast<C>(acc, aff_cneg_lazy<C>(p, ent & 1), rare);
                if (__builtin_expect(rare, 0)) nxt = jac_madd<C>(acc, aff_cneg_lazy<C>(load_aff_dev(seg_base_ptr(segs, ent >> 1)), ent & 1));
                acc = nxt;
            }
            ent = ent_n; ent_n = ent_nn;
            raw = raw_n;
        }
    }
    store_jac_ws<C>(out + (size_t)j * 24, acc);
}

// 4b. A/B arm of the accumulate study — the shape BASELINE.json's north_star prescribes: ONE WAVEFRONT PER BUCKET.  Lane l of the
// wave gathers entries l, l + 64, ... of its bucket (the entry words are one coalesced read, the bases are 64 B gathers), sums
// them with mixed adds, and the 64 partial sums meet in a 6-level tree through LDS.  The bucket's single sum lands where the last
// tree level of the chunked path would put it (off_last[b]), so no reduce level follows.  Selected by ARKBP_MSM_ACCUM=wave;
// measured against the lane-per-16-entry-chunk kernel in profiles/r02_accum_ab_*.  Its weakness is arithmetic, not memory: with
// c = 14-15 a bucket holds ~64-128 entries, i.e. 1-2 mixed adds per lane, and then the tree runs 6 full-price Jacobian additions
// on 32, 16, 8, 4, 2, 1 active lanes — ~1.5 modular-product slots per entry against 0.17 for the chunked kernel.
template <class C> __global__ void __launch_bounds__(256)
k_msm_accum_wave(BaseSegs segs, const u32* __restrict__ entries, const u32* __restrict__ off0, const u32* __restrict__ off_last, u32* __restrict__ out, u32 B,
                 int slotted, SlotPlan sp, u32 NB, const u32* __restrict__ boff) {
    __shared__ u32 sh[4][64 * 27];
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 b = blockIdx.x * 4 + wv;
    if (b >= B) return;                                  // (whole waves leave together)
    const u32 cnt = off0[b + 1] - off0[b];
    if (!cnt) return;
    u32 beg;
    if (slotted == 2) beg = boff[b];
    else if (slotted) { const u32 w = b / NB, v = b - w * NB; beg = sp.base[w] + v * sp.cap[w]; }
    else beg = off0[b];
    Jac acc = jac_inf<C>();
    for (u32 e = lane; e < cnt; e += 64) {
        const u32 ent = entries[beg + e];
        acc = jac_madd<C>(acc, aff_cneg_lazy<C>(load_aff_dev(seg_base_ptr(segs, ent >> 1)), ent & 1));
    }
    u32* s = sh[wv];
    for (u32 stride = 32; stride >= 1; stride >>= 1) {
        if (lane >= stride && lane < 2 * stride) {
#pragma unroll
            for (int i = 0; i < 9; i++) { s[i * 64 + lane] = acc.X.l[i]; s[(9 + i) * 64 + lane] = acc.Y.l[i]; s[(18 + i) * 64 + lane] = acc.Z.l[i]; }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (lane < stride) {
            Jac o;
#pragma unroll
            for (int i = 0; i < 9; i++) { o.X.l[i] = s[i * 64 + lane + stride]; o.Y.l[i] = s[(9 + i) * 64 + lane + stride]; o.Z.l[i] = s[(18 + i) * 64 + lane + stride]; }
            acc = jac_add<C>(acc, o);
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
    if (lane == 0) store_jac_ws<C>(out + (size_t)off_last[b] * 24, acc);
}

// 5. level k >= 2: lane j sums chunk j of level k-1 partials
template <class C> __global__ void __launch_bounds__(256)
k_msm_reduce(const u32* __restrict__ in, const u32* __restrict__ off_prev, const u32* __restrict__ off_cur, u32* __restrict__ out, u32 B,
             u32 nchunks, int chl, u32 skip_from /* buckets >= skip_from are left to k_msm_reduce_special at this level */) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nchunks) return;
    const u32 b = find_bucket(off_cur, B, j);
    if (b >= skip_from) return;
    const u32 beg = off_prev[b] + ((j - off_cur[b]) << chl);
    const u32 end = min(beg + (1u << chl), off_prev[b + 1]);
    Jac acc = load_jac_ws(in + (size_t)beg * 24);
    for (u32 e = beg + 1; e < end; e++) acc = jac_add<C>(acc, load_jac_ws(in + (size_t)e * 24));
    store_jac_ws<C>(out + (size_t)j * 24, acc);
}

// sum of the 256 lanes' points, valid in lane 0.  LDS tree: limbs stored limb-major (27 rows of 256 words) so lanes hit distinct banks
template <class C, u32 NT = 256> __device__ __forceinline__ Jac block_sum_jac(Jac acc, u32* __restrict__ sh /* NT * 27 words */) {
    const u32 tid = threadIdx.x;
    for (u32 stride = NT / 2; stride >= 1; stride >>= 1) {
        if (tid >= stride && tid < 2 * stride) {
#pragma unroll
            for (int i = 0; i < 9; i++) { sh[i * NT + tid] = acc.X.l[i]; sh[(9 + i) * NT + tid] = acc.Y.l[i]; sh[(18 + i) * NT + tid] = acc.Z.l[i]; }
        }
        __syncthreads();
        if (tid < stride) {
            Jac o;
#pragma unroll
            for (int i = 0; i < 9; i++) { o.X.l[i] = sh[i * NT + tid + stride]; o.Y.l[i] = sh[(9 + i) * NT + tid + stride]; o.Z.l[i] = sh[(18 + i) * NT + tid + stride]; }
            acc = jac_add<C>(acc, o);
        }
        __syncthreads();
    }
    return acc;
}
// The same sum with QUAD-COOPERATIVE additions (ecq.cuh): every lane parks its point in LDS, then m points become m/2 per level with
// one quad (4 lanes, one modular product per lane and dependency level) per addition — a level costs ~3 us instead of ~9 for the
// lane-per-addition tree above (tools/ubench_coop.hip, profiles/r03_ubench_coop.txt).  Valid in lane 0.
__device__ __forceinline__ void lds_put_jac(u32* __restrict__ sh, u32 NT, u32 slot, const Jac& a) {
#pragma unroll
    for (int i = 0; i < 9; i++) { sh[i * NT + slot] = a.X.l[i]; sh[(9 + i) * NT + slot] = a.Y.l[i]; sh[(18 + i) * NT + slot] = a.Z.l[i]; }
}
__device__ __forceinline__ Jac lds_get_jac(const u32* __restrict__ sh, u32 NT, u32 slot) {
    Jac o;
#pragma unroll
    for (int i = 0; i < 9; i++) { o.X.l[i] = sh[i * NT + slot]; o.Y.l[i] = sh[(9 + i) * NT + slot]; o.Z.l[i] = sh[(18 + i) * NT + slot]; }
    return o;
}
template <class C, u32 NT = 256> __device__ __forceinline__ Jac block_sum_jac_quad(const Jac& acc, u32* __restrict__ sh /* NT * 27 words */) {
    const u32 tid = threadIdx.x, q = tid & 3u, quad = tid >> 2;
    lds_put_jac(sh, NT, tid, acc);
    __syncthreads();
#pragma unroll 1
    for (u32 half = NT / 2; half >= 1; half >>= 1) {
        // pair j = (slot j, slot j + half) -> slot j.  A pass reads only slots its own quad writes or slots no quad of the pass
        // writes, so the (at most two) passes of a level need no barrier between them.
#pragma unroll 1
        for (u32 j = quad; j < half; j += NT / 4) {
            const Jac r = qjac_add<C>(lds_get_jac(sh, NT, j), lds_get_jac(sh, NT, j + half), q);
            if (q == 0) lds_put_jac(sh, NT, j, r);
        }
        __syncthreads();
    }
    return lds_get_jac(sh, NT, 0);
}
// a lane's point from `o` lanes further down its group of `width` lanes
__device__ __forceinline__ Jac jac_shfl_down(const Jac& a, int o, int width = 64) {
    Jac r;
#pragma unroll
    for (int i = 0; i < 9; i++) { r.X.l[i] = __shfl_down(a.X.l[i], o, width); r.Y.l[i] = __shfl_down(a.Y.l[i], o, width); r.Z.l[i] = __shfl_down(a.Z.l[i], o, width); }
    return r;
}
// 5b. special buckets: one workgroup sums ALL level-spl partials of its bucket into the bucket's single slot of the next level
template <class C> __global__ void __launch_bounds__(256)
k_msm_reduce_special(const u32* __restrict__ in, const u32* __restrict__ off1, const u32* __restrict__ off2, u32* __restrict__ out, u32 b_gen, u32 B) {
    __shared__ u32 sh[256 * 27];
    const u32 b = b_gen + blockIdx.x;
    if (b >= B) return;
    const u32 beg = off1[b], end = off1[b + 1];
    if (beg == end) return;   // (uniform for the workgroup)
    Jac acc = jac_inf<C>();
    for (u32 e = beg + threadIdx.x; e < end; e += 256) acc = jac_add<C>(acc, load_jac_ws(in + (size_t)e * 24));
    acc = block_sum_jac<C>(acc, sh);
    if (threadIdx.x == 0) store_jac_ws<C>(out + (size_t)off2[b] * 24, acc);
}

// 6. marginal sums: block (w, k) adds every bucket of window w whose value v = idx+1 has bit k set.
// `off` is the last level's offsets (<= 1 partial per bucket).  Output: Jacobian, ark Montgomery words
// (3 x 8 words) so the host tail reads them directly.
template <class C> __global__ void __launch_bounds__(256)
k_msm_marginals(const u32* __restrict__ sums, const u32* __restrict__ off, u32* __restrict__ T_out, MsmPlan pl) {
    typedef typename C::Fq F;
    __shared__ u32 sh[256 * 27];
    const int w = blockIdx.x, k = blockIdx.y;
    const u32 tid = threadIdx.x;
    Jac acc = jac_inf<C>();
    for (u32 idx = tid; idx < (u32)pl.NB; idx += 256) {
        if (!(((idx + 1) >> k) & 1)) continue;
        const u32 b = (u32)w * pl.NB + idx;
        const u32 o = off[b];
        if (off[b + 1] == o) continue;  // empty bucket
        acc = jac_add<C>(acc, load_jac_ws(sums + (size_t)o * 24));
    }
    acc = block_sum_jac<C>(acc, sh);
    if (tid == 0) {
        u32* o = T_out + ((size_t)w * pl.c + k) * 24;
        u32 wd[8];
        const bool inf = jac_is_inf(acc);
        fe_store_ark<F>(wd, acc.X); store_words8(o, wd);
        fe_store_ark<F>(wd, acc.Y); store_words8(o + 8, wd);
        if (inf) { for (int i = 0; i < 8; i++) wd[i] = 0; } else fe_store_ark<F>(wd, acc.Z);
        store_words8(o + 16, wd);
    }
}

// 6b. window sums by running sums (replaces the bit marginals): sum_v (v+1) * B[w][v] for a segment [lo, hi) of MSM_SEG buckets is
//   acc + lo * run,   run = sum_v B[v],   acc = sum_{i in [lo,hi)} sum_{v in [i,hi)} B[v]   (two additions per bucket),
// and lo * run is a <= 15-bit double-and-add.  A workgroup tree-sums its 256 segments; T_out[w][j] (Jacobian, ark Montgomery words)
// is the partial of workgroup j of window w, summed per window by the host before its Horner step.  ~60 modular products per
// bucket instead of ~120 for the marginals, but a serial chain of ~40 group operations per lane: used for large bucket counts,
// where the marginals' work dominates; small MSMs keep the marginals (shorter chains).
static constexpr u32 MSM_SEG = 8;
template <class C> __global__ void __launch_bounds__(256)
k_msm_window_sums(const u32* __restrict__ sums, const u32* __restrict__ off, u32* __restrict__ T_out, MsmPlan pl, u32 nblk) {
    typedef typename C::Fq F;
    __shared__ u32 sh[256 * 27];
    const u32 w = blockIdx.y, j = blockIdx.x, tid = threadIdx.x;
    const u32 lo = (j * 256u + tid) * MSM_SEG;
    Jac contrib = jac_inf<C>();
    if (lo < (u32)pl.NB) {
        const u32 hi = min((u32)pl.NB, lo + MSM_SEG);
        Jac run = jac_inf<C>(), acc = jac_inf<C>();
        for (u32 v = hi; v-- > lo;) {
            const u32 b = w * (u32)pl.NB + v;
            const u32 o = off[b];
            if (off[b + 1] != o) run = jac_add<C>(run, load_jac_ws(sums + (size_t)o * 24));
            acc = jac_add<C>(acc, run);
        }
        Jac m = jac_inf<C>();
        if (lo && !jac_is_inf(run)) {
#pragma unroll 1
            for (int bit = 31 - __clz((int)lo); bit >= 0; bit--) {
                m = jac_dbl<C>(m);
                if ((lo >> bit) & 1u) m = jac_add<C>(m, run);
            }
        }
        contrib = jac_add<C>(acc, m);
    }
    contrib = block_sum_jac<C>(contrib, sh);
    if (tid == 0) {
        u32* o = T_out + ((size_t)w * nblk + j) * 24;
        u32 wd[8];
        const bool inf = jac_is_inf(contrib);
        fe_store_ark<F>(wd, contrib.X); store_words8(o, wd);
        fe_store_ark<F>(wd, contrib.Y); store_words8(o + 8, wd);
        if (inf) { for (int i = 0; i < 8; i++) wd[i] = 0; } else fe_store_ark<F>(wd, contrib.Z);
        store_words8(o + 16, wd);
    }
}

// sum of `count` Jacobian points given as ark Montgomery words (the partials of k_msm_window_sums) -> out (same form), one workgroup
template <class C> __global__ void __launch_bounds__(256)
k_msm_sum_partials(const u32* __restrict__ T_in, u32 count, u32* __restrict__ T_out) {
    typedef typename C::Fq F;
    __shared__ u32 sh[256 * 27];
    Jac acc = jac_inf<C>();
    for (u32 j = threadIdx.x; j < count; j += 256) {
        u32 w[8];
        Jac p;
        load_words8(w, T_in + (size_t)j * 24); p.X = fe_load_ark<F>(w);
        load_words8(w, T_in + (size_t)j * 24 + 8); p.Y = fe_load_ark<F>(w);
        load_words8(w, T_in + (size_t)j * 24 + 16);
        bool z = true;
        for (int q = 0; q < 8; q++) z = z && w[q] == 0;
        if (z) continue;
        p.Z = fe_load_ark<F>(w);
        acc = jac_add<C>(acc, p);
    }
    acc = block_sum_jac<C>(acc, sh);
    if (threadIdx.x == 0) {
        u32 wd[8];
        const bool inf = jac_is_inf(acc);
        fe_store_ark<F>(wd, acc.X); store_words8(T_out, wd);
        fe_store_ark<F>(wd, acc.Y); store_words8(T_out + 8, wd);
        if (inf) { for (int i = 0; i < 8; i++) wd[i] = 0; } else fe_store_ark<F>(wd, acc.Z);
        store_words8(T_out + 16, wd);
    }
}

// ---- 7. The fixed-shape pipeline for mid-size MSMs (2^6 .. 2^18 terms, bit-marginal aggregation) ---------------------------------------
// The general path above sizes its launches from the level totals, i.e. the host waits for the scans in the middle of every MSM, and
// walks a tree of K launches whose depth is set by the fullest bucket.  For spread scalars (the Fiat-Shamir-derived vectors of the
// IPA rounds, the usual case) the shape is known in advance:
//   k_msm_bin_partition  as above
//   k_msm_bin_sort_fs    the bin sort; it also emits each bucket's population, its chunk offset INSIDE the bin and the bin's chunk total
//                        — no scan launches: a consumer scans the <= 4096 bin totals in LDS itself
//   k_msm_accum_fs       one lane per 16-entry chunk; the grid is the upper bound n*W/16 + B, lanes past the device-side total leave
//   k_msm_reduce_fs      one lane per bucket of the binned windows: <= 16 partials -> dense bucket sums
//   k_msm_marginals_fs   bit marginals of the binned windows from the dense sums; the narrow top window (a few buckets holding n/2^bits
//                        entries each) is summed straight from its level-1 partials, 4 .. 32 workgroups per bit, instead of
//                        stretching a tree for its sake; copies and clears the overflow flag into the result block
// One D2H copy, one wait, the host Horner tail.  A bucket above 256 entries or a full bin raises the flag: the MSM is then redone by
// the general path (skew-tolerant).  Results are identical either way (a sum of the same group elements).
static constexpr u32 MSM_FS_MAXBINS = 4096;
static constexpr u32 MSM_FS_BUCKET_MAX = 256;   // entries of a binned bucket; with chcap >= 8 at most 32 partials reach k_msm_reduce_fs
static constexpr u32 MSM_TOP_PARTS_MAX = 32;   // (a bit of the slot window collects ~n/32 partials: 4 .. 32 workgroups share them)
struct FsPlan {
    u32 nbins;      // wb * NBIN (+ 1 when the slot window exists: the last "bin" is that window)
    u32 has_top;    // the slot window (w = wb) exists
    u32 top_bits;   // bit length of its largest |digit|
    u32 max_chunks; // grid bound of k_msm_accum_fs
    u32 top_parts;  // workgroups per bit of the slot window in k_msm_marginals_fs
};
// exclusive scan of src[0..n) (n <= MSM_FS_MAXBINS) into sh[0..n], sh[n] = total; 256 lanes; sh has MSM_FS_MAXBINS + 1 words, ws 4
__device__ __forceinline__ void fs_block_scan(const u32* __restrict__ src, u32 n, u32* __restrict__ sh, u32* __restrict__ ws) {
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const u32 per = (n + 255u) >> 8;
    const u32 x0 = tid * per, x1 = min(x0 + per, n);
    u32 run = 0;
    for (u32 x = x0; x < x1; x++) run += src[x];
    u32 incl = run;
    for (int o = 1; o < 64; o <<= 1) { const u32 t2 = __shfl_up(incl, o); if ((int)lane >= o) incl += t2; }
    if (lane == 63) ws[wv] = incl;
    __syncthreads();
    u32 excl = incl - run;
    for (u32 q = 0; q < wv; q++) excl += ws[q];
    for (u32 x = x0; x < x1; x++) { sh[x] = excl; excl += src[x]; }
    if (tid == 255) sh[n] = excl;   // (the last lane's range ends at n, or is empty and then excl is the grand total)
    __syncthreads();
}
// grid (NBIN, wb + has_top).  Row wb (bin 0 only) is the slot window: hist -> bcnt / loff / boff, hist restored to zero.
__global__ void __launch_bounds__(256)
k_msm_bin_sort_fs(u32* __restrict__ ent, u32* __restrict__ bin_cur, u32* __restrict__ hist, u32* __restrict__ boff, u32* __restrict__ bcnt,
                  u32* __restrict__ loff, u32* __restrict__ bin_chunks, u32* __restrict__ overflow, MsmPlan pl, BinPlan bp, SlotPlan sp, u32 chcap) {
    extern __shared__ u32 lds[];
    const u32 w = blockIdx.y, bin = blockIdx.x, tid = threadIdx.x;
    const u32 lane = tid & 63u, wv = tid >> 6;
    const u32 chm = chcap - 1u;   // chunks of a bucket: ceil(population / chcap) (chcap: 8 .. 64, any value — the host picks it per MSM)
    if (w >= bp.wb) {
        if (bin) return;
        u32* wsum = lds;   // 4 wave sums
        const u32 wt = (u32)pl.W - 1u, nbt = bp.top_nb;
        const u32 per = (nbt + 255u) / 256u;
        u32 runc = 0;
        const u32 capt = sp.cap[wt];   // (a bucket that outgrew its slots raised the overflow flag; only the stored entries may be read)
        for (u32 x = tid * per; x < min((tid + 1) * per, nbt); x++) runc += (min(hist[wt * pl.NB + x], capt) + chm) / chcap;
        u32 incl = runc;
        for (int o = 1; o < 64; o <<= 1) { const u32 t2 = __shfl_up(incl, o); if ((int)lane >= o) incl += t2; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        u32 excl = incl - runc;
        for (u32 q = 0; q < wv; q++) excl += wsum[q];
        for (u32 x = tid * per; x < min((tid + 1) * per, nbt); x++) {
            const u32 b = wt * pl.NB + x;
            const u32 cn = min(hist[b], capt);
            hist[b] = 0;
            bcnt[b] = cn;
            loff[b] = excl;
            boff[b] = sp.base[wt] + x * sp.cap[wt];
            excl += (cn + chm) / chcap;
        }
        if (tid == 255) bin_chunks[bp.wb * bp.NBIN] = excl;
        return;
    }
    const u32 NF = 1u << bp.LB;
    u32* cnt = lds;            // NF counters, then cursors
    u32* wsum = lds + NF;      // 4 + 4 wave sums
    u32* buf = lds + NF + 8;   // cap entries
    const size_t region = ((size_t)w * bp.NBIN + bin) * bp.cap;
    const u32 n = min(bin_cur[w * bp.NBIN + bin], bp.cap);
    for (u32 x = tid; x < NF; x += 256) cnt[x] = 0;
    __syncthreads();
    if (tid == 0) bin_cur[w * bp.NBIN + bin] = 0;
    const u32 fmask = NF - 1u;
    for (u32 x = tid; x < n; x += 256) {
        const u32 e = ent[region + x];
        buf[x] = e;
        atomicAdd(&cnt[(e >> 1) & fmask], 1u);
    }
    __syncthreads();
    const u32 per = (NF + 255u) / 256u;
    u32 run = 0, runc = 0, big = 0;
    for (u32 x = tid * per; x < min((tid + 1) * per, NF); x++) { const u32 cn = cnt[x]; run += cn; runc += (cn + chm) / chcap; big = max(big, cn); }
    u32 incl = run, inclc = runc;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t2 = __shfl_up(incl, o), t3 = __shfl_up(inclc, o);
        if ((int)lane >= o) { incl += t2; inclc += t3; }
    }
    if (lane == 63) { wsum[wv] = incl; wsum[4 + wv] = inclc; }
    __syncthreads();
    u32 excl = incl - run, exclc = inclc - runc;
    for (u32 q = 0; q < wv; q++) { excl += wsum[q]; exclc += wsum[4 + q]; }
    const u32 b0 = w * pl.NB + (bin << bp.LB);
    for (u32 x = tid * per; x < min((tid + 1) * per, NF); x++) {
        const u32 cn = cnt[x];
        bcnt[b0 + x] = cn;
        boff[b0 + x] = (u32)region + excl;
        loff[b0 + x] = exclc;
        cnt[x] = excl;
        excl += cn;
        exclc += (cn + chm) / chcap;
    }
    if (tid == 255) bin_chunks[w * bp.NBIN + bin] = exclc;
    if (big > MSM_FS_BUCKET_MAX) *overflow = 1u;   // more partials than k_msm_reduce_fs takes in one step
    __syncthreads();
    for (u32 x = tid; x < n; x += 256) {
        const u32 e = buf[x];
        const u32 pos = atomicAdd(&cnt[(e >> 1) & fmask], 1u);
        ent[region + pos] = ((e >> (bp.LB + 1)) << 1) | (e & 1u);
    }
}
// buckets of bin x: first bucket index and count
__device__ __forceinline__ void fs_bin_range(u32 x, const MsmPlan& pl, const BinPlan& bp, u32& b0, u32& nb) {
    if (x < bp.wb * bp.NBIN) { const u32 w = x / bp.NBIN, bin = x - w * bp.NBIN; b0 = w * (u32)pl.NB + (bin << bp.LB); nb = 1u << bp.LB; }
    else { b0 = ((u32)pl.W - 1u) * (u32)pl.NB; nb = bp.top_nb; }
}
template <class C> __global__ void __launch_bounds__(256)
k_msm_accum_fs(BaseSegs segs, const u32* __restrict__ entries, const u32* __restrict__ bcnt, const u32* __restrict__ loff, const u32* __restrict__ boff,
               const u32* __restrict__ bin_chunks, u32* __restrict__ out, MsmPlan pl, BinPlan bp, FsPlan fp, u32 chcap, u32* __restrict__ info) {
    __shared__ u32 base[MSM_FS_MAXBINS + 1];
    __shared__ u32 ws[4];
    fs_block_scan(bin_chunks, fp.nbins, base, ws);
    const u32 total = min(base[fp.nbins], fp.max_chunks);
    if (blockIdx.x == 0 && threadIdx.x == 0) { info[0] = base[fp.nbins]; info[1] = base[fp.nbins - 1]; }   // chunks; first chunk of the last bin
    const u32 j = blockIdx.x * 256u + threadIdx.x;
    if (j >= total) return;
    u32 lo = 0, hi = fp.nbins;   // base[lo] <= j < base[hi]
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (base[mid] <= j) lo = mid; else hi = mid; }
    const u32 jj = j - base[lo];
    u32 b0, nb;
    fs_bin_range(lo, pl, bp, b0, nb);
    u32 l2 = 0, h2 = nb;         // loff[b0 + l2] <= jj; (h2 == nb or loff[b0 + h2] > jj)
    while (h2 - l2 > 1) { const u32 mid = (l2 + h2) >> 1; if (loff[b0 + mid] <= jj) l2 = mid; else h2 = mid; }
    const u32 b = b0 + l2;
    const u32 s0 = boff[b];
    const u32 cnt = bcnt[b], r = jj - loff[b];
    const u32 k = (cnt + chcap - 1u) / chcap;   // the bucket's chunks share its entries evenly (see k_msm_accum)
    const u32 per = k > 1 ? (cnt + k - 1) / k : chcap;
    const u32 beg = s0 + min(r * per, cnt);
    const u32 end = min(beg + per, s0 + cnt);
    Jac acc = jac_inf<C>();
    if (beg < end) {
        u32 ent = entries[beg];
        u32 ent_n = beg + 1 < end ? entries[beg + 1] : 0u;
        Raw16 raw = load_raw16(seg_base_ptr(segs, ent >> 1));
        for (u32 e = beg; e < end; e++) {   // (raw gather one entry ahead, entry word two ahead: see k_msm_accum)
            Raw16 raw_n = raw;
            u32 ent_nn = 0;
            if (e + 1 < end) raw_n = load_raw16(seg_base_ptr(segs, ent_n >> 1));
            if (e + 2 < end) ent_nn = entries[e + 2];
            Aff p = aff_from_raw(raw);
            if constexpr (C::HAS_GLV) {   // odd half-terms of a split scalar stand for phi(base) = (beta * x, y)
                if (segs.glv && (ent & 2u) && !aff_is_inf(p)) p.x = fe_mul<typename C::Fq>(p.x, fe_const<typename C::Fq, C::BETA29>());
            }
            if (e == beg) {     // the first entry is lifted, not added
                acc = jac_from_aff<C>(aff_cneg_lazy<C>(p, ent & 1));
                acc.Y = fe_wred<typename C::Fq>(acc.Y);
            } else {            // one-block additions; the rare exceptional ones are redone with the complete formulas (ec.cuh jac_madd_fast)
                bool rare;
                Jac nxt = jac_madd_fast<C>(acc, aff_cneg_lazy<C>(p, ent & 1), rare);
                if (__builtin_expect(rare, 0)) {
                    Aff q = load_aff_dev(seg_base_ptr(segs, ent >> 1));
                    if constexpr (C::HAS_GLV) {
                        if (segs.glv && (ent & 2u) && !aff_is_inf(q)) q.x = fe_mul<typename C::Fq>(q.x, fe_const<typename C::Fq, C::BETA29>());
                    }
                    nxt = jac_madd<C>(acc, aff_cneg_lazy<C>(q, ent & 1));
                }
                acc = nxt;
            }
            ent = ent_n; ent_n = ent_nn;
            raw = raw_n;
        }
    }
    store_jac_ws<C>(out + (size_t)j * 24, acc);
}
// G lanes per bucket of the binned windows (G = 4, or 1 when the buckets alone fill the chip): lane q of the group sums partials
// q, q + G, .. (<= 32 in all), lg G shuffle levels join them -> sums[b] (the identity when the bucket is empty)
template <class C, bool QUAD = false> __global__ void __launch_bounds__(256)
k_msm_reduce_fs(const u32* __restrict__ part, const u32* __restrict__ bcnt, const u32* __restrict__ loff, const u32* __restrict__ bin_chunks,
                u32* __restrict__ sums, MsmPlan pl, BinPlan bp, FsPlan fp, u32 chcap, u32 G) {
    __shared__ u32 base[MSM_FS_MAXBINS + 1];
    __shared__ u32 ws[4];
    fs_block_scan(bin_chunks, fp.nbins, base, ws);
    const u32 g = blockIdx.x * 256u + threadIdx.x;
    const u32 nbk = bp.wb * (u32)pl.NB;
    const u32 gb = G == 4 ? g >> 2 : g;
    const u32 b = min(gb, nbk - 1u), q = G == 4 ? g & 3u : 0u;   // (whole groups stay in the shuffles; the surplus groups of the last block store nothing)
    const u32 w = b / (u32)pl.NB, v = b - w * (u32)pl.NB;
    const u32 first = base[w * bp.NBIN + (v >> bp.LB)] + loff[b];
    u32 nch = min((bcnt[b] + chcap - 1u) / chcap, MSM_FS_BUCKET_MAX / 8u);
    if (first + nch > fp.max_chunks) nch = 0;
    if (QUAD) {
        // G == 4: the four lanes of a bucket are a DPP quad and SHARE every addition (ecq.cuh): the bucket's <= 16 partials as one
        // serial chain of quad additions (~3 us each) instead of 1-4 lane additions (~8 us each) plus two shuffle levels
        Jac acc = jac_inf<C>();
        if (nch) acc = load_jac_ws(part + (size_t)first * 24);
#pragma unroll 1
        for (u32 e = 1; e < nch; e++) acc = qjac_add<C>(acc, load_jac_ws(part + (size_t)(first + e) * 24), q);   // (trip count is quad-uniform)
        if (q == 0 && gb < nbk) store_jac_ws<C>(sums + (size_t)b * 24, acc);
        return;
    }
    // ONE inlined addition serves the serial steps and the shuffle levels: a Jacobian addition is ~50 KB of straight-line code
    Jac acc = jac_inf<C>();
    u32 e = q;
    int o = 0;   // 0: serial steps; 2, 1: shuffle levels; -1: done
#pragma unroll 1
    while (o >= 0) {
        Jac other;
        bool doit;
        if (o == 0) {
            doit = e < nch;
            if (!__any(doit)) { o = G == 4 ? 2 : -1; continue; }
            if (doit) other = load_jac_ws(part + (size_t)(first + e) * 24);
            e += G;
        } else {
            other = jac_shfl_down(acc, o, 4);
            doit = (int)q < o && q + (u32)o < nch;
            o = o == 2 ? 1 : -1;
        }
        if (doit) acc = jac_add<C>(acc, other);
    }
    if (q == 0 && gb < nbk) store_jac_ws<C>(sums + (size_t)b * 24, acc);
}
template <class C> __device__ __forceinline__ void store_T_ark(u32* __restrict__ o, const Jac& acc) {
    typedef typename C::Fq F;
    u32 wd[8];
    const bool inf = jac_is_inf(acc);
    fe_store_ark<F>(wd, acc.X); store_words8(o, wd);
    fe_store_ark<F>(wd, acc.Y); store_words8(o + 8, wd);
    if (inf) { for (int i = 0; i < 8; i++) wd[i] = 0; } else fe_store_ark<F>(wd, acc.Z);
    store_words8(o + 16, wd);
}
// blocks [0, wb * c): (w, k) of the binned windows over the dense sums; then top_bits * top_parts blocks for the slot window, summing
// the level-1 partials of the buckets whose value has bit k.  T_out[wb * c + k * top_parts + part]; info[2] = overflow flag (then cleared).
template <class C, u32 NT, bool QUAD = false> __global__ void __launch_bounds__(NT)
k_msm_marginals_fs(const u32* __restrict__ sums, const u32* __restrict__ part, const u32* __restrict__ bcnt, const u32* __restrict__ loff,
                   u32* __restrict__ T_out, MsmPlan pl, BinPlan bp, FsPlan fp, u32 chcap, u32* __restrict__ info, u32* __restrict__ overflow) {
    __shared__ u32 pre[2052];      // slot window: prefix counts of the partials of the buckets with bit k (top_nb <= 2048)
    __shared__ u32 wsum[NT / 64];
    __shared__ u32 tree[NT * 27];
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const u32 ngen = bp.wb * (u32)pl.c;
    if (blockIdx.x == 0 && tid == 0) { info[2] = *overflow; *overflow = 0; }
    const bool generic = blockIdx.x < ngen;
    u32 w = 0, k = 0, it, it_end, it_step;
    const u32 b0t = ((u32)pl.W - 1u) * (u32)pl.NB, nbt = bp.top_nb;
    u32 top_first = 0;
    if (generic) {
        w = blockIdx.x / (u32)pl.c; k = blockIdx.x - w * (u32)pl.c;
        it = tid; it_end = (u32)pl.NB; it_step = NT;
    } else {
        // slot window: a flat list of the partials of the buckets with bit k, dealt to fp.top_parts * 256 lanes
        const u32 t = blockIdx.x - ngen;
        k = t / fp.top_parts;
        const u32 prt = t - k * fp.top_parts;
        const u32 per = (nbt + NT - 1u) / NT;
        const u32 chm = chcap - 1u;
        u32 run = 0;
        for (u32 x = tid * per; x < min((tid + 1) * per, nbt); x++) run += (((x + 1) >> k) & 1u) ? (bcnt[b0t + x] + chm) / chcap : 0u;
        u32 incl = run;
        for (int o = 1; o < 64; o <<= 1) { const u32 t2 = __shfl_up(incl, o); if ((int)lane >= o) incl += t2; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        u32 excl = incl - run;
        for (u32 q = 0; q < wv; q++) excl += wsum[q];
        for (u32 x = tid * per; x < min((tid + 1) * per, nbt); x++) { pre[x] = excl; excl += (((x + 1) >> k) & 1u) ? (bcnt[b0t + x] + chm) / chcap : 0u; }
        if (tid == NT - 1u) pre[nbt] = excl;
        __syncthreads();
        it = prt * NT + tid; it_end = pre[nbt]; it_step = fp.top_parts * NT;
        top_first = info[1];   // (written by k_msm_accum_fs, an earlier launch on this stream)
    }
    // serial steps (one addition site for both kinds of block), then the LDS tree over the 256 lanes
    Jac acc = jac_inf<C>();
#pragma unroll 1
    while (true) {
        const u32* src = nullptr;
        if (generic) {
            while (it < it_end && !src) {
                if (((it + 1) >> k) & 1u) { const u32 b = w * (u32)pl.NB + it; if (bcnt[b]) src = sums + (size_t)b * 24; }
                it += it_step;
            }
        } else if (it < it_end) {
            u32 lo = 0, hi = nbt;   // pre[lo] <= it < pre[hi]
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (pre[mid] <= it) lo = mid; else hi = mid; }
            const u32 idx = top_first + loff[b0t + lo] + (it - pre[lo]);
            if (idx < fp.max_chunks) src = part + (size_t)idx * 24;
            it += it_step;
        }
        if (!__any(src != nullptr)) break;
        if (src) acc = jac_add<C>(acc, load_jac_ws(src));
    }
    acc = QUAD ? block_sum_jac_quad<C, NT>(acc, tree) : block_sum_jac<C, NT>(acc, tree);
    if (tid == 0) store_T_ark<C>(T_out + (size_t)blockIdx.x * 24, acc);
}

// ---- wire codec: ark-serialize compressed SW points (x as 8 Montgomery words + flag byte) -> affine, ark layout ----------
// Replaces the per-point square root of `R1CSProof::from_bytes` (src/r1cs/proof.rs:83-91; ark-ec get_point_from_x) for a
// whole batch.  flags: bit 7 = y is the larger root, bit 6 = identity (then x must be 0: checked by the host parser).
// ok[i] = 0 when x is not on the curve (FormatError upstream).
template <class C> __global__ void __launch_bounds__(256)
k_points_decompress(const u32* __restrict__ x_ark, const u32* __restrict__ flags, u32* __restrict__ out_xy_ark, u32* __restrict__ ok, u32 n) {
    typedef typename C::Fq F;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[16];
    const u32 fl = flags[i];
    if (fl & 0x40) {
        for (int j = 0; j < 16; j++) w[j] = 0;
        ok[i] = 1;
    } else {
        load_words8(w, x_ark + (size_t)i * 8);
        const Fe x = fe_canon<F>(fe_load_ark<F>(w));
        Aff p;
        const bool good = aff_from_x<C>(p, x, (fl & 0x80) != 0);
        ok[i] = good ? 1u : 0u;
        if (good) aff_store_ark<C>(w, p); else for (int j = 0; j < 16; j++) w[j] = 0;
    }
    store_words8(out_xy_ark + (size_t)i * 16, w);
    store_words8(out_xy_ark + (size_t)i * 16 + 8, w + 8);
}

// gathers the selected points (ark layout) and writes them in the engine's resident layout
template <class C> __global__ void k_points_gather_import(const u32* __restrict__ src_ark, const u32* __restrict__ idx, u32* __restrict__ dst, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[16];
    const u32* p = src_ark + (size_t)idx[i] * 16;
    load_words8(w, p);
    load_words8(w + 8, p + 8);
    Aff a = aff_load_ark<C>(w);
    u32 o[16];
    aff_store_dev(o, a);
    store_words8(dst + (size_t)i * 16, o);
    store_words8(dst + (size_t)i * 16 + 8, o + 8);
}

// ---- format conversion kernels -------------------------------------------------------------------
// ark layout (x||y Montgomery R=2^256, identity = zeros) -> device layout (packed R' form)
template <class C> __global__ void k_points_ark_to_dev(const u32* __restrict__ in, u32* __restrict__ out, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[16];
    load_words8(w, in + (size_t)i * 16);
    load_words8(w + 8, in + (size_t)i * 16 + 8);
    Aff p = aff_load_ark<C>(w);
    u32 o[16];
    aff_store_dev(o, p);
    store_words8(out + (size_t)i * 16, o);
    store_words8(out + (size_t)i * 16 + 8, o + 8);
}
template <class C> __global__ void k_points_dev_to_ark(const u32* __restrict__ in, u32* __restrict__ out, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff p = load_aff_dev(in + (size_t)i * 16);
    u32 o[16];
    aff_store_ark<C>(o, p);
    store_words8(out + (size_t)i * 16, o);
    store_words8(out + (size_t)i * 16 + 8, o + 8);
}

}  // namespace arkbp
