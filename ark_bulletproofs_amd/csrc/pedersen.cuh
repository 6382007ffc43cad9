// Batched Pedersen commitments  V_i = v_i * B + r_i * B_blinding  (PedersenGens::commit, src/generators.rs:39-44; called once
// per committed input by Prover::commit, src/r1cs/prover.rs:327-341 -- 2^20+2 times for cfg3's shuffle statement).
// Both bases are fixed, so the ladder is replaced by fixed-base window tables resident in L2:
//   T[b][w][d] = d * 2^(8w) * base_b,   b in {B, B_blinding}, w < 32, d < 256   (1 MiB, d = 0 is the identity)
// and a commitment is 64 table look-ups + mixed additions and one inversion; no doublings.
#pragma once
#include "ipa.cuh"

namespace arkbp {

static constexpr u32 PC_WINDOWS = 32, PC_DIGITS = 256;
static constexpr size_t PC_TABLE_BYTES = (size_t)2 * PC_WINDOWS * PC_DIGITS * 64;

// one thread per table entry: d * base by an 8-bit ladder, then 8w doublings
template <class C> __global__ void __launch_bounds__(256)
k_pc_table_build(const u32* __restrict__ pc /* B, B_blinding: resident affine layout */, u32* __restrict__ table) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * PC_WINDOWS * PC_DIGITS) return;
    const u32 b = t / (PC_WINDOWS * PC_DIGITS), w = (t / PC_DIGITS) % PC_WINDOWS, d = t % PC_DIGITS;
    const Aff P = load_aff_dev(pc + (size_t)b * 16);
    Jac acc = jac_inf<C>();
#pragma unroll 1
    for (int bit = 7; bit >= 0; bit--) {
        acc = jac_dbl<C>(acc);
        if ((d >> bit) & 1) acc = jac_madd<C>(acc, P);
    }
#pragma unroll 1
    for (u32 i = 0; i < 8 * w; i++) acc = jac_dbl<C>(acc);
    const Aff o = jac_to_aff<C>(acc);
    u32 wd[16];
    aff_store_dev(wd, o);
    store_words8(table + (size_t)t * 16, wd);
    store_words8(table + (size_t)t * 16 + 8, wd + 8);
}

// one thread per commitment.  v, blind: ark layout (Montgomery words, as the caller's Fr values lie in memory);
// out: affine points in ark layout (x || y Montgomery words, identity = zeros)
template <class C> __global__ void __launch_bounds__(256)
k_pc_commit(const u32* __restrict__ table, const u32* __restrict__ v, const u32* __restrict__ blind, u32 m, u32* __restrict__ out) {
    typedef typename C::Fr Fr;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Jac acc = jac_inf<C>();
#pragma unroll 1
    for (int b = 0; b < 2; b++) {
        u32 k[8];
        load_words8(k, (b ? blind : v) + (size_t)i * 8);
        fe_store_canon<Fr>(k, fe_load_ark<Fr>(k));
        const u32* T = table + (size_t)b * PC_WINDOWS * PC_DIGITS * 16;
#pragma unroll 1
        for (u32 w = 0; w < PC_WINDOWS; w++) {
            const u32 d = (k[w >> 2] >> (8 * (w & 3))) & 255u;
            if (d) acc = jac_madd<C>(acc, load_aff_dev(T + ((size_t)w * PC_DIGITS + d) * 16));
        }
    }
    const Aff o = jac_to_aff<C>(acc);
    u32 wd[16];
    aff_store_ark<C>(wd, o);
    store_words8(out + (size_t)i * 16, wd);
    store_words8(out + (size_t)i * 16 + 8, wd + 8);
}

}  // namespace arkbp
