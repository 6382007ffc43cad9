// Small device helpers shared by both translation units (arkbp.hip, vfe.hip): 32-byte loads / stores of packed field elements
// in the resident (radix-2^29 Montgomery) form, lazy sums, the workgroup tree-sum and the power-table look-up.
#pragma once
#include "ec.cuh"

namespace arkbp {

__device__ __forceinline__ void load_words8(u32 w[8], const u32* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void store_words8(u32* p, const u32 w[8]) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

template <class F> __device__ __forceinline__ Fe load_fe_dev(const u32* p) {
    u32 w[8];
    load_words8(w, p);
    return fe_unpack(w);
}
template <class F> __device__ __forceinline__ void store_fe_dev(u32* p, const Fe& a) {  // any L = 1, V < 32 value
    u32 w[8];
    fe_pack(w, fe_canon<F>(a));
    store_words8(p, w);
}
template <class F> __device__ __forceinline__ void store_fe_canon(u32* p, const Fe& a) {
    u32 w[8];
    fe_store_canon<F>(w, a);
    store_words8(p, w);
}
// lazy sum kept at L = 1, V <= 2
template <class F> __device__ __forceinline__ Fe fe_addr(const Fe& a, const Fe& b) { return fe_wred<F>(fe_norm(fe_add(a, b))); }

// workgroup tree-sum of one field element per lane (256 lanes); result valid in lane 0
template <class F> __device__ __forceinline__ Fe block_sum_fe(Fe v, u32* sh /* 9*256 words */) {
    const u32 tid = threadIdx.x;
    for (u32 stride = 128; stride >= 1; stride >>= 1) {
        if (tid >= stride && tid < 2 * stride) {
#pragma unroll
            for (int i = 0; i < 9; i++) sh[i * 256 + tid] = v.l[i];
        }
        __syncthreads();
        if (tid < stride) {
            Fe o;
#pragma unroll
            for (int i = 0; i < 9; i++) o.l[i] = sh[i * 256 + tid + stride];
            v = fe_addr<F>(v, o);
        }
        __syncthreads();
    }
    return v;
}

// x^e from a table of x^(2^k) (resident words), e < 2^32
template <class F> __device__ __forceinline__ Fe pow_table(const u32* __restrict__ tab, u32 e) {
    if (!e) return fe_one<F>();
    int k = __ffs((int)e) - 1;
    Fe r = load_fe_dev<F>(tab + (size_t)k * 8);   // the lowest set bit costs a load, not a product
    e >>= k + 1; k++;
#pragma unroll 1
    for (; e; k++, e >>= 1)
        if (e & 1) r = fe_mul<F>(r, load_fe_dev<F>(tab + (size_t)k * 8));
    return r;
}

}  // namespace arkbp
