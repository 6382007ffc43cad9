// Host-side protocol layer of the PRODUCT (C++ mirror of the reference's Rust interface for the hot path):
// Merlin transcript + TranscriptProtocol (src/transcript.rs:45-102), generator derivation
// (src/generators.rs:47-121,174-221), the constraint-system recorder the prover/verifier need
// (src/r1cs/prover.rs:96-268, verifier.rs:69-224, linear_combination.rs), proof wire codec
// (src/r1cs/proof.rs:74-91).  These are sequential / O(1)-sized or run once per statement; every O(N)
// vector operation of prove / verify runs in HIP kernels (r1cs.cuh, ipa.cuh, msm.cuh).
// Byte-level behaviour follows SURVEY.md Appendix A (merlin 3.0, rand_chacha 0.3, sha3 0.10,
// ark-serialize 0.4).  Independent of the test oracle.
#pragma once
#include <algorithm>
#include <cstdlib>
#if defined(__linux__)
#include <sys/mman.h>
#endif
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>
#include "../../include/arkbp.h"
#include "host_math.hpp"

namespace arkbp {
namespace host {

// ---- Keccak-f[1600], 64-bit lanes, theta/rho-pi/chi/iota per round -----------------------------------
#include "keccak_unrolled.inc"
static inline void keccakf(u64 s[25]) { keccakf_unrolled((uint64_t*)s); }
static inline void sha3_512(u8 out[64], const u8* m, size_t n) {
    u64 s[25] = {0};
    u8* b = (u8*)s;
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) { b[pos++] ^= m[i]; if (pos == 72) { keccakf(s); pos = 0; } }
    b[pos] ^= 0x06; b[71] ^= 0x80;
    keccakf(s);
    memcpy(out, b, 64);
}

// Large host vectors (witness, constraints, blinding draws: 32-170 MB each at 2^20 multipliers) come from fresh mmap'd pages in
// every statement; with 4 KiB pages that is ~75 k page faults per statement, all serialised on the process's mmap lock against
// the other threads' faults and unmaps — with a dozen proofs in flight this lock, not the GPU, bounded the prover pipeline
// (two processes sharing one GPU were 29 % faster than one).  Asking for transparent huge pages on these ranges cuts the fault
// count 512-fold.  Call between reserve() and the first touch.  (Process-wide alternative: GLIBC_TUNABLES=glibc.malloc.hugetlb=1.)
static inline void advise_huge(const void* p, size_t bytes) {
#if defined(__linux__) && defined(MADV_HUGEPAGE)
    const uintptr_t lo = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
    if (bytes >= ((size_t)4 << 20) && hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
#else
    (void)p; (void)bytes;
#endif
}
template <class T> static inline void reserve_huge(std::vector<T>& v, size_t n) {
    if (n <= v.capacity()) return;
    v.reserve(n);
    advise_huge(v.data(), v.capacity() * sizeof(T));
}

// size of the library's own host thread pools: the machine's hardware threads, capped (one process per GPU shares the host
// with 7 others), overridable with ARKBP_HOST_THREADS
static inline unsigned host_pool_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    if (n > 32) n = 32;
    if (const char* e = getenv("ARKBP_HOST_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 256) n = (unsigned)v; }
    return n;
}

// A persistent pool for the library's data-parallel host loops (the per-instance transcript replays of batch verification run
// once per block of 512 instances: spawning 32 threads per block cost ~0.8 ms of the ~2 ms a block takes, and thread creation
// maps stacks under the process's mmap lock, so concurrent batches serialised on it).  run(lo, hi, fn): fn(k) for every k in
// [lo, hi), dynamically scheduled over the workers and the calling thread; returns when all are done.  One job at a time per pool.
class HostPool {
public:
    explicit HostPool(unsigned workers) {
        for (unsigned t = 0; t < workers; t++) th_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> g(m_); quit_ = true; gen_++; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void run(size_t lo, size_t hi, const std::function<void(size_t)>& fn) {
        if (hi <= lo) return;
        if (hi - lo == 1 || th_.empty()) { for (size_t k = lo; k < hi; k++) fn(k); return; }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn; next_.store(lo); hi_ = hi; pending_ = th_.size(); gen_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    void work() { for (;;) { const size_t k = next_.fetch_add(1); if (k >= hi_) break; (*fn_)(k); } }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (quit_) return;
            }
            work();
            { std::lock_guard<std::mutex> g(m_); if (--pending_ == 0) done_.notify_one(); }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    std::atomic<size_t> next_{0};
    size_t hi_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool quit_ = false;
};

// ---- ChaCha20Rng (rand_chacha 0.3: 64-bit counter, stream 0, sequential u32 word stream) ------------------
struct ChaChaRng {
    u32 k[8]; u64 ctr; u32 blk[16]; int at;
    explicit ChaChaRng(const u8 seed[32]) { memcpy(k, seed, 32); ctr = 0; at = 16; }
    static inline u32 r(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
    void block() {
        u32 in[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574, k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], (u32)ctr, (u32)(ctr >> 32), 0, 0};
        u32 x[16]; memcpy(x, in, 64);
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = r(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = r(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = r(x[d] ^ x[a], 8); x[c] += x[d]; x[b] = r(x[b] ^ x[c], 7);
        };
        for (int i = 0; i < 10; i++) { qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15); qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14); }
        for (int i = 0; i < 16; i++) blk[i] = x[i] + in[i];
        ctr++; at = 0;
    }
    u32 next_u32() { if (at == 16) block(); return blk[at++]; }
    u64 next_u64() { u64 lo = next_u32(); return lo | ((u64)next_u32() << 32); }
    void fill_bytes(u8* d, size_t n) { while (n) { u32 w = next_u32(); size_t c = n < 4 ? n : 4; memcpy(d, &w, c); d += c; n -= c; } }
};

// ---- STROBE-128 / merlin ----------------------------------------------------------------------------
struct Strobe {
    union { u64 w[25]; u8 b[200]; } st;
    u8 pos, pos_begin, cur;
    enum { RATE = 166, fI = 1, fA = 2, fC = 4, fM = 16, fK = 32 };
    void init(const u8* label, size_t n) {
        memset(&st, 0, sizeof st);
        const u8 h[18] = {1, RATE + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
        memcpy(st.b, h, 18);
        keccakf(st.w);
        pos = pos_begin = cur = 0;
        meta_ad(label, n, false);
    }
    inline void runf() { st.b[pos] ^= pos_begin; st.b[pos + 1] ^= 0x04; st.b[RATE + 1] ^= 0x80; keccakf(st.w); pos = 0; pos_begin = 0; }
    inline void absorb(const u8* d, size_t n) {   // runs up to the rate boundary at a time (the inner loop vectorises)
        while (n) {
            const size_t take = n < (size_t)(RATE - pos) ? n : (size_t)(RATE - pos);
            u8* dst = st.b + pos;
            for (size_t i = 0; i < take; i++) dst[i] ^= d[i];
            pos = (u8)(pos + take); d += take; n -= take;
            if (pos == RATE) runf();
        }
    }
    inline void overwrite(const u8* d, size_t n) { for (size_t i = 0; i < n; i++) { st.b[pos++] = d[i]; if (pos == RATE) runf(); } }
    inline void squeeze(u8* d, size_t n) { for (size_t i = 0; i < n; i++) { d[i] = st.b[pos]; st.b[pos++] = 0; if (pos == RATE) runf(); } }
    inline void begin(u8 flags, bool more) {
        if (more) return;
        u8 hdr[2] = {pos_begin, flags};
        pos_begin = pos + 1; cur = flags;
        absorb(hdr, 2);
        if ((flags & (fC | fK)) && pos != 0) runf();
    }
    void meta_ad(const u8* d, size_t n, bool more) { begin(fM | fA, more); absorb(d, n); }
    void ad(const u8* d, size_t n, bool more) { begin(fA, more); absorb(d, n); }
    void prf(u8* d, size_t n, bool more) { begin(fI | fA | fC, more); squeeze(d, n); }
    void key(const u8* d, size_t n, bool more) { begin(fA | fC, more); overwrite(d, n); }
};

struct Transcript {
    Strobe s;
    Transcript() {}
    Transcript(const u8* label, size_t n) { s.init((const u8*)"Merlin v1.0", 11); append_message("dom-sep", label, n); }
    explicit Transcript(const char* label) : Transcript((const u8*)label, strlen(label)) {}
    void append_message(const char* label, const u8* m, size_t n) {
        u32 len = (u32)n;
        s.meta_ad((const u8*)label, strlen(label), false); s.meta_ad((const u8*)&len, 4, true); s.ad(m, n, false);
    }
    void append_message(const char* label, const char* m) { append_message(label, (const u8*)m, strlen(m)); }
    void append_u64(const char* label, u64 x) { append_message(label, (const u8*)&x, 8); }
    void challenge_bytes(const char* label, u8* d, size_t n) {
        u32 len = (u32)n;
        s.meta_ad((const u8*)label, strlen(label), false); s.meta_ad((const u8*)&len, 4, true); s.prf(d, n, false);
    }
};
// merlin TranscriptRng as the prover uses it (src/r1cs/prover.rs:483-494)
struct TranscriptRng {
    Strobe s;
    explicit TranscriptRng(const Transcript& t) : s(t.s) {}
    void rekey(const char* label, const u8* w, size_t n) {
        u32 len = (u32)n;
        s.meta_ad((const u8*)label, strlen(label), false); s.meta_ad((const u8*)&len, 4, true); s.key(w, n, false);
    }
    void finalize_bytes(const u8 b[32]) { s.meta_ad((const u8*)"rng", 3, false); s.key(b, 32, false); }
    template <class R> void finalize(R& ext) { u8 b[32]; ext.fill_bytes(b, 32); finalize_bytes(b); }
    // fill_bytes(8).  Steady state (pos = 8, pos_begin = 0, i.e. right after a previous 8-byte draw): the six STROBE steps
    // meta_ad(LE32(8)) ; prf(8) collapse to three constant XORs, one permutation and reading lane 0.
    inline u64 next_u64() {
        if (s.pos == 8 && s.pos_begin == 0) {
            s.st.w[1] ^= 0x0709000000081200ULL;  // bytes 8..15: [0,M|A] [8,0,0,0] [9,I|A|C]
            s.st.w[2] ^= 0x040FULL;              // run_f: st[16] ^= pos_begin (15), st[17] ^= 0x04
            s.st.w[20] ^= 0x8000000000000000ULL; //        st[R+1 = 167] ^= 0x80
            keccakf(s.st.w);
            const u64 x = s.st.w[0];
            s.st.w[0] = 0;                       // squeeze zeroes what it reads; pos = 8, pos_begin = 0 again
            s.cur = Strobe::fI | Strobe::fA | Strobe::fC;
            return x;
        }
        u32 len = 8; u64 x; s.meta_ad((const u8*)&len, 4, false); s.prf((u8*)&x, 8, false); return x;
    }
    inline u32 next_u32() { u32 len = 4; u32 x; s.meta_ad((const u8*)&len, 4, false); s.prf((u8*)&x, 4, false); return x; }
};

// Eight independent TranscriptRngs advanced in lockstep, one per 64-bit lane of AVX-512 registers (Keccak-f x8 needs no
// cross-lane traffic).  All eight must be in the steady state above; their word streams are exactly what eight separate
// next_u64() sequences would produce.  out[t * 8 + j] = word t of instance j.
static inline bool cpu_has_avx512() {
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx512f");
#else
    return false;
#endif
}
#if defined(__x86_64__)
__attribute__((target("avx512f"))) static inline void transcript_rng_x8_words(TranscriptRng* const rngs[8], u64* out, size_t nwords) {
    alignas(64) u64 lanes[25][8];
    for (int i = 0; i < 25; i++) for (int j = 0; j < 8; j++) lanes[i][j] = rngs[j]->s.st.w[i];
    __m512i st[25];
    for (int i = 0; i < 25; i++) st[i] = _mm512_load_si512((const void*)lanes[i]);
    const __m512i k1 = _mm512_set1_epi64((long long)0x0709000000081200ULL), k2 = _mm512_set1_epi64(0x040FLL),
                  k20 = _mm512_set1_epi64((long long)0x8000000000000000ULL);
    for (size_t t = 0; t < nwords; t++) {
        st[1] = _mm512_xor_si512(st[1], k1);
        st[2] = _mm512_xor_si512(st[2], k2);
        st[20] = _mm512_xor_si512(st[20], k20);
        keccakf_x8_avx512(st);
        _mm512_storeu_si512((void*)(out + t * 8), st[0]);
        st[0] = _mm512_setzero_si512();
    }
    for (int i = 0; i < 25; i++) _mm512_store_si512((void*)lanes[i], st[i]);
    for (int i = 0; i < 25; i++) for (int j = 0; j < 8; j++) rngs[j]->s.st.w[i] = lanes[i][j];
    for (int j = 0; j < 8; j++) rngs[j]->s.cur = Strobe::fI | Strobe::fA | Strobe::fC;
}
#endif

// Eight transcripts driven through the SAME operation sequence (same labels, same message lengths: the verifier transcripts of
// eight same-shaped statements while their commitments are appended — 74 bytes per point, i.e. ~114 of the ~150 permutations a
// 256-commitment verification replays).  The sponge states live interleaved (one 64-bit lane of eight AVX-512 registers each); the
// STROBE positions are shared.  scatter() hands the states back to ordinary Transcripts.
#if defined(__x86_64__)
struct StrobeX8 {
    alignas(64) u64 lanes[25][8];
    u8 pos = 0, pos_begin = 0, cur = 0;
    void broadcast(const Strobe& p) {
        for (int i = 0; i < 25; i++) for (int j = 0; j < 8; j++) lanes[i][j] = p.st.w[i];
        pos = p.pos; pos_begin = p.pos_begin; cur = p.cur;
    }
    void scatter(Strobe* const out[8], int count) const {
        for (int j = 0; j < count; j++) {
            for (int i = 0; i < 25; i++) out[j]->st.w[i] = lanes[i][j];
            out[j]->pos = pos; out[j]->pos_begin = pos_begin; out[j]->cur = cur;
        }
    }
    inline void xor_same(unsigned byte, u8 v) { const u64 x = (u64)v << (8 * (byte & 7)); u64* w = lanes[byte >> 3]; for (int j = 0; j < 8; j++) w[j] ^= x; }
    __attribute__((target("avx512f"))) void runf() {
        xor_same(pos, pos_begin); xor_same(pos + 1u, 0x04); xor_same(Strobe::RATE + 1u, 0x80);
        __m512i st[25];
        for (int i = 0; i < 25; i++) st[i] = _mm512_load_si512((const void*)lanes[i]);
        keccakf_x8_avx512(st);
        for (int i = 0; i < 25; i++) _mm512_store_si512((void*)lanes[i], st[i]);
        pos = 0; pos_begin = 0;
    }
    __attribute__((target("avx512f"))) void absorb_same(const u8* d, size_t n) {
        for (size_t i = 0; i < n; i++) { xor_same(pos, d[i]); if (++pos == Strobe::RATE) runf(); }
    }
    __attribute__((target("avx512f"))) void absorb_each(const u8* const d[8], size_t n) {
        size_t i = 0;
        while (i < n) {
            if ((pos & 7) == 0 && n - i >= 8 && pos + 8 <= Strobe::RATE) {   // a whole 64-bit word of every lane
                u64* w = lanes[pos >> 3];
                for (int j = 0; j < 8; j++) { u64 x; memcpy(&x, d[j] + i, 8); w[j] ^= x; }
                i += 8; pos = (u8)(pos + 8);
            } else {
                const unsigned sh = 8 * (pos & 7);
                u64* w = lanes[pos >> 3];
                for (int j = 0; j < 8; j++) w[j] ^= (u64)d[j][i] << sh;
                i++; pos++;
            }
            if (pos == Strobe::RATE) runf();
        }
    }
    __attribute__((target("avx512f"))) void begin(u8 flags, bool more) {
        if (more) return;
        const u8 hdr[2] = {pos_begin, flags};
        pos_begin = pos + 1; cur = flags;
        absorb_same(hdr, 2);
        if ((flags & (Strobe::fC | Strobe::fK)) && pos != 0) runf();
    }
    // the states of eight ordinary transcripts that went through the same operations so far (false when their positions differ)
    bool gather(const Strobe* const in[8]) {
        for (int j = 1; j < 8; j++) if (in[j]->pos != in[0]->pos || in[j]->pos_begin != in[0]->pos_begin || in[j]->cur != in[0]->cur) return false;
        for (int i = 0; i < 25; i++) for (int j = 0; j < 8; j++) lanes[i][j] = in[j]->st.w[i];
        pos = in[0]->pos; pos_begin = in[0]->pos_begin; cur = in[0]->cur;
        return true;
    }
    __attribute__((target("avx512f"))) void squeeze_each(u8* const d[8], size_t n) {   // Strobe::squeeze on all eight: reads and zeroes
        size_t i = 0;
        while (i < n) {
            if ((pos & 7) == 0 && n - i >= 8 && pos + 8 <= Strobe::RATE) {
                u64* w = lanes[pos >> 3];
                for (int j = 0; j < 8; j++) { memcpy(d[j] + i, &w[j], 8); w[j] = 0; }
                i += 8; pos = (u8)(pos + 8);
            } else {
                const unsigned sh = 8 * (pos & 7);
                u64* w = lanes[pos >> 3];
                for (int j = 0; j < 8; j++) { d[j][i] = (u8)(w[j] >> sh); w[j] &= ~((u64)0xff << sh); }
                i++; pos++;
            }
            if (pos == Strobe::RATE) runf();
        }
    }
    // Transcript::append_message(label, m, n) with the same message on all eight
    __attribute__((target("avx512f"))) void append_message_same(const char* label, const u8* m, size_t n) {
        const u32 len = (u32)n;
        begin(Strobe::fM | Strobe::fA, false); absorb_same((const u8*)label, strlen(label));
        begin(Strobe::fM | Strobe::fA, true); absorb_same((const u8*)&len, 4);
        begin(Strobe::fA, false); absorb_same(m, n);
    }
    // Transcript::challenge_bytes(label, d_j, n) on all eight
    __attribute__((target("avx512f"))) void challenge_bytes_each(const char* label, u8* const d[8], size_t n) {
        const u32 len = (u32)n;
        begin(Strobe::fM | Strobe::fA, false); absorb_same((const u8*)label, strlen(label));
        begin(Strobe::fM | Strobe::fA, true); absorb_same((const u8*)&len, 4);
        begin(Strobe::fI | Strobe::fA | Strobe::fC, false); squeeze_each(d, n);
    }
    // Transcript::append_message(label, m_j, n) on all eight
    __attribute__((target("avx512f"))) void append_message_each(const char* label, const u8* const m[8], size_t n) {
        const u32 len = (u32)n;
        begin(Strobe::fM | Strobe::fA, false); absorb_same((const u8*)label, strlen(label));
        begin(Strobe::fM | Strobe::fA, true); absorb_same((const u8*)&len, 4);
        begin(Strobe::fA, false); absorb_each(m, n);
    }
};
#endif

// ark-ff Fp::rand: raw limbs (top limb masked to the modulus width) accepted iff < p; they ARE the Montgomery form
template <class P, class R> static inline F4 rand_fe(R& rng) {
    for (;;) {
        F4 x;
        for (int i = 0; i < 4; i++) x.v[i] = rng.next_u64();
        if (P::BITS < 256) x.v[3] &= (~(u64)0) >> (256 - P::BITS);
        if (!Fld<P>::geq_p(x.v)) return x;
    }
}
// ark-ec Affine::rand (cofactor 1): x <- Fq::rand, greatest <- bool, retry until x is on the curve
template <class C, class R> static inline A4 rand_point(R& rng) {
    for (;;) {
        F4 x = rand_fe<typename C::Fq>(rng);
        bool greatest = ((int32_t)rng.next_u32()) < 0;
        A4 p;
        if (Grp<C>::from_x(p, x, greatest)) return p;
    }
}

// ---- TranscriptProtocol (src/transcript.rs:45-102) -------------------------------------------------------
template <class C> struct TP {
    typedef Fld<typename C::Fr> S;
    static void append_scalar(Transcript& t, const char* label, const F4& x) { u8 b[32]; S::to_bytes(b, x); t.append_message(label, b, 32); }
    static void append_point(Transcript& t, const char* label, const A4& p) { u8 b[65]; Grp<C>::ser_uncompressed(b, p); t.append_message(label, b, 65); }
    static bool validate_and_append_point(Transcript& t, const char* label, const A4& p) { if (p.is_inf()) return false; append_point(t, label, p); return true; }
    static F4 challenge_scalar(Transcript& t, const char* label) {
        u8 buf[32]; t.challenge_bytes(label, buf, 32);
        ChaChaRng prng(buf);
        return rand_fe<typename C::Fr>(prng);
    }
    static void innerproduct_domain_sep(Transcript& t, u64 n) { t.append_message("dom-sep", "ipp v1"); t.append_u64("n", n); }
    static void r1cs_domain_sep(Transcript& t) { t.append_message("dom-sep", "r1cs v1"); }
    static void r1cs_1phase_domain_sep(Transcript& t) { t.append_message("dom-sep", "r1cs-1phase"); }
    static void r1cs_2phase_domain_sep(Transcript& t) { t.append_message("dom-sep", "r1cs-2phase"); }
};

// ---- generators ------------------------------------------------------------------------------------------
template <class C> struct PedersenGens {
    A4 B, B_blinding;
    static PedersenGens make_default() {  // src/generators.rs:47-66
        PedersenGens g; g.B = Grp<C>::generator();
        u8 ser[65], h[64];
        Grp<C>::ser_uncompressed(ser, g.B);
        sha3_512(h, ser, 65);
        ChaChaRng prng(h);
        g.B_blinding = rand_point<C>(prng);
        return g;
    }
    // Host fixed-base tables of the DEFAULT generators: d * 16^w * B for w < 64, d = 1..15 (affine).  A commitment of single scalars —
    // the T_1..T_6 of every proof, the inputs of small statements — is then 2 x 64 additions instead of two 256-step double-and-add
    // ladders (0.24 ms -> ~0.06 ms on one core; at the reference's own benchmark sizes, k-shuffles of 2 .. 1024 inputs, these
    // commitments were a third of a proof's wall time).  Built once per curve, on first use (~1 ms).
    struct FixedTables { A4 B, Bb; std::vector<A4> tB, tBb; };
    static const FixedTables& fixed_tables() {
        static FixedTables ft;
        static std::once_flag once;
        std::call_once(once, [] {
            typedef Grp<C> G; typedef Fld<typename C::Fq> F;
            const PedersenGens d = make_default();
            ft.B = d.B; ft.Bb = d.B_blinding;
            auto build = [](const A4& base, std::vector<A4>& out) {
                std::vector<J4> jac(64 * 15);
                J4 pw = G::from_aff(base);
                for (int w = 0; w < 64; w++) {
                    J4 acc = pw;
                    for (int dgt = 1; dgt <= 15; dgt++) { jac[w * 15 + dgt - 1] = acc; acc = G::add(acc, pw); }
                    pw = acc;            // 16 * pw
                }
                // to affine with one inversion (Montgomery's trick); no entry is the identity (the order is prime and > 2^250)
                std::vector<F4> pref(jac.size());
                F4 run = F::one();
                for (size_t i = 0; i < jac.size(); i++) { pref[i] = run; run = F::mul(run, jac[i].Z); }
                F4 inv = F::inv(run);
                out.resize(jac.size());
                for (size_t i = jac.size(); i-- > 0;) {
                    const F4 zi = F::mul(inv, pref[i]);
                    inv = F::mul(inv, jac[i].Z);
                    const F4 zi2 = F::sqr(zi);
                    out[i] = A4{F::mul(jac[i].X, zi2), F::mul(jac[i].Y, F::mul(zi2, zi))};
                }
            };
            build(ft.B, ft.tB); build(ft.Bb, ft.tBb);
        });
        return ft;
    }
    static J4 fixed_mul_acc(J4 acc, const std::vector<A4>& tab, const F4& s) {
        u64 k[4]; Fld<typename C::Fr>::to_canon(k, s);
        for (int w = 0; w < 64; w++) {
            const unsigned dgt = (unsigned)(k[w >> 4] >> (4 * (w & 15))) & 15u;
            if (dgt) acc = Grp<C>::madd(acc, tab[w * 15 + dgt - 1]);
        }
        return acc;
    }
    // the same before the final inversion (several commitments share one: Montgomery's trick at the caller)
    J4 commit_jac(const F4& v, const F4& blind) const {
        const FixedTables& ft = fixed_tables();
        if (B == ft.B && B_blinding == ft.Bb) return fixed_mul_acc(fixed_mul_acc(Grp<C>::inf(), ft.tB, v), ft.tBb, blind);
        return Grp<C>::add(Grp<C>::mul(B, v), Grp<C>::mul(B_blinding, blind));
    }
    A4 commit(const F4& v, const F4& blind) const {  // src/generators.rs:39-44
        const FixedTables& ft = fixed_tables();
        if (B == ft.B && B_blinding == ft.Bb) return Grp<C>::to_aff(fixed_mul_acc(fixed_mul_acc(Grp<C>::inf(), ft.tB, v), ft.tBb, blind));
        return Grp<C>::to_aff(Grp<C>::add(Grp<C>::mul(B, v), Grp<C>::mul(B_blinding, blind)));
    }
    // optional: all commitments of a statement in one call (the engine installs its GPU fixed-base kernel here)
    std::function<int(const F4* v, const F4* blind, size_t m, A4* out)> batch;
    int commit_many(const F4* v, const F4* blind, size_t m, A4* out) const {
        if (batch && m > 3) return batch(v, blind, m, out);   // (up to three commitments: the host's fixed-base tables, ~25 us each, beat a launch + a wait)
        for (size_t i = 0; i < m; i++) out[i] = commit(v[i], blind[i]);
        return BP_OK;
    }
};

// GeneratorsChain (src/generators.rs:71-121) for label 'G'|'H' || LE32(party).  The ChaCha20 word stream is
// consumed strictly in order, but the expensive part of each draw (a square root) does not feed back into
// the stream position, so: (1) one thread walks the stream and records every (x, greatest) attempt,
// (2) all threads test attempts in parallel, (3) successes are compacted in stream order.
template <class C> static void derive_generators(std::vector<A4>& out, char which, u32 party, size_t count, unsigned nthreads = 0) {
    u8 msg[20]; memcpy(msg, "GeneratorsChain", 15); msg[15] = (u8)which; memcpy(msg + 16, &party, 4);
    u8 h[64]; sha3_512(h, msg, 20);
    ChaChaRng prng(h);
    if (!nthreads) nthreads = host_pool_threads();
    out.clear(); out.reserve(count);
    struct Attempt { F4 x; bool greatest; };
    while (out.size() < count) {
        size_t want = (count - out.size()) * 2 + 64;
        std::vector<Attempt> att(want);
        for (auto& a : att) { a.x = rand_fe<typename C::Fq>(prng); a.greatest = ((int32_t)prng.next_u32()) < 0; }
        std::vector<A4> pts(want); std::vector<u8> ok(want);
        auto work = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) ok[i] = Grp<C>::from_x(pts[i], att[i].x, att[i].greatest) ? 1 : 0; };
        if (nthreads <= 1 || want < 256) work(0, want);
        else {
            std::vector<std::thread> th;
            size_t per = (want + nthreads - 1) / nthreads;
            for (unsigned t = 0; t < nthreads; t++) { size_t lo = t * per, hi = std::min(want, lo + per); if (lo < hi) th.emplace_back(work, lo, hi); }
            for (auto& t : th) t.join();
        }
        size_t used = want;
        for (size_t i = 0; i < want; i++) if (ok[i]) { out.push_back(pts[i]); if (out.size() == count) { used = i + 1; break; } }
        if (out.size() == count && used < want) {
            // the chain is only ever extended from a fresh stream (increase_capacity re-derives), so the
            // over-read attempts past `used` are simply dropped
        }
    }
}

// ---- R1CS recording (src/r1cs/linear_combination.rs, constraint_system.rs) -----------------------------------
enum VKind : u8 { VK_COMMITTED = 0, VK_LEFT = 1, VK_RIGHT = 2, VK_OUT = 3, VK_ONE = 4 };
struct Var { VKind k; u32 i; };
struct Term { Var v; F4 c; };
typedef std::vector<Term> LinComb;

enum { SC_SHUFFLE = 0, SC_RANGE = 1, SC_EXAMPLE = 2, SC_SQUARE_CHAIN = 3, SC_MULTI_RANGE = 4 };
static inline const char* scenario_label(int sc) {
    switch (sc) {
        case SC_SHUFFLE: return "ShuffleBenchmark";
        case SC_RANGE: return "RangeProofTest";
        case SC_EXAMPLE: return "R1CSExampleGadget";
        case SC_SQUARE_CHAIN: return "SquareChainBenchmark";
        default: return "MultiRangeBenchmark";
    }
}

// ---- canonical structure of a recording ---------------------------------------------------------------------
// Two verifier instances of one gadget record the same constraint STRUCTURE (which variable appears in which constraint) but
// may differ in coefficient VALUES: a public input enters as a constant, a randomized (phase-2) constraint carries the
// instance's own challenge (benches/r1cs_secq256k1.rs:47-52: `x - z`).  CanonState walks a recording in order and numbers the
// distinct coefficient values by first occurrence (+1 and -1 are flags, not numbers): instances of one gadget then produce the
// same id sequence — hence the same 128-bit structure digest — and a small per-instance table id -> value.  The GPU-resident
// constraint matrices ("templates", r1cs_host.inc) are keyed by the digest and evaluated with the instance's table.
static constexpr u32 CID_ONE = 0x80000000u, CID_MONE = 0x40000000u, CID_MASK = 0x3fffffffu;
// The key of a circuit template: two multiply/xorshift lanes over the canonical stream, a third, independently built fingerprint
// (position-weighted sum through a splitmix64 finaliser: not a function of the first two lanes' states) and the exact number of
// mixed words — four 64-bit quantities compared on every template lookup, on top of the (n, n1, q, ncoef) check there.
struct Digest {
    u64 a = 0, b = 0, c = 0, cnt = 0;
    bool operator==(const Digest& o) const { return a == o.a && b == o.b && c == o.c && cnt == o.cnt; }
    bool operator<(const Digest& o) const { return a != o.a ? a < o.a : b != o.b ? b < o.b : c != o.c ? c < o.c : cnt < o.cnt; }
};
struct CanonState {
    u64 ha = 0x9E3779B97F4A7C15ULL, hb = 0xC2B2AE3D27D4EB4FULL, hc = 0, cnt = 0;
    std::vector<F4> coefs;      // id -> value
    std::vector<u32> slots;     // open addressing over coefs (power-of-two size, 0xFFFFFFFF = empty)
    inline void mix(u64 x) {
        ha = (ha ^ x) * 0xFF51AFD7ED558CCDULL; ha ^= ha >> 32;
        hb = (hb + x) * 0x9FB21C651E98DF25ULL; hb = (hb << 27) | (hb >> 37);
        u64 z = x + (++cnt) * 0x9E3779B97F4A7C15ULL;   // splitmix64 of (word, position), summed: order-sensitive through the position
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        hc += z ^ (z >> 31);
    }
    static inline u64 hash_fe(const F4& c) { u64 h = c.v[0] * 0x9E3779B97F4A7C15ULL ^ c.v[1] * 0xC2B2AE3D27D4EB4FULL ^ c.v[2] * 0x165667B19E3779F9ULL ^ c.v[3]; return h ^ (h >> 29); }
    void grow() {
        const size_t nsz = slots.empty() ? 64 : slots.size() * 2;
        slots.assign(nsz, 0xFFFFFFFFu);
        for (u32 id = 0; id < coefs.size(); id++) { size_t p = hash_fe(coefs[id]) & (nsz - 1); while (slots[p] != 0xFFFFFFFFu) p = (p + 1) & (nsz - 1); slots[p] = id; }
    }
    inline u32 id_of(const F4& c) {
        if (coefs.size() * 2 >= slots.size()) grow();
        size_t p = hash_fe(c) & (slots.size() - 1);
        for (;;) {
            const u32 id = slots[p];
            if (id == 0xFFFFFFFFu) { slots[p] = (u32)coefs.size(); coefs.push_back(c); return (u32)coefs.size() - 1; }
            if (coefs[id] == c) return id;
            p = (p + 1) & (slots.size() - 1);
        }
    }
    inline u32 classify(const F4& c, const F4& one, const F4& mone) { return c == one ? CID_ONE : c == mone ? CID_MONE : id_of(c); }
    // fn(q, term, cid) for every term of constraints [0, nq) of (terms, off), numbered q0 + local index
    template <class Fn> void walk(const Term* terms, const size_t* off, size_t nq, size_t q0, const F4& one, const F4& mone, Fn&& fn) {
        for (size_t q = 0; q < nq; q++) {
            mix(0xC0000000ULL + (off[q + 1] - off[q]));
            for (size_t k = off[q]; k < off[q + 1]; k++) {
                const Term& t = terms[k];
                const u32 cid = classify(t.c, one, mone);
                mix(((u64)t.v.k << 32) | t.v.i); mix(cid);
                fn(q0 + q, t, cid);
            }
        }
    }
    Digest digest(size_t n1, size_t n, size_t nq) const {
        CanonState t; t.ha = ha; t.hb = hb; t.hc = hc; t.cnt = cnt;
        t.mix(n1); t.mix(n); t.mix(nq); t.mix(coefs.size());
        Digest d; d.a = t.ha; d.b = t.hb; d.c = t.hc; d.cnt = t.cnt;
        return d;
    }
};
struct VTerm { u32 j, q; F4 c; };   // committed variable j in constraint q with coefficient c: feeds wV (verifier.rs:334-338, prover.rs:385-387)

// A phase-1 recording that several verifier instances share (bp_verifier_new_like): immutable once frozen.
struct FrozenRecording {
    std::vector<Term> terms;
    std::vector<size_t> off{0};
    size_t num_vars = 0;
    bool has_pending = false; size_t pending = 0;
    CanonState canon;              // state after walking every constraint of this part
    std::vector<VTerm> vterms;     // in constraint order
    size_t nq() const { return off.size() - 1; }
};

// The recorder shared by prover and verifier: the verifier records the same constraints without
// assignments.  Constraints are stored flat (CSR-like) because circuits reach 2^22 multipliers.
template <class C> struct ConstraintSystem {
    typedef Fld<typename C::Fr> S;
    Transcript* tr = nullptr;
    bool proving = false;
    void* owner = nullptr;         // the C-ABI handle that wraps this recorder (randomized closures of a C caller receive it)
    // an optional shared phase-1 part (constraints [0, base->nq()), multipliers [0, base->num_vars)); what this instance records
    // itself follows: constraint base->nq() + q owns terms [cs_off[q], cs_off[q+1])
    std::shared_ptr<const FrozenRecording> base;
    std::vector<Term> cs_terms;
    std::vector<size_t> cs_off{0};
    size_t num_vars = 0;           // all multipliers, the base's included
    size_t n1 = 0;                 // multipliers allocated before the randomized phase (set by run_randomized)
    bool phase2 = false;
    bool has_pending = false; size_t pending = 0;
    std::vector<std::function<int(ConstraintSystem&)>> deferred;
    // prover secrets
    std::vector<F4> v, v_blinding, a_L, a_R, a_O;
    // verifier commitments
    std::vector<A4> V;

    F4 one() const { return S::one(); }
    LinComb lc_var(Var x) const { return LinComb{Term{x, S::one()}}; }
    LinComb lc_const(const F4& c) const { return LinComb{Term{Var{VK_ONE, 0}, c}}; }
    static void lc_sub(LinComb& a, const LinComb& b) { for (auto& t : b) a.push_back(Term{t.v, S::neg(t.c)}); }
    static void lc_add(LinComb& a, const LinComb& b) { a.insert(a.end(), b.begin(), b.end()); }

    // The recording primitives take (terms, count): gadget code keeps its short linear combinations on the stack, and a C caller's
    // arrays arrive in this shape (bp_cs_multiply / bp_cs_constrain); the LinComb overloads below are conveniences.
    F4 eval(const Term* lc, size_t cnt) const {  // prover.rs:399-414
        F4 acc = S::zero();
        const F4 one = S::one();
        for (size_t j = 0; j < cnt; j++) {
            const Term& t = lc[j];
            const F4* val;
            switch (t.v.k) {
                case VK_LEFT: val = &a_L[t.v.i]; break;
                case VK_RIGHT: val = &a_R[t.v.i]; break;
                case VK_OUT: val = &a_O[t.v.i]; break;
                case VK_COMMITTED: val = &v[t.v.i]; break;
                default: val = &one; break;
            }
            acc = S::add(acc, t.c == one ? *val : S::mul(t.c, *val));
        }
        return acc;
    }
    F4 eval(const LinComb& lc) const { return eval(lc.data(), lc.size()); }
    // room for a gadget of known size (the vectors otherwise grow by doubling: ~2x peak memory at 2^20+ multipliers)
    void reserve(size_t multipliers, size_t constraints, size_t terms) {
        reserve_huge(cs_terms, cs_terms.size() + terms); reserve_huge(cs_off, cs_off.size() + constraints);
        if (proving) { reserve_huge(a_L, a_L.size() + multipliers); reserve_huge(a_R, a_R.size() + multipliers); reserve_huge(a_O, a_O.size() + multipliers); }
    }
    void constrain(const Term* lc, size_t cnt) { cs_terms.insert(cs_terms.end(), lc, lc + cnt); cs_off.push_back(cs_terms.size()); }
    void constrain(const LinComb& lc) { constrain(lc.data(), lc.size()); }
    // multiply (prover.rs:103-133 / verifier.rs:74-98): constraints left - l = 0 and right - r = 0
    void multiply(const Term* left, size_t nl, const Term* right, size_t nr, Var out[3]) {
        u32 i = (u32)num_vars++;
        if (proving) { F4 l = eval(left, nl), r = eval(right, nr); a_L.push_back(l); a_R.push_back(r); a_O.push_back(S::mul(l, r)); }
        out[0] = Var{VK_LEFT, i}; out[1] = Var{VK_RIGHT, i}; out[2] = Var{VK_OUT, i};
        const F4 m1 = S::neg(S::one());
        cs_terms.insert(cs_terms.end(), left, left + nl); cs_terms.push_back(Term{out[0], m1}); cs_off.push_back(cs_terms.size());
        cs_terms.insert(cs_terms.end(), right, right + nr); cs_terms.push_back(Term{out[1], m1}); cs_off.push_back(cs_terms.size());
    }
    void multiply(const LinComb& left, const LinComb& right, Var out[3]) { multiply(left.data(), left.size(), right.data(), right.size(), out); }
    // allocate (prover.rs:135-157 / verifier.rs:100-116)
    int allocate(const F4* assignment, Var& out) {
        if (proving && !assignment) return BP_E_MISSING;
        if (!has_pending) {
            u32 i = (u32)num_vars++;
            has_pending = true; pending = i;
            if (proving) { a_L.push_back(*assignment); a_R.push_back(S::zero()); a_O.push_back(S::zero()); }
            out = Var{VK_LEFT, i};
        } else {
            has_pending = false;
            if (proving) { a_R[pending] = *assignment; a_O[pending] = S::mul(a_L[pending], a_R[pending]); }
            out = Var{VK_RIGHT, (u32)pending};
        }
        return BP_OK;
    }
    // allocate_multiplier (prover.rs:159-183 / verifier.rs:118-138)
    int allocate_multiplier(const F4* l, const F4* r, Var out[3]) {
        if (proving && (!l || !r)) return BP_E_MISSING;
        u32 i = (u32)num_vars++;
        if (proving) { a_L.push_back(*l); a_R.push_back(*r); a_O.push_back(S::mul(*l, *r)); }
        out[0] = Var{VK_LEFT, i}; out[1] = Var{VK_RIGHT, i}; out[2] = Var{VK_OUT, i};
        return BP_OK;
    }
    void specify_randomized_constraints(std::function<int(ConstraintSystem&)> cb) { deferred.push_back(std::move(cb)); }
    F4 challenge_scalar(const char* label) { return TP<C>::challenge_scalar(*tr, label); }
    // create_randomized_constraints (prover.rs:418-441 / verifier.rs:353-376)
    int run_randomized() {
        has_pending = false;
        n1 = num_vars; phase2 = true;
        if (deferred.empty()) { TP<C>::r1cs_1phase_domain_sep(*tr); return BP_OK; }
        TP<C>::r1cs_2phase_domain_sep(*tr);
        std::vector<std::function<int(ConstraintSystem&)>> cbs; cbs.swap(deferred);
        for (auto& cb : cbs) { int rc = cb(*this); if (rc) return rc; }
        return BP_OK;
    }
    size_t base_nq() const { return base ? base->nq() : 0; }
    size_t num_constraints() const { return base_nq() + cs_off.size() - 1; }
    size_t num_terms() const { return (base ? base->terms.size() : 0) + cs_terms.size(); }
    // fn(q, term) over every recorded term, the shared part first
    template <class Fn> void for_each_term(Fn&& fn) const {
        if (base) for (size_t q = 0; q < base->nq(); q++) for (size_t k = base->off[q]; k < base->off[q + 1]; k++) fn(q, base->terms[k]);
        const size_t q0 = base_nq();
        for (size_t q = 0; q + 1 < cs_off.size(); q++) for (size_t k = cs_off[q]; k < cs_off[q + 1]; k++) fn(q0 + q, cs_terms[k]);
    }
    // Freezes what has been recorded so far (phase 1) into a shareable part; this instance continues on top of it.
    int freeze() {
        if (phase2 || proving) return BP_E_ARG;
        if (base && cs_terms.empty() && cs_off.size() == 1 && num_vars == base->num_vars) return BP_OK;   // already frozen, nothing new
        if (base) return BP_E_ARG;                         // (a second layer is not needed: like-instances record nothing in phase 1)
        auto f = std::make_shared<FrozenRecording>();
        f->terms.swap(cs_terms); f->off.swap(cs_off); f->num_vars = num_vars; f->has_pending = has_pending; f->pending = pending;
        cs_off.assign(1, 0);
        const F4 one = S::one(), mone = S::neg(S::one());
        f->canon.walk(f->terms.data(), f->off.data(), f->nq(), 0, one, mone, [&](size_t q, const Term& t, u32) {
            if (t.v.k == VK_COMMITTED) f->vterms.push_back(VTerm{t.v.i, (u32)q, t.c});
        });
        base = f;
        return BP_OK;
    }
    // a fresh verifier instance of the same gadget: shares `src`'s frozen phase-1 recording and its deferred closures
    void init_like(const ConstraintSystem& src) {
        base = src.base; num_vars = src.num_vars; has_pending = src.has_pending; pending = src.pending; deferred = src.deferred; proving = false;
    }
    // Canonical form of the whole recording (shared part + own part).  st_own is scratch for instances that recorded something
    // themselves; the returned state is the base's cached one otherwise (no copy).  own_vterms: committed-variable terms of the own part.
    const CanonState& canonical(CanonState& st_own, std::vector<VTerm>* own_vterms) const {
        const bool own_empty = cs_off.size() == 1;
        if (base && own_empty) return base->canon;
        if (base) st_own = base->canon; else st_own = CanonState();
        const F4 one = S::one(), mone = S::neg(S::one());
        st_own.walk(cs_terms.data(), cs_off.data(), cs_off.size() - 1, base_nq(), one, mone, [&](size_t q, const Term& t, u32) {
            if (own_vterms && t.v.k == VK_COMMITTED) own_vterms->push_back(VTerm{t.v.i, (u32)q, t.c});
        });
        return st_own;
    }
    // flattened_constraints (prover.rs:354-397 / verifier.rs:304-349): w* = sum_q z^(q+1) * W*[q, .]
    void flatten(const F4& z, std::vector<F4>& wL, std::vector<F4>& wR, std::vector<F4>& wO, std::vector<F4>& wV, F4& wc, size_t m) const {
        size_t n = num_vars;
        wL.assign(n, S::zero()); wR.assign(n, S::zero()); wO.assign(n, S::zero()); wV.assign(m, S::zero()); wc = S::zero();
        F4 ez = z, nez = S::neg(z);
        size_t cur_q = 0;
        const F4 one = S::one(), minus_one = S::neg(S::one());
        for_each_term([&](size_t q, const Term& t) {
            while (cur_q < q) { ez = S::mul(ez, z); cur_q++; nez = S::neg(ez); }
            // most coefficients of real circuits are +-1: no multiplication needed for those
            F4 p = t.c == one ? ez : t.c == minus_one ? nez : S::mul(ez, t.c);
            switch (t.v.k) {
                case VK_LEFT: wL[t.v.i] = S::add(wL[t.v.i], p); break;
                case VK_RIGHT: wR[t.v.i] = S::add(wR[t.v.i], p); break;
                case VK_OUT: wO[t.v.i] = S::add(wO[t.v.i], p); break;
                case VK_COMMITTED: wV[t.v.i] = S::sub(wV[t.v.i], p); break;
                default: wc = S::sub(wc, p); break;
            }
        });
    }
};

// ---- gadgets (benches/r1cs_secq256k1.rs:34-76; tests/r1cs_secq256k1.rs:216-228, 361-393) -----------------------
template <class C> static int shuffle_gadget(ConstraintSystem<C>& cs, std::vector<Var> x, std::vector<Var> y) {
    typedef ConstraintSystem<C> CS;
    size_t k = x.size();
    if (k != y.size()) return BP_E_ARG;
    if (k == 1) { LinComb lc = cs.lc_var(y[0]); CS::lc_sub(lc, cs.lc_var(x[0])); cs.constrain(lc); return BP_OK; }
    cs.specify_randomized_constraints([x, y, k](CS& cs) -> int {
        F4 z = cs.challenge_scalar("shuffle challenge");
        auto mz = [&](Var v) { LinComb l = cs.lc_var(v); CS::lc_sub(l, cs.lc_const(z)); return l; };
        Var o[3];
        cs.multiply(mz(x[k - 1]), mz(x[k - 2]), o);
        Var prev = o[2];
        for (size_t i = k - 2; i-- > 0;) { cs.multiply(cs.lc_var(prev), mz(x[i]), o); prev = o[2]; }
        Var fx = prev;
        cs.multiply(mz(y[k - 1]), mz(y[k - 2]), o);
        prev = o[2];
        for (size_t i = k - 2; i-- > 0;) { cs.multiply(cs.lc_var(prev), mz(y[i]), o); prev = o[2]; }
        LinComb lc = cs.lc_var(fx); CS::lc_sub(lc, cs.lc_var(prev)); cs.constrain(lc);
        return BP_OK;
    });
    return BP_OK;
}
template <class C> static int range_gadget(ConstraintSystem<C>& cs, LinComb v, const u64* assignment, size_t nbits) {
    typedef ConstraintSystem<C> CS; typedef Fld<typename C::Fr> S;
    F4 exp2 = S::one();
    for (size_t i = 0; i < nbits; i++) {
        Var abo[3]; int rc;
        if (assignment) { u64 bit = (*assignment >> i) & 1; F4 l = S::from_u64(1 - bit), r = S::from_u64(bit); rc = cs.allocate_multiplier(&l, &r, abo); }
        else rc = cs.allocate_multiplier(nullptr, nullptr, abo);
        if (rc) return rc;
        cs.constrain(cs.lc_var(abo[2]));
        LinComb ab = cs.lc_var(abo[0]); CS::lc_add(ab, cs.lc_var(abo[1])); CS::lc_sub(ab, cs.lc_const(S::one())); cs.constrain(ab);
        v.push_back(Term{abo[1], S::neg(exp2)});
        exp2 = S::dbl(exp2);
    }
    cs.constrain(v);
    return BP_OK;
}

struct StatementIO {
    std::vector<A4> commitments;
    std::vector<F4> publics;
};

// Statement construction, prover side.  RNG consumption order is part of the scenario definition.
template <class C> static int scenario_prover(ConstraintSystem<C>& cs, const PedersenGens<C>& pc, ChaChaRng& prng, int sc, const u64* prm, StatementIO& io) {
    typedef Fld<typename C::Fr> S; typedef typename C::Fr FrP; typedef ConstraintSystem<C> CS;
    auto commit = [&](const F4& val, const F4& blind) {  // Prover::commit (prover.rs:327-341)
        u32 i = (u32)cs.v.size();
        cs.v.push_back(val); cs.v_blinding.push_back(blind);
        A4 Vp = pc.commit(val, blind);
        TP<C>::append_point(*cs.tr, "V", Vp);
        io.commitments.push_back(Vp);
        return Var{VK_COMMITTED, i};
    };
    // a run of Prover::commit calls whose points do not depend on each other: the group work is one batch, the transcript
    // appends and variable numbering keep the reference's order
    auto commit_many = [&](const std::vector<F4>& vals, const std::vector<F4>& blinds, std::vector<Var>& vars) -> int {
        std::vector<A4> pts(vals.size());
        int rc = pc.commit_many(vals.data(), blinds.data(), vals.size(), pts.data());
        if (rc) return rc;
        vars.resize(vals.size());
        for (size_t i = 0; i < vals.size(); i++) {
            u32 idx = (u32)cs.v.size();
            cs.v.push_back(vals[i]); cs.v_blinding.push_back(blinds[i]);
            TP<C>::append_point(*cs.tr, "V", pts[i]);
            io.commitments.push_back(pts[i]);
            vars[i] = Var{VK_COMMITTED, idx};
        }
        return BP_OK;
    };
    switch (sc) {
        case SC_SHUFFLE: {
            size_t k = prm[0];
            std::vector<F4> in(k), out(k), bl(k);
            for (auto& x : in) x = S::from_u64(prng.next_u64());
            for (size_t i = 0; i < k; i++) out[i] = in[(i + 1) % k];
            cs.tr->append_message("dom-sep", "ShuffleProof"); cs.tr->append_u64("k", k);
            // Prover::commit for the k inputs, then the k outputs (benches/r1cs_secq256k1.rs:160-175): the points do not depend on each
            // other, so all 2k go to the engine as ONE batch; the blinding draws, the transcript appends and the variable numbering
            // keep the reference's order
            std::vector<F4> vals(2 * k), bls(2 * k);
            for (size_t i = 0; i < k; i++) { vals[i] = in[i]; bls[i] = rand_fe<FrP>(prng); }
            for (size_t i = 0; i < k; i++) { vals[k + i] = out[i]; bls[k + i] = rand_fe<FrP>(prng); }
            std::vector<Var> all;
            int rc = commit_many(vals, bls, all); if (rc) return rc;
            std::vector<Var> xv(all.begin(), all.begin() + k), yv(all.begin() + k, all.end());
            return shuffle_gadget<C>(cs, xv, yv);
        }
        case SC_RANGE: {
            u64 val = prm[1];
            F4 b = rand_fe<FrP>(prng);
            Var var = commit(S::from_u64(val), b);
            return range_gadget<C>(cs, cs.lc_var(var), &val, prm[0]);
        }
        case SC_EXAMPLE: {
            Var vars[5];
            for (int i = 0; i < 5; i++) { F4 b = rand_fe<FrP>(prng); vars[i] = commit(S::from_u64(prm[i]), b); }
            LinComb a = cs.lc_var(vars[0]); CS::lc_add(a, cs.lc_var(vars[1]));
            LinComb bb = cs.lc_var(vars[2]); CS::lc_add(bb, cs.lc_var(vars[3]));
            Var o[3]; cs.multiply(a, bb, o);
            LinComb c = cs.lc_var(vars[4]); CS::lc_add(c, cs.lc_const(S::from_u64(prm[5]))); CS::lc_sub(c, cs.lc_var(o[2]));
            cs.constrain(c);
            return BP_OK;
        }
        case SC_SQUARE_CHAIN: {
            size_t N = prm[0];
            F4 x0 = S::from_u64(prng.next_u64()), b = rand_fe<FrP>(prng), last = x0;
            for (size_t i = 0; i < N; i++) last = S::sqr(last);
            if (prm[1]) last = S::add(last, S::one());
            Var cur = commit(x0, b), o[3];
            io.publics.push_back(last);
            cs.reserve(N, 2 * N + 1, 4 * N + 2);
            for (size_t i = 0; i < N; i++) { const Term t{cur, S::one()}; cs.multiply(&t, 1, &t, 1, o); cur = o[2]; }
            LinComb lc = cs.lc_var(cur); CS::lc_sub(lc, cs.lc_const(last)); cs.constrain(lc);
            return BP_OK;
        }
        case SC_MULTI_RANGE: {
            size_t count = prm[0], nbits = prm[1];
            std::vector<u64> raw(count);
            std::vector<F4> vals(count), bl(count);
            for (size_t j = 0; j < count; j++) {   // the gadget draws nothing and appends nothing in phase 1: commits can go first
                u64 val = prng.next_u64();
                if (nbits < 64) val &= (((u64)1 << nbits) - 1);
                if (prm[2] && j == count - 1) val = nbits < 64 ? ((u64)1 << nbits) : val;
                raw[j] = val; vals[j] = S::from_u64(val); bl[j] = rand_fe<FrP>(prng);
            }
            std::vector<Var> vars;
            int rc = commit_many(vals, bl, vars); if (rc) return rc;
            for (size_t j = 0; j < count; j++) {
                rc = range_gadget<C>(cs, cs.lc_var(vars[j]), &raw[j], nbits);
                if (rc) return rc;
            }
            return BP_OK;
        }
    }
    return BP_E_ARG;
}
template <class C> static int scenario_verifier(ConstraintSystem<C>& cs, int sc, const u64* prm, const StatementIO& io) {
    typedef Fld<typename C::Fr> S; typedef ConstraintSystem<C> CS;
    auto commit = [&](const A4& Vp) {  // Verifier::commit (verifier.rs:279-287)
        u32 i = (u32)cs.V.size(); cs.V.push_back(Vp); TP<C>::append_point(*cs.tr, "V", Vp); return Var{VK_COMMITTED, i};
    };
    switch (sc) {
        case SC_SHUFFLE: {
            size_t k = prm[0];
            if (io.commitments.size() != 2 * k) return BP_E_ARG;
            cs.tr->append_message("dom-sep", "ShuffleProof"); cs.tr->append_u64("k", k);
            std::vector<Var> xv(k), yv(k);
            for (size_t i = 0; i < k; i++) xv[i] = commit(io.commitments[i]);
            for (size_t i = 0; i < k; i++) yv[i] = commit(io.commitments[k + i]);
            return shuffle_gadget<C>(cs, xv, yv);
        }
        case SC_RANGE: { if (io.commitments.size() != 1) return BP_E_ARG; Var var = commit(io.commitments[0]); return range_gadget<C>(cs, cs.lc_var(var), nullptr, prm[0]); }
        case SC_EXAMPLE: {
            if (io.commitments.size() != 5) return BP_E_ARG;
            Var vars[5];
            for (int i = 0; i < 5; i++) vars[i] = commit(io.commitments[i]);
            LinComb a = cs.lc_var(vars[0]); CS::lc_add(a, cs.lc_var(vars[1]));
            LinComb bb = cs.lc_var(vars[2]); CS::lc_add(bb, cs.lc_var(vars[3]));
            Var o[3]; cs.multiply(a, bb, o);
            LinComb c = cs.lc_var(vars[4]); CS::lc_add(c, cs.lc_const(S::from_u64(prm[5]))); CS::lc_sub(c, cs.lc_var(o[2]));
            cs.constrain(c);
            return BP_OK;
        }
        case SC_SQUARE_CHAIN: {
            if (io.commitments.size() != 1 || io.publics.size() != 1) return BP_E_ARG;
            Var cur = commit(io.commitments[0]), o[3];
            cs.reserve(prm[0], 2 * prm[0] + 1, 4 * prm[0] + 2);
            for (size_t i = 0; i < prm[0]; i++) { const Term t{cur, S::one()}; cs.multiply(&t, 1, &t, 1, o); cur = o[2]; }
            LinComb lc = cs.lc_var(cur); CS::lc_sub(lc, cs.lc_const(io.publics[0])); cs.constrain(lc);
            return BP_OK;
        }
        case SC_MULTI_RANGE: {
            if (io.commitments.size() != prm[0]) return BP_E_ARG;
            for (size_t j = 0; j < prm[0]; j++) { Var var = commit(io.commitments[j]); int rc = range_gadget<C>(cs, cs.lc_var(var), nullptr, prm[1]); if (rc) return rc; }
            return BP_OK;
        }
    }
    return BP_E_ARG;
}

// The transcript side of statement construction alone (Verifier::commit appends, plus the scenario's own domain separator):
// what remains per instance when the constraint matrices come from a cached template.  Only for single-phase scenarios.
// what a scenario's verifier puts into the transcript before its commitments; *expect = the number of commitments it takes
template <class C> static int scenario_verifier_prefix(Transcript& tr, int sc, const u64* prm, const StatementIO& io, size_t* expect) {
    *expect = 0;
    switch (sc) {
        case SC_SHUFFLE:   // only k = 1 is single-phase (benches/r1cs_secq256k1.rs:43-46); same prefix as scenario_verifier
            *expect = 2 * prm[0];
            tr.append_message("dom-sep", "ShuffleProof"); tr.append_u64("k", prm[0]);
            break;
        case SC_RANGE: *expect = 1; break;
        case SC_EXAMPLE: *expect = 5; break;
        case SC_SQUARE_CHAIN: *expect = 1; if (io.publics.size() != 1) return BP_E_ARG; break;
        case SC_MULTI_RANGE: *expect = prm[0]; break;
        default: return BP_E_ARG;
    }
    return BP_OK;
}
template <class C> static int scenario_verifier_transcript(Transcript& tr, int sc, const u64* prm, const StatementIO& io) {
    size_t expect = 0;
    const int rc = scenario_verifier_prefix<C>(tr, sc, prm, io, &expect);
    if (rc) return rc;
    if (io.commitments.size() != expect) return BP_E_ARG;
    for (auto& Vp : io.commitments) TP<C>::append_point(tr, "V", Vp);
    return BP_OK;
}

// ---- R1CSProof (src/r1cs/proof.rs:27-91) ---------------------------------------------------------------------
struct ProofData {
    A4 A_I1, A_O1, S1, A_I2, A_O2, S2, T_1, T_3, T_4, T_5, T_6;
    F4 t_x, t_x_blinding, e_blinding;
    std::vector<A4> L_vec, R_vec;
    F4 a, b;
};
template <class C> static std::vector<u8> proof_to_bytes(const ProofData& p) {
    typedef Fld<typename C::Fr> S;
    std::vector<u8> out;
    auto pt = [&](const A4& q) { u8 b[33]; Grp<C>::ser_compressed(b, q); out.insert(out.end(), b, b + 33); };
    auto sc = [&](const F4& s) { u8 b[32]; S::to_bytes(b, s); out.insert(out.end(), b, b + 32); };
    auto vec = [&](const std::vector<A4>& v) { u64 n = v.size(); out.insert(out.end(), (u8*)&n, (u8*)&n + 8); for (auto& q : v) pt(q); };
    pt(p.A_I1); pt(p.A_O1); pt(p.S1); pt(p.A_I2); pt(p.A_O2); pt(p.S2); pt(p.T_1); pt(p.T_3); pt(p.T_4); pt(p.T_5); pt(p.T_6);
    sc(p.t_x); sc(p.t_x_blinding); sc(p.e_blinding); vec(p.L_vec); vec(p.R_vec); sc(p.a); sc(p.b);
    return out;
}
template <class C> static int proof_from_bytes(ProofData& p, const u8* d, size_t n) {
    typedef Fld<typename C::Fr> S;
    size_t pos = 0;
    auto pt = [&](A4& q) { if (pos + 33 > n) return false; bool ok = Grp<C>::deser_compressed(q, d + pos); pos += 33; return ok; };
    auto sc = [&](F4& s) { if (pos + 32 > n) return false; bool ok = S::from_bytes(s, d + pos); pos += 32; return ok; };
    auto vec = [&](std::vector<A4>& v) {
        if (pos + 8 > n) return false;
        u64 len; memcpy(&len, d + pos, 8); pos += 8;
        if (len > (n - pos) / 33) return false;
        v.resize(len);
        for (auto& q : v) if (!pt(q)) return false;
        return true;
    };
    bool ok = pt(p.A_I1) && pt(p.A_O1) && pt(p.S1) && pt(p.A_I2) && pt(p.A_O2) && pt(p.S2) && pt(p.T_1) && pt(p.T_3) && pt(p.T_4) && pt(p.T_5) &&
              pt(p.T_6) && sc(p.t_x) && sc(p.t_x_blinding) && sc(p.e_blinding) && vec(p.L_vec) && vec(p.R_vec) && sc(p.a) && sc(p.b);
    return ok ? BP_OK : BP_E_FORMAT;
}

// Two-step decoding for batches: parse_lazy validates the framing and extracts (x, flag) of every compressed point without
// taking square roots; the y coordinates come back from the GPU (k_points_decompress) and finish_lazy assembles ProofData.
struct LazyProof {
    std::vector<F4> xs;        // Montgomery words of x, point order: A_I1,A_O1,S1,A_I2,A_O2,S2,T_1,T_3,T_4,T_5,T_6, L[k], R[k]
    std::vector<u32> flags;    // ark-serialize flag byte per point
    F4 t_x, t_x_blinding, e_blinding, a, b;
    size_t k = 0;
};
template <class C> static int proof_parse_lazy(LazyProof& lp, const u8* d, size_t n) {
    typedef Fld<typename C::Fq> Fq; typedef Fld<typename C::Fr> S;
    size_t pos = 0;
    auto pt = [&]() {
        if (pos + 33 > n) return false;
        const u8 fl = d[pos + 32];
        F4 x;
        bool ok = !(fl & 0x3f) && (fl & 0xc0) != 0xc0 && Fq::from_bytes(x, d + pos);
        if (ok && (fl & 0x40) && !x.is_zero()) ok = false;
        pos += 33;
        if (ok) { lp.xs.push_back(x); lp.flags.push_back(fl); }
        return ok;
    };
    auto sc = [&](F4& s) { if (pos + 32 > n) return false; bool ok = S::from_bytes(s, d + pos); pos += 32; return ok; };
    auto vec = [&](size_t& len_out) {
        if (pos + 8 > n) return false;
        u64 len; memcpy(&len, d + pos, 8); pos += 8;
        if (len > (n - pos) / 33) return false;
        for (u64 i = 0; i < len; i++) if (!pt()) return false;
        len_out = (size_t)len;
        return true;
    };
    for (int i = 0; i < 11; i++) if (!pt()) return BP_E_FORMAT;
    size_t kl = 0, kr = 0;
    bool ok = sc(lp.t_x) && sc(lp.t_x_blinding) && sc(lp.e_blinding) && vec(kl) && vec(kr) && sc(lp.a) && sc(lp.b);
    if (!ok) return BP_E_FORMAT;
    lp.k = kl;
    // L_vec and R_vec lengths may differ in a malformed proof; the verifier then fails (zip in verification_scalars)
    if (kl != kr) { lp.k = (size_t)-1; }
    return BP_OK;
}
static inline void proof_finish_lazy(ProofData& p, const LazyProof& lp, const A4* pts) {
    p.A_I1 = pts[0]; p.A_O1 = pts[1]; p.S1 = pts[2]; p.A_I2 = pts[3]; p.A_O2 = pts[4]; p.S2 = pts[5];
    p.T_1 = pts[6]; p.T_3 = pts[7]; p.T_4 = pts[8]; p.T_5 = pts[9]; p.T_6 = pts[10];
    p.t_x = lp.t_x; p.t_x_blinding = lp.t_x_blinding; p.e_blinding = lp.e_blinding; p.a = lp.a; p.b = lp.b;
    const size_t total = lp.xs.size() - 11;
    const size_t kl = lp.k == (size_t)-1 ? total : lp.k;   // unequal lengths: keep everything in L so the length check fails
    p.L_vec.assign(pts + 11, pts + 11 + std::min(kl, total));
    p.R_vec.assign(pts + 11 + std::min(kl, total), pts + 11 + total);
}

static inline size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

}  // namespace host
}  // namespace arkbp
