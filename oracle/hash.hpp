// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).  "parity unpinned".
//
// Byte-level primitives the reference takes from un-vendored crates:
//   sha3 0.10 (SHA3-512; src/generators.rs:52-54,79-82), rand_chacha 0.3 (ChaCha20Rng;
//   src/transcript.rs:99, src/generators.rs:59,87), merlin 3.0 (STROBE-128 transcript;
//   src/transcript.rs:45-102, src/r1cs/prover.rs:484-493).
// Pinned by: NIST SHA3-512 (hashlib), the ChaCha20 zero-key keystream, merlin's
// "test protocol" vector (tests/test_oracle_vectors.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include "field.hpp"

namespace orc {

// ---- Keccak-f[1600] ---------------------------------------------------------------------
static inline u64 rotl64(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static inline void keccak_f1600(u64 st[25]) {
    static const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; round++) {
        u64 C[5], D[5], B[25];
        for (int x = 0; x < 5; x++) C[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) st[i] ^= D[i % 5];
        // rho + pi: B[y, 2x+3y] = rot(A[x,y])
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(st[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) st[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
        st[0] ^= RC[round];
    }
}

// ---- SHA3-512 (rate 72, domain 0x06) -------------------------------------------------------
static inline void sha3_512(u8 out[64], const u8* msg, size_t len) {
    u64 st[25];
    memset(st, 0, sizeof st);
    u8* sb = (u8*)st;
    const size_t rate = 72;
    size_t pos = 0;
    for (size_t i = 0; i < len; i++) {
        sb[pos++] ^= msg[i];
        if (pos == rate) { keccak_f1600(st); pos = 0; }
    }
    sb[pos] ^= 0x06;
    sb[rate - 1] ^= 0x80;
    keccak_f1600(st);
    memcpy(out, sb, 64);
}

// ---- ChaCha20Rng (rand_chacha 0.3): key = seed, 64-bit block counter (words 12,13), stream 0;
// the u32 word stream is consumed strictly sequentially; next_u64 = lo word then hi word.
struct ChaCha20Rng {
    u32 key[8];
    u64 counter;
    u32 buf[16];
    int idx;
    void seed(const u8 s[32]) {
        memcpy(key, s, 32);
        counter = 0;
        idx = 16;
    }
    static inline u32 rotl(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
    void refill() {
        u32 s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
        for (int i = 0; i < 8; i++) s[4 + i] = key[i];
        s[12] = (u32)counter; s[13] = (u32)(counter >> 32); s[14] = 0; s[15] = 0;
        u32 x[16];
        memcpy(x, s, 64);
#define ORC_QR(a, b, c, d) \
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        for (int r = 0; r < 10; r++) {
            ORC_QR(0, 4, 8, 12) ORC_QR(1, 5, 9, 13) ORC_QR(2, 6, 10, 14) ORC_QR(3, 7, 11, 15)
            ORC_QR(0, 5, 10, 15) ORC_QR(1, 6, 11, 12) ORC_QR(2, 7, 8, 13) ORC_QR(3, 4, 9, 14)
        }
#undef ORC_QR
        for (int i = 0; i < 16; i++) buf[i] = x[i] + s[i];
        counter++;
        idx = 0;
    }
    u32 next_u32() {
        if (idx == 16) refill();
        return buf[idx++];
    }
    u64 next_u64() {
        u64 lo = next_u32();
        u64 hi = next_u32();
        return lo | (hi << 32);
    }
    void fill_bytes(u8* dst, size_t n) {  // rand_core fill_via_u32_chunks: whole words, LE
        while (n) {
            u32 w = next_u32();
            size_t k = n < 4 ? n : 4;
            memcpy(dst, &w, k);
            dst += k; n -= k;
        }
    }
};

// ---- STROBE-128 as used by merlin 3.0 ------------------------------------------------------
struct Strobe128 {
    u8 st[200];
    u8 pos, pos_begin, cur_flags;
    static const int R = 166;
    enum { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };

    void init(const u8* label, size_t n) {
        memset(st, 0, 200);
        const u8 hdr[6] = {1, R + 2, 1, 0, 1, 96};
        memcpy(st, hdr, 6);
        memcpy(st + 6, "STROBEv1.0.2", 12);
        u64 w[25]; memcpy(w, st, 200); keccak_f1600(w); memcpy(st, w, 200);
        pos = 0; pos_begin = 0; cur_flags = 0;
        meta_ad(label, n, false);
    }
    void run_f() {
        st[pos] ^= pos_begin;
        st[pos + 1] ^= 0x04;
        st[R + 1] ^= 0x80;
        u64 w[25]; memcpy(w, st, 200); keccak_f1600(w); memcpy(st, w, 200);
        pos = 0; pos_begin = 0;
    }
    void absorb(const u8* d, size_t n) {
        for (size_t i = 0; i < n; i++) { st[pos++] ^= d[i]; if (pos == R) run_f(); }
    }
    void overwrite(const u8* d, size_t n) {
        for (size_t i = 0; i < n; i++) { st[pos++] = d[i]; if (pos == R) run_f(); }
    }
    void squeeze(u8* d, size_t n) {
        for (size_t i = 0; i < n; i++) { d[i] = st[pos]; st[pos++] = 0; if (pos == R) run_f(); }
    }
    void begin_op(u8 flags, bool more) {
        if (more) return;  // continuation of the current op (flags must match)
        u8 old_begin = pos_begin;
        pos_begin = pos + 1;
        cur_flags = flags;
        u8 hdr[2] = {old_begin, flags};
        absorb(hdr, 2);
        bool force_f = (flags & (FLAG_C | FLAG_K)) != 0;
        if (force_f && pos != 0) run_f();
    }
    void meta_ad(const u8* d, size_t n, bool more) { begin_op(FLAG_M | FLAG_A, more); absorb(d, n); }
    void ad(const u8* d, size_t n, bool more) { begin_op(FLAG_A, more); absorb(d, n); }
    void prf(u8* d, size_t n, bool more) { begin_op(FLAG_I | FLAG_A | FLAG_C, more); squeeze(d, n); }
    void key(const u8* d, size_t n, bool more) { begin_op(FLAG_A | FLAG_C, more); overwrite(d, n); }
};

// ---- merlin::Transcript / TranscriptRng ------------------------------------------------------
struct Transcript {
    Strobe128 s;
    explicit Transcript(const char* label) { init((const u8*)label, strlen(label)); }
    Transcript(const u8* label, size_t n) { init(label, n); }
    void init(const u8* label, size_t n) {
        s.init((const u8*)"Merlin v1.0", 11);
        append_message("dom-sep", label, n);
    }
    void append_message(const char* label, const u8* msg, size_t n) {
        u32 len = (u32)n;
        s.meta_ad((const u8*)label, strlen(label), false);
        s.meta_ad((const u8*)&len, 4, true);
        s.ad(msg, n, false);
    }
    void append_message(const char* label, const char* msg) { append_message(label, (const u8*)msg, strlen(msg)); }
    void append_u64(const char* label, u64 x) { append_message(label, (const u8*)&x, 8); }
    void challenge_bytes(const char* label, u8* dst, size_t n) {
        u32 len = (u32)n;
        s.meta_ad((const u8*)label, strlen(label), false);
        s.meta_ad((const u8*)&len, 4, true);
        s.prf(dst, n, false);
    }
};

// merlin TranscriptRngBuilder + TranscriptRng (src/r1cs/prover.rs:483-494)
struct TranscriptRng {
    Strobe128 s;
    explicit TranscriptRng(const Transcript& t) : s(t.s) {}
    void rekey_with_witness_bytes(const char* label, const u8* w, size_t n) {
        u32 len = (u32)n;
        s.meta_ad((const u8*)label, strlen(label), false);
        s.meta_ad((const u8*)&len, 4, true);
        s.key(w, n, false);
    }
    template <class Rng> void finalize(Rng& rng) {
        u8 bytes[32];
        rng.fill_bytes(bytes, 32);
        s.meta_ad((const u8*)"rng", 3, false);
        s.key(bytes, 32, false);
    }
    void fill_bytes(u8* dst, size_t n) {
        u32 len = (u32)n;
        s.meta_ad((const u8*)&len, 4, false);
        s.prf(dst, n, false);
    }
    u32 next_u32() { u32 x; fill_bytes((u8*)&x, 4); return x; }
    u64 next_u64() { u64 x; fill_bytes((u8*)&x, 8); return x; }
};

// ---- ark-ff `Fp::rand` (UniformRand / Standard): 4 x next_u64 -> limbs, mask the top limb to the
// modulus bit length, accept if < p; the accepted limbs ARE the Montgomery representation.
template <class Rng> static inline Fe fe_rand(const Field& F, Rng& rng) {
    for (;;) {
        Fe r;
        for (int i = 0; i < 4; i++) r.v[i] = rng.next_u64();
        int shave = 256 - F.bits;
        if (shave) r.v[3] &= (~(u64)0) >> shave;
        if (cmp4(r.v, F.p) < 0) return r;
    }
}

}  // namespace orc
