// Verifier front end on the device, part 1 (host only): the STROBE-128 / merlin replay of a single-phase R1CS verification as a
// DATA-INDEPENDENT schedule.
//
// What a verifier appends to its transcript (src/r1cs/verifier.rs:279-287, 403-460, 516-519; src/inner_product_proof.rs:266-280
// through src/transcript.rs:45-102) is the statement's and the proof's points and scalars; the challenges are outputs only.  The
// sponge's byte positions — where every label, length, operation header and padding byte lands, where a message crosses a rate
// block, where Keccak-f runs — therefore depend on the SHAPE of the statement alone (number of commitments, rounds, starting
// position), not on its values.  SchedStrobe runs merlin's operation sequence symbolically and records, per Keccak-f permutation:
//   * the constant bytes XORed into the rate block (labels, lengths, STROBE headers, the padding of run_f) as a 168-byte image,
//   * the pieces of per-proof messages ("items": serialized points and scalars) with their byte ranges and destinations,
//   * whether the 32 bytes squeezed after this permutation are a challenge (STROBE's PRF: read and zero the first 32 bytes).
// The device kernel (vfe.hip: k_vfe_sponge, one lane per proof) then knows nothing about STROBE: it XORs, permutes and squeezes.
// Strobe mirrored here: csrc/host_proto.hpp (merlin 3.0 src/strobe.rs); the CPU interpreter at the bottom runs a schedule with the
// host's own Keccak-f so that tests without a GPU compare it with host::Transcript byte for byte.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace arkbp {
namespace vfe {

static constexpr uint32_t RATE = 166;           // STROBE-128 over Keccak-f[1600]
static constexpr uint32_t ITEM_WORDS = 9;       // an item = 72 bytes: a 65-byte uncompressed point or a 32-byte scalar, zero padded
static constexpr uint32_t NO_SQUEEZE = 0xffffu;

struct Piece { uint32_t item, src_off, len, dst_off; };
struct Block {
    uint8_t cimg[168];
    std::vector<Piece> pieces;
    uint32_t squeeze = NO_SQUEEZE;   // index of the challenge squeezed after this block's permutation
    Block() { memset(cimg, 0, sizeof cimg); }
};

struct Schedule {
    std::vector<Block> blocks;
    uint32_t nchal = 0;
    // flat form for the device: [nblocks] then per block [npieces | squeeze << 16] [42 words of constant image] [2 words per piece]
    std::vector<uint32_t> encode() const {
        std::vector<uint32_t> o;
        o.push_back((uint32_t)blocks.size());
        for (const Block& b : blocks) {
            o.push_back((uint32_t)b.pieces.size() | (b.squeeze << 16));
            for (int i = 0; i < 42; i++) { uint32_t w; memcpy(&w, b.cimg + 4 * i, 4); o.push_back(w); }
            for (const Piece& p : b.pieces) { o.push_back(p.item); o.push_back(p.src_off | (p.len << 8) | (p.dst_off << 16)); }
        }
        return o;
    }
};

// merlin's Strobe128 with symbolic messages: same operations, same position arithmetic as host::Strobe
class SchedStrobe {
public:
    enum { fI = 1, fA = 2, fC = 4, fM = 16, fK = 32 };
    Schedule out;
    bool ok = true;               // false: an operation this front end does not cover was requested
    SchedStrobe(uint8_t pos, uint8_t pos_begin) : pos_(pos), pos_begin_(pos_begin) {}
    uint8_t pos() const { return (uint8_t)pos_; }
    uint8_t pos_begin() const { return (uint8_t)pos_begin_; }

    void append_message_const(const char* label, const void* m, uint32_t n) {
        const uint32_t len = n;
        begin(fM | fA, false); absorb_const(label, (uint32_t)strlen(label));
        begin(fM | fA, true); absorb_const(&len, 4);
        begin(fA, false); absorb_const(m, n);
    }
    void append_u64(const char* label, uint64_t x) { append_message_const(label, &x, 8); }
    void append_message_item(const char* label, uint32_t item, uint32_t n) {
        const uint32_t len = n;
        begin(fM | fA, false); absorb_const(label, (uint32_t)strlen(label));
        begin(fM | fA, true); absorb_const(&len, 4);
        begin(fA, false); absorb_item(item, n);
    }
    // Transcript::challenge_bytes(label, 32 bytes): returns the challenge's index
    uint32_t challenge32(const char* label) {
        const uint32_t len = 32;
        begin(fM | fA, false); absorb_const(label, (uint32_t)strlen(label));
        begin(fM | fA, true); absorb_const(&len, 4);
        begin(fI | fA | fC, false);
        // squeeze: the forced run_f of begin() (or the one the header ran into) left pos = 0; the 32 bytes are the head of the state
        if (pos_ != 0 || out.blocks.empty() || out.blocks.back().squeeze != NO_SQUEEZE) { ok = false; return 0; }
        out.blocks.back().squeeze = out.nchal;
        pos_ = 32;
        return out.nchal++;
    }
    // call at the end: the bytes absorbed since the last permutation are dropped (nothing reads the state afterwards)
    void finish() { cur_ = Block(); }

private:
    uint32_t pos_, pos_begin_;
    Block cur_;
    void runf() {
        cur_.cimg[pos_] ^= (uint8_t)pos_begin_; cur_.cimg[pos_ + 1] ^= 0x04; cur_.cimg[RATE + 1] ^= 0x80;
        out.blocks.push_back(cur_);
        cur_ = Block();
        pos_ = 0; pos_begin_ = 0;
    }
    void absorb_const(const void* d_, uint32_t n) {
        const uint8_t* d = (const uint8_t*)d_;
        while (n) {
            const uint32_t take = n < RATE - pos_ ? n : RATE - pos_;
            for (uint32_t i = 0; i < take; i++) cur_.cimg[pos_ + i] ^= d[i];
            pos_ += take; d += take; n -= take;
            if (pos_ == RATE) runf();
        }
    }
    void absorb_item(uint32_t item, uint32_t n) {
        uint32_t src = 0;
        while (n) {
            const uint32_t take = n < RATE - pos_ ? n : RATE - pos_;
            cur_.pieces.push_back(Piece{item, src, take, pos_});
            pos_ += take; src += take; n -= take;
            if (pos_ == RATE) runf();
        }
    }
    void begin(uint8_t flags, bool more) {
        if (more) return;
        const uint8_t hdr[2] = {(uint8_t)pos_begin_, flags};
        pos_begin_ = pos_ + 1;
        absorb_const(hdr, 2);
        if ((flags & (fC | fK)) && pos_ != 0) runf();
    }
};

// Item numbering of one verification: [0, nV) commitments (only when the device absorbs them), then the proof's points in wire order
// A_I1, A_O1, S1, A_I2, A_O2, S2, T_1, T_3, T_4, T_5, T_6, L[k], R[k], then the scalars t_x, t_x_blinding, e_blinding.
struct ItemMap {
    uint32_t nV, k;
    uint32_t commitment(uint32_t j) const { return j; }
    uint32_t point(uint32_t j) const { return nV + j; }             // j < 11 + 2k
    uint32_t L(uint32_t i) const { return nV + 11 + i; }
    uint32_t R(uint32_t i) const { return nV + 11 + k + i; }
    uint32_t scalar(uint32_t j) const { return nV + 11 + 2 * k + j; }   // j < 3
    uint32_t count() const { return nV + 11 + 2 * k + 3; }
};

// The replay of verify_prepare_t (r1cs_host.inc; src/r1cs/verifier.rs:403-460 + :516-519, src/inner_product_proof.rs:266-280) for a
// single-phase statement with m commitments and k inner-product rounds over N = 2^k padded multipliers.  The transcript stands
// at (pos, pos_begin): before the commitments when absorb_commitments (they are then items 0 .. m-1), else right after them.
// Challenges come out in the order y, z, u, x, w, u_1 .. u_k, r (6 + k of them).
static inline bool build_verifier_schedule(Schedule& out, uint8_t pos, uint8_t pos_begin, bool absorb_commitments, uint64_t m, uint32_t k, uint64_t N) {
    SchedStrobe s(pos, pos_begin);
    const ItemMap im{absorb_commitments ? (uint32_t)m : 0u, k};
    if (absorb_commitments) for (uint32_t j = 0; j < (uint32_t)m; j++) s.append_message_item("V", im.commitment(j), 65);   // Verifier::commit (verifier.rs:279-287)
    s.append_u64("m", m);
    s.append_message_item("A_I1", im.point(0), 65); s.append_message_item("A_O1", im.point(1), 65); s.append_message_item("S1", im.point(2), 65);
    s.append_message_const("dom-sep", "r1cs-1phase", 11);
    s.append_message_item("A_I2", im.point(3), 65); s.append_message_item("A_O2", im.point(4), 65); s.append_message_item("S2", im.point(5), 65);
    s.challenge32("y"); s.challenge32("z");
    s.append_message_item("T_1", im.point(6), 65); s.append_message_item("T_3", im.point(7), 65); s.append_message_item("T_4", im.point(8), 65);
    s.append_message_item("T_5", im.point(9), 65); s.append_message_item("T_6", im.point(10), 65);
    s.challenge32("u"); s.challenge32("x");
    s.append_message_item("t_x", im.scalar(0), 32); s.append_message_item("t_x_blinding", im.scalar(1), 32); s.append_message_item("e_blinding", im.scalar(2), 32);
    s.challenge32("w");
    s.append_message_const("dom-sep", "ipp v1", 6);
    s.append_u64("n", N);
    for (uint32_t i = 0; i < k; i++) {
        s.append_message_item("L", im.L(i), 65); s.append_message_item("R", im.R(i), 65);
        s.challenge32("u");
    }
    s.challenge32("r");   // drawn from a clone upstream (verifier.rs:516-519): nothing reads the original afterwards
    s.finish();
    if (!s.ok || s.out.nchal != 6 + k) return false;
    out = std::move(s.out);
    return true;
}

// CPU interpreter of a schedule (tests; the device kernel does the same per lane): state = 25 words, items = count x 72 bytes;
// seeds[c * 32 ..] receives the 32 bytes of challenge c.
template <class KeccakF> static inline void run_schedule_cpu(const Schedule& sc, uint64_t state[25], const uint8_t* items, uint8_t* seeds, KeccakF&& keccakf) {
    for (const Block& b : sc.blocks) {
        uint8_t* st = (uint8_t*)state;
        for (uint32_t i = 0; i < 168; i++) st[i] ^= b.cimg[i];
        for (const Piece& p : b.pieces) for (uint32_t i = 0; i < p.len; i++) st[p.dst_off + i] ^= items[(size_t)p.item * 72 + p.src_off + i];
        keccakf(state);
        if (b.squeeze != NO_SQUEEZE) { memcpy(seeds + (size_t)b.squeeze * 32, st, 32); memset(st, 0, 32); }
    }
}

}  // namespace vfe
}  // namespace arkbp
