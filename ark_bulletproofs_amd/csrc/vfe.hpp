// Verifier front end on the device, part 2: launch interface of vfe.hip (its own translation unit: the kernels need nothing from
// the engine but the field / curve headers).  Everything is enqueued on the caller's stream; nothing here waits for the GPU.
// Layouts are documented at the kernels (vfe.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace arkbp {
namespace vfe {

static constexpr uint32_t PB_WORDS = 832;   // words per proof parameter block: r1cs.cuh VFY_PB_WORDS (checked where both are visible)

struct Shape {
    uint32_t P;        // proofs in the batch
    uint32_t m;        // commitments per statement
    uint32_t nV;       // commitments the sponge absorbs itself (m, or 0 when the transcripts already hold them)
    uint32_t k;        // inner-product rounds: N = 2^k
    uint32_t plen;     // bytes per compressed proof: 539 + 66 k
    uint32_t tail;     // tail terms per proof in the mega-check: 6 + m + 5 + 2 k
    uint32_t nitems;   // ItemMap::count()
};

// status bits raised by the kernels (any bit: the batch goes through the host replay instead, which reports the reference's error)
static constexpr uint32_t ST_FORMAT = 1;        // a point or scalar the reference's deserializer rejects (proof.rs:83-91)
static constexpr uint32_t ST_IDENTITY = 2;      // an identity point where validate_and_append_point rejects it (transcript.rs:81-93)
static constexpr uint32_t ST_FRAMING = 4;       // L_vec / R_vec length fields disagree with the proof's size

// k_vfe_points: decompression, validation and serialization of every point of the batch.
//   d_proofs   P x plen bytes (R1CSProof::to_bytes)
//   d_V        P x m x 16 words, ark layout
//   d_msg      nitems x 9 x P 64-bit words: word w of item `it` of proof p at [(it * 9 + w) * P + p]
//   d_tail_pts P x tail x 16 words, resident layout, the mega-check's per-proof bases in verification_scalars' order
//   d_status   one word, OR of ST_*
int launch_points(int curve, hipStream_t st, const Shape& sh, const uint8_t* d_proofs, const uint32_t* d_V, uint64_t* d_msg, uint32_t* d_tail_pts, uint32_t* d_status);

// k_vfe_sponge: one lane per proof runs the schedule (vfe_sched.hpp) from its transcript state; d_state0: 25 words per proof
// (state_stride = 25) or one shared state (0).  d_chal: P x (6 + k) x 8 words, resident form, order y z u x w u_1..u_k r.
int launch_sponge(int curve, hipStream_t st, const Shape& sh, const uint32_t* d_sched, const uint64_t* d_state0, uint32_t state_stride, const uint64_t* d_msg, uint32_t* d_chal,
                  uint8_t* d_seeds_or_null);

// k_vfe_consts + k_vfe_wv + k_vfe_sum2: everything of verification_scalars that is not of length N (verifier.rs:462-541 minus the
// g / h scalars).  d_alpha: P x 8 resident.  d_pb: P parameter blocks (r1cs.cuh VFY_PB_WORDS, resident form).  d_tail_sc: P x tail x 8
// canonical integers (alpha-scaled).  d_voff / d_vq / d_vc: the statement's terms on committed variables by commitment (CSR; vc
// resident).  d_ws: P x 32 x 8 words scratch.  d_sums: 2 x 8 words (ark form): sum_p alpha_p sB_p, sum_p alpha_p sBb_p.
int launch_prepare(int curve, hipStream_t st, const Shape& sh, const uint8_t* d_proofs, const uint32_t* d_chal, const uint32_t* d_alpha, const uint32_t* d_voff,
                   const uint32_t* d_vq, const uint32_t* d_vc, uint32_t* d_pb, uint32_t* d_tail_sc, uint32_t* d_ws, uint32_t* d_sums);

}  // namespace vfe
}  // namespace arkbp
