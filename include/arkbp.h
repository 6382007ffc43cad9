/* arkbp.h — C ABI of libarkbp_hip.so: the MI355X (gfx950) engine behind the MSM / inner-product hot path of
 * FindoraNetwork/ark-bulletproofs.  Every entry point names the reference interface it replaces
 * (paths are into the reference repository).  A Rust shim binds these with `extern "C"`
 * (see INTEGRATION.md); the repo's own tests bind them with ctypes.
 *
 * Data conventions (identical to the reference's in-memory types, so a shim passes pointers):
 *   field element  4 x uint64_t little-endian limbs, Montgomery form with R = 2^256 — ark-ff's
 *                  Fp256<MontBackend<_,4>>.  "scalar" = element of G::ScalarField, coordinates are in
 *                  the curve's base field.
 *   affine point   8 x uint64_t = x || y.  ark's Affine{x,y,infinity} packs to this with the identity
 *                  encoded as all-zero (0,0 is on neither curve).
 *   curve          0 = secq256k1 (ark-secq256k1), 1 = zorro (src/curve/zorro/)
 * All functions return 0 on success or a negative BP_E_* code; nothing throws or aborts across the
 * boundary.  A bp_ctx owns one HIP stream and its workspaces: calls on one ctx are serialised, distinct
 * contexts are independent (one per device / host thread).
 */
#ifndef ARKBP_H
#define ARKBP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BP_CURVE_SECQ256K1 0
#define BP_CURVE_ZORRO 1

#define BP_OK 0
#define BP_E_ARG (-1)          /* bad argument (null pointer, unknown curve, length mismatch) */
#define BP_E_HIP (-2)          /* HIP runtime error; bp_last_error() has the text */
#define BP_E_NO_DEVICE (-3)    /* no gfx950 device visible: the engine has no CPU fallback */
#define BP_E_VERIFICATION (-4) /* R1CSError::VerificationError / ProofError::VerificationError (src/errors.rs) */
#define BP_E_GENS_LENGTH (-5)  /* R1CSError::InvalidGeneratorsLength */
#define BP_E_FORMAT (-6)       /* R1CSError::FormatError */
#define BP_E_MISSING (-7)      /* R1CSError::MissingAssignment */

typedef struct bp_ctx bp_ctx;

const char* bp_last_error(void);
/* number of HIP devices visible (0 => every compute entry point returns BP_E_NO_DEVICE) */
int bp_device_count(void);
int bp_ctx_create(int curve, int device, bp_ctx** out);
void bp_ctx_destroy(bp_ctx* ctx);
int bp_ctx_sync(bp_ctx* ctx);

/* ---- device memory (HBM-resident vectors, so repeated calls do not re-cross PCIe) ---------------- */
int bp_dev_alloc(bp_ctx* ctx, size_t bytes, void** dptr);
int bp_dev_free(bp_ctx* ctx, void* dptr);
int bp_dev_upload(bp_ctx* ctx, void* dptr, const void* host, size_t bytes);
int bp_dev_download(bp_ctx* ctx, void* host, const void* dptr, size_t bytes);
/* affine points, ark layout -> the engine's resident layout (64 B/point, radix-2^29 Montgomery, packed),
 * device to device; in == out is allowed.  bp_points_export is the inverse. */
int bp_points_import(bp_ctx* ctx, const void* d_in_ark, void* d_out, size_t n);
int bp_points_export(bp_ctx* ctx, const void* d_in, void* d_out_ark, size_t n);

/* ---- VariableBaseMSM::msm -------------------------------------------------------------------------
 * Replaces `<G::Group as VariableBaseMSM>::msm(&bases, &scalars).unwrap().into_affine()`
 * (src/inner_product_proof.rs:104,124,187,202,375; src/r1cs/prover.rs:516,532,546,607,622,635;
 *  src/r1cs/verifier.rs:574,685 — the verifier only tests `is_zero()`, i.e. out == all-zero).
 * Host buffers: bases_xy n points (ark layout), scalars n field elements (Montgomery, or canonical
 * integers if scalars_canonical != 0 — what ark's msm_bigint takes).  out_xy = affine result. */
int bp_msm(bp_ctx* ctx, const uint64_t* bases_xy, const uint64_t* scalars, size_t n, int scalars_canonical, uint64_t out_xy[8]);
/* Same with operands already resident: d_bases in the engine's layout (bp_points_import), d_scalars n x
 * 32 B.  This is the call the timed region of bench.py makes. */
int bp_msm_dev(bp_ctx* ctx, const void* d_bases, const void* d_scalars, size_t n, int scalars_canonical, uint64_t out_xy[8]);

/* Window-sharded MSM for one large MSM across the GPUs of a node (every GPU holds all bases and scalars): the call
 * accumulates only Pippenger windows [w_lo, w_hi) of the bp_msm_window_count(curve, n) windows and returns that partial
 * sum already weighted by 2^(c*w); the ranks' partials add up to the full MSM (all-gather of one 64-byte point per rank,
 * then bp_host_points_sum).  The window schedule depends only on (curve, n), so all ranks agree on it. */
int bp_msm_window_count(int curve, size_t n, int* windows, int* window_bits);
int bp_msm_dev_windows(bp_ctx* ctx, const void* d_bases, const void* d_scalars, size_t n, int scalars_canonical, int w_lo, int w_hi,
                       uint64_t out_xy[8]);

/* MSM over the RESIDENT generator tables (bp_gens_derive / bp_gens_upload), for the prover's and verifier's own msm call sites
 * (e.g. A_I = msm([B_blinding] ++ G[..n] ++ H[..n], ..), src/r1cs/prover.rs:516-559): bases = G[off..off+n) if use_G, then
 * H[off..off+n) if use_H, then n_extra caller points (ark layout); scalars in the same order, (use_G ? n : 0) + (use_H ? n : 0) +
 * n_extra of them (ark Montgomery words, or canonical integers when scalars_canonical).  Nothing but the scalars and the extra
 * points crosses PCIe.  BP_E_GENS_LENGTH when the range exceeds the installed tables. */
int bp_msm_gens(bp_ctx* ctx, int use_G, int use_H, size_t off, size_t n, const uint64_t* extra_bases_xy, size_t n_extra, const uint64_t* scalars,
                int scalars_canonical, uint64_t out_xy[8]);

/* Window-sharded mode for whole computations (north_star: "large proofs shard Pippenger windows across the GPUs of one node with
 * a final point-reduce"): after bp_ctx_set_window_shard(ctx, rank, world, cb, user) EVERY MSM this ctx runs (bp_msm*, the L/R
 * MSMs of bp_ipa_create, the commitment MSMs of the prover, the verifier's mega-check) accumulates only the windows of `rank`
 * and then calls cb(user, xy): the callback must replace the partial point in xy by the sum of all ranks' partials
 * (all-gather over RCCL + bp_host_points_sum) and return 0.  All ranks run the same call sequence on the same inputs and obtain
 * identical results (same proof bytes).  world = 1 switches the mode off.  Calls with an explicit window range
 * (bp_msm_dev_windows) are not affected. */
typedef int (*bp_point_reduce_cb)(void* user, uint64_t xy[8]);
int bp_ctx_set_window_shard(bp_ctx* ctx, int rank, int world, bp_point_reduce_cb cb, void* user);
/* Optional second collective of the sharded mode: cb(user, send, bytes, recv) must fill recv with the `world` ranks' `bytes`-long
 * blocks in rank order (an all-gather) and return 0.  With it (and a power-of-two world) the prover also partitions the
 * inner-product argument: rank r keeps the elements i = r + j*world of a, b, G, H (src/inner_product_proof.rs:139-156,216-225 fold
 * element i with i + n/2 — both on the same rank), a round's L and R are sums of per-rank partial MSMs through the point-reduce
 * callback, and when the vectors are down to the frozen-tail length (BP_TUNE_IPA_FREEZE_LEN) the ranks all-gather the rest once
 * and finish replicated.  Every fold launch then does 1/world of the single-GPU work; proofs stay byte-identical on every rank. */
typedef int (*bp_allgather_cb)(void* user, const void* send, size_t bytes, void* recv);
int bp_ctx_set_shard_allgather(bp_ctx* ctx, bp_allgather_cb cb, void* user);
/* Native collectives: RCCL over xGMI, inside the library (no host callback, no interpreter lock on the data path).
 * One rank calls bp_rccl_unique_id (ncclGetUniqueId) and hands the 128 bytes to the others through whatever bootstrap the host
 * has (a torch.distributed broadcast in bench.py; a Rust host would use its own); then EVERY rank calls bp_ctx_rccl_init on its
 * ctx (ncclCommInitRank: collective, blocks until all ranks arrive).  The ctx is then in the sharded mode of
 * bp_ctx_set_window_shard with BOTH exchanges served by ncclAllGather on the ctx's stream: the per-MSM point-reduce
 * (64-byte affine partials, summed on the host: group addition is not an RCCL reduce op) and the one vector gather of the
 * index-cyclic inner-product argument.  Ranks must issue the same calls in the same order.  The library binds RCCL at run time
 * (the copy already in the process, else the ROCm installation's); BP_E_HIP with a message when there is none.
 * bp_ctx_collective_stats: number of native collectives since init and the host wall time spent in them. */
int bp_rccl_unique_id(uint8_t out[128]);
int bp_ctx_rccl_init(bp_ctx* ctx, const uint8_t unique_id[128], int rank, int world);
int bp_ctx_rccl_shutdown(bp_ctx* ctx);
int bp_ctx_collective_stats(bp_ctx* ctx, uint64_t* count, double* seconds);
/* test hook: one native all-gather of `bytes` bytes per rank (recv: world x bytes) */
int bp_debug_rccl_allgather(bp_ctx* ctx, const uint8_t* send, size_t bytes, uint8_t* recv);

/* ---- InnerProductProof::create -------------------------------------------------------------------
 * Replaces `InnerProductProof::create(transcript, &Q, &G_factors, &H_factors, G_vec, H_vec, a_vec, b_vec)`
 * (src/inner_product_proof.rs:37-239).  Host buffers of length n (a power of two, as the reference
 * asserts at :58-66).  The Merlin transcript stays with the caller: once per halving round the engine
 * hands the callback the affine L, R it would append (`append_point(b"L"|b"R")`, :132-133) and the
 * callback returns u = `challenge_scalar(b"u")` (:135); u^-1 is derived by the engine.  A non-zero
 * return from the callback aborts the call.  Outputs: L_vec, R_vec (lg n points each), a, b. */
typedef int (*bp_challenge_cb)(void* user, const uint64_t L_xy[8], const uint64_t R_xy[8], uint64_t u_out[4]);
int bp_ipa_create(bp_ctx* ctx, const uint64_t Q_xy[8], const uint64_t* G_factors, const uint64_t* H_factors, const uint64_t* G_xy,
                  const uint64_t* H_xy, const uint64_t* a, const uint64_t* b, size_t n, bp_challenge_cb cb, void* user, uint64_t* L_out_xy,
                  uint64_t* R_out_xy, uint64_t a_out[4], uint64_t b_out[4]);

/* The same computation cut at the Fiat-Shamir step, for hosts that keep the transcript on their side without a callback and
 * for the index-cyclic multi-GPU partition (each rank holds the elements i = rank mod world of every vector; a round's L and R
 * are then the sums of the ranks' partial points, see ark_bulletproofs_amd/parallel.py):
 *   bp_ipa_begin       copies the instance to the GPU (the arguments of `create`, :37-46; n a power of two)
 *   bp_ipa_round_LR    L, R of the current round (:78-131 / :166-213), affine ark layout; the host appends them (:132-133)
 *   bp_ipa_round_fold  folds a, b, G, H with the challenge u (:137-155 / :214-224); u^-1 is derived by the engine
 *   bp_ipa_finish      a[0], b[0] once the vectors have length 1 (:226-231)
 *   bp_ipa_export      the current vectors between rounds (n_cur elements each) with the scalars the engine still owes the
 *                      generator vectors: G_true[i] = gamma_G * G[i], H_true[i] = gamma_H * H[i] (uniform rounds fold
 *                      G_R + u^-2 G_L and carry the common factor u instead of multiplying every point by it)
 * One stepping instance per ctx; bp_ipa_begin restarts it. */
int bp_ipa_begin(bp_ctx* ctx, const uint64_t Q_xy[8], const uint64_t* G_factors, const uint64_t* H_factors, const uint64_t* G_xy, const uint64_t* H_xy,
                 const uint64_t* a, const uint64_t* b, size_t n);
int bp_ipa_round_LR(bp_ctx* ctx, uint64_t L_xy[8], uint64_t R_xy[8]);
int bp_ipa_round_fold(bp_ctx* ctx, const uint64_t u[4]);
int bp_ipa_finish(bp_ctx* ctx, uint64_t a[4], uint64_t b[4]);
int bp_ipa_export(bp_ctx* ctx, uint64_t* a, uint64_t* b, uint64_t* G_xy, uint64_t* H_xy, uint64_t gamma_G[4], uint64_t gamma_H[4], size_t* n_cur);

/* ---- InnerProductProof::verify -------------------------------------------------------------------
 * Replaces `proof.verify(n, transcript, G_factors, H_factors, &P, &Q, &G, &H)` (src/inner_product_proof.rs:321-382).
 * The caller replays its transcript (`innerproduct_domain_sep`, `validate_and_append_point(L|R)`, `challenge_scalar(u)`,
 * :266-277) and passes the lg_n challenges in creation order; the engine derives u^2, u^-2, the s vector (:279-311), builds
 * the 1 + 2n + 2 lg_n scalars on the GPU, runs the MSM and compares with P.  Returns BP_OK or BP_E_VERIFICATION. */
int bp_ipa_verify(bp_ctx* ctx, size_t n, const uint64_t* G_factors, const uint64_t* H_factors, const uint64_t P_xy[8], const uint64_t Q_xy[8],
                  const uint64_t* G_xy, const uint64_t* H_xy, const uint64_t* L_xy, const uint64_t* R_xy, size_t lg_n, const uint64_t* challenges,
                  const uint64_t a[4], const uint64_t b[4]);

/* ---- generators -----------------------------------------------------------------------------------
 * bp_gens_derive replaces `BulletproofGens::new(cap, 1)` + `PedersenGens::default()` (src/generators.rs:174-221,
 * 47-66): derives party 0's G/H tables on the host cores and installs them in HBM (resident for the life of
 * the ctx; the reference clones them into fresh Vecs per proof, src/r1cs/prover.rs:796-797).
 * bp_gens_upload installs tables the caller already holds (ark layout). */
int bp_gens_derive(bp_ctx* ctx, size_t gens_capacity);
int bp_gens_upload(bp_ctx* ctx, const uint64_t* G_xy, const uint64_t* H_xy, size_t gens_capacity);
int bp_gens_download(bp_ctx* ctx, uint64_t* G_xy, uint64_t* H_xy, size_t n);
/* Optional: fixed-base tables of G[0..count), H[0..count) for the FIRST fold round of the prover's inner-product argument
 * (src/inner_product_proof.rs:139-156: there the bases are the generator tables themselves, the same for every proof, and the
 * multiplier is one scalar for all elements) — e * 2^(w*j) * G[i] for all windows j and digits e, so that the round becomes
 * ~34 look-ups + mixed adds per point (w = 8) instead of a 130-step double-and-add ladder.  count = the left half of the largest
 * proof: N/2.  window_bits 2..8, or 0 = the widest whose tables fit in budget_bytes (0 = 3/4 of the free device memory); the
 * choice and the table size come back in *window_bits_out / *bytes_out (2^19 bases, w = 8: 146 GB for both vectors on secq256k1).
 * Built on the ctx that owns the generators; bp_gens_share hands them on.  count = 0 frees them.  Results never depend on it. */
int bp_gens_fold_tables(bp_ctx* ctx, size_t count, int window_bits, size_t budget_bytes, int* window_bits_out, size_t* bytes_out);
/* The same for ONE RANK of a sharded prover (bp_ctx_set_shard, index-cyclic inner-product argument): only the generators
 * rank + i * world, i < count / world — 1/world of the memory (count = 3N/4 of a 2^22 proof on 8 GPUs at w = 8: 110 GB per rank instead of
 * 876), which is all that rank's slice ever looks up.  These tables belong to the ctx that builds them, whoever owns the generators
 * (bp_gens_share first, then this); bp_gens_tables_check and bp_gens_fold_tables(.., 0, ..) treat them like the whole ones.
 * count must be a multiple of world.  Results never depend on it. */
int bp_gens_fold_tables_slice(bp_ctx* ctx, size_t count, int window_bits, size_t budget_bytes, int rank, int world, int* window_bits_out, size_t* bytes_out);
/* Optional: fixed-base rows 2^(4r) * G[i], 2^(4r) * H[i] (r < 64, i < count) and the same for PedersenGens, 4 KiB per generator
 * (count = 2^20: 8.6 GB).  With them every MSM the prover runs over the generator tables themselves — the commitments A_I, A_O, S
 * (src/r1cs/prover.rs:516-559, 604-649) and the first round's L, R (src/inner_product_proof.rs:83-131), 7/9 of a proof's MSM
 * terms — sorts the digits of ALL Pippenger windows into ONE bucket set (the row supplies the power of two): the per-window costs
 * are paid once, the window is wider (c = 20 at 2^21 terms: 13 mixed adds per term instead of 17-18) and the result needs no
 * doublings.  Skewed scalars (0/1 witness vectors) fall back to the ordinary schedule.  Results never depend on it. */
int bp_gens_msm_tables(bp_ctx* ctx, size_t count, size_t* bytes_out);
/* Optional: build the DIRECT WINDOW TABLES of the small-statement path now (d * 16^w * base for B, B_blinding and the first `count`
 * generators of G and H, 60 KiB per base: count = 8192 is 1 GB) instead of inside the first small proof — e.g. before bp_gens_share,
 * which hands them on (a ctx that shares generators without them builds its own copy on first use).  count is rounded down to what
 * BP_TUNE_DIRECT_MAX and the installed generators allow; count = 0 frees them.  As with the other tables: rebuilding them (a larger
 * count, other generators) on a ctx whose tables are shared is the owner's responsibility.  Results never depend on it. */
int bp_gens_direct_tables(bp_ctx* ctx, size_t count, size_t* bytes_out);
/* Integrity check of both kinds of precomputed tables: every entry is re-derived from its stored neighbours by the chain rule
 * (e * 2^(w*j) * P = (e-1) * 2^(w*j) * P + 2^(w*j) * P, next window = the doubled last entry; MSM rows: next row = 16 * row) with
 * mixed additions / doublings compared projectively, anchored at the resident generators; *bad_fold_entries / *bad_msm_rows (either
 * may be NULL) receive the number of entries that do not fit.  A proof touches only the few rows its challenges select, so a
 * corrupted entry (a bit flip in 150 GB of HBM, a bad build) cannot be found by proving; this finds it, in about the time of one
 * proof.  Tables that are not installed count as 0. */
int bp_gens_tables_check(bp_ctx* ctx, uint64_t* bad_fold_entries, uint64_t* bad_msm_rows);
/* test hook: device address and size in bytes of a table (which: 0 / 1 fold tables of G / H, layout [window][digit-1][i] x 64 B;
 * 2 / 3 MSM rows of G / H, layout [row][i] x 64 B); NULL / 0 when not installed */
int bp_debug_tables_ptr(bp_ctx* ctx, int which, void** dptr, size_t* nbytes);
/* PedersenGens::default() -> (B, B_blinding); host only */
int bp_pedersen_gens(int curve, uint64_t B_xy[8], uint64_t B_blinding_xy[8]);
/* GeneratorsChain for label 'G'|'H' || LE32(party) (src/generators.rs:71-121), first `count` points; host only */
int bp_host_derive_generators(int curve, int which_H, uint32_t party, size_t count, uint64_t* out_xy);

/* ---- host transcript: merlin::Transcript + TranscriptProtocol (src/transcript.rs:45-102); host only ---- */
void* bp_transcript_new(const uint8_t* label, size_t n);
void bp_transcript_free(void* t);
void bp_transcript_append_message(void* t, const char* label, const uint8_t* msg, size_t n);
void bp_transcript_challenge_bytes(void* t, const char* label, uint8_t* out, size_t n);
int bp_transcript_append_point(int curve, void* t, const char* label, const uint64_t xy[8]);
int bp_transcript_challenge_scalar(int curve, void* t, const char* label, uint64_t out[4]);
int bp_host_sha3_512(const uint8_t* msg, size_t n, uint8_t out[64]);
/* test hook, host only: the prover's TranscriptRng (src/r1cs/prover.rs:483-494) then `count` Fr::rand draws; lanes = 1 scalar
 * path (one 32-byte seed), lanes = 8 the AVX-512 x8 stream (8 seeds; out = 8 x count scalars; BP_E_ARG if unavailable) */
int bp_debug_rng_draws(int curve, void* transcript, const uint64_t* witness, size_t nw, const uint8_t* seeds, int lanes, size_t count,
                       uint64_t* out);
/* Test hook of the lockstep transcript path of batch verification (host::StrobeX8; the eight verifier transcripts of a group take
 * their commitments through AVX-512 Keccak-f x8): appends `npts` points under `label` to each of the `lanes` (2..8) transcripts, which
 * must all be in the state of transcripts[0]; points_xy = [lane][npts][8].  The result must equal bp_transcript_append_point applied
 * lane by lane.  BP_E_ARG when the host has no AVX-512 (the product then takes the scalar path). */
int bp_debug_append_points_x8(int curve, void* const* transcripts, int lanes, const char* label, const uint64_t* points_xy, size_t npts);
/* Test hook of the rest of the lockstep replay (gather / same-message append / challenge / scatter of host::StrobeX8): the EIGHT
 * transcripts — any contents, but at the same STROBE position — take `msg` under `msg_label`, then give `nbytes` (1..64) challenge bytes
 * each under `chal_label`; out = [8][nbytes].  Must equal bp_transcript_append_message + bp_transcript_challenge_bytes lane by lane,
 * and leave the transcripts in the same states.  BP_E_ARG without AVX-512 or when the positions differ. */
int bp_debug_challenge_x8(void* const* transcripts, const char* msg_label, const uint8_t* msg, size_t msg_len, const char* chal_label, size_t nbytes,
                          uint8_t* out);
/* group sum of affine points on the host (the point-reduce after an all-gather of per-GPU partials) */
int bp_host_points_sum(int curve, const uint64_t* pts_xy, size_t count, uint64_t out_xy[8]);

/* ---- r1cs::Prover::prove -------------------------------------------------------------------------
 * Replaces `Prover::new` + commits + gadget + `prove(prng, bp_gens)` (src/r1cs/prover.rs:291-341, 444-831)
 * for the statement families the reference's tests and benches build ("scenarios": 0 k-shuffle
 * benches/r1cs_secq256k1.rs:34-112, 1 range proof tests/r1cs_secq256k1.rs:361-445, 2 example gadget
 * :216-267, 3 square chain and 4 multi-range: this build's synthetic large circuits).  params: 8 u64
 * (scenario-specific), seed: ChaCha20 seed of the external prng.  Outputs: compressed proof bytes
 * (`R1CSProof::to_bytes`, src/r1cs/proof.rs:74-81), the V commitments, scenario publics.
 * timing[8] (seconds): [0] inside prove(), [1] statement construction, [2] TranscriptRng draws, [3] uploads,
 * [4] commitment MSMs, [5] flattened_constraints, [6] polynomial kernels, [7] inner-product argument. */
int bp_r1cs_prove_scenario(bp_ctx* ctx, int scenario, const uint64_t* params, const uint8_t seed[32], uint8_t* proof_out, size_t* proof_len,
                           uint64_t* commit_xy, size_t m_cap, size_t* m_out, uint64_t* publics, size_t* npub, double* timing);

/* `PedersenGens::commit` (src/generators.rs:39-44) for m (value, blinding) pairs at once: out[i] = v[i]*B + blind[i]*B_blinding
 * with PedersenGens::default()'s bases, by fixed-base window tables on the GPU.  v, blind: m x 4 words (ark layout);
 * out_xy: m x 8 words (affine, ark layout; the identity is all-zero). */
int bp_pedersen_commit_batch(bp_ctx* ctx, const uint64_t* v, const uint64_t* blind, size_t m, uint64_t* out_xy);

/* Statement handles: `Prover::new` + `commit`s + gadget (src/r1cs/prover.rs:291-341; host only, no GPU) separated from
 * `prove()` so that a service can keep several proofs in flight: the per-proof TranscriptRng chain (8 Keccak-f per
 * multiplier, strictly sequential inside one proof) of one statement overlaps the GPU work of the others.  Each in-flight
 * proof uses its own bp_ctx (stream + workspaces); bp_gens_share lets them all read one resident copy of the tables.
 * bp_stmt_prove consumes the statement (`prove(self, ..)`); timing as in bp_r1cs_prove_scenario ([1] = 0). */
typedef struct bp_stmt bp_stmt;
struct bp_cs;
/* a statement IS a prover handle (scenario code records through the same recorder as bp_cs_*): bp_stmt_prove / _precompute are
 * bp_prover_prove / _precompute on it */
struct bp_cs* bp_stmt_as_prover(bp_stmt* stmt);
int bp_stmt_prover_create(int curve, int scenario, const uint64_t* params, const uint8_t seed[32], bp_stmt** out);
/* same, with the statement's Pedersen commitments (`Prover::commit`, src/r1cs/prover.rs:327-341) computed on ctx's GPU in one
 * batch; the curve is ctx's.  Identical statement (same transcript, same commitments) as the host-only constructor. */
int bp_stmt_prover_create_dev(bp_ctx* ctx, int scenario, const uint64_t* params, const uint8_t seed[32], bp_stmt** out);
void bp_stmt_free(bp_stmt* stmt);
int bp_stmt_info(bp_stmt* stmt, uint64_t* commit_xy, size_t m_cap, size_t* m_out, uint64_t* publics, size_t* npub, size_t* multipliers,
                 size_t* constraints);
/* Optional, host only (no ctx): runs the head of prove() — `m`, TranscriptRng construction and the phase-1 blinding draws
 * (src/r1cs/prover.rs:466-513) — so it can overlap other proofs' GPU work; bp_stmt_prove continues from that state. */
int bp_stmt_precompute(bp_stmt* stmt);
/* Batch form: statements of the same curve; every group of 8 with equal multiplier / commitment counts advances its eight
 * TranscriptRng chains in lockstep in AVX-512 lanes (Keccak-f x8), producing exactly the per-statement streams. */
int bp_stmt_precompute_batch(bp_stmt** stmts, size_t count);
int bp_stmt_prove(bp_ctx* ctx, bp_stmt* stmt, uint8_t* proof_out, size_t* proof_len, double* timing);
int bp_gens_share(bp_ctx* dst, bp_ctx* src);

/* ---- r1cs::ConstraintSystem / Prover / Verifier for the caller's OWN gadgets ------------------------------------------------
 * The reference's public surface for circuits is the trait API of src/r1cs/constraint_system.rs:19-135 (`multiply`, `allocate`,
 * `allocate_multiplier`, `constrain`, `specify_randomized_constraints`, `challenge_scalar`), recorded by `Prover` (src/r1cs/
 * prover.rs:96-268) and `Verifier` (src/r1cs/verifier.rs:69-224) and consumed by `prove` (prover.rs:444-831), `verify`
 * (verifier.rs:549-600) and `batch_verify` (verifier.rs:604-691).  A bp_cs is one such recorder; everything it records reaches the
 * same GPU path the scenario entry points use (they are callers of this API): the fused prover with GPU flattened constraints, the
 * TranscriptRng x8 pipeline, circuit templates and the block pipeline of batch verification — for single-phase and randomized
 * (two-phase) circuits alike.
 *
 *   Variable            bp_var {kind, index}: r1cs::Variable::{Committed(i), MultiplierLeft(i), MultiplierRight(i),
 *                       MultiplierOutput(i), One()} (src/r1cs/linear_combination.rs:12-24)
 *   LinearCombination   n terms as two parallel arrays: vars[n], coefs[4n] (ark Montgomery words)
 *   transcript          a bp_transcript_* handle, BORROWED for the life of the bp_cs like `T: BorrowMut<Transcript>`: the recorder
 *                       appends to it exactly what the reference appends, so afterwards it is in the state
 *                       `prove_and_return_transcript` / `verify_and_return_transcript` would return.  Hosts with their own merlin
 *                       move the 203-byte STROBE state across with bp_transcript_export_state / bp_transcript_import_state.
 *   external rng        `prove(prng, ..)` uses its rng for ONE thing: the 32 bytes of TranscriptRngBuilder::finalize (prover.rs:493);
 *                       the caller draws them (`prng.fill_bytes(&mut [0u8; 32])`) and passes them as rng_bytes.
 *   alphas              batch_verify's `G::ScalarField::rand(prng)` per instance (verifier.rs:649): drawn by the caller, in
 *                       instance order (ark Montgomery words); NULL = all ones (what Verifier::verify amounts to).
 * A bp_cs is not thread-safe; distinct handles are independent.  Errors are the reference's: BP_E_MISSING <-> MissingAssignment,
 * BP_E_VERIFICATION, BP_E_FORMAT, BP_E_GENS_LENGTH; a non-zero return of a randomized-constraint callback is passed through
 * (`R1CSError::GadgetError`). */
typedef struct bp_cs bp_cs;
#define BP_VAR_COMMITTED 0
#define BP_VAR_MULT_LEFT 1
#define BP_VAR_MULT_RIGHT 2
#define BP_VAR_MULT_OUT 3
#define BP_VAR_ONE 4
typedef struct bp_var { uint32_t kind; uint32_t index; } bp_var;

/* `Prover::new(&pc_gens, transcript)` (prover.rs:291-308) / `Verifier::new(transcript)` (verifier.rs:252-263): appends the r1cs
 * domain separator.  pc_gens is PedersenGens::default() (bp_pedersen_gens). */
int bp_prover_new(int curve, void* transcript, bp_cs** out);
int bp_verifier_new(int curve, void* transcript, bp_cs** out);
void bp_cs_free(bp_cs* cs);
/* `ConstraintSystem::transcript()` (constraint_system.rs:21): the recorder's transcript, for gadgets that bind extra data */
void* bp_cs_transcript(bp_cs* cs);
/* `Prover::commit(v, v_blinding) -> (V, Variable)` (prover.rs:327-341) for `count` values in order; with a ctx the Pedersen
 * commitments are one GPU batch (bp_pedersen_commit_batch), otherwise host arithmetic.  V_xy_out / vars_out may be NULL. */
int bp_prover_commit(bp_cs* prover, bp_ctx* ctx_or_null, const uint64_t* v, const uint64_t* v_blinding, size_t count, uint64_t* V_xy_out, bp_var* vars_out);
/* `Verifier::commit(V) -> Variable` (verifier.rs:279-287) for `count` commitments in order */
int bp_verifier_commit(bp_cs* verifier, const uint64_t* V_xy, size_t count, bp_var* vars_out);
/* `multiply(left, right) -> (l, r, o)` (constraint_system.rs:30-41; prover.rs:103-133, verifier.rs:74-98) */
int bp_cs_multiply(bp_cs* cs, const bp_var* left_vars, const uint64_t* left_coefs, size_t n_left, const bp_var* right_vars, const uint64_t* right_coefs,
                   size_t n_right, bp_var out[3]);
/* `allocate(assignment)` (constraint_system.rs:43-54; prover.rs:135-157, verifier.rs:100-116); NULL = None (BP_E_MISSING for a prover) */
int bp_cs_allocate(bp_cs* cs, const uint64_t* assignment_or_null, bp_var* out);
/* `allocate_multiplier(input_assignments)` (constraint_system.rs:56-65; prover.rs:159-183, verifier.rs:118-138) */
int bp_cs_allocate_multiplier(bp_cs* cs, const uint64_t* left_or_null, const uint64_t* right_or_null, bp_var out[3]);
/* `constrain(lc)` (constraint_system.rs:70-75) */
int bp_cs_constrain(bp_cs* cs, const bp_var* vars, const uint64_t* coefs, size_t n);
/* bulk forms for large gadgets (one FFI call instead of one per gate): `count` allocate_multiplier calls (left / right: count x 4
 * words, NULL for a verifier; the multipliers get indices first_index .. first_index + count - 1), and `nconstraints` constrain
 * calls in CSR form (constraint q owns terms [offsets[q], offsets[q+1]) of vars / coefs) */
int bp_cs_allocate_multipliers(bp_cs* cs, const uint64_t* left, const uint64_t* right, size_t count, uint32_t* first_index);
int bp_cs_constrain_many(bp_cs* cs, const bp_var* vars, const uint64_t* coefs, const size_t* offsets, size_t nconstraints);
/* `specify_randomized_constraints(callback)` (constraint_system.rs:96-109; prover.rs:211-217, verifier.rs:172-178): cb runs in the
 * randomized phase of prove / verify (prover.rs:418-441, verifier.rs:353-376) with the recorder it belongs to; inside it
 * bp_cs_challenge_scalar is `RandomizedConstraintSystem::challenge_scalar(label)` (constraint_system.rs:127-134).  In batch
 * verification the callbacks of different instances run concurrently on the library's host threads. */
typedef int (*bp_randomize_cb)(void* user, bp_cs* cs);
int bp_cs_specify_randomized_constraints(bp_cs* cs, bp_randomize_cb cb, void* user);
int bp_cs_challenge_scalar(bp_cs* cs, const char* label, uint64_t out[4]);
int bp_cs_metrics(bp_cs* cs, size_t* multipliers, size_t* constraints, size_t* commitments);
/* `prover.prove(prng, &bp_gens)` (prover.rs:444-451): consumes the prover.  rng_bytes: see "external rng" above (may be NULL when
 * bp_prover_set_rng / bp_prover_precompute supplied them).  proof_out / *proof_len: `R1CSProof::to_bytes` (proof.rs:74-81).
 * timing (8 doubles, may be NULL) as in bp_r1cs_prove_scenario.  bp_gens are the ctx's resident tables.
 * bp_prover_precompute[_batch]: the host-only head of prove() (TranscriptRng chain), see bp_stmt_precompute. */
int bp_prover_set_rng(bp_cs* prover, const uint8_t rng_bytes[32]);
int bp_prover_precompute(bp_cs* prover, const uint8_t rng_bytes_or_null[32]);
int bp_prover_precompute_batch(bp_cs** provers, size_t count);
int bp_prover_prove(bp_ctx* ctx, bp_cs* prover, const uint8_t rng_bytes_or_null[32], uint8_t* proof_out, size_t* proof_len, double* timing);
/* `verifier.verify(&proof, &pc_gens, &bp_gens)` (verifier.rs:549-557): consumes the verifier */
int bp_verifier_verify(bp_ctx* ctx, bp_cs* verifier, const uint8_t* proof, size_t proof_len);
/* `batch_verify(prng, instances, &pc_gens, &bp_gens)` (verifier.rs:604-691): instance k = (verifiers[k], the k-th of the
 * concatenated compressed proofs); every verifier is consumed.  Instances whose recordings have the same STRUCTURE (same gadget;
 * coefficient values may differ: public constants, challenge-dependent randomized constraints) are evaluated by one kernel launch
 * per block of 512 from a GPU-resident circuit template.  alphas: count x 4 words (ark Montgomery form), the weights the reference
 * draws per instance from its prng (verifier.rs:649) — REQUIRED for count > 1 (BP_E_ARG otherwise: with equal weights the errors of
 * two invalid proofs can cancel in the one mega-check); NULL is accepted for count == 1 only (Verifier::verify, weight 1).  The same
 * handle twice in one batch is BP_E_ARG at any count.  A verifier created by bp_verifier_new_like that holds fewer commitments than
 * the shared constraints name is BP_E_ARG (the reference would index past its V vector).  check_point_xy (may be NULL): the value
 * of the mega-check MSM, all-zero iff it is the identity.  timing (5 doubles, may be NULL) as in bp_r1cs_batch_verify_scenarios. */
int bp_r1cs_batch_verify(bp_ctx* ctx, size_t count, bp_cs* const* verifiers, const uint8_t* proofs, const size_t* proof_lens, const uint64_t* alphas,
                         double* timing, uint64_t* check_point_xy);
/* The next instance of the SAME gadget without recording it again: a verifier that shares `of`'s phase-1 constraints and its
 * randomized-constraint callbacks (ref-counted, immutable from now on: `of` and the new verifier accept commits and randomized
 * constraints only).  The caller replays just the instance's own transcript traffic: bp_verifier_commit in the same order as on
 * `of`, plus whatever the gadget appends through bp_cs_transcript.  The reference has no such call — every `Verifier` there records
 * its gadget anew (verifier.rs:69-224); with 4096 instances of a 2^14-multiplier circuit that recording is the whole cost. */
int bp_verifier_new_like(bp_cs* of, void* transcript, bp_cs** out);
/* merlin::Transcript <-> bp_transcript handle: Strobe128 {state[200], pos, pos_begin, cur_flags} (merlin 3.0 src/strobe.rs) */
int bp_transcript_export_state(const void* transcript, uint8_t out[203]);
int bp_transcript_import_state(void* transcript, const uint8_t in[203]);
void* bp_transcript_clone(const void* transcript);

/* ---- r1cs::Verifier::verify / batch_verify ------------------------------------------------------------
 * bp_r1cs_verify_scenario replaces `Verifier::new` + commits + gadget + `verify(&proof, &pc_gens, &bp_gens)`
 * (src/r1cs/verifier.rs:252-287, 549-600) for the same scenarios as the prover: returns BP_OK, or
 * BP_E_VERIFICATION / BP_E_FORMAT / BP_E_GENS_LENGTH exactly where the reference returns the matching R1CSError.
 * bp_r1cs_batch_verify_scenarios replaces `batch_verify(prng, instances, pc_gens, bp_gens)` (:604-691):
 * `count` instances with concatenated proofs / commitments / publics, params 8 u64 per instance; the per-proof
 * alpha is `Fr::rand` of a ChaCha20 rng seeded with alpha_seed.  The instances go through in blocks: the host replays the
 * transcripts of one block while the GPU evaluates the previous one.  timing[5] (s): [0] the whole call, [1] host replay
 * inside the block loop (overlapped with the GPU), [2] wait for the GPU after the last block + tail scaling, [3] final MSM,
 * [4] proof decoding (framing + point decompression).  Proof-sharded multi-GPU use: rank r passes its slice of the instances, alpha_skip = number
 * of instances on lower ranks (their alphas are drawn and discarded), and receives the affine value of ITS mega-check
 * in check_point_xy (may be NULL).  The batch is valid iff EVERY rank's call returned BP_OK (then every check point is the
 * identity, and so is their sum — by linearity the reference's single MSM, :685).  A call can fail before its MSM runs (an
 * identity A_I1 / T_i / L_j, mismatched L_vec, malformed bytes: verifier.rs:420-470 return early); check_point_xy is then
 * all-zero, so the point sum alone must never decide — reduce the statuses first (parallel.sharded_batch_verify). */
int bp_r1cs_verify_scenario(bp_ctx* ctx, int scenario, const uint64_t* params, const uint8_t* proof, size_t proof_len, const uint64_t* commit_xy,
                            size_t m, const uint64_t* publics, size_t npub);
int bp_r1cs_batch_verify_scenarios(bp_ctx* ctx, size_t count, const int* scenarios, const uint64_t* params, const uint8_t* proofs,
                                   const size_t* proof_lens, const uint64_t* commit_xy, const size_t* ms, const uint64_t* publics,
                                   const size_t* npubs, const uint8_t alpha_seed[32], double* timing, size_t alpha_skip,
                                   uint64_t* check_point_xy);

/* ---- profiling: HIP-event time of the dominant kernel of the last call, on the ctx stream ---------- */
#define BP_K_MSM_ACCUM 0   /* bucket accumulation (k_msm_accum) */
#define BP_K_MSM_TOTAL 1   /* all MSM kernels of the call, first launch to last */
#define BP_K_IPA_SCALARS 2 /* k_ipa_scalars + k_ipa_ip_finish of one round */
#define BP_K_IPA_FOLD 3    /* k_ipa_fold_ab + k_ipa_fold_pts of one round */
#define BP_K_R1CS_POLY 4   /* k_r1cs_poly_t / k_r1cs_poly_eval */
#define BP_K_VFY_SCALARS 5 /* k_vfy_scalars */
#define BP_K_FOLD_TAB 6     /* k_ipa_fold_tab: the first fold round through the fixed-base tables */
#define BP_K_FOLD_LADDER 7  /* k_ipa_fold_glv / k_ipa_fold_uniform / k_ipa_fold_pts: the scalar-multiplication ladders of a round */
#define BP_K_FOLD_FINISH 8  /* k_ipa_fold_finish: Jacobian -> affine with shared inversions */
#define BP_K_MSM_ACCUM_FS 9 /* k_msm_accum_fs: accumulate of the fixed-shape pipeline (mid-size MSMs) */
#define BP_K_MSM_AGG 10     /* bucket reduction + aggregation (k_msm_reduce*, k_msm_marginals*, k_msm_window_sums) */
#define BP_K_VFY_TABLES 11  /* k_vfy_tables */
#define BP_K_VFE_POINTS 12  /* k_vfe_points: decompression + serialization of a batch's points (device front end of batch verification) */
#define BP_K_VFE_SPONGE 13  /* k_vfe_sponge: transcript replay + challenge derivation, one lane per proof */
#define BP_K_VFE_PREPARE 14 /* k_vfe_consts + k_vfe_wv + k_vfe_sum2: challenge arithmetic, parameter blocks, tail scalars */
#define BP_K_COUNT 15
int bp_ctx_set_profiling(bp_ctx* ctx, int enabled);
/* accumulated milliseconds and launch count since the last reset */
int bp_ctx_kernel_time(bp_ctx* ctx, int which, double* ms_total, uint64_t* launches);
int bp_ctx_reset_profiling(bp_ctx* ctx);

/* Size thresholds at which the engine switches kernels (results never depend on them; the tests lower them to drive the
 * large-input paths with small inputs).  FOLD_BATCH_MIN: output points per IPA fold round from which the affine conversion
 * shares inversions (default 65536); MSM_BIN_MIN: terms from which the MSM uses the two-level sort (default 64);
 * IPA_FREEZE_LEN: vector length from which bp_ipa_create / the prover stop folding G and H and fold per-element coefficients
 * over the frozen vectors instead (default 8192; 0 or 1 = never); MSM_WSUM_MIN: total bucket count from which an MSM aggregates
 * its buckets by running sums instead of bit marginals (default 2^18). */
#define BP_TUNE_FOLD_BATCH_MIN 0
#define BP_TUNE_MSM_BIN_MIN 1
#define BP_TUNE_IPA_FREEZE_LEN 2
#define BP_TUNE_MSM_WSUM_MIN 3
#define BP_TUNE_CYCLIC_MIN 4   /* padded size from which a sharded prover partitions the IPA index-cyclically (default 2^14) */
#define BP_TUNE_MSM_FIXED_MIN 5   /* terms from which an MSM over the generator tables uses the fixed-base rows (default 2^20; >= 4096) */
#define BP_TUNE_HOST_THREADS 6    /* host threads of this ctx's pool for the per-instance transcript replays of batch verification (0 = default:
                                     the machine's hardware threads, at most 32, or ARKBP_HOST_THREADS); callers that keep several batches in
                                     flight on several ctxs divide the cores among them */
#define BP_TUNE_FOLD_QUAD_MAX 8    /* fold rounds of the inner-product argument with at most this many output points (G and H together) run with FOUR lanes per
                                     point (quad-cooperative group arithmetic): shorter rounds, more arithmetic; default 0 = never (a prover that keeps the GPU
                                     full gains nothing); a latency-bound single prover sets 2^14 */
#define BP_TUNE_MSM_GLV_MIN 7     /* terms from which a variable-base MSM on secq256k1 splits its scalars with the endomorphism (default 256; a huge value turns it off) */
#define BP_TUNE_WAIT_SLEEP 10     /* microseconds this ctx's host waits sleep between polls of the GPU event, instead of busy-waiting (what the HIP
                                     synchronisation calls do on this stack: one core per waiting thread); 0 = busy wait (default), 1 = 30 us, at most 1000.
                                     30 us: a third of the host CPU per proof for ~2 % of latency, for a prover that keeps several proofs in flight;
                                     batch verification waits often and shortly: 10 us */
#define BP_TUNE_MSM_CHUNK_CAP 9   /* entries per first-level chunk of the mid-size (fixed-shape) MSM pipeline: 8 .. 64; 0 = default (16 for callers that keep
                                     the GPU full; fitted per MSM to whole waves per SIMD for the bp_msm* entry points).  Results never depend on it */
#define BP_TUNE_VFY_DEVICE 11    /* 1 (default): batch verification of like-instances of one single-phase statement runs its per-proof front end on the
                                  * GPU (see "verifier front end on the device" below); 0: always the host replay (A/B, tests) */
#define BP_TUNE_DIRECT_MAX 12    /* statements whose padded size is at most this (default 8192, at most 2^16; 0 = never) are proved over DIRECT WINDOW
                                  * TABLES of the first generators (d * 16^w * base, 60 KiB per base, built by the first such proof of the ctx): every
                                  * MSM of Prover::prove (src/r1cs/prover.rs:516-649) and of InnerProductProof::create
                                  * (src/inner_product_proof.rs:86-213) becomes a sum of table entries, G and H are never folded.  This is the
                                  * latency path for the reference's own benchmark range (benches/r1cs_secq256k1.rs:152-250, 2 .. 2046 multipliers) */
int bp_ctx_set_tuning(bp_ctx* ctx, int knob, uint64_t value);

/* The O(N) part of `Verifier::verification_scalars` (src/r1cs/verifier.rs:465-514, s from inner_product_proof.rs:279-311) for a
 * caller that records constraints and replays the transcript itself: from the flattened wL, wR, wO (n each; zero beyond n), the
 * challenges y, x, u (phase-2 separator), the proof's a, b and the k inner-product challenges u_j (creation order), with
 * n1 phase-1 multipliers and padded size N = 2^k:
 *   g[i] = u_or_1(i) * (x * y^-i * wR[i] - a * s[i]),   h[i] = u_or_1(i) * (y^-i * (x * wL[i] + wO[i] - b * s[N-1-i]) - 1),  i < N
 * (u_or_1 = 1 for i < n1, u otherwise), written as CANONICAL integers — the form the mega-check MSM consumes (bp_msm_gens with
 * scalars_canonical = 1).  All inputs ark Montgomery words. */
int bp_r1cs_verification_gh(bp_ctx* ctx, size_t n, size_t n1, const uint64_t* wL, const uint64_t* wR, const uint64_t* wO, const uint64_t y[4],
                            const uint64_t x[4], const uint64_t u[4], const uint64_t a[4], const uint64_t b[4], const uint64_t* ipa_challenges, size_t k,
                            uint64_t* g_out, uint64_t* h_out);

/* ---- unit-test hooks: one field / group operation per element on the GPU -------------------------- */
/* field: 2*curve + (0 base field | 1 scalar field); op: 0 mul, 1 add, 2 sub, 3 sqr, 4 inv */
int bp_debug_field_op(bp_ctx* ctx, int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* the reference's two fixed-value unit tests, through the kernels' own code paths: out[i] = x^i, i < n (`exp_iter`, src/util.rs:55-58,
 * test :147-157) by the power-table routine the prover / verifier kernels use; and <a, b> over the ctx's scalar field
 * (`inner_product`, src/inner_product_proof.rs:390-399, tests :556-562 and src/util.rs:160-166) by the kernels that form
 * c_L = <a_L, b_R> in InnerProductProof::create.  Inputs / outputs: ark Montgomery words. */
int bp_debug_exp_iter(bp_ctx* ctx, const uint64_t x[4], size_t n, uint64_t* out);
int bp_debug_inner_product(bp_ctx* ctx, const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]);
/* ark-serialize compressed SW points (33 bytes each: x LE || flag byte) -> affine, on the GPU (the square roots of
 * `R1CSProof::from_bytes`, src/r1cs/proof.rs:83-91); out_ok[i] = 0 for malformed or off-curve encodings */
int bp_debug_decompress(bp_ctx* ctx, const uint8_t* compressed33, size_t n, uint64_t* out_xy, uint32_t* out_ok);
/* host only: GLV split of a fold multiplier t (ark words, curve 0 only): masks = p1[5] m1[5] p2[5] m2[5] signed-digit bit masks
 * with t = t1 + lambda*t2 (mod r); lambda_out = canonical lambda.  The uniform IPA fold (src/inner_product_proof.rs:139-150)
 * runs its ladder over these 130-digit halves on secq256k1. */
int bp_debug_glv_decompose(int curve, const uint64_t t[4], uint32_t masks[20], uint64_t lambda_out[4]);
/* ---- verifier front end on the device (csrc/vfe.hip, csrc/vfe_sched.hpp) ------------------------------------------------------
 * For a batch of like-instances of ONE single-phase statement bp_r1cs_batch_verify / bp_r1cs_batch_verify_scenarios run the
 * per-proof front of `batch_verify` (src/r1cs/verifier.rs:604-691) on the GPU: R1CSProof::from_bytes' point decompression
 * (src/r1cs/proof.rs:83-91), the merlin transcript replay of verification_scalars (verifier.rs:403-460, 516-519;
 * src/inner_product_proof.rs:266-280; src/transcript.rs:45-102: Keccak-f[1600] / STROBE-128 / ChaCha20 -> Fr::rand) and the
 * O(k + m) challenge arithmetic (:462-541).  Anything unusual (malformed bytes, an identity point that
 * validate_and_append_point rejects, mixed shapes, randomized constraints) takes the host replay instead, which reports the
 * reference's error.  ARKBP_VFY_HOST=1 in the environment forces the host replay (A/B).
 *
 * bp_debug_vfe_schedule_replay (host only, no GPU): builds the data-independent sponge schedule of one verification — transcript
 * at state203 (bp_transcript_export_state), m commitments (absorbed as items 0..m-1 when absorb_commitments), k rounds, n = the
 * padded multiplier count appended under "n" — and runs it on the CPU over `items` (72 bytes each: a 65-byte uncompressed point
 * or a 32-byte scalar, zero padded; order: [commitments] A_I1 A_O1 S1 A_I2 A_O2 S2 T_1 T_3 T_4 T_5 T_6 L[k] R[k] t_x t_x_blinding
 * e_blinding).  seeds_out: (6 + k) x 32 bytes, the challenge_bytes outputs y z u x w u_1..u_k r; *nblocks_out = permutations.
 * bp_debug_vfe_challenges (GPU): the same through k_vfe_points + k_vfe_sponge for `count` proofs of one shape (proof_len bytes
 * each; commit_xy count x m x 8 words, ark layout; states203: count x 203 bytes, or one when shared_state): seeds_out as above per
 * proof, chal_out the derived challenge scalars (count x (6 + k) x 4 ark words), *status_out the OR of the kernels' reject bits
 * (1 malformed, 2 identity under validation, 4 framing).
 * bp_ctx_vfe_stats: batches the device front end completed / batches it handed to the host replay on this ctx. */
int bp_debug_vfe_schedule_replay(const uint8_t state203[203], int absorb_commitments, uint64_t m, uint32_t k, uint64_t n, const uint8_t* items, uint8_t* seeds_out,
                                 uint32_t* nblocks_out);
int bp_debug_vfe_challenges(bp_ctx* ctx, size_t count, const uint8_t* proofs, size_t proof_len, const uint64_t* commit_xy, size_t m, const uint8_t* states203,
                            int shared_state, int absorb_commitments, uint8_t* seeds_out, uint64_t* chal_out, uint32_t* status_out);
int bp_ctx_vfe_stats(bp_ctx* ctx, uint64_t* device_batches, uint64_t* host_fallbacks);
/* MSMs over the generator tables that took the fixed-base schedule (bp_gens_msm_tables) on this ctx, and how many of those ran on a
 * rank's share of the terms of a sharded proof (blocks of a commitment's terms, the strided slice of the first IPA round) */
int bp_ctx_msm_stats(bp_ctx* ctx, uint64_t* fixed_base_runs, uint64_t* fixed_base_runs_sharded);
/* The small-statement path (BP_TUNE_DIRECT_MAX): MSMs this ctx answered from its direct window tables, and the number of generators
 * per vector those tables cover (0 = not built yet) */
int bp_ctx_direct_stats(bp_ctx* ctx, uint64_t* direct_msms, size_t* bases_per_vector);
/* Two fold rounds from the tables (bp_gens_fold_tables over 3N/4 bases; src/inner_product_proof.rs:143-155, 219-224): first folds this
 * ctx deferred, and second folds it then produced straight from the tables — on one GPU and, since round 4, on the index-cyclic slice
 * of a sharded prover */
int bp_ctx_fold_stats(bp_ctx* ctx, uint64_t* deferred_first_folds, uint64_t* second_folds_from_tables);
/* A ctx WITHOUT a device for sanitizer runs of the host layer on machines with no GPU (tools/sanitize/): only bp_r1cs_batch_verify,
 * bp_r1cs_batch_verify_scenarios, bp_ctx_set_tuning and bp_ctx_destroy accept it.  They run the complete host side of batch
 * verification — framing, square roots (on the host here), thread pools, shared recordings, transcript replay (live and lockstep),
 * template cache and eviction, staging — and skip every kernel, copy and the mega-check: NOTHING IS VERIFIED.  The status is that of
 * the host replay (format / validation errors in instance order); check_point_xy receives [checksum of everything the host staged,
 * count, launched groups, 0 ..] so that runs with different thread counts and replay modes can be compared.  gens_capacity: the
 * generator count the ctx pretends to hold. */
int bp_debug_ctx_create_hostonly(int curve, size_t gens_capacity, bp_ctx** out);
/* host only (no GPU): the Fiat-Shamir challenges y z u x w u_1..u_k r of up to eight scenario verifications (flat arrays as in
 * bp_r1cs_batch_verify_scenarios), derived by the per-proof live transcript replay (use_x8 = 0: verify_prepare over the instance's own
 * merlin transcript, src/r1cs/verifier.rs:403-460 + :516-519) or by the lockstep replay of eight same-shaped single-phase instances
 * (use_x8 = 1; AVX-512 Keccak-f x8).  out: count x 40 x 4 ark words, nchal[j] = challenges of instance j.  Returns 1 (not an error
 * code) when the lockstep replay does not apply: fewer than eight instances, differing shapes, a randomized phase, no AVX-512. */
int bp_debug_verify_challenges(int curve, size_t count, const int* scenarios, const uint64_t* params, const uint8_t* proofs, const size_t* proof_lens,
                               const uint64_t* commit_xy, const size_t* ms, const uint64_t* publics, const size_t* npubs, int use_x8, uint64_t* out,
                               size_t* nchal);
/* op: 0 P+Q (general add), 1 P+Q (mixed add), 2 2P, 3 k*P (k canonical, one per element) */
int bp_debug_point_op(bp_ctx* ctx, int op, const uint64_t* p_xy, const uint64_t* q_xy, const uint64_t* k, uint64_t* out_xy, size_t n);

#ifdef __cplusplus
}
#endif
#endif
