// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).  "parity unpinned".
// C entry points (ctypes) over the CPU restatement.  Built by oracle/Makefile into
// oracle/liboracle.so; loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
//
// Data conventions (same as include/arkbp.h so buffers can be handed to both sides unchanged):
//   field element = 4 x u64 little-endian limbs, Montgomery form (R = 2^256), ark's in-memory Fp256;
//   affine point  = 8 x u64 (x || y); the identity is encoded as all-zero.
#include <chrono>
#include <map>
#include "gadgets.hpp"

using namespace orc;

static Curve g_curves[2];
static bool g_init = false;
static PedersenGens g_pc[2];
static BulletproofGens g_bp[2];

static void parse_dec(const char* s, u64 out[4]) {
    memset(out, 0, 32);
    for (; *s; s++) {
        u128 carry = (u128)(*s - '0');
        for (int i = 0; i < 4; i++) { u128 t = (u128)out[i] * 10 + carry; out[i] = (u64)t; carry = t >> 64; }
    }
}
static Fe fe_dec(const Field& F, const char* s) { u64 c[4]; parse_dec(s, c); return F.from_canon(c); }

static void init_once() {
    if (g_init) return;
    u64 m[4];
    {   // secq256k1 (ark-secq256k1 0.4.0): Fq = secp256k1 group order, Fr = secp256k1 base prime
        Curve& C = g_curves[0]; C.id = 0;
        parse_dec("115792089237316195423570985008687907852837564279074904382605163141518161494337", m); C.fq.init(m);
        parse_dec("115792089237316195423570985008687907853269984665640564039457584007908834671663", m); C.fr.init(m);
        C.a = C.fq.Z; C.a_zero = true; C.b = C.fq.from_u64(7);
        C.gen.x = fe_dec(C.fq, "53718550993811904772965658690407829053653678808745171666022356150019200052646");
        C.gen.y = fe_dec(C.fq, "28941648020349172432234515805717979317553499307621291159490218670604692907903");
        C.gen.inf = false;
    }
    {   // zorro: src/curve/zorro/fq.rs:4, fr.rs:1 (= ed25519 base field 2^255-19), g1.rs:24-46
        Curve& C = g_curves[1]; C.id = 1;
        parse_dec("57896044618658097711785492504343953927116110621106131396339151912985063395361", m); C.fq.init(m);
        parse_dec("57896044618658097711785492504343953926634992332820282019728792003956564819949", m); C.fr.init(m);
        C.a = C.fq.from_u64(6); C.a_zero = false;
        C.b = fe_dec(C.fq, "7277470329389939148381533754641607518092114590371880995609984561067837624798");
        C.gen.x = C.fq.from_u64(2);
        C.gen.y = fe_dec(C.fq, "19711758720854384559191066596451394956860102304684364148268676039962145446511");
        C.gen.inf = false;
    }
    for (int c = 0; c < 2; c++) { g_pc[c] = pedersen_default(g_curves[c]); g_bp[c].build(g_curves[c], 0, 1); }
    g_init = true;
}
static const Field& field_of(int fid) { init_once(); return fid & 1 ? g_curves[fid >> 1].fr : g_curves[fid >> 1].fq; }

static Aff aff_in(const u64* xy) {
    Aff p; memcpy(p.x.v, xy, 32); memcpy(p.y.v, xy + 4, 32);
    p.inf = p.x.is_zero() && p.y.is_zero();
    return p;
}
static void aff_out(u64* xy, const Aff& p) {
    if (p.inf) { memset(xy, 0, 64); return; }
    memcpy(xy, p.x.v, 32); memcpy(xy + 4, p.y.v, 32);
}
static const BulletproofGens& gens_for(int curve, size_t cap) {
    init_once();
    g_bp[curve].increase_capacity(g_curves[curve], cap);
    return g_bp[curve];
}

extern "C" {

int orc_init() { init_once(); return 0; }
// threads over the Pippenger windows of every msm call (the reference's `parallel` feature, Cargo.toml:76); 1 = default features
void orc_set_msm_threads(int n) { Curve::msm_threads() = n < 1 ? 1 : n; }
int orc_get_msm_threads() { return Curve::msm_threads(); }

// ---- field primitives: fid = 2*curve + (0: Fq base field, 1: Fr scalar field) ---------------
void orc_fe_modulus(int fid, u64* out) { memcpy(out, field_of(fid).p, 32); }
void orc_fe_mul(int fid, const u64* a, const u64* b, u64* o) { field_of(fid).mul(*(Fe*)o, *(const Fe*)a, *(const Fe*)b); }
void orc_fe_add(int fid, const u64* a, const u64* b, u64* o) { field_of(fid).add(*(Fe*)o, *(const Fe*)a, *(const Fe*)b); }
void orc_fe_sub(int fid, const u64* a, const u64* b, u64* o) { field_of(fid).sub(*(Fe*)o, *(const Fe*)a, *(const Fe*)b); }
int orc_fe_inv(int fid, const u64* a, u64* o) { return field_of(fid).inv(*(Fe*)o, *(const Fe*)a) ? 0 : -1; }
int orc_fe_sqrt(int fid, const u64* a, u64* o) { return field_of(fid).sqrt(*(Fe*)o, *(const Fe*)a) ? 0 : -1; }
void orc_fe_from_canon(int fid, const u64* c, u64* o) { *(Fe*)o = field_of(fid).from_canon(c); }
void orc_fe_to_canon(int fid, const u64* a, u64* c) { field_of(fid).to_canon(c, *(const Fe*)a); }
// Fp::rand from a ChaCha20 seed: `count` consecutive draws
void orc_fe_rand(int fid, const u8* seed, size_t count, u64* out) {
    ChaCha20Rng r; r.seed(seed);
    for (size_t i = 0; i < count; i++) *(Fe*)(out + 4 * i) = fe_rand(field_of(fid), r);
}

// util::exp_iter (src/util.rs:55-58): 1, x, x^2, ... by repeated multiplication, as the reference iterates; and
// inner_product (src/inner_product_proof.rs:390-399).  The reference pins both: exp_iter(2) -> 1,2,4,8 (util.rs:147-157),
// inner_product([1,2,3,4],[2,3,4,5]) = 40 over secq256k1's Fr (util.rs:160-166) and secp256k1's Fr = secq256k1's Fq (:556-562).
void orc_exp_iter(int fid, const u64* x, size_t n, u64* out) {
    const Field& F = field_of(fid);
    Fe cur = F.R1;
    for (size_t i = 0; i < n; i++) { *(Fe*)(out + 4 * i) = cur; F.mul(cur, cur, *(const Fe*)x); }
}
void orc_inner_product(int fid, const u64* a, const u64* b, size_t n, u64* out) { *(Fe*)out = inner_product(field_of(fid), (const Fe*)a, (const Fe*)b, n); }

// ---- byte primitives -----------------------------------------------------------------------
void orc_sha3_512(const u8* msg, size_t n, u8* out) { sha3_512(out, msg, n); }
void orc_chacha20_words(const u8* seed, size_t nwords, u32* out) {
    ChaCha20Rng r; r.seed(seed);
    for (size_t i = 0; i < nwords; i++) out[i] = r.next_u32();
}
void* orc_transcript_new(const u8* label, size_t n) { return new Transcript(label, n); }
void* orc_transcript_clone(void* t) { return new Transcript(*(Transcript*)t); }
void orc_transcript_free(void* t) { delete (Transcript*)t; }
void orc_transcript_append_message(void* t, const char* label, const u8* msg, size_t n) { ((Transcript*)t)->append_message(label, msg, n); }
void orc_transcript_append_u64(void* t, const char* label, u64 x) { ((Transcript*)t)->append_u64(label, x); }
void orc_transcript_challenge_bytes(void* t, const char* label, u8* out, size_t n) { ((Transcript*)t)->challenge_bytes(label, out, n); }
void orc_transcript_append_point(int curve, void* t, const char* label, const u64* xy) { init_once(); TP::append_point(g_curves[curve], *(Transcript*)t, label, aff_in(xy)); }
void orc_transcript_append_scalar(int curve, void* t, const char* label, const u64* s) { init_once(); TP::append_scalar(g_curves[curve], *(Transcript*)t, label, *(const Fe*)s); }
void orc_transcript_challenge_scalar(int curve, void* t, const char* label, u64* out) {
    init_once();
    *(Fe*)out = TP::challenge_scalar(g_curves[curve], *(Transcript*)t, label);
}
// TranscriptRng: build_rng().rekey("v_blinding", w_i)...finalize(ChaCha20(seed)), then `count` Fr::rand draws
void orc_transcript_rng_draws(int curve, void* t, const u64* witness, size_t nw, const u8* seed, size_t count, u64* out) {
    init_once();
    const Field& F = g_curves[curve].fr;
    TranscriptRng rng(*(Transcript*)t);
    for (size_t i = 0; i < nw; i++) { u8 b[32]; F.to_bytes(b, *(const Fe*)(witness + 4 * i)); rng.rekey_with_witness_bytes("v_blinding", b, 32); }
    ChaCha20Rng ext; ext.seed(seed);
    rng.finalize(ext);
    for (size_t i = 0; i < count; i++) *(Fe*)(out + 4 * i) = fe_rand(F, rng);
}

// ---- group primitives -----------------------------------------------------------------------
void orc_generator(int curve, u64* xy) { init_once(); aff_out(xy, g_curves[curve].gen); }
int orc_on_curve(int curve, const u64* xy) { init_once(); return g_curves[curve].on_curve(aff_in(xy)) ? 1 : 0; }
void orc_point_add(int curve, const u64* p, const u64* q, u64* out) {
    init_once(); const Curve& C = g_curves[curve];
    Jac r; C.add(r, C.to_jac(aff_in(p)), C.to_jac(aff_in(q)));
    aff_out(out, C.to_affine(r));
}
void orc_scalar_mul(int curve, const u64* p, const u64* s, u64* out) {
    init_once(); const Curve& C = g_curves[curve];
    aff_out(out, C.to_affine(C.mul(aff_in(p), *(const Fe*)s)));
}
void orc_point_ser(int curve, const u64* p, int compressed, u8* out) {
    init_once(); const Curve& C = g_curves[curve];
    if (compressed) C.ser_compressed(out, aff_in(p)); else C.ser_uncompressed(out, aff_in(p));
}
int orc_point_deser_compressed(int curve, const u8* in, u64* out) {
    init_once(); Aff p;
    if (!g_curves[curve].deser_compressed(p, in)) return -1;
    aff_out(out, p); return 0;
}
// VariableBaseMSM::msm(bases, scalars).into_affine()
void orc_msm(int curve, const u64* bases_xy, const u64* scalars, size_t n, u64* out_xy) {
    init_once(); const Curve& C = g_curves[curve];
    std::vector<Aff> b(n);
    for (size_t i = 0; i < n; i++) b[i] = aff_in(bases_xy + 8 * i);
    aff_out(out_xy, C.to_affine(C.msm(b.data(), (const Fe*)scalars, n)));
}
// same, returns seconds spent inside msm (cpu_baseline leg of bench.py)
double orc_msm_timed(int curve, const u64* bases_xy, const u64* scalars, size_t n, u64* out_xy) {
    init_once(); const Curve& C = g_curves[curve];
    std::vector<Aff> b(n);
    for (size_t i = 0; i < n; i++) b[i] = aff_in(bases_xy + 8 * i);
    auto t0 = std::chrono::steady_clock::now();
    Jac r = C.msm(b.data(), (const Fe*)scalars, n);
    auto t1 = std::chrono::steady_clock::now();
    aff_out(out_xy, C.to_affine(r));
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---- generators ---------------------------------------------------------------------------
void orc_pedersen_default(int curve, u64* B_xy, u64* Bb_xy) { init_once(); aff_out(B_xy, g_pc[curve].B); aff_out(Bb_xy, g_pc[curve].B_blinding); }
void orc_pedersen_commit(int curve, const u64* v, const u64* blind, u64* out) {
    init_once(); aff_out(out, g_pc[curve].commit(g_curves[curve], *(const Fe*)v, *(const Fe*)blind));
}
// BulletproofGens::new(cap, 1).share(0).G(cap) / .H(cap)
void orc_bp_gens(int curve, size_t cap, u64* G_xy, u64* H_xy) {
    const BulletproofGens& g = gens_for(curve, cap);
    for (size_t i = 0; i < cap; i++) { aff_out(G_xy + 8 * i, g.G_vec[0][i]); aff_out(H_xy + 8 * i, g.H_vec[0][i]); }
}
// generic party: BulletproofGens::new(cap, party+1) tables of one party (not cached)
void orc_bp_gens_party(int curve, size_t cap, size_t party, u64* G_xy, u64* H_xy) {
    init_once();
    BulletproofGens g; g.build(g_curves[curve], cap, party + 1);
    for (size_t i = 0; i < cap; i++) { aff_out(G_xy + 8 * i, g.G_vec[party][i]); aff_out(H_xy + 8 * i, g.H_vec[party][i]); }
}

// ---- inner-product proof ------------------------------------------------------------------
// InnerProductProof::create; L_out/R_out sized lg n points; returns lg n
int orc_ipa_create(int curve, void* transcript, const u64* Q, const u64* Gf, const u64* Hf, const u64* G, const u64* H, const u64* a,
                   const u64* b, size_t n, u64* L_out, u64* R_out, u64* a_out, u64* b_out) {
    init_once(); const Curve& C = g_curves[curve];
    std::vector<Fe> gf((const Fe*)Gf, (const Fe*)Gf + n), hf((const Fe*)Hf, (const Fe*)Hf + n), av((const Fe*)a, (const Fe*)a + n),
        bv((const Fe*)b, (const Fe*)b + n);
    std::vector<Aff> Gv(n), Hv(n);
    for (size_t i = 0; i < n; i++) { Gv[i] = aff_in(G + 8 * i); Hv[i] = aff_in(H + 8 * i); }
    InnerProductProof pf = ipa_create(C, *(Transcript*)transcript, aff_in(Q), gf, hf, Gv, Hv, av, bv);
    for (size_t i = 0; i < pf.L_vec.size(); i++) { aff_out(L_out + 8 * i, pf.L_vec[i]); aff_out(R_out + 8 * i, pf.R_vec[i]); }
    *(Fe*)a_out = pf.a; *(Fe*)b_out = pf.b;
    return (int)pf.L_vec.size();
}
int orc_ipa_verify(int curve, void* transcript, size_t n, const u64* Gf, const u64* Hf, const u64* P, const u64* Q, const u64* G, const u64* H,
                   const u64* L, const u64* R, size_t lgn, const u64* a, const u64* b) {
    init_once(); const Curve& C = g_curves[curve];
    InnerProductProof pf;
    for (size_t i = 0; i < lgn; i++) { pf.L_vec.push_back(aff_in(L + 8 * i)); pf.R_vec.push_back(aff_in(R + 8 * i)); }
    pf.a = *(const Fe*)a; pf.b = *(const Fe*)b;
    std::vector<Fe> gf((const Fe*)Gf, (const Fe*)Gf + n), hf((const Fe*)Hf, (const Fe*)Hf + n);
    std::vector<Aff> Gv(n), Hv(n);
    for (size_t i = 0; i < n; i++) { Gv[i] = aff_in(G + 8 * i); Hv[i] = aff_in(H + 8 * i); }
    return ipa_verify(C, pf, n, *(Transcript*)transcript, gf, hf, aff_in(P), aff_in(Q), Gv, Hv);
}
// InnerProductProof::verification_scalars: outputs u_sq[lgn], u_inv_sq[lgn], s[n]
int orc_ipa_verification_scalars(int curve, void* transcript, size_t n, const u64* L, const u64* R, size_t lgn, u64* u_sq, u64* u_inv_sq, u64* s) {
    init_once(); const Curve& C = g_curves[curve];
    InnerProductProof pf;
    for (size_t i = 0; i < lgn; i++) { pf.L_vec.push_back(aff_in(L + 8 * i)); pf.R_vec.push_back(aff_in(R + 8 * i)); }
    pf.a = C.fr.Z; pf.b = C.fr.Z;
    std::vector<Fe> us, uis, sv;
    Err e = ipa_verification_scalars(C, pf, n, *(Transcript*)transcript, us, uis, sv);
    if (e) return e;
    memcpy(u_sq, us.data(), 32 * lgn); memcpy(u_inv_sq, uis.data(), 32 * lgn); memcpy(s, sv.data(), 32 * n);
    return 0;
}

// ---- R1CS scenarios ------------------------------------------------------------------------
// prove: returns Err; proof bytes (compressed wire format) -> proof_out/proof_len (capacity in *proof_len),
// commitments -> commit_xy (capacity m_cap points) and *m_out, scenario publics -> publics (4 u64 each), *npub.
// timing[0] = seconds inside Prover::prove (the reference's `prove()`), timing[1] = seconds in commits + gadget.
int orc_r1cs_prove(int curve, int scenario, const u64* params, const u8* seed, size_t gens_cap, u8* proof_out, size_t* proof_len, u64* commit_xy,
                   size_t m_cap, size_t* m_out, u64* publics, size_t* npub, double* timing) {
    const BulletproofGens& bp = gens_for(curve, gens_cap);
    const Curve& C = g_curves[curve];
    ChaCha20Rng prng; prng.seed(seed);
    Transcript tr(scenario_label(scenario));
    Prover p(C, g_pc[curve], tr);
    ScenarioIO io;
    auto t0 = std::chrono::steady_clock::now();
    Err e = scenario_prover_setup(p, prng, scenario, params, io);
    if (e) return e;
    auto t1 = std::chrono::steady_clock::now();
    R1CSProof proof;
    e = p.prove(prng, bp, proof);
    auto t2 = std::chrono::steady_clock::now();
    if (timing) { timing[0] = std::chrono::duration<double>(t2 - t1).count(); timing[1] = std::chrono::duration<double>(t1 - t0).count(); }
    if (e) return e;
    std::vector<u8> bytes = proof.to_bytes(C);
    if (bytes.size() > *proof_len || io.commitments.size() > m_cap) return -2;
    memcpy(proof_out, bytes.data(), bytes.size()); *proof_len = bytes.size();
    for (size_t i = 0; i < io.commitments.size(); i++) aff_out(commit_xy + 8 * i, io.commitments[i]);
    *m_out = io.commitments.size();
    for (size_t i = 0; i < io.publics.size(); i++) memcpy(publics + 4 * i, io.publics[i].v, 32);
    *npub = io.publics.size();
    return 0;
}

static Err build_verifier(Verifier& v, int scenario, const u64* params, const u64* commit_xy, size_t m, const u64* publics, size_t npub) {
    ScenarioIO io;
    for (size_t i = 0; i < m; i++) io.commitments.push_back(aff_in(commit_xy + 8 * i));
    for (size_t i = 0; i < npub; i++) io.publics.push_back(*(const Fe*)(publics + 4 * i));
    return scenario_verifier_setup(v, scenario, params, io);
}

int orc_r1cs_verify(int curve, int scenario, const u64* params, size_t gens_cap, const u8* proof_bytes, size_t proof_len, const u64* commit_xy,
                    size_t m, const u64* publics, size_t npub, double* timing) {
    const BulletproofGens& bp = gens_for(curve, gens_cap);
    const Curve& C = g_curves[curve];
    R1CSProof proof;
    Err e = R1CSProof::from_bytes(C, proof_bytes, proof_len, proof);
    if (e) return e;
    Transcript tr(scenario_label(scenario));
    Verifier v(C, tr);
    e = build_verifier(v, scenario, params, commit_xy, m, publics, npub);
    if (e) return e;
    auto t0 = std::chrono::steady_clock::now();
    e = v.verify(proof, g_pc[curve], bp);
    auto t1 = std::chrono::steady_clock::now();
    if (timing) timing[0] = std::chrono::duration<double>(t1 - t0).count();
    return e;
}

// Verifier::verification_scalars: writes the full scalar vector (2 + 2N + 6 + m + 5 + 2k), returns Err; *count = length
int orc_r1cs_verification_scalars(int curve, int scenario, const u64* params, size_t gens_cap, const u8* proof_bytes, size_t proof_len,
                                  const u64* commit_xy, size_t m, const u64* publics, size_t npub, u64* scalars_out, size_t cap, size_t* count) {
    const BulletproofGens& bp = gens_for(curve, gens_cap);
    const Curve& C = g_curves[curve];
    R1CSProof proof;
    Err e = R1CSProof::from_bytes(C, proof_bytes, proof_len, proof);
    if (e) return e;
    Transcript tr(scenario_label(scenario));
    Verifier v(C, tr);
    e = build_verifier(v, scenario, params, commit_xy, m, publics, npub);
    if (e) return e;
    std::vector<Fe> sc;
    e = v.verification_scalars(proof, bp, sc);
    if (e) return e;
    if (sc.size() > cap) return -2;
    memcpy(scalars_out, sc.data(), 32 * sc.size());
    *count = sc.size();
    return 0;
}

// batch_verify over `count` instances of (scenario, params) with concatenated proofs/commitments/publics
int orc_batch_verify(int curve, size_t count, const int* scenarios, const u64* params /*8 per instance*/, size_t gens_cap, const u8* proofs,
                     const size_t* proof_lens, const u64* commit_xy, const size_t* ms, const u64* publics, const size_t* npubs, const u8* alpha_seed,
                     double* timing) {
    const BulletproofGens& bp = gens_for(curve, gens_cap);
    const Curve& C = g_curves[curve];
    std::vector<std::unique_ptr<Transcript>> trs;
    std::vector<std::unique_ptr<Verifier>> vs;
    std::vector<R1CSProof> pfs(count);
    std::vector<std::pair<Verifier*, const R1CSProof*>> inst;
    size_t poff = 0, coff = 0, uoff = 0;
    for (size_t k = 0; k < count; k++) {
        Err e = R1CSProof::from_bytes(C, proofs + poff, proof_lens[k], pfs[k]);
        if (e) return e;
        trs.emplace_back(new Transcript(scenario_label(scenarios[k])));
        vs.emplace_back(new Verifier(C, *trs.back()));
        e = build_verifier(*vs.back(), scenarios[k], params + 8 * k, commit_xy + 8 * coff, ms[k], publics + 4 * uoff, npubs[k]);
        if (e) return e;
        poff += proof_lens[k]; coff += ms[k]; uoff += npubs[k];
    }
    for (size_t k = 0; k < count; k++) inst.push_back({vs[k].get(), &pfs[k]});
    ChaCha20Rng prng; prng.seed(alpha_seed);
    auto t0 = std::chrono::steady_clock::now();
    Err e = batch_verify(C, prng, inst, g_pc[curve], bp);
    auto t1 = std::chrono::steady_clock::now();
    if (timing) timing[0] = std::chrono::duration<double>(t1 - t0).count();
    return e;
}

// same, also returning the value of the mega-check MSM (identity = all-zero) so a failing batch can be compared point for point
int orc_batch_verify_point(int curve, size_t count, const int* scenarios, const u64* params, size_t gens_cap, const u8* proofs, const size_t* proof_lens,
                           const u64* commit_xy, const size_t* ms, const u64* publics, const size_t* npubs, const u8* alpha_seed, u64* point_out) {
    const BulletproofGens& bp = gens_for(curve, gens_cap);
    const Curve& C = g_curves[curve];
    std::vector<std::unique_ptr<Transcript>> trs;
    std::vector<std::unique_ptr<Verifier>> vs;
    std::vector<R1CSProof> pfs(count);
    std::vector<std::pair<Verifier*, const R1CSProof*>> inst;
    size_t poff = 0, coff = 0, uoff = 0;
    for (size_t k = 0; k < count; k++) {
        Err e = R1CSProof::from_bytes(C, proofs + poff, proof_lens[k], pfs[k]);
        if (e) return e;
        trs.emplace_back(new Transcript(scenario_label(scenarios[k])));
        vs.emplace_back(new Verifier(C, *trs.back()));
        e = build_verifier(*vs.back(), scenarios[k], params + 8 * k, commit_xy + 8 * coff, ms[k], publics + 4 * uoff, npubs[k]);
        if (e) return e;
        poff += proof_lens[k]; coff += ms[k]; uoff += npubs[k];
    }
    for (size_t k = 0; k < count; k++) inst.push_back({vs[k].get(), &pfs[k]});
    ChaCha20Rng prng; prng.seed(alpha_seed);
    Aff mega; mega.inf = true;
    Err e = batch_verify(C, prng, inst, g_pc[curve], bp, nullptr, &mega);
    if (point_out) aff_out(point_out, mega);
    return e;
}

// ---- the reference's ConstraintSystem trait as handles (src/r1cs/constraint_system.rs:19-135), so that a test can run ANY
// gadget — not only the scenarios above — on the restated Prover / Verifier and compare with the product's own recorder.
// Variables cross the boundary as (kind, index) pairs of u32; linear combinations as parallel arrays (vars, coefficients).
struct OrcCs {
    int curve; bool proving;
    Transcript tr;
    std::unique_ptr<Prover> p;
    std::unique_ptr<Verifier> v;
    CS* cs() { return proving ? (CS*)p.get() : (CS*)v.get(); }
    OrcCs(int c, bool pr, const u8* label, size_t n) : curve(c), proving(pr), tr(label, n) {}
};
struct FixedBytesRng {   // the external prng of Prover::prove is only ever asked for 32 bytes (TranscriptRngBuilder::finalize)
    const u8* b;
    void fill_bytes(u8* dst, size_t n) { memcpy(dst, b, n); }
};
typedef int (*orc_randomize_cb)(void* user, void* cs_handle);
static LC lc_in(const u32* vars, const u64* coefs, size_t n) {
    LC l;
    for (size_t i = 0; i < n; i++) { Variable v; v.k = (VarKind)vars[2 * i]; v.i = vars[2 * i + 1]; l.terms.push_back({v, *(const Fe*)(coefs + 4 * i)}); }
    return l;
}
static void var_out(u32* o, const Variable& v) { o[0] = (u32)v.k; o[1] = (u32)v.i; }

// Prover::new / Verifier::new on a transcript created with `label` (the caller may append to it first through orc_cs_transcript)
void* orc_cs_new(int curve, int proving, void* transcript_or_null, const u8* label, size_t n) {
    init_once();
    OrcCs* h = new OrcCs(curve, proving != 0, label, n);
    if (transcript_or_null) h->tr = *(Transcript*)transcript_or_null;
    if (proving) h->p.reset(new Prover(g_curves[curve], g_pc[curve], h->tr)); else h->v.reset(new Verifier(g_curves[curve], h->tr));
    return h;
}
void orc_cs_free(void* h) { delete (OrcCs*)h; }
void* orc_cs_transcript(void* h) { return &((OrcCs*)h)->tr; }
int orc_prover_commit(void* hh, const u64* val, const u64* blind, u64* V_xy, u32* var) {
    OrcCs* h = (OrcCs*)hh; if (!h->proving) return -1;
    Variable v; Aff V = h->p->commit(*(const Fe*)val, *(const Fe*)blind, v);
    aff_out(V_xy, V); var_out(var, v); return 0;
}
int orc_verifier_commit(void* hh, const u64* V_xy, u32* var) {
    OrcCs* h = (OrcCs*)hh; if (h->proving) return -1;
    var_out(var, h->v->commit(aff_in(V_xy))); return 0;
}
int orc_cs_multiply(void* hh, const u32* lv, const u64* lc, size_t nl, const u32* rv, const u64* rc, size_t nr, u32* out3) {
    Variable o[3]; ((OrcCs*)hh)->cs()->multiply(lc_in(lv, lc, nl), lc_in(rv, rc, nr), o);
    for (int i = 0; i < 3; i++) var_out(out3 + 2 * i, o[i]);
    return 0;
}
int orc_cs_allocate(void* hh, const u64* assignment, u32* out) {
    Variable o; Err e = ((OrcCs*)hh)->cs()->allocate((const Fe*)assignment, o);
    if (e) return e;
    var_out(out, o); return 0;
}
int orc_cs_allocate_multiplier(void* hh, const u64* l, const u64* r, u32* out3) {
    Variable o[3]; Err e = ((OrcCs*)hh)->cs()->allocate_multiplier((const Fe*)l, (const Fe*)r, o);
    if (e) return e;
    for (int i = 0; i < 3; i++) var_out(out3 + 2 * i, o[i]);
    return 0;
}
int orc_cs_constrain(void* hh, const u32* vars, const u64* coefs, size_t n) { ((OrcCs*)hh)->cs()->constrain(lc_in(vars, coefs, n)); return 0; }
int orc_cs_specify_randomized_constraints(void* hh, orc_randomize_cb cb, void* user) {
    OrcCs* h = (OrcCs*)hh;
    return h->cs()->specify_randomized_constraints([cb, user, h](CS&) -> Err { return (Err)cb(user, h); });
}
int orc_cs_challenge_scalar(void* hh, const char* label, u64* out) { *(Fe*)out = ((OrcCs*)hh)->cs()->challenge_scalar(label); return 0; }
// Prover::prove(prng, bp_gens) with the 32 bytes the external prng yields
int orc_prover_prove(void* hh, const u8* rng_bytes32, size_t gens_cap, u8* proof_out, size_t* proof_len) {
    OrcCs* h = (OrcCs*)hh; if (!h->proving) return -1;
    const BulletproofGens& bp = gens_for(h->curve, gens_cap);
    FixedBytesRng prng{rng_bytes32};
    R1CSProof proof;
    Err e = h->p->prove(prng, bp, proof);
    if (e) return e;
    std::vector<u8> bytes = proof.to_bytes(g_curves[h->curve]);
    if (bytes.size() > *proof_len) return -2;
    memcpy(proof_out, bytes.data(), bytes.size()); *proof_len = bytes.size();
    return 0;
}
int orc_verifier_verify(void* hh, size_t gens_cap, const u8* proof_bytes, size_t proof_len) {
    OrcCs* h = (OrcCs*)hh; if (h->proving) return -1;
    const BulletproofGens& bp = gens_for(h->curve, gens_cap);
    R1CSProof proof;
    Err e = R1CSProof::from_bytes(g_curves[h->curve], proof_bytes, proof_len, proof);
    if (e) return e;
    return h->v->verify(proof, g_pc[h->curve], bp);
}
// batch_verify over verifier handles with the caller's alphas (count x 4 words); point_out = the mega-check value
int orc_cs_batch_verify(int curve, size_t count, void** handles, size_t gens_cap, const u8* proofs, const size_t* proof_lens, const u64* alphas, u64* point_out) {
    const BulletproofGens& bp = gens_for(curve, gens_cap);
    const Curve& C = g_curves[curve];
    std::vector<R1CSProof> pfs(count);
    std::vector<std::pair<Verifier*, const R1CSProof*>> inst;
    size_t poff = 0;
    for (size_t k = 0; k < count; k++) {
        Err e = R1CSProof::from_bytes(C, proofs + poff, proof_lens[k], pfs[k]);
        if (e) return e;
        poff += proof_lens[k];
        OrcCs* h = (OrcCs*)handles[k];
        if (h->proving || h->curve != curve) return -1;
        inst.push_back({h->v.get(), &pfs[k]});
    }
    ChaCha20Rng unused; u8 z[32] = {0}; unused.seed(z);
    Aff mega; mega.inf = true;
    Err e = batch_verify(C, unused, inst, g_pc[curve], bp, (const Fe*)alphas, &mega);
    if (point_out) aff_out(point_out, mega);
    return e;
}

}  // extern "C"
