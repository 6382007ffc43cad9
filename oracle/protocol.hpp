// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).  "parity unpinned".
//
// CPU restatement of the reference's protocol layer, function by function:
//   generators      src/generators.rs:30-66 (PedersenGens), :71-121 (GeneratorsChain), :150-244 (BulletproofGens)
//   transcript      src/transcript.rs:45-102 (TranscriptProtocol for merlin::Transcript)
//   IPA             src/inner_product_proof.rs:37-239 (create), :244-314 (verification_scalars),
//                   :321-382 (verify), :390-399 (inner_product)
//   util            src/util.rs:55-58 (exp_iter), :75-93 (special_inner_product), :95-109 (evals)
//   R1CS prover     src/r1cs/prover.rs:96-268 (CS recording), :327-341 (commit), :354-397
//                   (flattened_constraints), :454-831 (prove)
//   R1CS verifier   src/r1cs/verifier.rs:69-224, :279-287, :304-349, :394-541, :559-600, :604-691
//   proof codec     src/r1cs/proof.rs:27-91
#pragma once
#include <functional>
#include <memory>
#include <utility>
#include <vector>
#include "curve.hpp"
#include "hash.hpp"

namespace orc {

enum Err { OK = 0, E_VERIFICATION = 1, E_GENS_LENGTH = 2, E_MISSING_ASSIGNMENT = 3, E_FORMAT = 4, E_GADGET = 5 };

// ---- random group element: ark-ec `Distribution<Affine<P>> for Standard` ------------------------
template <class Rng> static inline Aff aff_rand(const Curve& C, Rng& rng) {
    for (;;) {
        Fe x = fe_rand(C.fq, rng);
        bool greatest = ((int32_t)rng.next_u32()) < 0;  // rand 0.8 Standard for bool
        Aff p;
        if (C.point_from_x(p, x, greatest)) return p;   // cofactor 1: mul_by_cofactor is the identity map
    }
}

// ---- generators ------------------------------------------------------------------------
struct PedersenGens {
    Aff B, B_blinding;
    // src/generators.rs:39-44
    Aff commit(const Curve& C, const Fe& value, const Fe& blinding) const {
        Jac a = C.mul(B, value), b = C.mul(B_blinding, blinding), s;
        C.add(s, a, b);
        return C.to_affine(s);
    }
};
// src/generators.rs:47-66
static inline PedersenGens pedersen_default(const Curve& C) {
    u8 bytes[65], h[64];
    C.ser_uncompressed(bytes, C.gen);
    sha3_512(h, bytes, 65);
    ChaCha20Rng prng; prng.seed(h);
    PedersenGens g; g.B = C.gen; g.B_blinding = aff_rand(C, prng);
    return g;
}
// src/generators.rs:77-93
static inline ChaCha20Rng generators_chain(const u8* label, size_t n) {
    std::vector<u8> m((const u8*)"GeneratorsChain", (const u8*)"GeneratorsChain" + 15);
    m.insert(m.end(), label, label + n);
    u8 h[64];
    sha3_512(h, m.data(), m.size());
    ChaCha20Rng prng; prng.seed(h);
    return prng;
}
struct BulletproofGens {
    size_t gens_capacity = 0, party_capacity = 0;
    std::vector<std::vector<Aff>> G_vec, H_vec;
    // src/generators.rs:174-221 (new + increase_capacity from 0)
    void build(const Curve& C, size_t cap, size_t parties) {
        party_capacity = parties; gens_capacity = 0;
        G_vec.assign(parties, {}); H_vec.assign(parties, {});
        increase_capacity(C, cap);
    }
    void increase_capacity(const Curve& C, size_t new_cap) {
        if (gens_capacity >= new_cap) return;
        for (size_t i = 0; i < party_capacity; i++) {
            u32 pi = (u32)i;
            u8 label[5] = {'G', 0, 0, 0, 0};
            memcpy(label + 1, &pi, 4);
            ChaCha20Rng g = generators_chain(label, 5);
            for (size_t k = 0; k < gens_capacity; k++) (void)aff_rand(C, g);  // fast_forward
            for (size_t k = gens_capacity; k < new_cap; k++) G_vec[i].push_back(aff_rand(C, g));
            label[0] = 'H';
            ChaCha20Rng h = generators_chain(label, 5);
            for (size_t k = 0; k < gens_capacity; k++) (void)aff_rand(C, h);
            for (size_t k = gens_capacity; k < new_cap; k++) H_vec[i].push_back(aff_rand(C, h));
        }
        gens_capacity = new_cap;
    }
};

// ---- TranscriptProtocol (src/transcript.rs:45-102) ---------------------------------------------
struct TP {
    static void innerproduct_domain_sep(Transcript& t, u64 n) { t.append_message("dom-sep", "ipp v1"); t.append_u64("n", n); }
    static void r1cs_domain_sep(Transcript& t) { t.append_message("dom-sep", "r1cs v1"); }
    static void r1cs_1phase_domain_sep(Transcript& t) { t.append_message("dom-sep", "r1cs-1phase"); }
    static void r1cs_2phase_domain_sep(Transcript& t) { t.append_message("dom-sep", "r1cs-2phase"); }
    static void append_scalar(const Curve& C, Transcript& t, const char* label, const Fe& s) {
        u8 b[32]; C.fr.to_bytes(b, s); t.append_message(label, b, 32);
    }
    static void append_point(const Curve& C, Transcript& t, const char* label, const Aff& p) {
        u8 b[65]; C.ser_uncompressed(b, p); t.append_message(label, b, 65);
    }
    static bool validate_and_append_point(const Curve& C, Transcript& t, const char* label, const Aff& p) {
        if (p.inf) return false;
        append_point(C, t, label, p);
        return true;
    }
    static Fe challenge_scalar(const Curve& C, Transcript& t, const char* label) {
        u8 buf[32];
        t.challenge_bytes(label, buf, 32);
        ChaCha20Rng prng; prng.seed(buf);
        return fe_rand(C.fr, prng);
    }
};

// src/inner_product_proof.rs:390-399
static inline Fe inner_product(const Field& F, const Fe* a, const Fe* b, size_t n) {
    Fe out = F.Z, t;
    for (size_t i = 0; i < n; i++) { F.mul(t, a[i], b[i]); F.add(out, out, t); }
    return out;
}

// ---- inner-product proof ------------------------------------------------------------------
struct InnerProductProof {
    std::vector<Aff> L_vec, R_vec;
    Fe a, b;
};

// src/inner_product_proof.rs:37-239.  Vectors are taken by value and folded in place, like the
// reference.  The G/H fold is the reference's per-element 2-term msm + into_affine (:143-155, :219-224).
static inline InnerProductProof ipa_create(const Curve& C, Transcript& tr, const Aff& Q, const std::vector<Fe>& G_factors,
                                           const std::vector<Fe>& H_factors, std::vector<Aff> G, std::vector<Aff> H,
                                           std::vector<Fe> a, std::vector<Fe> b) {
    const Field& F = C.fr;
    size_t n = G.size();
    if (H.size() != n || a.size() != n || b.size() != n || G_factors.size() != n || H_factors.size() != n || (n & (n - 1)) || n == 0) {
        fprintf(stderr, "ipa_create: bad lengths\n"); abort();
    }
    TP::innerproduct_domain_sep(tr, n);
    InnerProductProof pf;
    bool first = true;
    std::vector<Aff> bases; std::vector<Fe> scal;
    while (n != 1) {
        n /= 2;
        Fe* aL = a.data(); Fe* aR = a.data() + n; Fe* bL = b.data(); Fe* bR = b.data() + n;
        Aff* GL = G.data(); Aff* GR = G.data() + n; Aff* HL = H.data(); Aff* HR = H.data() + n;
        Fe cL = inner_product(F, aL, bR, n), cR = inner_product(F, aR, bL, n);
        bases.clear(); scal.clear();
        for (size_t i = 0; i < n; i++) bases.push_back(GR[i]);
        for (size_t i = 0; i < n; i++) bases.push_back(HL[i]);
        bases.push_back(Q);
        for (size_t i = 0; i < n; i++) { Fe t = aL[i]; if (first) F.mul(t, aL[i], G_factors[n + i]); scal.push_back(t); }
        for (size_t i = 0; i < n; i++) { Fe t = bR[i]; if (first) F.mul(t, bR[i], H_factors[i]); scal.push_back(t); }
        scal.push_back(cL);
        Aff L = C.to_affine(C.msm(bases.data(), scal.data(), bases.size()));
        bases.clear(); scal.clear();
        for (size_t i = 0; i < n; i++) bases.push_back(GL[i]);
        for (size_t i = 0; i < n; i++) bases.push_back(HR[i]);
        bases.push_back(Q);
        for (size_t i = 0; i < n; i++) { Fe t = aR[i]; if (first) F.mul(t, aR[i], G_factors[i]); scal.push_back(t); }
        for (size_t i = 0; i < n; i++) { Fe t = bL[i]; if (first) F.mul(t, bL[i], H_factors[n + i]); scal.push_back(t); }
        scal.push_back(cR);
        Aff R = C.to_affine(C.msm(bases.data(), scal.data(), bases.size()));
        pf.L_vec.push_back(L); pf.R_vec.push_back(R);
        TP::append_point(C, tr, "L", L);
        TP::append_point(C, tr, "R", R);
        Fe u = TP::challenge_scalar(C, tr, "u"), u_inv;
        F.inv(u_inv, u);
        for (size_t i = 0; i < n; i++) {
            Fe t1, t2;
            F.mul(t1, aL[i], u); F.mul(t2, u_inv, aR[i]); F.add(aL[i], t1, t2);
            F.mul(t1, bL[i], u_inv); F.mul(t2, u, bR[i]); F.add(bL[i], t1, t2);
            Aff pb[2]; Fe ps[2];
            pb[0] = GL[i]; pb[1] = GR[i]; ps[0] = u_inv; ps[1] = u;
            if (first) { F.mul(ps[0], u_inv, G_factors[i]); F.mul(ps[1], u, G_factors[n + i]); }
            GL[i] = C.to_affine(C.msm(pb, ps, 2));
            pb[0] = HL[i]; pb[1] = HR[i]; ps[0] = u; ps[1] = u_inv;
            if (first) { F.mul(ps[0], u, H_factors[i]); F.mul(ps[1], u_inv, H_factors[n + i]); }
            HL[i] = C.to_affine(C.msm(pb, ps, 2));
        }
        first = false;
    }
    pf.a = a[0]; pf.b = b[0];
    return pf;
}

// src/inner_product_proof.rs:244-314
static inline Err ipa_verification_scalars(const Curve& C, const InnerProductProof& pf, size_t n, Transcript& tr,
                                           std::vector<Fe>& u_sq, std::vector<Fe>& u_inv_sq, std::vector<Fe>& s) {
    const Field& F = C.fr;
    size_t lg_n = pf.L_vec.size();
    if (lg_n >= 32) return E_VERIFICATION;
    if (n != ((size_t)1 << lg_n)) return E_VERIFICATION;
    TP::innerproduct_domain_sep(tr, n);
    std::vector<Fe> ch;
    for (size_t i = 0; i < lg_n; i++) {
        if (!TP::validate_and_append_point(C, tr, "L", pf.L_vec[i])) return E_VERIFICATION;
        if (!TP::validate_and_append_point(C, tr, "R", pf.R_vec[i])) return E_VERIFICATION;
        ch.push_back(TP::challenge_scalar(C, tr, "u"));
    }
    std::vector<Fe> chi = ch;
    F.batch_inv(chi.data(), chi.size());
    Fe allinv = F.R1;
    for (auto& f : chi) if (!f.is_zero()) F.mul(allinv, allinv, f);
    for (size_t i = 0; i < lg_n; i++) { F.sqr(ch[i], ch[i]); F.sqr(chi[i], chi[i]); }
    u_sq = ch; u_inv_sq = chi;
    s.clear(); s.reserve(n);
    s.push_back(allinv);
    for (size_t i = 1; i < n; i++) {
        int lg_i = 31 - __builtin_clz((u32)i);
        size_t k = (size_t)1 << lg_i;
        Fe t; F.mul(t, s[i - k], u_sq[(lg_n - 1) - lg_i]);
        s.push_back(t);
    }
    return OK;
}

// src/inner_product_proof.rs:321-382
static inline Err ipa_verify(const Curve& C, const InnerProductProof& pf, size_t n, Transcript& tr, const std::vector<Fe>& G_factors,
                             const std::vector<Fe>& H_factors, const Aff& P, const Aff& Q, const std::vector<Aff>& G, const std::vector<Aff>& H) {
    const Field& F = C.fr;
    std::vector<Fe> u_sq, u_inv_sq, s;
    Err e = ipa_verification_scalars(C, pf, n, tr, u_sq, u_inv_sq, s);
    if (e) return e;
    std::vector<Aff> bases; std::vector<Fe> scal;
    bases.push_back(Q);
    Fe ab; F.mul(ab, pf.a, pf.b); scal.push_back(ab);
    for (size_t i = 0; i < G.size(); i++) { Fe t; F.mul(t, pf.a, s[i]); F.mul(t, t, G_factors[i]); scal.push_back(t); bases.push_back(G[i]); }
    for (size_t i = 0; i < H.size(); i++) { Fe t; F.mul(t, pf.b, s[n - 1 - i]); F.mul(t, t, H_factors[i]); scal.push_back(t); bases.push_back(H[i]); }
    for (size_t i = 0; i < u_sq.size(); i++) { Fe t; F.neg(t, u_sq[i]); scal.push_back(t); bases.push_back(pf.L_vec[i]); }
    for (size_t i = 0; i < u_inv_sq.size(); i++) { Fe t; F.neg(t, u_inv_sq[i]); scal.push_back(t); bases.push_back(pf.R_vec[i]); }
    Aff expect = C.to_affine(C.msm(bases.data(), scal.data(), bases.size()));
    return expect == P ? OK : E_VERIFICATION;
}

// ---- R1CS: variables, linear combinations (src/r1cs/linear_combination.rs) -------------------------
enum VarKind : u8 { V_COMMITTED = 0, V_MUL_LEFT = 1, V_MUL_RIGHT = 2, V_MUL_OUT = 3, V_ONE = 4 };
struct Variable { VarKind k; size_t i; };
static inline Variable var_one() { Variable v; v.k = V_ONE; v.i = 0; return v; }

struct LC {
    std::vector<std::pair<Variable, Fe>> terms;
    LC() {}
    static LC from_var(const Field& F, Variable v) { LC l; l.terms.push_back({v, F.R1}); return l; }
    static LC from_scalar(const Fe& s) { LC l; l.terms.push_back({var_one(), s}); return l; }
    static LC term(Variable v, const Fe& s) { LC l; l.terms.push_back({v, s}); return l; }
    LC& add(const LC& o) { terms.insert(terms.end(), o.terms.begin(), o.terms.end()); return *this; }
    LC& sub(const Field& F, const LC& o) {
        for (auto& t : o.terms) { Fe n; F.neg(n, t.second); terms.push_back({t.first, n}); }
        return *this;
    }
    LC& neg(const Field& F) { for (auto& t : terms) F.neg(t.second, t.second); return *this; }
    LC& scale(const Field& F, const Fe& s) { for (auto& t : terms) F.mul(t.second, t.second, s); return *this; }
};

// ConstraintSystem + RandomizableConstraintSystem + RandomizedConstraintSystem in one interface
// (src/r1cs/constraint_system.rs:19-135).  `challenge_scalar` is only legal inside a randomized callback.
struct CS {
    const Curve& C;
    explicit CS(const Curve& c) : C(c) {}
    virtual ~CS() {}
    virtual Transcript& transcript() = 0;
    virtual void multiply(LC left, LC right, Variable out[3]) = 0;
    virtual Err allocate(const Fe* assignment, Variable& out) = 0;
    virtual Err allocate_multiplier(const Fe* l, const Fe* r, Variable out[3]) = 0;
    virtual size_t multipliers_len() const = 0;
    virtual void constrain(LC lc) = 0;
    virtual Err specify_randomized_constraints(std::function<Err(CS&)> cb) = 0;
    virtual Fe challenge_scalar(const char* label) = 0;
};

struct R1CSProof {
    Aff A_I1, A_O1, S1, A_I2, A_O2, S2, T_1, T_3, T_4, T_5, T_6;
    Fe t_x, t_x_blinding, e_blinding;
    InnerProductProof ipp;

    // src/r1cs/proof.rs:74-81 (serialize_compressed of the derived struct, field order as declared)
    std::vector<u8> to_bytes(const Curve& C) const {
        std::vector<u8> out;
        auto pt = [&](const Aff& p) { u8 b[33]; C.ser_compressed(b, p); out.insert(out.end(), b, b + 33); };
        auto sc = [&](const Fe& s) { u8 b[32]; C.fr.to_bytes(b, s); out.insert(out.end(), b, b + 32); };
        auto vec = [&](const std::vector<Aff>& v) { u64 n = v.size(); out.insert(out.end(), (u8*)&n, (u8*)&n + 8); for (auto& p : v) pt(p); };
        pt(A_I1); pt(A_O1); pt(S1); pt(A_I2); pt(A_O2); pt(S2); pt(T_1); pt(T_3); pt(T_4); pt(T_5); pt(T_6);
        sc(t_x); sc(t_x_blinding); sc(e_blinding);
        vec(ipp.L_vec); vec(ipp.R_vec); sc(ipp.a); sc(ipp.b);
        return out;
    }
    // src/r1cs/proof.rs:83-91
    static Err from_bytes(const Curve& C, const u8* d, size_t n, R1CSProof& pf) {
        size_t pos = 0;
        auto pt = [&](Aff& p) { if (pos + 33 > n) return false; bool ok = C.deser_compressed(p, d + pos); pos += 33; return ok; };
        auto sc = [&](Fe& s) { if (pos + 32 > n) return false; bool ok = C.fr.from_bytes(s, d + pos); pos += 32; return ok; };
        auto vec = [&](std::vector<Aff>& v) {
            if (pos + 8 > n) return false;
            u64 len; memcpy(&len, d + pos, 8); pos += 8;
            if (len > (n - pos) / 33) return false;
            v.resize(len);
            for (auto& p : v) if (!pt(p)) return false;
            return true;
        };
        bool ok = pt(pf.A_I1) && pt(pf.A_O1) && pt(pf.S1) && pt(pf.A_I2) && pt(pf.A_O2) && pt(pf.S2) && pt(pf.T_1) && pt(pf.T_3) &&
                  pt(pf.T_4) && pt(pf.T_5) && pt(pf.T_6) && sc(pf.t_x) && sc(pf.t_x_blinding) && sc(pf.e_blinding) && vec(pf.ipp.L_vec) &&
                  vec(pf.ipp.R_vec) && sc(pf.ipp.a) && sc(pf.ipp.b);
        return ok ? OK : E_FORMAT;
    }
};

static inline size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }  // 0 -> 1 like Rust

// ---- Prover (src/r1cs/prover.rs) ------------------------------------------------------------
struct Prover : CS {
    Transcript& tr;
    const PedersenGens& pc;
    std::vector<LC> constraints;
    std::vector<Fe> v, v_blinding, a_L, a_R, a_O;
    std::vector<std::function<Err(CS&)>> deferred;
    bool has_pending = false; size_t pending = 0;
    bool randomizing = false;

    // :291-308
    Prover(const Curve& c, const PedersenGens& pcg, Transcript& t) : CS(c), tr(t), pc(pcg) { TP::r1cs_domain_sep(tr); }
    Transcript& transcript() override { return tr; }
    // :327-341
    Aff commit(const Fe& val, const Fe& blind, Variable& var) {
        size_t i = v.size();
        v.push_back(val); v_blinding.push_back(blind);
        Aff V = pc.commit(C, val, blind);
        TP::append_point(C, tr, "V", V);
        var.k = V_COMMITTED; var.i = i;
        return V;
    }
    // :399-414
    Fe eval(const LC& lc) const {
        const Field& F = C.fr;
        Fe sum = F.Z;
        for (auto& t : lc.terms) {
            Fe val;
            switch (t.first.k) {
                case V_MUL_LEFT: val = a_L[t.first.i]; break;
                case V_MUL_RIGHT: val = a_R[t.first.i]; break;
                case V_MUL_OUT: val = a_O[t.first.i]; break;
                case V_COMMITTED: val = v[t.first.i]; break;
                default: val = F.R1; break;
            }
            Fe p; F.mul(p, t.second, val); F.add(sum, sum, p);
        }
        return sum;
    }
    // :103-133
    void multiply(LC left, LC right, Variable out[3]) override {
        const Field& F = C.fr;
        Fe l = eval(left), r = eval(right), o;
        F.mul(o, l, r);
        size_t i = a_L.size();
        out[0] = {V_MUL_LEFT, i}; out[1] = {V_MUL_RIGHT, i}; out[2] = {V_MUL_OUT, i};
        a_L.push_back(l); a_R.push_back(r); a_O.push_back(o);
        Fe m1; F.neg(m1, F.R1);
        left.terms.push_back({out[0], m1});
        right.terms.push_back({out[1], m1});
        constrain(std::move(left)); constrain(std::move(right));
    }
    // :135-157
    Err allocate(const Fe* assignment, Variable& out) override {
        if (!assignment) return E_MISSING_ASSIGNMENT;
        if (!has_pending) {
            size_t i = a_L.size();
            has_pending = true; pending = i;
            a_L.push_back(*assignment); a_R.push_back(C.fr.Z); a_O.push_back(C.fr.Z);
            out = {V_MUL_LEFT, i};
        } else {
            size_t i = pending;
            has_pending = false;
            a_R[i] = *assignment;
            C.fr.mul(a_O[i], a_L[i], a_R[i]);
            out = {V_MUL_RIGHT, i};
        }
        return OK;
    }
    // :159-183
    Err allocate_multiplier(const Fe* l, const Fe* r, Variable out[3]) override {
        if (!l || !r) return E_MISSING_ASSIGNMENT;
        Fe o; C.fr.mul(o, *l, *r);
        size_t i = a_L.size();
        out[0] = {V_MUL_LEFT, i}; out[1] = {V_MUL_RIGHT, i}; out[2] = {V_MUL_OUT, i};
        a_L.push_back(*l); a_R.push_back(*r); a_O.push_back(o);
        return OK;
    }
    size_t multipliers_len() const override { return a_L.size(); }
    void constrain(LC lc) override { constraints.push_back(std::move(lc)); }
    Err specify_randomized_constraints(std::function<Err(CS&)> cb) override { deferred.push_back(std::move(cb)); return OK; }
    Fe challenge_scalar(const char* label) override { return TP::challenge_scalar(C, tr, label); }

    // :354-397
    void flattened_constraints(const Fe& z, std::vector<Fe>& wL, std::vector<Fe>& wR, std::vector<Fe>& wO, std::vector<Fe>& wV) const {
        const Field& F = C.fr;
        size_t n = a_L.size(), m = v.size();
        wL.assign(n, F.Z); wR.assign(n, F.Z); wO.assign(n, F.Z); wV.assign(m, F.Z);
        Fe exp_z = z;
        for (auto& lc : constraints) {
            for (auto& t : lc.terms) {
                Fe p; F.mul(p, exp_z, t.second);
                switch (t.first.k) {
                    case V_MUL_LEFT: F.add(wL[t.first.i], wL[t.first.i], p); break;
                    case V_MUL_RIGHT: F.add(wR[t.first.i], wR[t.first.i], p); break;
                    case V_MUL_OUT: F.add(wO[t.first.i], wO[t.first.i], p); break;
                    case V_COMMITTED: F.sub(wV[t.first.i], wV[t.first.i], p); break;
                    default: break;
                }
            }
            F.mul(exp_z, exp_z, z);
        }
    }
    // :418-441
    Err create_randomized_constraints() {
        has_pending = false;
        if (deferred.empty()) { TP::r1cs_1phase_domain_sep(tr); return OK; }
        TP::r1cs_2phase_domain_sep(tr);
        std::vector<std::function<Err(CS&)>> cbs;
        cbs.swap(deferred);
        for (auto& cb : cbs) { Err e = cb(*this); if (e) return e; }
        return OK;
    }

    // :454-831
    template <class Rng> Err prove(Rng& prng, const BulletproofGens& bp, R1CSProof& proof) {
        const Field& F = C.fr;
        tr.append_u64("m", v.size());
        TranscriptRng rng(tr);
        for (auto& vb : v_blinding) { u8 b[32]; F.to_bytes(b, vb); rng.rekey_with_witness_bytes("v_blinding", b, 32); }
        rng.finalize(prng);

        size_t n1 = a_L.size();
        if (bp.gens_capacity < n1) return E_GENS_LENGTH;
        const std::vector<Aff>& Gg = bp.G_vec[0];
        const std::vector<Aff>& Hg = bp.H_vec[0];

        Fe i_b1 = fe_rand(F, rng), o_b1 = fe_rand(F, rng), s_b1 = fe_rand(F, rng);
        std::vector<Fe> s_L1(n1), s_R1(n1);
        for (auto& x : s_L1) x = fe_rand(F, rng);
        for (auto& x : s_R1) x = fe_rand(F, rng);

        std::vector<Aff> bases; std::vector<Fe> scal;
        auto commit3 = [&](const Fe& blind, const Fe* xs, const Fe* ys, size_t off, size_t cnt) {
            bases.clear(); scal.clear();
            bases.push_back(pc.B_blinding); scal.push_back(blind);
            for (size_t i = 0; i < cnt; i++) { bases.push_back(Gg[off + i]); scal.push_back(xs[i]); }
            if (ys) for (size_t i = 0; i < cnt; i++) { bases.push_back(Hg[off + i]); scal.push_back(ys[i]); }
            return C.to_affine(C.msm(bases.data(), scal.data(), bases.size()));
        };
        Aff A_I1 = commit3(i_b1, a_L.data(), a_R.data(), 0, n1);
        Aff A_O1 = commit3(o_b1, a_O.data(), nullptr, 0, n1);
        Aff S1 = commit3(s_b1, s_L1.data(), s_R1.data(), 0, n1);
        TP::append_point(C, tr, "A_I1", A_I1);
        TP::append_point(C, tr, "A_O1", A_O1);
        TP::append_point(C, tr, "S1", S1);

        Err e = create_randomized_constraints();
        if (e) return e;

        size_t n = a_L.size(), n2 = n - n1, padded_n = next_pow2(n);
        if (bp.gens_capacity < padded_n) return E_GENS_LENGTH;
        bool has2 = n2 > 0;
        Fe i_b2 = F.Z, o_b2 = F.Z, s_b2 = F.Z;
        if (has2) { i_b2 = fe_rand(F, rng); o_b2 = fe_rand(F, rng); s_b2 = fe_rand(F, rng); }
        std::vector<Fe> s_L2(n2), s_R2(n2);
        for (auto& x : s_L2) x = fe_rand(F, rng);
        for (auto& x : s_R2) x = fe_rand(F, rng);
        Aff A_I2 = C.aff_zero(), A_O2 = C.aff_zero(), S2 = C.aff_zero();
        if (has2) {
            A_I2 = commit3(i_b2, a_L.data() + n1, a_R.data() + n1, n1, n2);
            A_O2 = commit3(o_b2, a_O.data() + n1, nullptr, n1, n2);
            S2 = commit3(s_b2, s_L2.data(), s_R2.data(), n1, n2);
        }
        TP::append_point(C, tr, "A_I2", A_I2);
        TP::append_point(C, tr, "A_O2", A_O2);
        TP::append_point(C, tr, "S2", S2);

        Fe y = TP::challenge_scalar(C, tr, "y"), z = TP::challenge_scalar(C, tr, "z");
        std::vector<Fe> wL, wR, wO, wV;
        flattened_constraints(z, wL, wR, wO, wV);

        std::vector<Fe> l1(n), l2(n), l3(n), r0(n), r1(n), r3(n);
        Fe exp_y = F.R1, y_inv;
        F.inv(y_inv, y);
        std::vector<Fe> exp_y_inv(padded_n);
        { Fe cur = F.R1; for (size_t i = 0; i < padded_n; i++) { exp_y_inv[i] = cur; F.mul(cur, cur, y_inv); } }
        for (size_t i = 0; i < n; i++) {
            const Fe& sl = i < n1 ? s_L1[i] : s_L2[i - n1];
            const Fe& sr = i < n1 ? s_R1[i] : s_R2[i - n1];
            Fe t;
            F.mul(t, exp_y_inv[i], wR[i]); F.add(l1[i], a_L[i], t);
            l2[i] = a_O[i];
            l3[i] = sl;
            F.sub(r0[i], wO[i], exp_y);
            F.mul(t, exp_y, a_R[i]); F.add(r1[i], t, wL[i]);
            F.mul(r3[i], exp_y, sr);
            F.mul(exp_y, exp_y, y);
        }
        // util.rs:75-93 (l0 = 0, r2 = 0)
        auto ip = [&](const std::vector<Fe>& x, const std::vector<Fe>& yv) { return inner_product(F, x.data(), yv.data(), n); };
        Fe t1 = ip(l1, r0), t2, t3, t4, t5 = ip(l2, r3), t6 = ip(l3, r3), tmp;
        t2 = ip(l1, r1); tmp = ip(l2, r0); F.add(t2, t2, tmp);
        t3 = ip(l2, r1); tmp = ip(l3, r0); F.add(t3, t3, tmp);
        t4 = ip(l1, r3); tmp = ip(l3, r1); F.add(t4, t4, tmp);

        Fe t1b = fe_rand(F, rng), t3b = fe_rand(F, rng), t4b = fe_rand(F, rng), t5b = fe_rand(F, rng), t6b = fe_rand(F, rng);
        Aff T_1 = pc.commit(C, t1, t1b), T_3 = pc.commit(C, t3, t3b), T_4 = pc.commit(C, t4, t4b), T_5 = pc.commit(C, t5, t5b),
            T_6 = pc.commit(C, t6, t6b);
        TP::append_point(C, tr, "T_1", T_1);
        TP::append_point(C, tr, "T_3", T_3);
        TP::append_point(C, tr, "T_4", T_4);
        TP::append_point(C, tr, "T_5", T_5);
        TP::append_point(C, tr, "T_6", T_6);

        Fe u = TP::challenge_scalar(C, tr, "u"), x = TP::challenge_scalar(C, tr, "x");
        Fe t2b = F.Z;
        for (size_t j = 0; j < wV.size(); j++) { F.mul(tmp, v_blinding[j], wV[j]); F.add(t2b, t2b, tmp); }
        // Poly6::eval (util.rs:107-109)
        auto poly6 = [&](const Fe& c1, const Fe& c2, const Fe& c3, const Fe& c4, const Fe& c5, const Fe& c6) {
            Fe acc;
            F.mul(acc, x, c6); F.add(acc, acc, c5);
            F.mul(acc, acc, x); F.add(acc, acc, c4);
            F.mul(acc, acc, x); F.add(acc, acc, c3);
            F.mul(acc, acc, x); F.add(acc, acc, c2);
            F.mul(acc, acc, x); F.add(acc, acc, c1);
            F.mul(acc, acc, x);
            return acc;
        };
        Fe t_x = poly6(t1, t2, t3, t4, t5, t6);
        Fe t_x_blinding = poly6(t1b, t2b, t3b, t4b, t5b, t6b);
        std::vector<Fe> l_vec(padded_n, F.Z), r_vec(padded_n, F.Z);
        for (size_t i = 0; i < n; i++) {  // VecPoly3::eval (util.rs:95-102)
            Fe acc;
            F.mul(acc, x, l3[i]); F.add(acc, acc, l2[i]); F.mul(acc, acc, x); F.add(acc, acc, l1[i]); F.mul(l_vec[i], acc, x);
            F.mul(acc, x, r3[i]); /* r2 = 0 */ F.mul(acc, acc, x); F.add(acc, acc, r1[i]); F.mul(acc, acc, x); F.add(r_vec[i], acc, r0[i]);
        }
        for (size_t i = n; i < padded_n; i++) { F.neg(r_vec[i], exp_y); F.mul(exp_y, exp_y, y); }

        Fe i_b, o_b, s_b;
        F.mul(tmp, u, i_b2); F.add(i_b, i_b1, tmp);
        F.mul(tmp, u, o_b2); F.add(o_b, o_b1, tmp);
        F.mul(tmp, u, s_b2); F.add(s_b, s_b1, tmp);
        Fe e_blinding;
        F.mul(tmp, x, s_b); F.add(tmp, tmp, o_b); F.mul(tmp, tmp, x); F.add(tmp, tmp, i_b); F.mul(e_blinding, tmp, x);

        TP::append_scalar(C, tr, "t_x", t_x);
        TP::append_scalar(C, tr, "t_x_blinding", t_x_blinding);
        TP::append_scalar(C, tr, "e_blinding", e_blinding);
        Fe w = TP::challenge_scalar(C, tr, "w");
        Aff Q = C.to_affine(C.mul(pc.B, w));

        std::vector<Fe> G_factors(padded_n), H_factors(padded_n);
        for (size_t i = 0; i < padded_n; i++) {
            G_factors[i] = i < n1 ? F.R1 : u;
            F.mul(H_factors[i], exp_y_inv[i], G_factors[i]);
        }
        std::vector<Aff> Gv(Gg.begin(), Gg.begin() + padded_n), Hv(Hg.begin(), Hg.begin() + padded_n);
        proof.ipp = ipa_create(C, tr, Q, G_factors, H_factors, std::move(Gv), std::move(Hv), std::move(l_vec), std::move(r_vec));
        proof.A_I1 = A_I1; proof.A_O1 = A_O1; proof.S1 = S1; proof.A_I2 = A_I2; proof.A_O2 = A_O2; proof.S2 = S2;
        proof.T_1 = T_1; proof.T_3 = T_3; proof.T_4 = T_4; proof.T_5 = T_5; proof.T_6 = T_6;
        proof.t_x = t_x; proof.t_x_blinding = t_x_blinding; proof.e_blinding = e_blinding;
        return OK;
    }
};

// ---- Verifier (src/r1cs/verifier.rs) ----------------------------------------------------------
struct Verifier : CS {
    Transcript& tr;
    std::vector<LC> constraints;
    size_t num_vars = 0;
    std::vector<Aff> V;
    std::vector<std::function<Err(CS&)>> deferred;
    bool has_pending = false; size_t pending = 0;

    // :252-263
    Verifier(const Curve& c, Transcript& t) : CS(c), tr(t) { TP::r1cs_domain_sep(tr); }
    Transcript& transcript() override { return tr; }
    // :279-287
    Variable commit(const Aff& commitment) {
        size_t i = V.size();
        V.push_back(commitment);
        TP::append_point(C, tr, "V", commitment);
        return {V_COMMITTED, i};
    }
    // :74-98
    void multiply(LC left, LC right, Variable out[3]) override {
        size_t var = num_vars++;
        out[0] = {V_MUL_LEFT, var}; out[1] = {V_MUL_RIGHT, var}; out[2] = {V_MUL_OUT, var};
        Fe m1; C.fr.neg(m1, C.fr.R1);
        left.terms.push_back({out[0], m1});
        right.terms.push_back({out[1], m1});
        constrain(std::move(left)); constrain(std::move(right));
    }
    // :100-116
    Err allocate(const Fe*, Variable& out) override {
        if (!has_pending) { size_t i = num_vars++; has_pending = true; pending = i; out = {V_MUL_LEFT, i}; }
        else { has_pending = false; out = {V_MUL_RIGHT, pending}; }
        return OK;
    }
    // :118-138
    Err allocate_multiplier(const Fe*, const Fe*, Variable out[3]) override {
        size_t var = num_vars++;
        out[0] = {V_MUL_LEFT, var}; out[1] = {V_MUL_RIGHT, var}; out[2] = {V_MUL_OUT, var};
        return OK;
    }
    size_t multipliers_len() const override { return num_vars; }
    void constrain(LC lc) override { constraints.push_back(std::move(lc)); }
    Err specify_randomized_constraints(std::function<Err(CS&)> cb) override { deferred.push_back(std::move(cb)); return OK; }
    Fe challenge_scalar(const char* label) override { return TP::challenge_scalar(C, tr, label); }

    // :304-349
    void flattened_constraints(const Fe& z, std::vector<Fe>& wL, std::vector<Fe>& wR, std::vector<Fe>& wO, std::vector<Fe>& wV, Fe& wc) const {
        const Field& F = C.fr;
        size_t n = num_vars, m = V.size();
        wL.assign(n, F.Z); wR.assign(n, F.Z); wO.assign(n, F.Z); wV.assign(m, F.Z); wc = F.Z;
        Fe exp_z = z;
        for (auto& lc : constraints) {
            for (auto& t : lc.terms) {
                Fe p; F.mul(p, exp_z, t.second);
                switch (t.first.k) {
                    case V_MUL_LEFT: F.add(wL[t.first.i], wL[t.first.i], p); break;
                    case V_MUL_RIGHT: F.add(wR[t.first.i], wR[t.first.i], p); break;
                    case V_MUL_OUT: F.add(wO[t.first.i], wO[t.first.i], p); break;
                    case V_COMMITTED: F.sub(wV[t.first.i], wV[t.first.i], p); break;
                    case V_ONE: F.sub(wc, wc, p); break;
                }
            }
            F.mul(exp_z, exp_z, z);
        }
    }
    // :353-376
    Err create_randomized_constraints() {
        has_pending = false;
        if (deferred.empty()) { TP::r1cs_1phase_domain_sep(tr); return OK; }
        TP::r1cs_2phase_domain_sep(tr);
        std::vector<std::function<Err(CS&)>> cbs;
        cbs.swap(deferred);
        for (auto& cb : cbs) { Err e = cb(*this); if (e) return e; }
        return OK;
    }
    // :394-541.  Scalar order: B, B_blinding, g[N], h[N], A_I1,A_O1,S1,A_I2,A_O2,S2, V[m], T_1,3,4,5,6, L[k], R[k]
    Err verification_scalars(const R1CSProof& proof, const BulletproofGens& bp, std::vector<Fe>& scalars) {
        const Field& F = C.fr;
        tr.append_u64("m", V.size());
        size_t n1 = num_vars;
        if (!TP::validate_and_append_point(C, tr, "A_I1", proof.A_I1)) return E_VERIFICATION;
        if (!TP::validate_and_append_point(C, tr, "A_O1", proof.A_O1)) return E_VERIFICATION;
        if (!TP::validate_and_append_point(C, tr, "S1", proof.S1)) return E_VERIFICATION;
        Err e = create_randomized_constraints();
        if (e) return e;
        size_t n = num_vars, n2 = n - n1, padded_n = next_pow2(n), pad = padded_n - n;
        (void)n2; (void)pad;
        if (bp.gens_capacity < padded_n) return E_GENS_LENGTH;
        TP::append_point(C, tr, "A_I2", proof.A_I2);
        TP::append_point(C, tr, "A_O2", proof.A_O2);
        TP::append_point(C, tr, "S2", proof.S2);
        Fe y = TP::challenge_scalar(C, tr, "y"), z = TP::challenge_scalar(C, tr, "z");
        if (!TP::validate_and_append_point(C, tr, "T_1", proof.T_1)) return E_VERIFICATION;
        if (!TP::validate_and_append_point(C, tr, "T_3", proof.T_3)) return E_VERIFICATION;
        if (!TP::validate_and_append_point(C, tr, "T_4", proof.T_4)) return E_VERIFICATION;
        if (!TP::validate_and_append_point(C, tr, "T_5", proof.T_5)) return E_VERIFICATION;
        if (!TP::validate_and_append_point(C, tr, "T_6", proof.T_6)) return E_VERIFICATION;
        Fe u = TP::challenge_scalar(C, tr, "u"), x = TP::challenge_scalar(C, tr, "x");
        TP::append_scalar(C, tr, "t_x", proof.t_x);
        TP::append_scalar(C, tr, "t_x_blinding", proof.t_x_blinding);
        TP::append_scalar(C, tr, "e_blinding", proof.e_blinding);
        Fe w = TP::challenge_scalar(C, tr, "w");
        std::vector<Fe> wL, wR, wO, wV; Fe wc;
        flattened_constraints(z, wL, wR, wO, wV, wc);
        std::vector<Fe> u_sq, u_inv_sq, s;
        if (ipa_verification_scalars(C, proof.ipp, padded_n, tr, u_sq, u_inv_sq, s)) return E_VERIFICATION;
        Fe a = proof.ipp.a, b = proof.ipp.b;
        Fe y_inv; F.inv(y_inv, y);
        std::vector<Fe> y_inv_vec(padded_n), yneg_wR(padded_n, F.Z);
        { Fe cur = F.R1; for (size_t i = 0; i < padded_n; i++) { y_inv_vec[i] = cur; F.mul(cur, cur, y_inv); } }
        for (size_t i = 0; i < n; i++) F.mul(yneg_wR[i], wR[i], y_inv_vec[i]);
        Fe delta = inner_product(F, yneg_wR.data(), wL.data(), n);
        std::vector<Fe> g_scalars(padded_n), h_scalars(padded_n);
        for (size_t i = 0; i < padded_n; i++) {
            Fe u_or_1 = i < n1 ? F.R1 : u, t, t2;
            F.mul(t, x, yneg_wR[i]); F.mul(t2, a, s[i]); F.sub(t, t, t2); F.mul(g_scalars[i], u_or_1, t);
            Fe wLi = i < n ? wL[i] : F.Z, wOi = i < n ? wO[i] : F.Z;
            F.mul(t, x, wLi); F.add(t, t, wOi); F.mul(t2, b, s[padded_n - 1 - i]); F.sub(t, t, t2);
            F.mul(t, y_inv_vec[i], t); F.sub(t, t, F.R1); F.mul(h_scalars[i], u_or_1, t);
        }
        Transcript clone = tr;
        Fe r = TP::challenge_scalar(C, clone, "r");
        Fe xx, rxx, xxx, t, t2;
        F.mul(xx, x, x); F.mul(rxx, r, xx); F.mul(xxx, x, xx);
        Fe T_scalars[5];
        F.mul(T_scalars[0], r, x); F.mul(T_scalars[1], rxx, x); F.mul(T_scalars[2], rxx, xx); F.mul(T_scalars[3], rxx, xxx);
        F.mul(t, rxx, xx); F.mul(T_scalars[4], t, xx);
        scalars.clear();
        // w*(t_x - a*b) + r*(xx*(wc + delta) - t_x)
        F.mul(t, a, b); F.sub(t, proof.t_x, t); F.mul(t, w, t);
        F.add(t2, wc, delta); F.mul(t2, xx, t2); F.sub(t2, t2, proof.t_x); F.mul(t2, r, t2);
        F.add(t, t, t2); scalars.push_back(t);
        // -e_blinding - r*t_x_blinding
        F.neg(t, proof.e_blinding); F.mul(t2, r, proof.t_x_blinding); F.sub(t, t, t2); scalars.push_back(t);
        scalars.insert(scalars.end(), g_scalars.begin(), g_scalars.end());
        scalars.insert(scalars.end(), h_scalars.begin(), h_scalars.end());
        scalars.push_back(x); scalars.push_back(xx); scalars.push_back(xxx);
        F.mul(t, u, x); scalars.push_back(t); F.mul(t, u, xx); scalars.push_back(t); F.mul(t, u, xxx); scalars.push_back(t);
        for (auto& wVi : wV) { F.mul(t, wVi, rxx); scalars.push_back(t); }
        for (int i = 0; i < 5; i++) scalars.push_back(T_scalars[i]);
        scalars.insert(scalars.end(), u_sq.begin(), u_sq.end());
        scalars.insert(scalars.end(), u_inv_sq.begin(), u_inv_sq.end());
        return OK;
    }
    void tail_points(const R1CSProof& proof, std::vector<Aff>& out) const {
        out.push_back(proof.A_I1); out.push_back(proof.A_O1); out.push_back(proof.S1);
        out.push_back(proof.A_I2); out.push_back(proof.A_O2); out.push_back(proof.S2);
        out.insert(out.end(), V.begin(), V.end());
        out.push_back(proof.T_1); out.push_back(proof.T_3); out.push_back(proof.T_4); out.push_back(proof.T_5); out.push_back(proof.T_6);
        out.insert(out.end(), proof.ipp.L_vec.begin(), proof.ipp.L_vec.end());
        out.insert(out.end(), proof.ipp.R_vec.begin(), proof.ipp.R_vec.end());
    }
    // :559-600
    Err verify(const R1CSProof& proof, const PedersenGens& pc, const BulletproofGens& bp) {
        std::vector<Fe> scalars;
        Err e = verification_scalars(proof, bp, scalars);
        if (e) return e;
        size_t padded_n = next_pow2(num_vars);
        std::vector<Aff> bases;
        bases.push_back(pc.B); bases.push_back(pc.B_blinding);
        bases.insert(bases.end(), bp.G_vec[0].begin(), bp.G_vec[0].begin() + padded_n);
        bases.insert(bases.end(), bp.H_vec[0].begin(), bp.H_vec[0].begin() + padded_n);
        tail_points(proof, bases);
        if (bases.size() != scalars.size()) return E_VERIFICATION;  // ark msm would Err(min_len) -> unwrap panic
        Jac mega = C.msm(bases.data(), scalars.data(), bases.size());
        return C.is_inf(mega) ? OK : E_VERIFICATION;
    }
};

// src/r1cs/verifier.rs:604-691
// alphas_in: the per-instance weights the caller drew (instances.size() of them) instead of `Fr::rand(prng)`;
// mega_out: the value of the final MSM (the check accepts iff it is the identity) — both optional, for parity tests.
template <class Rng>
static inline Err batch_verify(const Curve& C, Rng& prng, std::vector<std::pair<Verifier*, const R1CSProof*>>& instances, const PedersenGens& pc,
                               const BulletproofGens& bp, const Fe* alphas_in = nullptr, Aff* mega_out = nullptr) {
    const Field& F = C.fr;
    size_t max_n = 0;
    std::vector<std::vector<Fe>> vs(instances.size());
    for (size_t k = 0; k < instances.size(); k++) {
        Err e = instances[k].first->verification_scalars(*instances[k].second, bp, vs[k]);
        if (e) return e;
        size_t n = next_pow2(instances[k].first->num_vars);
        if (n > max_n) max_n = n;
    }
    std::vector<Fe> all_scalars(2 * max_n + 2, F.Z);
    std::vector<Aff> all_elems;
    all_elems.push_back(pc.B); all_elems.push_back(pc.B_blinding);
    all_elems.insert(all_elems.end(), bp.G_vec[0].begin(), bp.G_vec[0].begin() + max_n);
    all_elems.insert(all_elems.end(), bp.H_vec[0].begin(), bp.H_vec[0].begin() + max_n);
    for (size_t k = 0; k < instances.size(); k++) {
        Fe alpha = alphas_in ? alphas_in[k] : fe_rand(F, prng);
        std::vector<Fe>& sc = vs[k];
        for (auto& s : sc) F.mul(s, alpha, s);
        size_t padded_n = next_pow2(instances[k].first->num_vars);
        F.add(all_scalars[0], all_scalars[0], sc[0]);
        F.add(all_scalars[1], all_scalars[1], sc[1]);
        for (size_t i = 0; i < padded_n; i++) F.add(all_scalars[2 + i], all_scalars[2 + i], sc[2 + i]);
        for (size_t i = 0; i < padded_n; i++) F.add(all_scalars[2 + max_n + i], all_scalars[2 + max_n + i], sc[2 + padded_n + i]);
        for (size_t i = 2 + 2 * padded_n; i < sc.size(); i++) all_scalars.push_back(sc[i]);
        instances[k].first->tail_points(*instances[k].second, all_elems);
    }
    Jac m = C.msm(all_elems.data(), all_scalars.data(), all_elems.size());
    if (mega_out) *mega_out = C.to_affine(m);
    return C.is_inf(m) ? OK : E_VERIFICATION;
}

}  // namespace orc
