// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).  "parity unpinned".
//
// The reference's test/bench gadgets restated on the oracle's ConstraintSystem:
//   shuffle        benches/r1cs_secq256k1.rs:34-147 (= tests/r1cs_secq256k1.rs:20-130)
//   example        tests/r1cs_secq256k1.rs:216-228 (+ proof/verify drivers :231-303)
//   range proof    tests/r1cs_secq256k1.rs:361-393 (+ helper :413-445)
// plus one synthetic low-commitment circuit of this build's own (SURVEY.md §8d cfg3 "square chain").
// A "scenario" fixes the witness derivation from a ChaCha20 seed so that the HIP product and this
// oracle can be driven with identical inputs; the proving algorithm itself is the reference's.
#pragma once
#include "protocol.hpp"

namespace orc {

enum Scenario { SC_SHUFFLE = 0, SC_RANGE = 1, SC_EXAMPLE = 2, SC_SQUARE_CHAIN = 3, SC_MULTI_RANGE = 4 };

static inline const char* scenario_label(int sc) {
    switch (sc) {
        case SC_SHUFFLE: return "ShuffleBenchmark";        // benches/r1cs_secq256k1.rs:173
        case SC_RANGE: return "RangeProofTest";            // tests/r1cs_secq256k1.rs:422
        case SC_EXAMPLE: return "R1CSExampleGadget";       // tests/r1cs_secq256k1.rs:241
        case SC_SQUARE_CHAIN: return "SquareChainBenchmark";
        default: return "MultiRangeBenchmark";
    }
}

// benches/r1cs_secq256k1.rs:34-76
static inline Err shuffle_gadget(CS& cs, const std::vector<Variable>& x, const std::vector<Variable>& y) {
    const Field& F = cs.C.fr;
    size_t k = x.size();
    if (k != y.size()) return E_GADGET;
    if (k == 1) {
        LC lc = LC::from_var(F, y[0]);
        lc.sub(F, LC::from_var(F, x[0]));
        cs.constrain(lc);
        return OK;
    }
    return cs.specify_randomized_constraints([x, y, k](CS& cs) -> Err {
        const Field& F = cs.C.fr;
        Fe z = cs.challenge_scalar("shuffle challenge");
        auto minus_z = [&](Variable v) { LC l = LC::from_var(F, v); l.sub(F, LC::from_scalar(z)); return l; };
        Variable o[3];
        cs.multiply(minus_z(x[k - 1]), minus_z(x[k - 2]), o);
        Variable prev = o[2];
        for (size_t i = k - 2; i-- > 0;) { cs.multiply(LC::from_var(F, prev), minus_z(x[i]), o); prev = o[2]; }
        Variable first_mulx_out = prev;
        cs.multiply(minus_z(y[k - 1]), minus_z(y[k - 2]), o);
        prev = o[2];
        for (size_t i = k - 2; i-- > 0;) { cs.multiply(LC::from_var(F, prev), minus_z(y[i]), o); prev = o[2]; }
        LC lc = LC::from_var(F, first_mulx_out);
        lc.sub(F, LC::from_var(F, prev));
        cs.constrain(lc);
        return OK;
    });
}

// tests/r1cs_secq256k1.rs:216-228: (a1 + a2) * (b1 + b2) = (c1 + c2)
static inline void example_gadget(CS& cs, LC a1, LC a2, LC b1, LC b2, LC c1, LC c2) {
    const Field& F = cs.C.fr;
    Variable o[3];
    a1.add(a2); b1.add(b2);
    cs.multiply(a1, b1, o);
    c1.add(c2); c1.sub(F, LC::from_var(F, o[2]));
    cs.constrain(c1);
}

// tests/r1cs_secq256k1.rs:361-393
static inline Err range_proof_gadget(CS& cs, LC v, const u64* v_assignment, size_t n) {
    const Field& F = cs.C.fr;
    Fe exp_2 = F.R1;
    for (size_t i = 0; i < n; i++) {
        Variable abo[3];
        Err e;
        if (v_assignment) {
            u64 bit = (*v_assignment >> i) & 1;
            Fe l = F.from_u64(1 - bit), r = F.from_u64(bit);
            e = cs.allocate_multiplier(&l, &r, abo);
        } else e = cs.allocate_multiplier(nullptr, nullptr, abo);
        if (e) return e;
        cs.constrain(LC::from_var(F, abo[2]));
        LC ab = LC::from_var(F, abo[0]);
        ab.add(LC::from_var(F, abo[1]));
        ab.sub(F, LC::from_scalar(F.R1));
        cs.constrain(ab);
        v.sub(F, LC::term(abo[1], exp_2));
        F.add(exp_2, exp_2, exp_2);
    }
    cs.constrain(v);
    return OK;
}

// this build's synthetic circuit: x_{i+1} = x_i^2 for N gates, 1 commitment, q = 2N + 1
static inline void square_chain_gadget(CS& cs, Variable x0, size_t N, const Fe& expected_last) {
    const Field& F = cs.C.fr;
    Variable cur = x0, o[3];
    for (size_t i = 0; i < N; i++) { cs.multiply(LC::from_var(F, cur), LC::from_var(F, cur), o); cur = o[2]; }
    LC lc = LC::from_var(F, cur);
    lc.sub(F, LC::from_scalar(expected_last));
    cs.constrain(lc);
}

// ---- scenario drivers -------------------------------------------------------------------
struct ScenarioIO {
    std::vector<Aff> commitments;  // V points in commit order
    std::vector<Fe> publics;       // scenario-specific public scalars (square chain: expected_last)
};

// Builds the statement on the prover side (commits + gadget).  RNG consumption order is part of the
// scenario definition and is mirrored by the product's host code.
static inline Err scenario_prover_setup(Prover& p, ChaCha20Rng& prng, int sc, const u64* prm, ScenarioIO& io) {
    const Curve& C = p.C; const Field& F = C.fr;
    switch (sc) {
        case SC_SHUFFLE: {  // benches/r1cs_secq256k1.rs:81-112 (prove)
            size_t k = prm[0];
            std::vector<Fe> input(k), output(k);
            for (auto& x : input) x = F.from_u64(prng.next_u64());
            for (size_t i = 0; i < k; i++) output[i] = input[(i + 1) % k];
            p.tr.append_message("dom-sep", "ShuffleProof");
            p.tr.append_u64("k", k);
            std::vector<Variable> xv(k), yv(k);
            for (size_t i = 0; i < k; i++) { Fe b = fe_rand(F, prng); io.commitments.push_back(p.commit(input[i], b, xv[i])); }
            for (size_t i = 0; i < k; i++) { Fe b = fe_rand(F, prng); io.commitments.push_back(p.commit(output[i], b, yv[i])); }
            return shuffle_gadget(p, xv, yv);
        }
        case SC_RANGE: {  // tests/r1cs_secq256k1.rs:413-431
            size_t nbits = prm[0]; u64 val = prm[1];
            Variable var; Fe b = fe_rand(F, prng);
            io.commitments.push_back(p.commit(F.from_u64(val), b, var));
            return range_proof_gadget(p, LC::from_var(F, var), &val, nbits);
        }
        case SC_EXAMPLE: {  // tests/r1cs_secq256k1.rs:231-267
            Variable vars[5];
            for (int i = 0; i < 5; i++) { Fe b = fe_rand(F, prng); io.commitments.push_back(p.commit(F.from_u64(prm[i]), b, vars[i])); }
            example_gadget(p, LC::from_var(F, vars[0]), LC::from_var(F, vars[1]), LC::from_var(F, vars[2]), LC::from_var(F, vars[3]),
                           LC::from_var(F, vars[4]), LC::from_scalar(F.from_u64(prm[5])));
            return OK;
        }
        case SC_SQUARE_CHAIN: {
            size_t N = prm[0];
            Fe x0 = F.from_u64(prng.next_u64()), b = fe_rand(F, prng), last = x0;
            for (size_t i = 0; i < N; i++) F.sqr(last, last);
            if (prm[1]) F.add(last, last, F.R1);  // negative case: wrong public output
            Variable var;
            io.commitments.push_back(p.commit(x0, b, var));
            io.publics.push_back(last);
            square_chain_gadget(p, var, N, last);
            return OK;
        }
        case SC_MULTI_RANGE: {  // cfg4 shape: `count` values of `nbits` bits in one circuit
            size_t count = prm[0], nbits = prm[1];
            for (size_t j = 0; j < count; j++) {
                u64 val = prng.next_u64();
                if (nbits < 64) val &= (((u64)1 << nbits) - 1);
                if (prm[2] && j == count - 1) val = nbits < 64 ? ((u64)1 << nbits) : val;  // negative: out of range (nbits<64 only)
                Variable var; Fe b = fe_rand(F, prng);
                io.commitments.push_back(p.commit(F.from_u64(val), b, var));
                Err e = range_proof_gadget(p, LC::from_var(F, var), &val, nbits);
                if (e) return e;
            }
            return OK;
        }
    }
    return E_GADGET;
}

static inline Err scenario_verifier_setup(Verifier& v, int sc, const u64* prm, const ScenarioIO& io) {
    const Curve& C = v.C; const Field& F = C.fr;
    switch (sc) {
        case SC_SHUFFLE: {  // benches/r1cs_secq256k1.rs:117-146 (verify)
            size_t k = prm[0];
            if (io.commitments.size() != 2 * k) return E_GADGET;
            v.tr.append_message("dom-sep", "ShuffleProof");
            v.tr.append_u64("k", k);
            std::vector<Variable> xv(k), yv(k);
            for (size_t i = 0; i < k; i++) xv[i] = v.commit(io.commitments[i]);
            for (size_t i = 0; i < k; i++) yv[i] = v.commit(io.commitments[k + i]);
            return shuffle_gadget(v, xv, yv);
        }
        case SC_RANGE: {
            Variable var = v.commit(io.commitments[0]);
            return range_proof_gadget(v, LC::from_var(F, var), nullptr, prm[0]);
        }
        case SC_EXAMPLE: {
            Variable vars[5];
            for (int i = 0; i < 5; i++) vars[i] = v.commit(io.commitments[i]);
            example_gadget(v, LC::from_var(F, vars[0]), LC::from_var(F, vars[1]), LC::from_var(F, vars[2]), LC::from_var(F, vars[3]),
                           LC::from_var(F, vars[4]), LC::from_scalar(F.from_u64(prm[5])));
            return OK;
        }
        case SC_SQUARE_CHAIN: {
            Variable var = v.commit(io.commitments[0]);
            square_chain_gadget(v, var, prm[0], io.publics[0]);
            return OK;
        }
        case SC_MULTI_RANGE: {
            for (size_t j = 0; j < prm[0]; j++) {
                Variable var = v.commit(io.commitments[j]);
                Err e = range_proof_gadget(v, LC::from_var(F, var), nullptr, prm[1]);
                if (e) return e;
            }
            return OK;
        }
    }
    return E_GADGET;
}

}  // namespace orc
