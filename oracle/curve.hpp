// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).  "parity unpinned".
//
// Short-Weierstrass group arithmetic restated from ark-ec ^0.4 (un-vendored dependency of the
// reference, Cargo.toml:28-30): `short_weierstrass::{Affine, Projective}` (Jacobian X,Y,Z),
// `CurveGroup::into_affine`, `AffineRepr::mul_bigint`, `VariableBaseMSM::msm`
// (call sites: src/inner_product_proof.rs:104,124,143,150,187,202,219,222,375;
//  src/r1cs/prover.rs:516,532,546,607,622,635; src/r1cs/verifier.rs:574,685).
// Curves: secq256k1 (ark-secq256k1 0.4.0: y^2 = x^3 + 7 over F_{n_secp}, scalar field F_{p_secp})
// and zorro (src/curve/zorro/g1.rs:11-46: a = 6).
#pragma once
#include <vector>
#include "field.hpp"

namespace orc {

struct Aff {
    Fe x, y;
    bool inf;
    bool operator==(const Aff& o) const { return inf == o.inf && (inf || (x == o.x && y == o.y)); }
};
struct Jac {
    Fe X, Y, Z;  // Z == 0 <=> identity
};

struct Curve {
    int id;
    Field fq, fr;
    Fe a, b;
    bool a_zero;
    Aff gen;

    Aff aff_zero() const { Aff z; z.x = fq.Z; z.y = fq.Z; z.inf = true; return z; }
    Jac jac_zero() const { Jac z; z.X = fq.R1; z.Y = fq.R1; z.Z = fq.Z; return z; }
    bool is_inf(const Jac& p) const { return p.Z.is_zero(); }
    Jac to_jac(const Aff& p) const {
        if (p.inf) return jac_zero();
        Jac j; j.X = p.x; j.Y = p.y; j.Z = fq.R1; return j;
    }
    Aff neg(const Aff& p) const { Aff r = p; if (!p.inf) fq.neg(r.y, p.y); return r; }

    bool on_curve(const Aff& p) const {
        if (p.inf) return true;
        Fe l, r, t;
        fq.sqr(l, p.y);
        fq.sqr(r, p.x); fq.mul(r, r, p.x);
        fq.mul(t, a, p.x); fq.add(r, r, t); fq.add(r, r, b);
        return l == r;
    }

    void dbl(Jac& o, const Jac& p) const {
        if (is_inf(p)) { o = p; return; }
        const Field& F = fq;
        if (a_zero) {  // dbl-2009-l
            Fe A, B, C, D, E, Fq_, t;
            F.sqr(A, p.X); F.sqr(B, p.Y); F.sqr(C, B);
            F.add(t, p.X, B); F.sqr(t, t); F.sub(t, t, A); F.sub(t, t, C); F.dbl(D, t);
            F.dbl(E, A); F.add(E, E, A);
            F.sqr(Fq_, E);
            Fe X3, Y3, Z3;
            F.mul(Z3, p.Y, p.Z); F.dbl(Z3, Z3);
            F.dbl(t, D); F.sub(X3, Fq_, t);
            F.sub(t, D, X3); F.mul(Y3, E, t);
            F.dbl(C, C); F.dbl(C, C); F.dbl(C, C);
            F.sub(Y3, Y3, C);
            o.X = X3; o.Y = Y3; o.Z = Z3;
        } else {  // dbl-2007-bl
            Fe XX, YY, YYYY, ZZ, S, M, t;
            F.sqr(XX, p.X); F.sqr(YY, p.Y); F.sqr(YYYY, YY); F.sqr(ZZ, p.Z);
            F.add(t, p.X, YY); F.sqr(t, t); F.sub(t, t, XX); F.sub(t, t, YYYY); F.dbl(S, t);
            F.dbl(M, XX); F.add(M, M, XX);
            F.sqr(t, ZZ); F.mul(t, t, a); F.add(M, M, t);
            Fe X3, Y3, Z3;
            F.sqr(X3, M); F.dbl(t, S); F.sub(X3, X3, t);
            F.add(Z3, p.Y, p.Z); F.sqr(Z3, Z3); F.sub(Z3, Z3, YY); F.sub(Z3, Z3, ZZ);
            F.sub(t, S, X3); F.mul(Y3, M, t);
            F.dbl(YYYY, YYYY); F.dbl(YYYY, YYYY); F.dbl(YYYY, YYYY);
            F.sub(Y3, Y3, YYYY);
            o.X = X3; o.Y = Y3; o.Z = Z3;
        }
    }
    // add-2007-bl with the exceptional cases ark handles (P+P -> double, P+(-P) -> identity)
    void add(Jac& o, const Jac& p, const Jac& q) const {
        if (is_inf(p)) { o = q; return; }
        if (is_inf(q)) { o = p; return; }
        const Field& F = fq;
        Fe Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, r, V, t;
        F.sqr(Z1Z1, p.Z); F.sqr(Z2Z2, q.Z);
        F.mul(U1, p.X, Z2Z2); F.mul(U2, q.X, Z1Z1);
        F.mul(S1, p.Y, q.Z); F.mul(S1, S1, Z2Z2);
        F.mul(S2, q.Y, p.Z); F.mul(S2, S2, Z1Z1);
        if (U1 == U2) {
            if (S1 == S2) { dbl(o, p); return; }
            o = jac_zero(); return;
        }
        F.sub(H, U2, U1);
        F.dbl(I, H); F.sqr(I, I);
        F.mul(J, H, I);
        F.sub(r, S2, S1); F.dbl(r, r);
        F.mul(V, U1, I);
        Fe X3, Y3, Z3;
        F.sqr(X3, r); F.sub(X3, X3, J); F.dbl(t, V); F.sub(X3, X3, t);
        F.sub(t, V, X3); F.mul(Y3, r, t); F.mul(t, S1, J); F.dbl(t, t); F.sub(Y3, Y3, t);
        F.add(Z3, p.Z, q.Z); F.sqr(Z3, Z3); F.sub(Z3, Z3, Z1Z1); F.sub(Z3, Z3, Z2Z2); F.mul(Z3, Z3, H);
        o.X = X3; o.Y = Y3; o.Z = Z3;
    }
    // madd-2007-bl
    void add_mixed(Jac& o, const Jac& p, const Aff& q) const {
        if (q.inf) { o = p; return; }
        if (is_inf(p)) { o = to_jac(q); return; }
        const Field& F = fq;
        Fe Z1Z1, U2, S2, H, HH, I, J, r, V, t;
        F.sqr(Z1Z1, p.Z);
        F.mul(U2, q.x, Z1Z1);
        F.mul(S2, q.y, p.Z); F.mul(S2, S2, Z1Z1);
        if (U2 == p.X) {
            if (S2 == p.Y) { dbl(o, p); return; }
            o = jac_zero(); return;
        }
        F.sub(H, U2, p.X);
        F.sqr(HH, H);
        F.dbl(I, HH); F.dbl(I, I);
        F.mul(J, H, I);
        F.sub(r, S2, p.Y); F.dbl(r, r);
        F.mul(V, p.X, I);
        Fe X3, Y3, Z3;
        F.sqr(X3, r); F.sub(X3, X3, J); F.dbl(t, V); F.sub(X3, X3, t);
        F.sub(t, V, X3); F.mul(Y3, r, t); F.mul(t, p.Y, J); F.dbl(t, t); F.sub(Y3, Y3, t);
        F.add(Z3, p.Z, H); F.sqr(Z3, Z3); F.sub(Z3, Z3, Z1Z1); F.sub(Z3, Z3, HH);
        o.X = X3; o.Y = Y3; o.Z = Z3;
    }
    // ark `CurveGroup::into_affine`: (X/Z^2, Y/Z^3)
    Aff to_affine(const Jac& p) const {
        if (is_inf(p)) return aff_zero();
        Fe zi, zi2, zi3;
        fq.inv(zi, p.Z);
        fq.sqr(zi2, zi); fq.mul(zi3, zi2, zi);
        Aff r; r.inf = false;
        fq.mul(r.x, p.X, zi2); fq.mul(r.y, p.Y, zi3);
        return r;
    }
    // ark `mul_bigint`: MSB-first double-and-add over a canonical 256-bit integer
    Jac mul_canon(const Aff& p, const u64 k[4]) const {
        Jac r = jac_zero();
        for (int i = 255; i >= 0; i--) {
            dbl(r, r);
            if ((k[i / 64] >> (i % 64)) & 1) add_mixed(r, r, p);
        }
        return r;
    }
    Jac mul(const Aff& p, const Fe& s) const {
        u64 k[4]; fr.to_canon(k, s);
        return mul_canon(p, k);
    }

    // ---- VariableBaseMSM::msm restated (ark-ec 0.4.2 `msm_bigint_wnaf`: signed-digit buckets,
    // c = 3 if n < 32 else floor(ceil(log2 n) * 69 / 100) + 2, 2^c bucket slots, running-sum
    // reduction, high->low window combine with c doublings).  The RESULT is a group element and
    // does not depend on this schedule; the schedule is kept for CPU-baseline timing fidelity.
    // threads over the windows of one msm call (1 = the reference's default features: single-threaded)
    static int& msm_threads() { static int t = 1; return t; }
    static int ark_window(size_t n) {
        if (n < 32) return 3;
        int lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        return lg * 69 / 100 + 2;
    }
    static void make_digits(std::vector<int64_t>& digits, const u64 scalar[4], int w, int num_bits) {
        u64 radix = (u64)1 << w, mask = radix - 1, carry = 0;
        int count = (num_bits + w - 1) / w;
        digits.assign(count, 0);
        for (int i = 0; i < count; i++) {
            int off = i * w, ui = off / 64, bi = off % 64;
            u64 buf;
            if (bi < 64 - w || ui == 3) buf = scalar[ui] >> bi;
            else buf = (scalar[ui] >> bi) | (scalar[ui + 1] << (64 - bi));
            u64 coef = carry + (buf & mask);
            carry = (coef + radix / 2) >> w;
            digits[i] = (int64_t)coef - (int64_t)(carry << w);
        }
        digits[count - 1] += (int64_t)(carry << w);
    }
    Jac msm(const Aff* bases, const Fe* scalars, size_t n) const {
        if (n == 0) return jac_zero();
        int c = ark_window(n);
        int num_bits = fr.bits;
        int W = (num_bits + c - 1) / c;
        std::vector<int64_t> all((size_t)W * n), d;
        for (size_t i = 0; i < n; i++) {
            u64 k[4]; fr.to_canon(k, scalars[i]);
            make_digits(d, k, c, num_bits);
            for (int w = 0; w < W; w++) all[i * W + w] = d[w];
        }
        std::vector<Jac> wsum(W);
        // one window: bucket accumulation + running-sum reduction (the closure ark-ec maps over `window_starts`)
        auto window = [&](int w, std::vector<Jac>& buckets) {
            for (auto& bk : buckets) bk = jac_zero();
            for (size_t i = 0; i < n; i++) {
                int64_t s = all[i * W + w];
                if (s > 0) add_mixed(buckets[s - 1], buckets[s - 1], bases[i]);
                else if (s < 0) add_mixed(buckets[-s - 1], buckets[-s - 1], neg(bases[i]));
            }
            Jac run = jac_zero(), res = jac_zero();
            for (size_t bi = buckets.size(); bi-- > 0;) {
                add(run, run, buckets[bi]);
                add(res, res, run);
            }
            wsum[w] = res;
        };
        const int nthreads = msm_threads();
        if (nthreads > 1) {
            // the reference's optional `parallel` feature (Cargo.toml:76 -> ark-ec/parallel): rayon over the Pippenger WINDOWS of
            // every msm call, nothing else [3P-mem]; restated with OpenMP for the all-cores CPU baseline
#pragma omp parallel num_threads(nthreads)
            {
                std::vector<Jac> buckets((size_t)1 << c);
#pragma omp for schedule(static)
                for (int w = 0; w < W; w++) window(w, buckets);
            }
        } else {
            std::vector<Jac> buckets((size_t)1 << c);
            for (int w = 0; w < W; w++) window(w, buckets);
        }
        Jac total = jac_zero();
        for (int w = W - 1; w >= 1; w--) {
            add(total, total, wsum[w]);
            for (int k = 0; k < c; k++) dbl(total, total);
        }
        add(total, total, wsum[0]);
        return total;
    }

    // ---- ark-serialize 0.4 encodings of SW affine points ---------------------------------
    // flag byte: 0x80 if y > -y (canonical integer order), 0x40 for the identity (x = y = 0).
    u8 y_flag(const Aff& p) const {
        if (p.inf) return 0x40;
        Fe ny; fq.neg(ny, p.y);
        return fq.cmp(p.y, ny) > 0 ? 0x80 : 0x00;
    }
    void ser_uncompressed(u8 out[65], const Aff& p) const {
        Aff q = p.inf ? aff_zero() : p;
        fq.to_bytes(out, q.x);
        fq.to_bytes(out + 32, q.y);
        out[64] = y_flag(p);
    }
    void ser_compressed(u8 out[33], const Aff& p) const {
        Aff q = p.inf ? aff_zero() : p;
        fq.to_bytes(out, q.x);
        out[32] = y_flag(p);
    }
    // ark-ec `get_point_from_x_unchecked(x, greatest)`: both roots ordered (smaller, larger)
    bool point_from_x(Aff& o, const Fe& x, bool greatest) const {
        Fe rhs, t, y, ny;
        fq.sqr(rhs, x); fq.mul(rhs, rhs, x);
        if (!a_zero) { fq.mul(t, a, x); fq.add(rhs, rhs, t); }
        fq.add(rhs, rhs, b);
        if (!fq.sqrt(y, rhs)) return false;
        fq.neg(ny, y);
        bool y_smaller = fq.cmp(y, ny) < 0;
        o.x = x; o.inf = false;
        o.y = (greatest == y_smaller) ? ny : y;
        return true;
    }
    // deserialize_compressed with validation (on-curve by construction; cofactor 1)
    bool deser_compressed(Aff& o, const u8 in[33]) const {
        u8 flags = in[32];
        if (flags & 0x3f) return false;
        if ((flags & 0xc0) == 0xc0) return false;
        Fe x;
        if (!fq.from_bytes(x, in)) return false;
        if (flags & 0x40) {
            if (!x.is_zero()) return false;
            o = aff_zero();
            return true;
        }
        return point_from_x(o, x, (flags & 0x80) != 0);
    }
};

}  // namespace orc
