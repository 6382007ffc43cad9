"""Test gadgets written ONCE against the reference's ConstraintSystem trait shape (src/r1cs/constraint_system.rs:19-135) and run
on both recorders: the product's (ark_bulletproofs_amd.engine.ProverCS / VerifierCS over the C ABI) and the oracle's
(oracle.pyoracle.ProverCS / VerifierCS).  None of them is one of the product's built-in scenarios.

A "program" is a random sparse constraint system, satisfiable by construction: its STRUCTURE comes from one seed (so several
instances share it), its witness from another.  With `two_phase` it adds randomized constraints whose coefficients depend on a
transcript challenge (the shape of the reference's shuffle gadget, benches/r1cs_secq256k1.rs:47-76)."""
import random

VAR_COMMITTED, VAR_MULT_LEFT, VAR_MULT_RIGHT, VAR_MULT_OUT, VAR_ONE = 0, 1, 2, 3, 4
ONE = (VAR_ONE, 0)


class Field:
    """scalar-field helper: python ints <-> 4 x u64 Montgomery words (conversion by the oracle: test infrastructure)"""

    def __init__(self, O, curve):
        self.O, self.fid = O, O.fid(curve, True)
        self.p = O.modulus(self.fid)
        self._cache = {}

    def w(self, x):
        x %= self.p
        r = self._cache.get(x)
        if r is None:
            r = self._cache[x] = self.O.fe_from_int(self.fid, x)
        return r

    def i(self, words):
        return self.O.fe_to_int(self.fid, words)


class Witness:
    """values of the variables a prover-side run has seen (None on the verifier side)"""

    def __init__(self, F):
        self.F, self.val = F, {ONE: 1}

    def eval(self, lc):
        return sum(c * self.val[v] for v, c in lc) % self.F.p


def random_program(cs, F, struct_seed, wit, committed_vars, n_mul=12, n_alloc=3, n_extra=6, two_phase=False, n_mul2=5, publics=None, dense_coefs=True):
    """Records a random satisfiable circuit on `cs`.  wit: Witness (prover) or None (verifier).  publics: list the prover run fills
    with the public constants it derives (ints) and the verifier run reads back in the same order.  Returns nothing: all state is
    in cs / publics."""
    rs = random.Random(struct_seed)
    proving = wit is not None
    pub_iter = iter(publics) if (publics is not None and not proving) else None
    pool = list(committed_vars) + [ONE]

    def coef():
        t = rs.random()
        if t < 0.35:
            return 1
        if t < 0.55:
            return F.p - 1
        if t < 0.75 or not dense_coefs:
            return rs.randrange(2, 50)
        return rs.randrange(F.p)

    def rand_lc(maxlen=3):
        return [(rs.choice(pool), coef()) for _ in range(rs.randint(1, maxlen))]

    def W(lc):
        return [(v, F.w(c)) for v, c in lc]

    def mul(left, right):
        l, r, o = cs.multiply(W(left), W(right))
        if proving:
            wit.val[l], wit.val[r] = wit.eval(left), wit.eval(right)
            wit.val[o] = wit.val[l] * wit.val[r] % F.p
        pool.extend([l, r, o])
        return l, r, o

    def pin(lc):
        """a satisfiable constraint on lc without a public constant: bind it to a fresh allocated variable"""
        val = wit.eval(lc) if proving else None
        x = cs.allocate(F.w(val) if proving else None)
        if proving:
            wit.val[x] = val
        pool.append(x)
        cs.constrain(W(lc + [(x, F.p - 1)]))

    for _ in range(n_mul):
        mul(rand_lc(), rand_lc())
    for _ in range(n_alloc):   # allocate_multiplier with explicit inputs
        a, b = (rs.randrange(F.p), rs.randrange(1 << 20))
        l, r, o = cs.allocate_multiplier((F.w(a), F.w(b)) if proving else None)
        if proving:
            wit.val[l], wit.val[r], wit.val[o] = a, b, a * b % F.p
        pool.extend([l, r, o])
    for j in range(n_extra):
        lc = rand_lc(4)
        if j % 2 == 0:
            pin(lc)
        else:     # lc - c = 0 with a PUBLIC constant c (differs per instance: the per-instance coefficient-table path)
            if proving:
                c = wit.eval(lc)
                publics.append(c)
            else:
                c = next(pub_iter)
            cs.constrain(W(lc + [(ONE, (F.p - c) % F.p)]))
    if two_phase:
        snapshot = list(pool)
        seed2 = rs.randrange(1 << 30)

        def randomized(cs2):
            r2 = random.Random(seed2)
            z = F.i(cs2.challenge_scalar(b"gadget challenge"))
            pool2 = list(snapshot)

            def coef2():
                t = r2.random()
                return 1 if t < 0.3 else F.p - 1 if t < 0.5 else z if t < 0.7 else (F.p - z) % F.p if t < 0.85 else (z * z + r2.randrange(5)) % F.p

            def lc2(maxlen=3):
                return [(r2.choice(pool2), coef2()) for _ in range(r2.randint(1, maxlen))]

            for _ in range(n_mul2):
                left, right = lc2(), lc2()
                l, r, o = cs2.multiply(W(left), W(right))
                if proving:
                    wit.val[l], wit.val[r] = wit.eval(left), wit.eval(right)
                    wit.val[o] = wit.val[l] * wit.val[r] % F.p
                pool2.extend([l, r, o])
            for _ in range(3):
                lc = lc2(4)
                val = wit.eval(lc) if proving else None
                x = cs2.allocate(F.w(val) if proving else None)
                if proving:
                    wit.val[x] = val
                pool2.append(x)
                cs2.constrain(W(lc + [(x, F.p - 1)]))

        cs.specify_randomized_constraints(randomized)


def make_witness(F, wit_seed, m):
    rw = random.Random(wit_seed)
    vals = [rw.randrange(F.p) if j % 2 else rw.randrange(1 << 32) for j in range(m)]
    blinds = [rw.randrange(F.p) for _ in range(m)]
    return vals, blinds
