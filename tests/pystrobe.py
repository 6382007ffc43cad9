"""Independent pure-Python model of what sits under the Fiat-Shamir layer of the path: Keccak-f[1600] (FIPS 202), STROBE-128/1600
in the duplex form merlin 3.0 uses (strobe.rs), merlin::Transcript and merlin::TranscriptRng.  Written from the specifications,
sharing no code with oracle/ or the product's host_proto.hpp; the model itself is pinned by hashlib (SHA3-512 through its own
permutation) and by merlin's published transcript vectors in tests/test_oracle_vectors.py."""

_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001, 0x8000000080008081,
       0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B,
       0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A, 0x8000000080008081,
       0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M if n else x


def keccak_f(state):
    """state: bytearray(200), permuted in place"""
    A = [[int.from_bytes(state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rnd in range(24):
        Cc = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [Cc[(x - 1) % 5] ^ _rol(Cc[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8] = (A[x][y] & _M).to_bytes(8, "little")


def sha3_512(msg):
    st, rate = bytearray(200), 72
    m = bytearray(msg) + b"\x06"
    m += bytes((-len(m)) % rate)
    m[-1] |= 0x80
    for off in range(0, len(m), rate):
        for i in range(rate):
            st[i] ^= m[off + i]
        keccak_f(st)
    return bytes(st[:64])


FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32
RATE = 166


class Strobe128:
    def __init__(self, label):
        self.st = bytearray(200)
        self.st[0:6] = bytes([1, RATE + 2, 1, 0, 1, 96])
        self.st[6:18] = b"STROBEv1.0.2"
        keccak_f(self.st)
        self.pos, self.pos_begin, self.cur = 0, 0, 0
        self.meta_ad(label, False)

    def clone(self):
        o = Strobe128.__new__(Strobe128)
        o.st, o.pos, o.pos_begin, o.cur = bytearray(self.st), self.pos, self.pos_begin, self.cur
        return o

    def _run_f(self):
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[RATE + 1] ^= 0x80
        keccak_f(self.st)
        self.pos, self.pos_begin = 0, 0

    def _absorb(self, data):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == RATE:
                self._run_f()

    def _overwrite(self, data):
        for b in data:
            self.st[self.pos] = b
            self.pos += 1
            if self.pos == RATE:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray()
        for _ in range(n):
            out.append(self.st[self.pos])
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == RATE:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur == flags
            return
        assert not (flags & FLAG_T)
        old = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur = flags
        self._absorb(bytes([old, flags]))
        if (flags & (FLAG_C | FLAG_K)) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n, more=False):
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)

    def key(self, data, more=False):
        self._begin_op(FLAG_A | FLAG_C, more)
        self._overwrite(data)


class Transcript:
    def __init__(self, label):
        self.s = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, msg):
        self.s.meta_ad(label, False)
        self.s.meta_ad(len(msg).to_bytes(4, "little"), True)
        self.s.ad(msg, False)

    def append_u64(self, label, x):
        self.append_message(label, int(x).to_bytes(8, "little"))

    def challenge_bytes(self, label, n):
        self.s.meta_ad(label, False)
        self.s.meta_ad(int(n).to_bytes(4, "little"), True)
        return self.s.prf(n)

    def build_rng(self):
        return TranscriptRngBuilder(self.s.clone())


class TranscriptRngBuilder:
    def __init__(self, strobe):
        self.s = strobe

    def rekey_with_witness_bytes(self, label, witness):
        self.s.meta_ad(label, False)
        self.s.meta_ad(len(witness).to_bytes(4, "little"), True)
        self.s.key(witness, False)
        return self

    def finalize(self, random_bytes32):
        self.s.meta_ad(b"rng", False)
        self.s.key(random_bytes32, False)
        return TranscriptRng(self.s)


class TranscriptRng:
    def __init__(self, strobe):
        self.s = strobe

    def fill_bytes(self, n):
        self.s.meta_ad(int(n).to_bytes(4, "little"), False)
        return self.s.prf(n)

    def next_u64(self):
        return int.from_bytes(self.fill_bytes(8), "little")


def chacha20_block(key32, counter, nonce12=bytes(12)):
    """RFC 7539 block function, written from the RFC (independent of oracle/hash.hpp): 64 keystream bytes"""
    def rotl(v, c):
        return ((v << c) & 0xFFFFFFFF) | (v >> (32 - c))

    def qr(s, a, b, c, d):
        s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = rotl(s[d] ^ s[a], 16)
        s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = rotl(s[b] ^ s[c], 12)
        s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = rotl(s[d] ^ s[a], 8)
        s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = rotl(s[b] ^ s[c], 7)

    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + [int.from_bytes(key32[4 * i: 4 * i + 4], "little") for i in range(8)] + [counter & 0xFFFFFFFF] + \
           [int.from_bytes(nonce12[4 * i: 4 * i + 4], "little") for i in range(3)]
    s = list(init)
    for _ in range(10):
        qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15)
        qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14)
    return b"".join(((s[i] + init[i]) & 0xFFFFFFFF).to_bytes(4, "little") for i in range(16))
