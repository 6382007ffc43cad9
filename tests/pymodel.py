"""Independent big-integer model (plain Python ints, affine formulas) used to cross-check the C++
oracle's field and group arithmetic.  Written for this build; constants are the published curve
parameters (secp256k1 order / base prime; zorro: /root/reference/src/curve/zorro/{fq,fr,g1}.rs)."""

CURVES = {
    0: dict(  # secq256k1: base field = secp256k1 group order, scalar field = secp256k1 base prime
        q=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
        r=2**256 - 2**32 - 977,
        a=0,
        b=7,
        gx=53718550993811904772965658690407829053653678808745171666022356150019200052646,
        gy=28941648020349172432234515805717979317553499307621291159490218670604692907903,
    ),
    1: dict(
        q=57896044618658097711785492504343953927116110621106131396339151912985063395361,
        r=2**255 - 19,
        a=6,
        b=7277470329389939148381533754641607518092114590371880995609984561067837624798,
        gx=2,
        gy=19711758720854384559191066596451394956860102304684364148268676039962145446511,
    ),
}
R = 1 << 256


def add(c, P, Q):
    q, a = CURVES[c]["q"], CURVES[c]["a"]
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % q == 0:
            return None
        lam = (3 * x1 * x1 + a) * pow(2 * y1, -1, q) % q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, q) % q
    x3 = (lam * lam - x1 - x2) % q
    return x3, (lam * (x1 - x3) - y1) % q


def mul(c, P, k):
    acc = None
    while k:
        if k & 1:
            acc = add(c, acc, P)
        P = add(c, P, P)
        k >>= 1
    return acc


def on_curve(c, P):
    p = CURVES[c]
    return P is None or (P[1] ** 2 - P[0] ** 3 - p["a"] * P[0] - p["b"]) % p["q"] == 0


def msm(c, pts, ks):
    acc = None
    for P, k in zip(pts, ks):
        acc = add(c, acc, mul(c, P, k))
    return acc
