"""Independent Python big-int model of BN254 (alt_bn128) used to cross-check the C oracle.

TEST INFRASTRUCTURE ONLY.  Deliberately shares no code and no algorithm with oracle/*.c or the HIP
kernels: fields are Python ints reduced with `%`, curve arithmetic is textbook AFFINE chord-tangent
(one modular inversion per addition), the DFT is the O(n^2) definition.  It restates only the
*conventions* the reference pins:
  - moduli:            pairing/src/bn256/fq.rs:5, fr.rs:4
  - Montgomery R=2^256, 4 little-endian u64 limbs:  fq.rs:39-50 (G1 generator literals)
  - curve y^2 = x^3 + 3 (fq.rs:9-16), twist y^2 = x^3 + 3/(9+u) (fq.rs:18-31)
  - generators:        G1 (1, 2) fq.rs:35-50; G2 decimals fq.rs:52-58
  - Fr: multiplicative generator 7 (fr.rs:5), two-adicity S = 28 (fr.rs:31-34)
"""
from __future__ import annotations

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MONT_R = 1 << 256
FR_S = 28
FR_GENERATOR = 7
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R_ORDER - 1) >> FR_S, R_ORDER)  # order 2^28

G1_GEN = (1, 2)
# fq.rs:52-58: x = x_c1*u + x_c0, y = y_c1*u + y_c0
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


# ---------------------------------------------------------------- limb helpers
def to_limbs(x: int) -> list[int]:
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_limbs(l) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def to_mont(x: int, p: int) -> int:
    return x * MONT_R % p


def from_mont(x: int, p: int) -> int:
    return x * pow(MONT_R, -1, p) % p


# ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2+1), tuples (c0, c1)
def f2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def f2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def f2_neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return (a[0] * n % Q, (-a[1]) * n % Q)


class Fp:
    """Field-op bundle so the affine group law below is written once for G1 and G2."""

    def __init__(self, add, sub, neg, mul, inv, zero, from_int):
        self.add, self.sub, self.neg, self.mul, self.inv, self.zero, self.from_int = add, sub, neg, mul, inv, zero, from_int


FQ_OPS = Fp(lambda a, b: (a + b) % Q, lambda a, b: (a - b) % Q, lambda a: (-a) % Q, lambda a, b: a * b % Q,
            lambda a: pow(a, -1, Q), 0, lambda v: v % Q)
FQ2_OPS = Fp(f2_add, f2_sub, f2_neg, f2_mul, f2_inv, (0, 0), lambda v: (v % Q, 0))

B_G1 = 3
B_G2 = f2_mul((3, 0), f2_inv((9, 1)))  # 3 / (9 + u)


# ---------------------------------------------------------------- affine group law; None == infinity
def ec_add(F: Fp, p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if y1 != y2 or y1 == F.zero:
            return None
        lam = F.mul(F.mul(F.from_int(3), F.mul(x1, x1)), F.inv(F.add(y1, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_neg(F: Fp, p):
    return None if p is None else (p[0], F.neg(p[1]))


def ec_mul(F: Fp, p, k: int):
    acc = None
    add = p
    while k:
        if k & 1:
            acc = ec_add(F, acc, add)
        add = ec_add(F, add, add)
        k >>= 1
    return acc


def on_curve_g1(p):
    return p is None or (p[1] * p[1] - p[0] ** 3 - B_G1) % Q == 0


def on_curve_g2(p):
    if p is None:
        return True
    x, y = p
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B_G2)) == (0, 0)


def msm(F: Fp, points, scalars):
    acc = None
    for p, k in zip(points, scalars):
        acc = ec_add(F, acc, ec_mul(F, p, k))
    return acc


# ---------------------------------------------------------------- raw (boundary) encodings
def g1_affine_to_raw(p) -> list[int]:
    """x||y Montgomery limbs (8 u64); infinity = all zero (ec.rs:673-675)."""
    if p is None:
        return [0] * 8
    return to_limbs(to_mont(p[0], Q)) + to_limbs(to_mont(p[1], Q))


def g1_affine_from_raw(l):
    l = [int(v) for v in l]
    if not any(l):
        return None
    return (from_mont(from_limbs(l[0:4]), Q), from_mont(from_limbs(l[4:8]), Q))


def g2_affine_to_raw(p) -> list[int]:
    """x.c0||x.c1||y.c0||y.c1 Montgomery limbs (16 u64); infinity = all zero."""
    if p is None:
        return [0] * 16
    out = []
    for c in (p[0][0], p[0][1], p[1][0], p[1][1]):
        out += to_limbs(to_mont(c, Q))
    return out


def g2_affine_from_raw(l):
    l = [int(v) for v in l]
    if not any(l):
        return None
    c = [from_mont(from_limbs(l[4 * i:4 * i + 4]), Q) for i in range(4)]
    return ((c[0], c[1]), (c[2], c[3]))


def g1_jac_from_raw(l):
    """Jacobian X,Y,Z Montgomery limbs (12 u64) -> affine tuple / None."""
    l = [int(v) for v in l]
    x, y, z = (from_mont(from_limbs(l[4 * i:4 * i + 4]), Q) for i in range(3))
    if z == 0:
        return None
    zi = pow(z, -1, Q)
    return (x * zi * zi % Q, y * zi * zi * zi % Q)


def g2_jac_from_raw(l):
    l = [int(v) for v in l]
    c = [from_mont(from_limbs(l[4 * i:4 * i + 4]), Q) for i in range(6)]
    x, y, z = (c[0], c[1]), (c[2], c[3]), (c[4], c[5])
    if z == (0, 0):
        return None
    zi = f2_inv(z)
    zi2 = f2_mul(zi, zi)
    return (f2_mul(x, zi2), f2_mul(y, f2_mul(zi2, zi)))


# ---------------------------------------------------------------- Fr DFT (definition)
def domain_omega(log_n: int) -> int:
    """omega of EvaluationDomain::from_coeffs (domain.rs:84-86): root_of_unity squared S-exp times."""
    w = FR_ROOT_OF_UNITY
    for _ in range(log_n, FR_S):
        w = w * w % R_ORDER
    return w


def dft(a, omega):
    """X[k] = sum_i a[i] * omega^(i*k)  -- what serial_fft (domain.rs:274-317) computes."""
    n = len(a)
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * omega % R_ORDER
    return [sum(a[i] * pw[(i * k) % n] for i in range(n)) % R_ORDER for k in range(n)]


def fft_recursive(a, omega):
    """O(n log n) recursive radix-2 (still independent of the oracle's iterative in-place form)."""
    n = len(a)
    if n == 1:
        return list(a)
    w2 = omega * omega % R_ORDER
    ev = fft_recursive(a[0::2], w2)
    od = fft_recursive(a[1::2], w2)
    out = [0] * n
    w = 1
    for k in range(n // 2):
        t = w * od[k] % R_ORDER
        out[k] = (ev[k] + t) % R_ORDER
        out[k + n // 2] = (ev[k] - t) % R_ORDER
        w = w * omega % R_ORDER
    return out


def domain_op(a, op: str):
    """fft / ifft / coset_fft / icoset_fft of domain.rs:154-203 on canonical ints."""
    n = len(a)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    w = domain_omega(log_n)
    f = fft_recursive
    if op == "fft":
        return f(a, w)
    if op == "ifft":
        minv = pow(n, -1, R_ORDER)
        return [v * minv % R_ORDER for v in f(a, pow(w, -1, R_ORDER))]
    if op == "coset_fft":
        g = FR_GENERATOR
        return f([v * pow(g, i, R_ORDER) % R_ORDER for i, v in enumerate(a)], w)
    if op == "icoset_fft":
        minv = pow(n, -1, R_ORDER)
        ginv = pow(FR_GENERATOR, -1, R_ORDER)
        return [v * minv % R_ORDER * pow(ginv, i, R_ORDER) % R_ORDER for i, v in enumerate(f(a, pow(w, -1, R_ORDER)))]
    raise ValueError(op)


# ---------------------------------------------------------------- rand 0.4 XorShiftRng (seeded tests of the reference)
class XorShiftRng:
    """xorshift128 as used by the reference's seeded tests (e.g. bellman/src/multiexp.rs:560:
    XorShiftRng::from_seed([0x3dbe6259, 0x8d313d76, 0x3237db17, 0xe5bc0654]))."""

    def __init__(self, seed=(0x3dbe6259, 0x8d313d76, 0x3237db17, 0xe5bc0654)):
        self.x, self.y, self.z, self.w = seed

    def next_u32(self) -> int:
        t = (self.x ^ (self.x << 11)) & 0xFFFFFFFF
        self.x, self.y, self.z = self.y, self.z, self.w
        self.w = (self.w ^ (self.w >> 19) ^ (t ^ (t >> 8))) & 0xFFFFFFFF
        return self.w

    def next_u64(self) -> int:
        hi = self.next_u32()
        return (hi << 32) | self.next_u32()

    def next_below(self, bound: int, bits: int = 254) -> int:
        """uniform in [0, bound) by rejection on `bits`-bit draws."""
        while True:
            v = 0
            for i in range(4):
                v |= self.next_u64() << (64 * i)
            v &= (1 << bits) - 1
            if v < bound:
                return v
