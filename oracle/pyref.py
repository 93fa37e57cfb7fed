"""Pure-Python (bigint) restatement of the arithmetic behind Bazuka's MPN Groth16 path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (bazuka_amd/, the C-ABI library) may import
this module; only tests/, oracle/ tooling, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

Role: an independent second opinion for the C oracle (oracle/*.c) and the generator of the small
golden fixtures under tests/golden/.  It is slow (CPython bigints) and used on small sizes only.

What it follows (paths relative to /root/reference):
  * Fr constants / ZkScalar semantics ............ src/zk/mod.rs:196-271
  * Poseidon permutation ......................... src/zk/poseidon/mod.rs:24-84
  * Poseidon parameters .......................... src/zk/poseidon/params/mod.rs:27-80 (files are the
    output of hadeshash `generate_params_poseidon.sage 1 0 255 t 5 128`, params/README.md; here they
    are RE-DERIVED with the public Grain-LFSR procedure and compared to the files in tests)
  * 4-ary state tree ............................. src/zk/state/mod.rs:310-420, src/zk/mod.rs:401-423
  * proof / VK byte formats ...................... src/zk/groth16/mod.rs:19-65
  * verify equation .............................. src/zk/groth16/mod.rs:67-121 (bellman verify_proof)
  * Groth16 setup/prove: bellman 0.14 (third-party, NOT under /root/reference; algorithm restated
    from the Groth16 paper + bellman's published layout).
  * BLS12-381: bls12_381 0.8 (third-party; public curve parameters).

Parity pinning: 16 Poseidon KATs (src/zk/poseidon/mod.rs:114-149) and the three hard-coded VK blobs
(src/config/blockchain.rs:32-37) are reproduced in tests/test_oracle_*.py.  Groth16 proof bytes are
NOT pinned by any reference vector ("parity unpinned" for proof bytes: every reference prove call
draws from OsRng) - see DESIGN.md.
"""
from __future__ import annotations

# --------------------------------------------------------------------------------------------------
# constants
# --------------------------------------------------------------------------------------------------
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # Fr modulus
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FR_GENERATOR = 7
FR_S = 32
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R_MOD - 1) >> FR_S, R_MOD)
FR_MONT_R = (1 << 256) % R_MOD
FP_MONT_R = (1 << 384) % P_MOD
BLS_X = 0xD201000000010000  # |x|, the curve parameter is -x

G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
G2_GEN = (
    (
        0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
    ),
    (
        0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
    ),
)


def inv_mod(a: int, m: int) -> int:
    return pow(a, -1, m)


# --------------------------------------------------------------------------------------------------
# Fr helpers (canonical ints) and byte formats
# --------------------------------------------------------------------------------------------------
def fr_to_mont_bytes(x: int) -> bytes:
    """32-byte LE Montgomery limbs = in-memory / bincode ZkScalar."""
    return ((x % R_MOD) * FR_MONT_R % R_MOD).to_bytes(32, "little")


def fr_from_mont_bytes(b: bytes) -> int:
    return int.from_bytes(b, "little") * inv_mod(FR_MONT_R, R_MOD) % R_MOD


def fr_to_canon_bytes(x: int) -> bytes:
    return (x % R_MOD).to_bytes(32, "little")


def fp_to_mont_bytes(x: int) -> bytes:
    return ((x % P_MOD) * FP_MONT_R % P_MOD).to_bytes(48, "little")


def fp_from_mont_bytes(b: bytes) -> int:
    return int.from_bytes(b, "little") * inv_mod(FP_MONT_R, P_MOD) % P_MOD


def zkscalar_new(b: bytes) -> int:
    """ZkScalar::new = LE integer mod r (src/zk/mod.rs:262-271)."""
    return int.from_bytes(b, "little") % R_MOD


# --------------------------------------------------------------------------------------------------
# Poseidon parameters (Grain LFSR, hadeshash) and permutation
# --------------------------------------------------------------------------------------------------
class _Grain:
    def __init__(self, t: int, r_f: int, r_p: int, n: int = 255, field: int = 1, sbox: int = 0):
        bits = []
        for val, width in ((field, 2), (sbox, 4), (n, 12), (t, 12), (r_f, 10), (r_p, 10)):
            bits += [(val >> (width - 1 - i)) & 1 for i in range(width)]
        bits += [1] * 30
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._step()

    def _step(self) -> int:
        s = self.s
        new = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(new)
        return new

    def bit(self) -> int:
        while True:
            a = self._step()
            b = self._step()
            if a:
                return b

    def raw(self, n: int = 255) -> int:
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v


_POSEIDON_CACHE: dict = {}


def poseidon_rounds(t: int):
    return 8, (56 if t <= 5 else 57)


def poseidon_params(t: int):
    """(round_constants[t*(R_F+R_P)], mds[t][t]) for width t in 2..17 (= arity + 1)."""
    if t in _POSEIDON_CACHE:
        return _POSEIDON_CACHE[t]
    r_f, r_p = poseidon_rounds(t)
    g = _Grain(t, r_f, r_p)
    rc = []
    while len(rc) < t * (r_f + r_p):
        v = g.raw()
        if v < R_MOD:  # rejection sampling for round constants
            rc.append(v)
    xs = [g.raw() % R_MOD for _ in range(t)]  # MDS draws: reduced, NOT rejected
    ys = [g.raw() % R_MOD for _ in range(t)]
    mds = [[inv_mod((xs[i] + ys[j]) % R_MOD, R_MOD) for j in range(t)] for i in range(t)]
    _POSEIDON_CACHE[t] = (rc, mds)
    return rc, mds


def poseidon(vals):
    """src/zk/poseidon/mod.rs:24-84: state=[0]+vals, 4 full, R_P partial (S-box on elem 0), 4 full,
    dense MDS each round, output state[1]."""
    t = len(vals) + 1
    assert 2 <= t <= 17
    rc, mds = poseidon_params(t)
    r_f, r_p = poseidon_rounds(t)
    st = [0] + [v % R_MOD for v in vals]
    off = 0

    def mix(st):
        return [sum(mds[j][k] * st[k] for k in range(t)) % R_MOD for j in range(t)]

    for rnd in range(r_f + r_p):
        st = [(st[i] + rc[off + i]) % R_MOD for i in range(t)]
        off += t
        if rnd < r_f // 2 or rnd >= r_f // 2 + r_p:
            st = [pow(x, 5, R_MOD) for x in st]
        else:
            st[0] = pow(st[0], 5, R_MOD)
        st = mix(st)
    return st[1]


def merkle4_root(leaves, log4: int, nodes_out=None):
    """Dense 4-ary Poseidon tree (== KvStoreStateManager root for a fully populated
    List{log4_size, Scalar}; src/zk/state/mod.rs:353-391).  nodes_out, if given, receives the heap
    layout (4^k-1)/3+i of every internal level (root at index 0)."""
    assert len(leaves) == 4 ** log4
    level = list(leaves)
    levels = [level]
    for _ in range(log4):
        level = [poseidon(level[4 * i:4 * i + 4]) for i in range(len(level) // 4)]
        levels.append(level)
    if nodes_out is not None:
        for k in range(log4):  # depth k has 4^k nodes
            nodes_out.extend(levels[log4 - k])
    return level[0]


# --------------------------------------------------------------------------------------------------
# Fp2 / Fp6 / Fp12 tower (u^2=-1, v^3=1+u, w^2=v)
# --------------------------------------------------------------------------------------------------
def f2_add(a, b): return ((a[0] + b[0]) % P_MOD, (a[1] + b[1]) % P_MOD)
def f2_sub(a, b): return ((a[0] - b[0]) % P_MOD, (a[1] - b[1]) % P_MOD)
def f2_neg(a): return ((-a[0]) % P_MOD, (-a[1]) % P_MOD)
def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P_MOD, (a[0] * b[1] + a[1] * b[0]) % P_MOD)
def f2_sqr(a): return f2_mul(a, a)
def f2_scale(a, k): return (a[0] * k % P_MOD, a[1] * k % P_MOD)
def f2_inv(a):
    d = inv_mod((a[0] * a[0] + a[1] * a[1]) % P_MOD, P_MOD)
    return (a[0] * d % P_MOD, (-a[1]) * d % P_MOD)
def f2_mul_xi(a): return ((a[0] - a[1]) % P_MOD, (a[0] + a[1]) % P_MOD)  # * (1+u)
F2_ZERO = (0, 0)
F2_ONE = (1, 0)

def f6_add(a, b): return tuple(f2_add(x, y) for x, y in zip(a, b))
def f6_sub(a, b): return tuple(f2_sub(x, y) for x, y in zip(a, b))
def f6_neg(a): return tuple(f2_neg(x) for x in a)
def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0, t1, t2 = f2_mul(a0, b0), f2_mul(a1, b1), f2_mul(a2, b2)
    c0 = f2_add(t0, f2_mul_xi(f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    c1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul_xi(t2))
    c2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a2, b0)), t1)
    return (c0, c1, c2)
def f6_mul_v(a): return (f2_mul_xi(a[2]), a[0], a[1])
def f6_inv(a):
    a0, a1, a2 = a
    c0 = f2_sub(f2_sqr(a0), f2_mul_xi(f2_mul(a1, a2)))
    c1 = f2_sub(f2_mul_xi(f2_sqr(a2)), f2_mul(a0, a1))
    c2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    t = f2_add(f2_mul(a0, c0), f2_mul_xi(f2_add(f2_mul(a2, c1), f2_mul(a1, c2))))
    ti = f2_inv(t)
    return (f2_mul(c0, ti), f2_mul(c1, ti), f2_mul(c2, ti))
F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)

def f12_mul(a, b):
    a0, a1 = a
    b0, b1 = b
    t0, t1 = f6_mul(a0, b0), f6_mul(a1, b1)
    c0 = f6_add(t0, f6_mul_v(t1))
    c1 = f6_sub(f6_sub(f6_mul(f6_add(a0, a1), f6_add(b0, b1)), t0), t1)
    return (c0, c1)
def f12_sqr(a): return f12_mul(a, a)
def f12_conj(a): return (a[0], f6_neg(a[1]))
def f12_inv(a):
    a0, a1 = a
    t = f6_inv(f6_sub(f6_mul(a0, a0), f6_mul_v(f6_mul(a1, a1))))
    return (f6_mul(a0, t), f6_neg(f6_mul(a1, t)))
F12_ONE = (F6_ONE, F6_ZERO)
def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_sqr(r)
        if bit == "1":
            r = f12_mul(r, a)
    return r


# --------------------------------------------------------------------------------------------------
# curves: generic short-Weierstrass (a=0) affine arithmetic over a field given by an op table
# points are None (identity) or (x, y)
# --------------------------------------------------------------------------------------------------
class _Field:
    def __init__(self, add, sub, mul, inv, neg, zero, one, eq=None):
        self.add, self.sub, self.mul, self.inv, self.neg = add, sub, mul, inv, neg
        self.zero, self.one = zero, one

FP = _Field(lambda a, b: (a + b) % P_MOD, lambda a, b: (a - b) % P_MOD, lambda a, b: a * b % P_MOD,
            lambda a: inv_mod(a, P_MOD), lambda a: (-a) % P_MOD, 0, 1)
FP2 = _Field(f2_add, f2_sub, f2_mul, f2_inv, f2_neg, F2_ZERO, F2_ONE)
G1_B = 4
G2_B = (4, 4)


def ec_add(F, p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if p[1] == q[1]:
            return ec_double(F, p)
        return None
    lam = F.mul(F.sub(q[1], p[1]), F.inv(F.sub(q[0], p[0])))
    x3 = F.sub(F.sub(F.mul(lam, lam), p[0]), q[0])
    y3 = F.sub(F.mul(lam, F.sub(p[0], x3)), p[1])
    return (x3, y3)


def ec_double(F, p):
    if p is None: return None
    if p[1] == F.zero: return None
    xx = F.mul(p[0], p[0])
    lam = F.mul(F.add(F.add(xx, xx), xx), F.inv(F.add(p[1], p[1])))
    x3 = F.sub(F.sub(F.mul(lam, lam), p[0]), p[0])
    y3 = F.sub(F.mul(lam, F.sub(p[0], x3)), p[1])
    return (x3, y3)


def ec_neg(F, p):
    return None if p is None else (p[0], F.neg(p[1]))


def ec_mul(F, p, k: int):
    if k < 0:
        return ec_mul(F, ec_neg(F, p), -k)
    r = None
    for bit in bin(k)[2:] if k else "":
        r = ec_double(F, r)
        if bit == "1":
            r = ec_add(F, r, p)
    return r


def ec_msm(F, pts, scalars):
    acc = None
    for p, s in zip(pts, scalars):
        acc = ec_add(F, acc, ec_mul(F, p, s % R_MOD))
    return acc


def g1_on_curve(p):
    return p is None or (p[1] * p[1] - p[0] ** 3 - G1_B) % P_MOD == 0


def g2_on_curve(p):
    if p is None: return True
    return f2_sub(f2_sqr(p[1]), f2_add(f2_mul(f2_sqr(p[0]), p[0]), G2_B)) == F2_ZERO


def g1_mul(p, k): return ec_mul(FP, p, k)
def g2_mul(p, k): return ec_mul(FP2, p, k)
def g1_add(p, q): return ec_add(FP, p, q)
def g2_add(p, q): return ec_add(FP2, p, q)


# byte formats: G1 = x(48) y(48) inf(1); identity = (0, 1, inf=1) in Montgomery form
def g1_to_bytes(p) -> bytes:
    if p is None:
        return fp_to_mont_bytes(0) + fp_to_mont_bytes(1) + b"\x01"
    return fp_to_mont_bytes(p[0]) + fp_to_mont_bytes(p[1]) + b"\x00"


def g1_from_bytes(b: bytes):
    assert len(b) == 97
    if b[96]:
        return None
    return (fp_from_mont_bytes(b[0:48]), fp_from_mont_bytes(b[48:96]))


def g2_to_bytes(p) -> bytes:
    if p is None:
        return (fp_to_mont_bytes(0) * 2) + fp_to_mont_bytes(1) + fp_to_mont_bytes(0) + b"\x01"
    (x0, x1), (y0, y1) = p
    return fp_to_mont_bytes(x0) + fp_to_mont_bytes(x1) + fp_to_mont_bytes(y0) + fp_to_mont_bytes(y1) + b"\x00"


def g2_from_bytes(b: bytes):
    assert len(b) == 193
    if b[192]:
        return None
    f = [fp_from_mont_bytes(b[48 * i:48 * i + 48]) for i in range(4)]
    return ((f[0], f[1]), (f[2], f[3]))


def g1_raw96(p) -> bytes:
    """MSM base format: x|y Montgomery, no infinity flag."""
    return fp_to_mont_bytes(p[0]) + fp_to_mont_bytes(p[1])


def g2_raw192(p) -> bytes:
    (x0, x1), (y0, y1) = p
    return fp_to_mont_bytes(x0) + fp_to_mont_bytes(x1) + fp_to_mont_bytes(y0) + fp_to_mont_bytes(y1)


# --------------------------------------------------------------------------------------------------
# pairing (optimal ate, M-type twist), used for the verify side only
# --------------------------------------------------------------------------------------------------
def _line(lam, xt, yt, p):
    """Line through T (twist coords, slope lam in Fp2) evaluated at P in G1, scaled by w^3 (an Fp4
    element, killed by the final exponentiation): (lam*xt - yt) + (-lam*xP) w^2 + yP w^3."""
    c0 = f2_sub(f2_mul(lam, xt), yt)
    c1 = f2_scale(f2_neg(lam), p[0])
    c3 = (p[1], 0)
    return ((c0, c1, F2_ZERO), (F2_ZERO, c3, F2_ZERO))


def miller_loop(p, q):
    if p is None or q is None:
        return F12_ONE
    f = F12_ONE
    t = q
    for bit in bin(BLS_X)[3:]:
        xx = f2_sqr(t[0])
        lam = f2_mul(f2_add(f2_add(xx, xx), xx), f2_inv(f2_add(t[1], t[1])))
        f = f12_mul(f12_sqr(f), _line(lam, t[0], t[1], p))
        t = ec_double(FP2, t)
        if bit == "1":
            lam = f2_mul(f2_sub(q[1], t[1]), f2_inv(f2_sub(q[0], t[0])))
            f = f12_mul(f, _line(lam, t[0], t[1], p))
            t = ec_add(FP2, t, q)
    return f12_conj(f)  # x is negative


_FINAL_EXP = (P_MOD ** 12 - 1) // R_MOD


def final_exp(f):
    # easy part via conj/inv, then the rest by plain exponentiation (simple, slow, robust)
    f = f12_mul(f12_conj(f), f12_inv(f))  # f^(p^6-1)
    return f12_pow(f, (P_MOD ** 6 + 1) // R_MOD)


def pairing(p, q):
    return final_exp(miller_loop(p, q))


# --------------------------------------------------------------------------------------------------
# NTT over Fr (bellman EvaluationDomain semantics)
# --------------------------------------------------------------------------------------------------
def omega_for(log_n: int) -> int:
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - log_n), R_MOD)


def ntt(a, log_n: int, inverse: bool = False, coset: bool = False):
    """forward: a_k -> sum_j a_j w^{jk}; coset forward scales a_j by g^j first (g=7);
    inverse: divides by n; coset inverse additionally scales by g^{-j} after."""
    n = 1 << log_n
    assert len(a) == n
    a = [x % R_MOD for x in a]
    w = omega_for(log_n)
    if inverse:
        w = inv_mod(w, R_MOD)
    if coset and not inverse:
        g = 1
        for j in range(n):
            a[j] = a[j] * g % R_MOD
            g = g * FR_GENERATOR % R_MOD
    # iterative radix-2 DIT
    rev = [0] * n
    for i in range(n):
        rev[i] = (rev[i >> 1] >> 1) | ((i & 1) << (log_n - 1)) if log_n else 0
    a = [a[rev[i]] for i in range(n)]
    m = 1
    while m < n:
        wm = pow(w, n // (2 * m), R_MOD)
        for k in range(0, n, 2 * m):
            t = 1
            for j in range(m):
                u = a[k + j]
                v = a[k + j + m] * t % R_MOD
                a[k + j] = (u + v) % R_MOD
                a[k + j + m] = (u - v) % R_MOD
                t = t * wm % R_MOD
        m *= 2
    if inverse:
        ninv = inv_mod(n, R_MOD)
        a = [x * ninv % R_MOD for x in a]
        if coset:
            gi = inv_mod(FR_GENERATOR, R_MOD)
            g = 1
            for j in range(n):
                a[j] = a[j] * g % R_MOD
                g = g * gi % R_MOD
    return a


# --------------------------------------------------------------------------------------------------
# Groth16 over a generic R1CS (bellman layout)
# R1CS: n_in (incl. ONE at index 0), n_aux, constraints = list of (A, B, C), each a list of
# (var, coeff) with var = ("in", i) / ("aux", i) encoded as int: i for inputs, n_in + i for aux.
# --------------------------------------------------------------------------------------------------
class R1CS:
    def __init__(self, n_in: int, n_aux: int, constraints):
        self.n_in, self.n_aux = n_in, n_aux
        self.constraints = list(constraints)
        # bellman appends input_i * 0 = 0 for every input (incl. ONE)
        for i in range(n_in):
            self.constraints.append(([(i, 1)], [], []))

    @property
    def n_vars(self):
        return self.n_in + self.n_aux

    def log_m(self):
        m, lg = 1, 0
        while m < len(self.constraints):
            m, lg = m * 2, lg + 1
        return lg

    def densities(self):
        a_d = [False] * self.n_vars
        b_d = [False] * self.n_vars
        for A, B, _ in self.constraints:
            for v, _c in A: a_d[v] = True
            for v, _c in B: b_d[v] = True
        return a_d, b_d

    def evaluate(self, z):
        az, bz, cz = [], [], []
        for A, B, C in self.constraints:
            az.append(sum(c * z[v] for v, c in A) % R_MOD)
            bz.append(sum(c * z[v] for v, c in B) % R_MOD)
            cz.append(sum(c * z[v] for v, c in C) % R_MOD)
        return az, bz, cz

    def is_satisfied(self, z):
        az, bz, cz = self.evaluate(z)
        return all((a * b - c) % R_MOD == 0 for a, b, c in zip(az, bz, cz))


def groth16_setup(r1cs: R1CS, tau, alpha, beta, gamma, delta):
    """Returns params dict with vk + h, l, a, b_g1, b_g2 (a/b filtered by LC-appearance density;
    inputs of `a` always present thanks to the input_i*0=0 constraints)."""
    lg = r1cs.log_m()
    m = 1 << lg
    # Lagrange basis at tau: L_j(tau) via inverse NTT of powers of tau
    pw = [pow(tau, i, R_MOD) for i in range(m)]
    # L_j(tau) = (1/m) sum_i tau^i w^{-ij}
    lag = ntt(pw, lg, inverse=True)
    at = [0] * r1cs.n_vars
    bt = [0] * r1cs.n_vars
    ct = [0] * r1cs.n_vars
    for k, (A, B, C) in enumerate(r1cs.constraints):
        for v, c in A: at[v] = (at[v] + c * lag[k]) % R_MOD
        for v, c in B: bt[v] = (bt[v] + c * lag[k]) % R_MOD
        for v, c in C: ct[v] = (ct[v] + c * lag[k]) % R_MOD
    zt = (pow(tau, m, R_MOD) - 1) % R_MOD
    dinv, ginv = inv_mod(delta, R_MOD), inv_mod(gamma, R_MOD)
    a_d, b_d = r1cs.densities()
    g1, g2 = G1_GEN, G2_GEN
    params = {
        "n_in": r1cs.n_in, "n_aux": r1cs.n_aux, "log_m": lg,
        "alpha_g1": g1_mul(g1, alpha), "beta_g1": g1_mul(g1, beta), "beta_g2": g2_mul(g2, beta),
        "gamma_g2": g2_mul(g2, gamma), "delta_g1": g1_mul(g1, delta), "delta_g2": g2_mul(g2, delta),
        "ic": [g1_mul(g1, (beta * at[v] + alpha * bt[v] + ct[v]) * ginv % R_MOD) for v in range(r1cs.n_in)],
        "h": [g1_mul(g1, pow(tau, i, R_MOD) * zt * dinv % R_MOD) for i in range(m - 1)],
        "l": [g1_mul(g1, (beta * at[v] + alpha * bt[v] + ct[v]) * dinv % R_MOD)
              for v in range(r1cs.n_in, r1cs.n_vars)],
        "a": [g1_mul(g1, at[v]) for v in range(r1cs.n_vars) if a_d[v]],
        "b_g1": [g1_mul(g1, bt[v]) for v in range(r1cs.n_vars) if b_d[v]],
        "b_g2": [g2_mul(g2, bt[v]) for v in range(r1cs.n_vars) if b_d[v]],
        "a_density": a_d, "b_density": b_d,
    }
    return params


def groth16_h_coeffs(az, bz, cz, lg):
    """h = (A*B - C)/Z via 3 iNTT + 3 coset NTT + pointwise + 1 inverse coset NTT; returns m-1 coeffs."""
    m = 1 << lg
    pad = lambda v: list(v) + [0] * (m - len(v))
    a = ntt(ntt(pad(az), lg, inverse=True), lg, coset=True)
    b = ntt(ntt(pad(bz), lg, inverse=True), lg, coset=True)
    c = ntt(ntt(pad(cz), lg, inverse=True), lg, coset=True)
    zinv = inv_mod((pow(FR_GENERATOR, m, R_MOD) - 1) % R_MOD, R_MOD)
    h = [((x * y - w) % R_MOD) * zinv % R_MOD for x, y, w in zip(a, b, c)]
    h = ntt(h, lg, inverse=True, coset=True)
    # (for a satisfying witness h[m-1] == 0; bellman does not check, neither do we)
    return h[: m - 1]


def groth16_prove(r1cs: R1CS, params, z, r, s):
    az, bz, cz = r1cs.evaluate(z)
    h = groth16_h_coeffs(az, bz, cz, params["log_m"])
    a_d, b_d = params["a_density"], params["b_density"]
    H = ec_msm(FP, params["h"], h)
    L = ec_msm(FP, params["l"], z[r1cs.n_in:])
    za = [z[v] for v in range(r1cs.n_vars) if a_d[v]]
    zb = [z[v] for v in range(r1cs.n_vars) if b_d[v]]
    A_ = ec_msm(FP, params["a"], za)
    B1 = ec_msm(FP, params["b_g1"], zb)
    B2 = ec_msm(FP2, params["b_g2"], zb)
    g_a = g1_add(g1_add(g1_mul(params["delta_g1"], r), params["alpha_g1"]), A_)
    g_b = g2_add(g2_add(g2_mul(params["delta_g2"], s), params["beta_g2"]), B2)
    g_c = g1_mul(params["delta_g1"], r * s % R_MOD)
    g_c = g1_add(g_c, g1_mul(params["alpha_g1"], s))
    g_c = g1_add(g_c, g1_mul(params["beta_g1"], r))
    g_c = g1_add(g_c, g1_mul(A_, s))
    g_c = g1_add(g_c, g1_mul(B1, r))
    g_c = g1_add(g_c, H)
    g_c = g1_add(g_c, L)
    return g_a, g_b, g_c


def proof_to_bytes(proof) -> bytes:
    return g1_to_bytes(proof[0]) + g2_to_bytes(proof[1]) + g1_to_bytes(proof[2])


def proof_from_bytes(b: bytes):
    assert len(b) == 387
    return g1_from_bytes(b[0:97]), g2_from_bytes(b[97:290]), g1_from_bytes(b[290:387])


def groth16_verify(vk, public_inputs, proof) -> bool:
    """e(A,B) * e(sum x_i IC_i, -gamma) * e(C, -delta) == e(alpha, beta); x_0 = 1."""
    a, b, c = proof
    if not (g1_on_curve(a) and g2_on_curve(b) and g1_on_curve(c)):
        return False
    xs = [1] + [x % R_MOD for x in public_inputs]
    if len(xs) != len(vk["ic"]):
        return False
    acc = ec_msm(FP, vk["ic"], xs)
    f = miller_loop(a, b)
    f = f12_mul(f, miller_loop(acc, ec_neg(FP2, vk["gamma_g2"])))
    f = f12_mul(f, miller_loop(c, ec_neg(FP2, vk["delta_g2"])))
    f = f12_mul(f, f12_inv(miller_loop(vk["alpha_g1"], vk["beta_g2"])))
    return final_exp(f) == F12_ONE


def vk_from_bytes(b: bytes):
    """Groth16VerifyingKey bincode layout (src/zk/groth16/mod.rs:22-31)."""
    o = 0
    def take(n):
        nonlocal o
        v = b[o:o + n]
        o += n
        return v
    vk = {}
    vk["alpha_g1"] = g1_from_bytes(take(97))
    vk["beta_g1"] = g1_from_bytes(take(97))
    vk["beta_g2"] = g2_from_bytes(take(193))
    vk["gamma_g2"] = g2_from_bytes(take(193))
    vk["delta_g1"] = g1_from_bytes(take(97))
    vk["delta_g2"] = g2_from_bytes(take(193))
    n = int.from_bytes(take(8), "little")
    vk["ic"] = [g1_from_bytes(take(97)) for _ in range(n)]
    assert o == len(b)
    return vk


def vk_to_bytes(vk) -> bytes:
    out = g1_to_bytes(vk["alpha_g1"]) + g1_to_bytes(vk["beta_g1"]) + g2_to_bytes(vk["beta_g2"])
    out += g2_to_bytes(vk["gamma_g2"]) + g1_to_bytes(vk["delta_g1"]) + g2_to_bytes(vk["delta_g2"])
    out += len(vk["ic"]).to_bytes(8, "little")
    for p in vk["ic"]:
        out += g1_to_bytes(p)
    return out


# --------------------------------------------------------------------------------------------------
# deterministic RNG shared by python + C (SplitMix64), seed "BAZUKA"
# --------------------------------------------------------------------------------------------------
SEED = 0x42415A554B41


class SplitMix64:
    def __init__(self, seed: int = SEED):
        self.s = seed & (2**64 - 1)

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)

    def fr(self) -> int:
        """uniform in [0, r) by rejection on 255 bits (limb 0 drawn first)."""
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << 255) - 1
            if v < R_MOD:
                return v


# --------------------------------------------------------------------------------------------------
# Jubjub + EdDSA (src/crypto/jubjub/curve.rs:19-164, mod.rs:112-167) - independent check of the host code
# --------------------------------------------------------------------------------------------------
JJ_D = 19257038036680949359750312669786877991949435402254120286184196891950884077233
JJ_BASE = (28867639725710769449342053336011988556061781325688749245863888315629457631946, 18)
JJ_ORDER = 6554484396890773809930967563523245729705921265872317281365359162392183254199


def jj_on_curve(p):
    x, y = p
    return (y * y - x * x - 1 - JJ_D * x * x % R_MOD * y * y) % R_MOD == 0


def jj_add(p, q):
    (x1, y1), (x2, y2) = p, q
    k = JJ_D * x1 * x2 * y1 * y2 % R_MOD
    x3 = (x1 * y2 + y1 * x2) * inv_mod((1 + k) % R_MOD, R_MOD) % R_MOD
    y3 = (y1 * y2 + x1 * x2) * inv_mod((1 - k) % R_MOD, R_MOD) % R_MOD
    return (x3, y3)


def jj_mul(p, k):
    r = (0, 1)
    for bit in bin(k)[2:] if k else "":
        r = jj_add(r, r)
        if bit == "1":
            r = jj_add(r, p)
    return r


def sha3_scalar(b: bytes) -> int:
    import hashlib
    return int.from_bytes(hashlib.sha3_256(b).digest(), "little") % R_MOD


def jj_generate_keys(seed: bytes):
    randomness = sha3_scalar(seed)
    scalar = sha3_scalar(randomness.to_bytes(32, "little"))
    return {"pub": jj_mul(JJ_BASE, scalar), "randomness": randomness, "scalar": scalar}


def jj_sign(sk, msg: int):
    r = poseidon([sk["randomness"], msg])
    rr = jj_mul(JJ_BASE, r)
    h = poseidon([rr[0], rr[1], sk["pub"][0], sk["pub"][1], msg])
    s = (r + h * sk["scalar"]) % JJ_ORDER
    return rr, s


def jj_verify(pub, msg: int, sig) -> bool:
    rr, s = sig
    if not (jj_on_curve(pub) and jj_on_curve(rr)):
        return False
    h = poseidon([rr[0], rr[1], pub[0], pub[1], msg])
    return jj_add(jj_mul(pub, h), rr) == jj_mul(JJ_BASE, s)


# --------------------------------------------------------------------------------------------------
# bellman `groth16::Parameters<Bls12>::write` [recalled: bellman 0.14 groth16/mod.rs, bls12_381 0.8 `to_uncompressed`] - the
# oracle's WRITER of the proving-key file format, for round-trip tests of the product's reader (bzk_bellman_params_*).
# Points: 48-byte big-endian canonical coordinates; G2 = x.c1 | x.c0 | y.c1 | y.c0; infinity = 0x40 then zeros.
# --------------------------------------------------------------------------------------------------
def g1_uncompressed(p) -> bytes:
    if p is None:
        return b"\x40" + bytes(95)
    return p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")


def g2_uncompressed(p) -> bytes:
    if p is None:
        return b"\x40" + bytes(191)
    (x0, x1), (y0, y1) = p
    return x1.to_bytes(48, "big") + x0.to_bytes(48, "big") + y1.to_bytes(48, "big") + y0.to_bytes(48, "big")


def bellman_params_bytes(vk, h, l, a, b_g1, b_g2) -> bytes:
    """vk: dict(alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2, ic) of affine points (ints; None = infinity);
    h, l, a, b_g1: lists of G1 points; b_g2: list of G2 points."""
    out = [g1_uncompressed(vk["alpha_g1"]), g1_uncompressed(vk["beta_g1"]), g2_uncompressed(vk["beta_g2"]), g2_uncompressed(vk["gamma_g2"]),
           g1_uncompressed(vk["delta_g1"]), g2_uncompressed(vk["delta_g2"]), len(vk["ic"]).to_bytes(4, "big")]
    out += [g1_uncompressed(p) for p in vk["ic"]]
    for arr, enc in ((h, g1_uncompressed), (l, g1_uncompressed), (a, g1_uncompressed), (b_g1, g1_uncompressed), (b_g2, g2_uncompressed)):
        out.append(len(arr).to_bytes(4, "big"))
        out += [enc(p) for p in arr]
    return b"".join(out)
