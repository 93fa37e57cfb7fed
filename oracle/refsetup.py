"""TEST INFRASTRUCTURE - an attempt to reproduce the toxic waste behind the reference's hard-coded verifying keys.

/root/reference/src/config/blockchain.rs:32-37 holds three 1460-byte Groth16 verifying keys whose first 870 bytes
(alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2) are equal - what a FIXED-SEED setup gives when every circuit
is set up with a fresh rng of the same seed.  The only seeded setup in the tree is the dev configuration,
`ChaChaRng::from_seed([0u8; 32])` feeding `bellman::groth16::generate_random_parameters` (:369-399).  This module restates
that path from the published crates [recalled - none of them is under /root/reference]:

  rand_chacha 0.3  ChaChaRng = ChaCha20, 64-bit block counter from 0, stream 0; a stream of little-endian u32 words
                   (rand_core BlockRng: fill_bytes consumes whole words when the lengths are multiples of four, as all here)
  bls12_381 0.8    Fp::random      96 bytes -> 12 big-endian u64 -> from_u768 -> value mod p
                   Fp2::random     c0 then c1
                   G1Projective::random   loop { x = Fp::random; flip = next_u32 % 2 != 0; y = sqrt(x^3 + 4) = (.)^((p+1)/4);
                                          if flip y = -y; p = clear_cofactor = P - [x]P = (1 + |x|) P; until p != identity }
                   G2Projective::random   same over Fp2 with Algorithm 9 of eprint 2012/685 as sqrt and the psi-based
                                          cofactor clearing (= multiplication by h_eff)
                   Scalar::random  64 bytes -> from_bytes_wide -> little-endian 512-bit value mod r
  bellman 0.14     generate_random_parameters draws g1, g2, alpha, beta, gamma, delta, tau in that order; the key holds
                   alpha*g1, beta*g1, beta*g2, gamma*g2, delta*g1, delta*g2 and ic_i = ((beta*A_i + alpha*B_i + C_i)(tau) / gamma) * g1

Only tests/ import this.  tests/test_reference_vk_cpu.py records what the attempt gives."""
import struct

from oracle import pyref as pr

P, R = pr.P_MOD, pr.R_MOD


class ChaCha20Rng:
    """rand_chacha::ChaCha20Rng (djb variant: 64-bit counter in words 12-13, 64-bit stream id in words 14-15)"""

    def __init__(self, seed: bytes, rounds: int = 20):
        assert len(seed) == 32
        self.key = struct.unpack("<8I", seed)
        self.counter = 0
        self.rounds = rounds
        self.buf = []

    def _block(self, ctr):
        c = (0x61707865, 0x3320646E, 0x79622D32, 0x6B206574)
        st = list(c + self.key + (ctr & 0xFFFFFFFF, ctr >> 32, 0, 0))
        x = st[:]

        def qr(a, b, c_, d):
            x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] ^= x[a]; x[d] = ((x[d] << 16) | (x[d] >> 16)) & 0xFFFFFFFF
            x[c_] = (x[c_] + x[d]) & 0xFFFFFFFF; x[b] ^= x[c_]; x[b] = ((x[b] << 12) | (x[b] >> 20)) & 0xFFFFFFFF
            x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] ^= x[a]; x[d] = ((x[d] << 8) | (x[d] >> 24)) & 0xFFFFFFFF
            x[c_] = (x[c_] + x[d]) & 0xFFFFFFFF; x[b] ^= x[c_]; x[b] = ((x[b] << 7) | (x[b] >> 25)) & 0xFFFFFFFF

        for _ in range(self.rounds // 2):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        return [(a + b) & 0xFFFFFFFF for a, b in zip(x, st)]

    def next_u32(self) -> int:
        if not self.buf:
            self.buf = self._block(self.counter)
            self.counter += 1
        return self.buf.pop(0)

    def fill_bytes(self, n: int) -> bytes:
        assert n % 4 == 0
        return b"".join(struct.pack("<I", self.next_u32()) for _ in range(n // 4))


def fp_random(rng, variant=0) -> int:
    b = rng.fill_bytes(96)
    limbs = [int.from_bytes(b[8 * i:8 * i + 8], "big") for i in range(12)]
    d1 = sum(l << (64 * k) for k, l in enumerate([limbs[11], limbs[10], limbs[9], limbs[8], limbs[7], limbs[6]]))
    d0 = sum(l << (64 * k) for k, l in enumerate([limbs[5], limbs[4], limbs[3], limbs[2], limbs[1], limbs[0]]))
    if variant == 0:   # from_u768 as published: d0 * R2 + d1 * R3 in Montgomery arithmetic = d0 + d1 * 2^384
        return (d0 + (d1 << 384)) % P
    return (d1 + (d0 << 384)) % P  # the plain big-endian reading of the 96 bytes


def scalar_random(rng) -> int:
    return int.from_bytes(rng.fill_bytes(64), "little") % R


def fp_sqrt(a):
    y = pow(a, (P + 1) // 4, P)
    return y if y * y % P == a % P else None


def f2_pow(a, e):
    out = pr.F2_ONE
    while e:
        if e & 1:
            out = pr.f2_mul(out, a)
        a = pr.f2_sqr(a)
        e >>= 1
    return out


def fp2_sqrt(a):
    """Algorithm 9 of eprint 2012/685 as bls12_381 0.8 fp2.rs has it"""
    if a == (0, 0):
        return (0, 0)
    a1 = f2_pow(a, (P - 3) // 4)
    alpha = pr.f2_mul(pr.f2_sqr(a1), a)
    x0 = pr.f2_mul(a1, a)
    if alpha == ((P - 1) % P, 0):
        s = ((-x0[1]) % P, x0[0])
    else:
        s = pr.f2_mul(f2_pow(pr.f2_add(alpha, pr.F2_ONE), (P - 1) // 2), x0)
    return s if pr.f2_sqr(s) == (a[0] % P, a[1] % P) else None


G2_H_EFF = 0xBC69F08F2EE75B3584C6A0EA91B352888E2A8E9145AD7689986FF031508FFE1329C2F178731DB956D82BF015D1212B02EC0EC69D7477C1AE954CBC06689F6A359894C0ADEBBF6B4E8020005AAA95551


def g1_random(rng, variant=0):
    while True:
        x = fp_random(rng, variant)
        flip = rng.next_u32() % 2 != 0
        y = fp_sqrt((x * x * x + 4) % P)
        if y is None:
            continue
        if flip:
            y = (-y) % P
        p = pr.g1_mul((x, y), 1 + pr.BLS_X)     # clear_cofactor: P - [x]P with x = -|x|
        if p is not None:
            return p


def g2_random(rng, variant=0):
    while True:
        x = (fp_random(rng, variant), fp_random(rng, variant))
        flip = rng.next_u32() % 2 != 0
        y = fp2_sqrt(pr.f2_add(pr.f2_mul(pr.f2_sqr(x), x), pr.G2_B))
        if y is None:
            continue
        if flip:
            y = pr.f2_neg(y)
        p = pr.g2_mul((x, y), G2_H_EFF)
        if p is not None:
            return p


def draw_setup(seed: bytes, variant=0, rounds=20):
    """-> (g1, g2, alpha, beta, gamma, delta, tau) as bellman's generate_random_parameters draws them"""
    rng = ChaCha20Rng(seed, rounds)
    g1 = g1_random(rng, variant)
    g2 = g2_random(rng, variant)
    alpha, beta, gamma, delta, tau = (scalar_random(rng) for _ in range(5))
    return g1, g2, alpha, beta, gamma, delta, tau


def vk_prefix(setup) -> bytes:
    """the 870 bytes the three hard-coded keys share: alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2"""
    g1, g2, alpha, beta, gamma, delta, _ = setup
    return (pr.g1_to_bytes(pr.g1_mul(g1, alpha)) + pr.g1_to_bytes(pr.g1_mul(g1, beta)) + pr.g2_to_bytes(pr.g2_mul(g2, beta))
            + pr.g2_to_bytes(pr.g2_mul(g2, gamma)) + pr.g1_to_bytes(pr.g1_mul(g1, delta)) + pr.g2_to_bytes(pr.g2_mul(g2, delta)))
