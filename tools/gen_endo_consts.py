#!/usr/bin/env python3
"""Prints the constant block of bazuka_amd/csrc/bzk_endo.cuh (between the GENERATED markers): the curve parameter |x| of BLS12-381,
its square, their Barrett reciprocals, and the coordinate multipliers of the endomorphism images  X^m P  of G1 / G2 points, derived
NUMERICALLY from the generators with the oracle's Python curve arithmetic (oracle/pyref.py) and validated there on random points
(tests/test_endo_cpu.py re-derives and compares them).  usage: python tools/gen_endo_consts.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyref as pr

P, R, X = pr.P_MOD, pr.R_MOD, pr.BLS_X


def limbs32(v, n):
    return "{" + ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xffffffff) for i in range(n)) + "}"


def fp28(v):
    m = v * (1 << 392) % P
    return "{{" + ", ".join("0x%07xu" % ((m >> (28 * i)) & 0xfffffff) for i in range(14)) + "}}"


def conj(a, i):
    return a if i % 2 == 0 else (a[0], (-a[1]) % P)


def constants():
    X2 = X * X
    g = 2
    while pow(g, (P - 1) // 3, P) == 1:
        g += 1
    b = pow(g, (P - 1) // 3, P)
    G = pr.G1_GEN
    Q = pr.g1_mul(G, X2 % R)
    beta = [bb for bb in (b, b * b % P) if Q[0] == bb * G[0] % P][0]
    assert Q == (beta * G[0] % P, (-G[1]) % P)
    out = {"X": X, "X2": X2, "MU_X": (1 << 320) // X, "MU_X2": (1 << 384) // X2, "BETA": beta, "G2": []}
    for i in (1, 2, 3):
        Qi = pr.g2_mul(pr.G2_GEN, pow(X, i, R))
        cx = pr.f2_mul(Qi[0], pr.f2_inv(conj(pr.G2_GEN[0], i)))
        cy = pr.f2_mul(Qi[1], pr.f2_inv(conj(pr.G2_GEN[1], i)))
        out["G2"].append((cx, cy))
    return out


def table_fn(name, v, n, what):
    vals = ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xffffffff) for i in range(n))
    return ("BZK_HD constexpr uint32_t %s(int i) {  // %s\n    constexpr uint32_t t[%d] = {%s};\n    return t[i];\n}" % (name, what, n, vals))


def block():
    c = constants()
    L = []
    L.append(table_fn("x_l", c["X"], 2, "|x| = 0xd201000000010000 (the curve parameter is -|x|); r = X^4 - X^2 + 1"))
    L.append(table_fn("xh_l", c["X"] // 2, 2, "X / 2"))
    L.append(table_fn("mu_x", c["MU_X"], 9, "floor(2^320 / X)"))
    L.append(table_fn("x2_l", c["X2"], 4, "X^2"))
    L.append(table_fn("x2h_l", c["X2"] // 2, 4, "X^2 / 2"))
    L.append(table_fn("mu_x2", c["MU_X2"], 9, "floor(2^384 / X^2)"))
    L.append("// G1: X^2 (x, y) = (BETA x, -y); Montgomery-2^392 limbs of BETA and of -1")
    L.append("static constexpr fp28::Consts G1_BETA = %s;" % fp28(c["BETA"]))
    L.append("static constexpr fp28::Consts FP_MINUS1 = %s;" % fp28(P - 1))
    L.append("// G2: X^m (x, y) = (CX_m conj^m(x), CY_m conj^m(y)), m = 1, 2, 3; components c0, c1 of each multiplier")
    for i, (cx, cy) in zip((1, 2, 3), c["G2"]):
        for nm, val in (("CX", cx), ("CY", cy)):
            for k in (0, 1):
                L.append("static constexpr fp28::Consts G2_%s%d_%d = %s;" % (nm, i, k, fp28(val[k])))
    return "\n".join(L)


if __name__ == "__main__":
    print(block())
