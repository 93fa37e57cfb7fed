"""LDS bank-conflict model of ntt_pass_kernel's access patterns (ds_*_b32: two lane groups of 32, bank = dword address mod 32),
with and without the element-index swizzle of ntt.hip (lds_sw).  Prints the worst conflict degree per phase for the tile shapes
the planner produces.  Pure Python, no GPU.   usage: python tools/ntt_bank_sim.py"""


def sw(idx):
    h = idx >> 5
    x = (h ^ ((h & 3) << 2) ^ ((h & 2) << 3)) & 31
    return idx ^ x


def bitrev(v, b):
    return int(format(v, "0%db" % b)[::-1], 2) if b else 0


def conflicts(idxs):
    worst = 0
    for g in (idxs[:32], idxs[32:]):
        banks = {}
        for i in g:
            banks.setdefault((9 * i) % 32, set()).add(i)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def report(b, lc, f):
    R, CC = 1 << b, 1 << lc
    tile_n = R << lc
    res = {}
    res["load_col"] = max(conflicts([f((bitrev(t >> lc, b) << lc) + (t & (CC - 1))) for t in range(base, base + 64)]) for base in range(0, tile_n, 64))
    res["load_final"] = max(conflicts([f((bitrev(t & (R - 1), b) << lc) + (t >> b)) for t in range(base, base + 64)]) for base in range(0, tile_n, 64))
    if b & 1:
        w = 0
        for base in range(0, tile_n // 2, 64):
            for e in range(2):
                w = max(w, conflicts([f(((((q >> lc) << 1) + e) << lc) + (q & (CC - 1))) for q in range(base, base + 64)]))
        res["lone_s0"] = w
    for s in range(b & 1, b, 2):
        m, w = 1 << s, 0
        for base in range(0, tile_n // 4, 64):
            for e in range(4):
                idxs = []
                for q in range(base, base + 64):
                    c, bq = q & (CC - 1), q >> lc
                    k0 = ((bq >> s) << (s + 2)) + (bq & (m - 1))
                    idxs.append(f(((k0 + e * m) << lc) + c))
                w = max(w, conflicts(idxs))
        res["s%d" % s] = w
    res["store"] = max(conflicts([f(t) for t in range(base, base + 64)]) for base in range(0, tile_n, 64))
    return res


if __name__ == "__main__":
    for b, lc in ((10, 0), (9, 1), (8, 2), (7, 3), (6, 4), (5, 5), (4, 6)):
        print(f"b={b} log_cc={lc} plain   ", report(b, lc, lambda x: x))
        print(f"b={b} log_cc={lc} swizzled", report(b, lc, sw))
