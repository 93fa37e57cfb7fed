"""A fixed, known number of operations for the PMC traffic passes of the kernels other than msm_accumulate<G1> (tools/pmc_kernels.py):
3 forward NTTs of 2^24 points, 2 re-hashes of a 2^24-leaf tree, 2 G2 MSMs of 2^20 points.  Nothing is timed here.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out_f -- python tools/pmc_ops.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out_w -- python tools/pmc_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bazuka_amd import Bzk

N_NTT, N_TREE, N_G2 = 3, 2, 2


def rand_fr(n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    t = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    t[:, 31] &= 0x3F
    return t.contiguous()


def main():
    ctx = Bzk(0)
    d = rand_fr(1 << 24, 1)
    torch.cuda.synchronize()
    for _ in range(N_NTT):
        ctx.ntt_dev(d, 24, False, False)
    for _ in range(N_TREE):
        ctx.merkle4_root_dev(d, 12)
    del d
    n = 1 << 20
    bases = torch.empty(n * 192, dtype=torch.uint8, device="cuda")
    ctx.g2_synth_bases_dev(1, 0, n, bases)
    sc = rand_fr(n, 2)
    torch.cuda.synchronize()
    for _ in range(N_G2):
        ctx.msm_g2_dev(bases, sc, n)
    ctx.close()


if __name__ == "__main__":
    main()
