"""CPU model of the two-level fold's index maps (msm_impl.cuh msm_fold_wide_kernel / msm_fold_wide_slot, msm_g2pair_tails.cuh twin): the chunk sums of the
giant buckets land in disjoint ranges of `wide` that fit its capacity, and the round-robin deal hands every chunk of every giant to exactly one workgroup.
The kernels themselves are pinned on the oracle by tests/test_gpu_msm.py::test_msm_giant_buckets_two_level_fold_vs_oracle."""
import random

WIDE_FROM, WIDE_POS, GRID = 640, 64, 512   # MSM_FOLD_WIDE_FROM / _POS / _GRID


def plan(ntasks):
    """ntasks[i] = tasks of the bucket at sorted position i -> (tbase, giants [(i, slot, chunks)])"""
    tbase, t = [], 0
    for nt in ntasks:
        tbase.append(t)
        t += nt
    giants = [(i, tbase[i] // 64 + i, (nt + 63) // 64) for i, nt in enumerate(ntasks[:WIDE_POS]) if nt > WIDE_FROM]
    return tbase, t, giants


def deal(giants, grid):
    """which workgroup folds chunk ch of giant g: the kernel's loop, workgroup by workgroup"""
    owner = {}
    for b in range(grid):
        before = 0
        for i, _, chunks in giants:
            first = (b + grid - before % grid) % grid
            before += chunks
            for ch in range(first, chunks, grid):
                assert (i, ch) not in owner, "a chunk dealt twice"
                owner[(i, ch)] = b
    return owner


def test_slots_are_disjoint_and_fit_the_buffer():
    rng = random.Random(5)
    for trial in range(300):
        nb = rng.randint(1, 400)
        ntasks = [1] * nb
        for _ in range(rng.randint(0, 70)):            # giants and near-giants anywhere among the first positions (the order ties beyond the key's clamp)
            ntasks[rng.randrange(min(nb, 90))] = rng.choice([639, 640, 641, 1000, 5700, rng.randint(2, 20000)])
        tbase, total, giants = plan(ntasks)
        t_cap = total + rng.randint(0, 1000)            # capacity of `partial` >= the tasks that exist
        wide_cap = t_cap // 32 + 2 * WIDE_POS + 2       # msm_run's allocation
        used = set()
        for i, slot, chunks in giants:
            rng_ = range(slot, slot + chunks)
            assert rng_[-1] < wide_cap, (trial, i, slot, chunks, wide_cap)
            assert not used.intersection(rng_), (trial, i)
            used.update(rng_)


def test_every_chunk_has_exactly_one_workgroup_and_the_deal_is_even():
    rng = random.Random(6)
    for trial in range(40):
        ntasks = [rng.choice([1, 2, 700, 1024, 5700, 20000]) for _ in range(rng.randint(1, 80))]
        _, _, giants = plan(ntasks)
        for grid in (GRID, 7, 1):
            owner = deal(giants, grid)
            want = {(i, ch) for i, _, chunks in giants for ch in range(chunks)}
            assert set(owner) == want, (trial, grid)
            if giants:
                load = [0] * grid
                for b in owner.values():
                    load[b] += 1
                assert max(load) - min(load) <= 1, (trial, grid, max(load), min(load))   # round-robin over ALL giants: eight giants of 16 chunks are one round, not eight
