"""What the compiler made of the hot kernels, read from the gfx950 code objects of the built library (no GPU needed).

The measured numbers in DESIGN.md rest on properties of the code objects that a compiler update, a flag or an innocent-looking edit can
silently take away: occupancy steps (a few registers too many halve the waves per SIMD), scratch (the rolled MDS loops of round 4 cost
2.2 GB of HBM writes per tree), and - since run 30 - the inlined products of the G1 accumulation.  This pins them; the values asserted are
the ones of profiles/r04_kernel_resources.txt with a little slack where slack is harmless."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.isdir(kr.OBJ) or not os.path.exists(os.path.join(kr.OBJ, "msm_g1.o")),
                                reason="bazuka_amd/csrc/_obj not built (build() compiles it)")


@pytest.fixture(scope="module")
def table():
    return {(r["object"], r["kernel"]): r for r in kr.resources()}


def row(table, obj, kernel):
    assert (obj, kernel) in table, f"{kernel} not found in {obj}: {[k for o, k in table if o == obj]}"
    return table[(obj, kernel)]


def test_g1_accumulate_keeps_two_waves_and_no_scratch(table):
    r = row(table, "msm_g1", "msm_accumulate_kernel<G1Fast, 2>")
    assert r["vgpr"] <= 256 and r["waves"] >= 2
    # round 6 run 24: the next base is staged through LDS by direct loads - 7 slabs of base pieces + 1 of index words per wave, two waves per workgroup;
    # four workgroups per CU (two waves per SIMD) take 64 of its 160 KiB
    assert r["spill"] == 0 and r["scratch"] == 0 and r["lds"] == 16384


def test_g1_accumulate_has_its_products_inlined():
    c = kr.instruction_counts("msm_g1", "msm_accumulate_kernel")
    mads, calls = c.get("v_mad_u64_u32", 0), c.get("s_swappc_b64", 0)
    # 6 products x 392 + 2 squares x 301 + the fused Y 589 = 3 543 multiply-adds in the loop body itself; with the products as calls the
    # kernel body holds ~600.  The calls that remain belong to the rare doubling / cancellation path (dbl_affine).
    assert mads >= 3500, c
    assert calls <= 10, c
    # the call ABI's argument / result moves are what the inlining removed (741 v_mov in the kernel before, ~450 outside the loop now)
    assert c.get("v_mov_b32", 0) + c.get("v_mov_b32_e32", 0) <= 520, c


def test_g2_accumulate_does_not_spill(table):
    r = row(table, "msm_g2", "msm_accumulate_kernel<G2Fast, 1>")
    assert r["spill"] == 0 and r["scratch"] == 0


@pytest.mark.parametrize("kernel,max_vgpr", [("tree4_rehash_kernel", 168), ("poseidon29_kernel<5>", 168), ("poseidon29_kernel<4>", 168),
                                             ("poseidon29_kernel<2>", 96), ("poseidon29_kernel<8>", 256)])
def test_poseidon_occupancy_steps(table, kernel, max_vgpr):
    r = row(table, "poseidon", kernel)
    assert r["vgpr"] <= max_vgpr, r
    assert r["scratch"] == 0, r  # the MDS rows stay in registers (BZK_POSEIDON_ROWS_STRAIGHT)


def test_poseidon_partial_rounds_are_inlined_for_the_tree_width():
    c = kr.instruction_counts("poseidon", "tree4_rehash_kernel")
    # full rounds keep their 15 S-box calls (two loops) + conversions; the 7 products of a partial round and the periodic renormalisation are inline
    assert c.get("s_swappc_b64", 0) <= 45, c
    assert c.get("v_mad_u64_u32", 0) >= 8000, c


def test_ntt_pass_keeps_four_waves(table):
    r = row(table, "ntt", "ntt_pass_kernel<4, 1, 1>")
    assert r["vgpr"] <= 128 and r["waves"] >= 4


def test_no_device_function_destroys_its_return_address():
    # the hang of the first GPU run of the pair-lane G2 tails (round 5, run 4): `s_getpc_b64 s[30:31]` as the scratch pair of a long branch
    # inside a called function (kernels have no return address; there the pair is free)
    assert kr.functions_clobbering_return_address() == []


def test_g2_pair_tails_do_not_spill(table):
    # the one-lane tails ran at 512 registers with 94 / 33 / 31 spilled ones
    for k in ("msm_fold_g2pair_kernel<0>", "msm_fold_small_g2pair_kernel<0>", "msm_reduce_g2pair_kernel<0>", "msm_window_partial_g2pair_kernel<128>",
              "msm_window_sum_g2pair_kernel<128>"):
        assert row(table, "msm_g2", k)["spill"] <= 2, k


@pytest.mark.parametrize("obj,kernel", [("msm_g1", "msm_accumulate_kernel"), ("msm_g2", "msm_accumulate_g2pair")])
def test_no_value_is_parked_in_scratch_per_addition(obj, kernel):
    """the G2 pair kernel of round 5 parked the prefetched base - 28 registers - in scratch on EVERY addition (8 scratch stores
    behind an s_waitcnt vmcnt(0) inside the loop: 3.77 GB of writes per 2^20-point launch).  One pass of the widest loop of either accumulation
    kernel = one point addition: no scratch store may be in it, and the few reloads of loop-invariant values stay few."""
    c = kr.loop_instruction_counts(obj, kernel)
    assert c.get("v_mad_u64_u32", 0) >= 3500, c           # it IS the addition loop
    stores = sum(v for k, v in c.items() if k.startswith("scratch_store"))
    loads = sum(v for k, v in c.items() if k.startswith("scratch_load"))
    assert stores == 0, {k: v for k, v in c.items() if k.startswith("scratch")}
    assert loads <= 12, {k: v for k, v in c.items() if k.startswith("scratch")}


def test_g2_pair_accumulation_stages_the_next_base_through_lds(table):
    """round 6: the next base (and the index word after next) travel global memory -> LDS by direct loads while the current addition runs"""
    c = kr.loop_instruction_counts("msm_g2", "msm_accumulate_g2pair")
    assert c.get("global_load_lds_dwordx4", 0) == 8 and c.get("global_load_lds_dword", 0) == 1, c
    assert sum(v for k, v in c.items() if k.startswith("global_load_dword")) == 0, c      # nothing of the gather lands in registers
    r = row(table, "msm_g2", "msm_accumulate_g2pair_kernel<2>")
    assert r["waves"] >= 2 and r["lds"] <= 20 * 1024 and r["scratch"] <= 64, r


def test_g1_accumulation_stages_the_next_base_through_lds():
    """round 6 run 24: the G1 accumulation requests the next base (7 x 16 B) and the index word after next as direct loads to LDS at the top of the
    addition - nothing of the gather is held in registers across the products"""
    c = kr.loop_instruction_counts("msm_g1", "msm_accumulate_kernel")
    assert c.get("global_load_lds_dwordx4", 0) == 7 and c.get("global_load_lds_dword", 0) == 1, c
    assert sum(v for k, v in c.items() if k.startswith("global_load_dword")) == 0, c


def test_witness_fill_kernels_fit_beside_other_work(table):
    """round 5's one-launch fill (wf_tx_kernel) is a 512-register body with 6.8 KB of scratch per lane that needs a drained CU.  The
    cooperative form of round 6 (eight lanes per hash): no scratch, no body that takes a whole SIMD's register file."""
    for k in ("wf_pass1_coop_kernel", "wf_trace_all_kernel"):
        r = row(table, "witfill", k)
        assert r["scratch"] == 0 and r["spill"] == 0, r
        assert r["vgpr"] <= 304, r      # a SIMD that one accumulation wave has left (512 - 176) takes it
    assert row(table, "witfill", "wf_small_kernel")["vgpr"] <= 64
