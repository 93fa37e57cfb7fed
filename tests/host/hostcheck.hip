// TEST HARNESS (not product): runs the __host__ __device__ field / curve code of libbzk on the CPU
// so that limb-level logic is checked against the oracle in this GPU-less container.
#define BZK_FP28_CHECK 1
#include "../../bazuka_amd/csrc/bzk_fp28.cuh"
#define BZK_G2P_HOST_EMU 1  // the pair-lane G2 arithmetic as host functions: two threads per pair, the DPP exchange a rendezvous (below)
#include "../../bazuka_amd/csrc/bzk_g2pair.cuh"
#include "../../bazuka_amd/csrc/bzk_endo.cuh"
#include "../../bazuka_amd/csrc/bzk_poseidon29.cuh"
#include "../../bazuka_amd/csrc/bzk_poseidon_opt.h"
#include "../../bazuka_amd/csrc/host_fp64.h"
#include "../../bazuka_amd/csrc/host_fr64.h"
#include "../../bazuka_amd/csrc/host_fr_ifma.h"
#include "../../bazuka_amd/csrc/host_pairing.h"
#include <string.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
using namespace bzk;

template <class P> static Fe<P> ld(const uint8_t* p) { Fe<P> a; memcpy(a.l, p, 4 * P::N); return a; }
template <class P> static void st(uint8_t* p, const Fe<P>& a) { memcpy(p, a.l, 4 * P::N); }

template <class P> static int field_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    Fe<P> x = ld<P>(a), y = b ? ld<P>(b) : Fe<P>::zero(), r;
    switch (op) {
        case 0: r = fe_add<P>(x, y); break;
        case 1: r = fe_sub<P>(x, y); break;
        case 2: r = fe_mul<P>(x, y); break;
        case 3: r = fe_inv<P>(x); break;
        case 4: r = fe_from_mont<P>(x); break;
        case 5: r = fe_to_mont<P>(x); break;
        case 6: r = fe_neg<P>(x); break;
        default: return -1;
    }
    st<P>(out, r);
    return 0;
}

static G1Affine ld_g1(const uint8_t* p) { return {ld<FpParams>(p), ld<FpParams>(p + 48)}; }
static G2Affine ld_g2(const uint8_t* p) {
    return {{ld<FpParams>(p), ld<FpParams>(p + 48)}, {ld<FpParams>(p + 96), ld<FpParams>(p + 144)}};
}
static void st_g1(uint8_t* o, const G1Xyzz& p) {
    G1Affine a; bool ok = xyzz_to_affine<FpOps>(p, a);
    st<FpParams>(o, a.x); st<FpParams>(o + 48, a.y); o[96] = ok ? 0 : 1;
}
static void st_g2(uint8_t* o, const G2Xyzz& p) {
    G2Affine a; bool ok = xyzz_to_affine<Fp2Ops>(p, a);
    st<FpParams>(o, a.x.c0); st<FpParams>(o + 48, a.x.c1); st<FpParams>(o + 96, a.y.c0); st<FpParams>(o + 144, a.y.c1);
    o[192] = ok ? 0 : 1;
}


// ---- bzk_g2pair.cuh on the CPU: lanes 2 k / 2 k + 1 are two threads, swp32 (the DPP quad_perm move on the device) a rendezvous
namespace {
struct PairEmu {
    std::mutex m;
    std::condition_variable cv;
    uint32_t slot[2] = {0, 0}, res[2] = {0, 0};
    int arrived = 0, gen = 0;
};
thread_local PairEmu* g_emu = nullptr;
thread_local int g_lane = 0;
void run_pair(const std::function<void(int)>& body) {
    PairEmu emu;
    auto th = [&](int lane) {
        g_emu = &emu;
        g_lane = lane;
        body(lane);
    };
    std::thread a(th, 0), b(th, 1);
    a.join();
    b.join();
}
}  // namespace
namespace bzk { namespace g2p {
uint32_t bzk_g2p_host_swp32(uint32_t v) {
    PairEmu& e = *g_emu;
    std::unique_lock<std::mutex> lk(e.m);
    const int my_gen = e.gen;
    e.slot[g_lane] = v;
    if (++e.arrived == 2) {
        e.res[0] = e.slot[1];
        e.res[1] = e.slot[0];
        e.arrived = 0;
        ++e.gen;
        e.cv.notify_all();
    } else {
        e.cv.wait(lk, [&] { return e.gen != my_gen; });
    }
    return e.res[g_lane];
}
bool bzk_g2p_host_lane_odd() { return g_lane != 0; }
} }  // namespace bzk::g2p
static g2p::Aff pair_aff(const G2A28& a, int lane) { return lane ? g2p::Aff{a.x.c1, a.y.c1} : g2p::Aff{a.x.c0, a.y.c0}; }
static void pair_store(G2X28& dst, const g2p::Pt& p, int lane) {
    (lane ? dst.X.c1 : dst.X.c0) = p.X;
    (lane ? dst.Y.c1 : dst.Y.c0) = p.Y;
    (lane ? dst.ZZ.c1 : dst.ZZ.c0) = p.ZZ;
    (lane ? dst.ZZZ.c1 : dst.ZZZ.c0) = p.ZZZ;
}
static g2p::Pt pair_load(const G2X28& src, int lane) {
    return lane ? g2p::Pt{src.X.c1, src.Y.c1, src.ZZ.c1, src.ZZZ.c1} : g2p::Pt{src.X.c0, src.Y.c0, src.ZZ.c0, src.ZZZ.c0};
}
// stored-point discipline of bzk_g2pair.cuh: X normalised and < 12 p, Y normalised and < 3 p, ZZ / ZZZ product outputs
static int pair_invariants(const g2p::Pt& p) {
    for (int i = 0; i < 13; ++i)
        if ((p.X.l[i] >> 28) || (p.Y.l[i] >> 28) || (p.ZZ.l[i] >> 28) || (p.ZZZ.l[i] >> 28)) return -2;
    if (p.X.l[13] >= 12 * 0x1a012u || p.Y.l[13] >= 3 * 0x1a012u || p.ZZ.l[13] >= 2 * 0x1a012u || p.ZZZ.l[13] >= 2 * 0x1a012u) return -1;
    return 0;
}

extern "C" {
// radix-2^28 field (device header run on the CPU with bound assertions)
int hc_fp28_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
    Fp x = ld<FpParams>(a), y = ld<FpParams>(b);
    st<FpParams>(out, fp28::from28(fp28::mul(fp28::to28(x), fp28::to28(y))));
    return 0;
}
// the dedicated square against the product of a value with itself - also on weakly reduced inputs (sums of up to
// 2^grow terms, i.e. limbs up to ~2^(28 + grow)): returns 0 iff every output LIMB is identical
int hc_fp28_sqr_equals_mul(const uint8_t* a, int grow, uint8_t* out) {
    Fp28 x = fp28::to28(ld<FpParams>(a));
    for (int i = 0; i < grow; ++i) x = fp28::add(x, x);
    const Fp28 s = fp28::sqr_body(x), m = fp28::mul_body(x, x);
    int diff = 0;
    for (int i = 0; i < fp28::N; ++i) diff |= s.l[i] != m.l[i];
    st<FpParams>(out, fp28::from28(s));
    return diff;
}
// binary-GCD inversion against the Fermat inversion, on a weakly reduced input (2^grow a); 0 iff the canonical results agree.
// *iters_needed (optional): outer iterations after which gcd_inv_plain's `a` reached 0 (re-run with a counting copy)
int hc_fp28_inv_gcd(const uint8_t* a, int grow, uint8_t* out) {
    Fp28 x = fp28::to28(ld<FpParams>(a));
    for (int i = 0; i < grow; ++i) x = fp28::add(x, x);
    const Fp r1 = fp28::from28(fp28::inv_gcd(x)), r2 = fp28::from28(fp28::inv_fermat(x));
    st<FpParams>(out, r1);
    return r1.equals(r2) ? 0 : 1;
}
int hc_fp28_roundtrip(const uint8_t* a, uint8_t* out) {
    st<FpParams>(out, fp28::from28(fp28::to28(ld<FpParams>(a))));
    return 0;
}
// sum_i (+/-) k_i * P_i through the XYZZ/Fp28 ops: mixed add (with negation), full add, doubling
int hc_g1x28_lincomb(const uint8_t* pts, const uint32_t* k, const uint8_t* neg, int n, uint8_t* out97) {
    G1X28 acc = g1x28::identity();
    for (int i = 0; i < n; ++i) {
        G1A28 a = g1x28::affine_to28(ld_g1(pts + 96 * i));
        G1X28 t = g1x28::identity();
        g1x28::add_mixed(t, a, neg[i] != 0);
        G1X28 m = g1x28::mul_u32(t, k[i]);
        g1x28::add_full(acc, m);
    }
    st_g1(out97, g1x28::to_std(acc));
    return 0;
}
// acc = sum of (+/-) points via repeated mixed add; long chains exercise the weak-reduction bounds
int hc_g1x28_sum_mixed(const uint8_t* pts, const uint8_t* neg, int n, uint8_t* out97) {
    G1X28 acc = g1x28::identity();
    for (int i = 0; i < n; ++i) g1x28::add_mixed(acc, g1x28::affine_to28(ld_g1(pts + 96 * i)), neg[i] != 0);
    st_g1(out97, g1x28::to_std(acc));
    return 0;
}
int hc_fp28_reduce_chain(const uint8_t* a, const uint8_t* b, int rounds, uint8_t* out) {
    // exercises add / sub<K> / reduce on growing values: r = ((a + b) * 2 - b + ...) tracked against the oracle by the caller
    Fp28 x = fp28::to28(ld<FpParams>(a)), y = fp28::to28(ld<FpParams>(b));
    for (int i = 0; i < rounds; ++i) {
        Fp28 t = fp28::add(fp28::add(x, y), fp28::add(x, x));   // 3x + y, unreduced
        t = fp28::add(t, t);                                     // 6x + 2y
        x = fp28::reduce(fp28::sub<6>(t, y));                    // 6x + y
        if (x.l[13] >> 19) return -1;                           // < 3p  => top limb < 2^19
    }
    st<FpParams>(out, fp28::from28(x));
    return 0;
}
int hc_g2x28_lincomb(const uint8_t* pts, const uint32_t* k, const uint8_t* neg, int n, uint8_t* out193) {
    G2X28 acc = xyzz_identity<Fp2x28Ops>();
    for (int i = 0; i < n; ++i) {
        G2A28 a = g2x28::affine_to28(ld_g2(pts + 192 * i));
        if (neg[i]) a.y = Fp2x28Ops::neg(a.y);
        G2X28 t = xyzz_identity<Fp2x28Ops>();
        xyzz_add_mixed<Fp2x28Ops>(t, a);
        G2X28 m = xyzz_mul_u32<Fp2x28Ops>(t, k[i]);
        xyzz_add<Fp2x28Ops>(acc, m);
    }
    st_g2(out193, g2x28::to_std(acc));
    return 0;
}
int hc_g2x28_sum_mixed(const uint8_t* pts, const uint8_t* neg, int n, uint8_t* out193) {
    G2X28 acc = xyzz_identity<Fp2x28Ops>();
    for (int i = 0; i < n; ++i) {
        G2A28 a = g2x28::affine_to28(ld_g2(pts + 192 * i));
        if (neg[i]) a.y = Fp2x28Ops::neg(a.y);
        xyzz_add_mixed<Fp2x28Ops>(acc, a);
    }
    st_g2(out193, g2x28::to_std(acc));
    return 0;
}
// the accumulation's own G2 mixed addition (g2x28::add_mixed: static bounds, Y under one reduction per component): a chain of
// (+/-) points, then - mode 1 - the chain's sum is handed to the GENERIC general addition (the tail kernels' consumer discipline)
int hc_g2x28_sum_mixed_fast(const uint8_t* pts, const uint8_t* neg, int n, int mode, uint8_t* out193) {
    G2X28 acc = xyzz_identity<Fp2x28Ops>();
    for (int i = 0; i < n; ++i) {
        G2A28 a = g2x28::affine_to28(ld_g2(pts + 192 * i));
        if (mode == 2) {  // bases as the table build / the de-duplication hand them over: outputs of Fp2 products (c0 < 5p, c1 < 8p)
            const Fp2x28 one = Fp2x28Ops::one();
            a.x = Fp2x28Ops::mul(a.x, one);
            a.y = Fp2x28Ops::mul(a.y, one);
        }
        g2x28::add_mixed(acc, a, neg[i] != 0);
    }
    // invariants the next addition relies on: X, Y < 3p (top limb below 3 * 0x1a012), every limb normalised
    if (!(fp28::limbs_all_zero(acc.ZZ.c0) && fp28::limbs_all_zero(acc.ZZ.c1))) {
        const Fp28* xy[4] = {&acc.X.c0, &acc.X.c1, &acc.Y.c0, &acc.Y.c1};
        for (int e = 0; e < 4; ++e) {
            if (xy[e]->l[13] >= 3 * 0x1a012u) return -1;
            for (int i = 0; i < 13; ++i)
                if (xy[e]->l[i] >> 28) return -2;
        }
    }
    if (mode == 1) {
        G2X28 twice = acc;
        xyzz_add<Fp2x28Ops>(twice, acc);  // generic consumer: 2 * sum
        acc = twice;
    }
    st_g2(out193, g2x28::to_std(acc));
    return 0;
}
// sum_i (+/-) k_i * P_i with EVERY general addition done by xyzz_add_mem (second operand read from memory - what the G2
// tail kernels use on the device), including the double-and-add of the scalar multiplication; mode 1 adds each term twice
// through a parked copy (P + P = doubling inside add_mem), mode 2 adds a term and its negation (cancellation -> identity)
// the same with the static-bound general addition / doubling of the tail kernels (g2x28::add_mem, g2x28::dbl); the terms enter through
// the accumulation's mixed addition, and the sum is finally handed to the GENERIC addition once (both disciplines must mix)
int hc_g2x28_lincomb_mem_fast(const uint8_t* pts, const uint32_t* k, const uint8_t* neg, int n, int mode, uint8_t* out193) {
    typedef Fp2x28Ops F;
    G2X28 acc = xyzz_identity<F>();
    for (int i = 0; i < n; ++i) {
        G2X28 t = xyzz_identity<F>();
        g2x28::add_mixed(t, g2x28::affine_to28(ld_g2(pts + 192 * i)), neg[i] != 0);
        G2X28 m = xyzz_identity<F>();
        for (int b = 31; b >= 0; --b) {
            m = g2x28::dbl(m);
            if ((k[i] >> b) & 1) g2x28::add_mem(m, &t);
        }
        const G2X28 parked = m;
        g2x28::add_mem(acc, &parked);
        if (mode == 1) {
            G2X28 d = parked;
            g2x28::add_mem(d, &parked);  // doubling branch
            g2x28::add_mem(acc, &d);
        } else if (mode == 2) {
            G2X28 c = parked;
            const G2X28 minus = xyzz_neg<F>(parked);
            g2x28::add_mem(c, &minus);   // cancellation branch
            if (!g2x28::is_identity(c)) return 1;
            g2x28::add_mem(acc, &c);
        }
        // invariants: X, Y < 3p and normalised
        if (!g2x28::is_identity(acc)) {
            const Fp28* xy[4] = {&acc.X.c0, &acc.X.c1, &acc.Y.c0, &acc.Y.c1};
            for (int e = 0; e < 4; ++e) {
                if (xy[e]->l[13] >= 3 * 0x1a012u) return -1;
                for (int j = 0; j < 13; ++j)
                    if (xy[e]->l[j] >> 28) return -2;
            }
        }
    }
    G2X28 z = xyzz_identity<F>();
    xyzz_add<F>(z, acc);  // generic consumer
    st_g2(out193, g2x28::to_std(z));
    return 0;
}
int hc_g2x28_lincomb_mem(const uint8_t* pts, const uint32_t* k, const uint8_t* neg, int n, int mode, uint8_t* out193) {
    typedef Fp2x28Ops F;
    G2X28 acc = xyzz_identity<F>();
    for (int i = 0; i < n; ++i) {
        G2A28 a = g2x28::affine_to28(ld_g2(pts + 192 * i));
        if (neg[i]) a.y = F::neg(a.y);
        G2X28 t = xyzz_identity<F>();
        xyzz_add_mixed<F>(t, a);
        G2X28 m = xyzz_identity<F>();
        for (int b = 31; b >= 0; --b) {
            m = xyzz_dbl<F>(m);
            if ((k[i] >> b) & 1) xyzz_add_mem<F>(m, &t);
        }
        const G2X28 parked = m;
        xyzz_add_mem<F>(acc, &parked);               // acc += m  (first term: acc is the identity; k = 0: m is the identity)
        if (mode == 1) {                              // m + m through add_mem: the doubling branch
            G2X28 d = parked;
            xyzz_add_mem<F>(d, &parked);
            xyzz_add_mem<F>(acc, &d);                 // acc += 2 m   (total 3 m)
        } else if (mode == 2) {                       // m + (-m) through add_mem: the cancellation branch
            G2X28 c = parked;
            const G2X28 minus = xyzz_neg<F>(parked);
            xyzz_add_mem<F>(c, &minus);
            if (!xyzz_is_identity<F>(c)) return 1;
            xyzz_add_mem<F>(acc, &c);                 // adding the identity from memory changes nothing
        }
    }
    st_g2(out193, g2x28::to_std(acc));
    return 0;
}
// what msm_accumulate_g2pair_kernel runs (g2p::add_mixed on a pair of lanes): a chain of (+/-) points; mode 2: bases that are Fp2 product
// outputs of the one-lane code (c0 < 5p, c1 < 8p: group sums, table entries); mode 1: the pair's sum is stored the way the kernel stores it
// (X brought below 3 p) and handed to the one-lane GENERIC general addition (the tail kernels' consumer discipline)
int hc_g2p_sum_mixed(const uint8_t* pts, const uint8_t* neg, int n, int mode, uint8_t* out193) {
    G2X28 sum;
    int bad[2] = {0, 0};
    run_pair([&](int lane) {
        g2p::Pt acc = g2p::identity();
        for (int i = 0; i < n; ++i) {
            G2A28 a = g2x28::affine_to28(ld_g2(pts + 192 * i));
            if (mode == 2) {
                const Fp2x28 one = Fp2x28Ops::one();
                a.x = Fp2x28Ops::mul(a.x, one);
                a.y = Fp2x28Ops::mul(a.y, one);
            }
            g2p::add_mixed(acc, pair_aff(a, lane), neg[i] != 0);
            if (!g2p::is_identity(acc) && pair_invariants(acc)) bad[lane] = pair_invariants(acc);
        }
        if (!g2p::is_identity(acc)) acc.X = fp28::reduce(acc.X);
        pair_store(sum, acc, lane);
    });
    if (bad[0] || bad[1]) return bad[0] ? bad[0] : bad[1];
    if (mode == 1) {
        G2X28 twice = sum;
        xyzz_add<Fp2x28Ops>(twice, sum);
        sum = twice;
    }
    st_g2(out193, g2x28::to_std(sum));
    return 0;
}
// sum_i (+/-) k_i * P_i with every general addition / doubling done by g2p::add / g2p::dbl on a pair of lanes (terms enter through
// g2p::add_mixed); mode 1 adds each term twice more through a copy (P + P = the doubling branch of add), mode 2 adds a term and its
// negation (cancellation -> identity, then the identity as an operand); the one-lane generic addition consumes the result once
int hc_g2p_lincomb(const uint8_t* pts, const uint32_t* k, const uint8_t* neg, int n, int mode, uint8_t* out193) {
    G2X28 sum;
    int bad[2] = {0, 0};
    run_pair([&](int lane) {
        g2p::Pt acc = g2p::identity();
        for (int i = 0; i < n; ++i) {
            g2p::Pt t = g2p::identity();
            const G2A28 a = g2x28::affine_to28(ld_g2(pts + 192 * i));
            g2p::add_mixed(t, pair_aff(a, lane), neg[i] != 0);
            g2p::Pt m = g2p::identity();
            for (int b = 31; b >= 0; --b) {
                m = g2p::dbl(m);
                if ((k[i] >> b) & 1) g2p::add(m, t);
            }
            g2p::add(acc, m);
            if (mode == 1) {
                g2p::Pt d = m;
                g2p::add(d, m);  // doubling branch
                g2p::add(acc, d);
            } else if (mode == 2) {
                g2p::Pt c = m, minus = m;
                minus.Y = fp28::norm(fp28::sub<3>(fp28::zero(), m.Y));  // 3 p - Y
                minus.Y = fp28::reduce(minus.Y);
                g2p::add(c, minus);  // cancellation branch
                if (!g2p::is_identity(c)) bad[lane] = 1;
                g2p::add(acc, c);
            }
            if (!g2p::is_identity(acc) && pair_invariants(acc)) bad[lane] = pair_invariants(acc);
        }
        if (!g2p::is_identity(acc)) acc.X = fp28::reduce(acc.X);
        pair_store(sum, acc, lane);
    });
    if (bad[0] || bad[1]) return bad[0] ? bad[0] : bad[1];
    G2X28 z = xyzz_identity<Fp2x28Ops>();
    xyzz_add<Fp2x28Ops>(z, sum);  // generic consumer
    st_g2(out193, g2x28::to_std(z));
    return 0;
}
// Fr in 9 x 29-bit limbs
// ---- bzk_endo.cuh: scalar split and point images of the endomorphism form of the MSM (round 4)
// k: 8 x u32 canonical scalar -> G2 split: 4 x int64
int hc_endo_decompose4(const uint32_t* k, int64_t* s4) {
    endo::decompose4(k, s4);
    return 0;
}
// -> G1 split: mag[2][4] (u32 limbs) and neg[2]
int hc_endo_decompose2(const uint32_t* k, uint32_t* mag8, uint8_t* neg2) {
    uint32_t mag[2][4];
    bool neg[2];
    endo::decompose2(k, mag, neg);
    for (int m = 0; m < 2; ++m) {
        for (int i = 0; i < 4; ++i) mag8[4 * m + i] = mag[m][i];
        neg2[m] = neg[m] ? 1 : 0;
    }
    return 0;
}
// image X^2 P of a G1 point / X^m P of a G2 point in the internal form, handed back through ONE mixed addition (what the accumulation
// does with an image): out = packed affine point.  neg: add -image
int hc_endo_g1_image(const uint8_t* p96, int neg, uint8_t* out97) {
    const G1A28 a = g1x28::affine_to28(ld_g1(p96));
    G1X28 acc = g1x28::identity();
    g1x28::add_mixed(acc, endo::g1_image(a), neg != 0);
    st_g1(out97, g1x28::to_std(acc));
    return 0;
}
// twice: image added to the point itself first (exercises the bounds of an image as the affine operand of a REAL addition)
int hc_endo_g2_image(const uint8_t* p192, int m, int neg, int onto_self, uint8_t* out193) {
    const G2A28 a = g2x28::affine_to28(ld_g2(p192));
    const G2A28 im = m == 1 ? endo::g2_image<1>(a) : m == 2 ? endo::g2_image<2>(a) : endo::g2_image<3>(a);
    G2X28 acc = xyzz_identity<Fp2x28Ops>();
    if (onto_self) g2x28::add_mixed(acc, a, false);
    g2x28::add_mixed(acc, im, neg != 0);
    st_g2(out193, g2x28::to_std(acc));
    return 0;
}
int hc_fr29_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
    st<FrParams>(out, fr29::from29(fr29::mul(fr29::to29(ld<FrParams>(a)), fr29::to29(ld<FrParams>(b)))));
    return 0;
}
// the dedicated square == the product of a value with itself, limb for limb, also on unnormalised inputs (value * 2^grow)
int hc_fr29_sqr_equals_mul(const uint8_t* a, int grow, uint8_t* out) {
    Fr29 x = fr29::to29(ld<FrParams>(a));
    for (int i = 0; i < grow; ++i) x = fr29::add(x, x);
    const Fr29 s = fr29::sqr_body(x), m = fr29::mul_body(x, x);
    int diff = 0;
    for (int i = 0; i < fr29::N; ++i) diff |= s.l[i] != m.l[i];
    st<FrParams>(out, fr29::from29(s));
    return diff;
}
// one Poseidon hash through the device function; consts = rc then mds in the reference's plain layout, 8 x 32-bit
// Montgomery (n_consts entries): the sparse-partial-round constants are derived here exactly as the library does
int hc_poseidon29(const uint8_t* in, int arity, const uint8_t* consts, int n_consts, int rf, int rp, uint8_t* out) {
    const int T = arity + 1;
    if (n_consts != (rf + rp) * T + T * T) return -2;
    Fr inp[8];
    for (int k = 0; k < arity; ++k) inp[k] = ld<FrParams>(in + 32 * k);
    std::vector<Fr> rc((size_t)(rf + rp) * T), mds((size_t)T * T), flat;
    for (size_t i = 0; i < rc.size(); ++i) rc[i] = ld<FrParams>(consts + 32 * i);
    for (size_t i = 0; i < mds.size(); ++i) mds[i] = ld<FrParams>(consts + 32 * (rc.size() + i));
    if (!poseidon_optimize(T, rf, rp, rc, mds, flat)) return -3;
    std::vector<Fr29> c(flat.size());
    for (size_t i = 0; i < flat.size(); ++i) c[i] = fr29::norm(fr29::to29(flat[i]));
    Fr r;
    switch (T) {
        case 2: r = poseidon29_hash<2>(inp, c.data(), rf, rp); break;
        case 3: r = poseidon29_hash<3>(inp, c.data(), rf, rp); break;
        case 4: r = poseidon29_hash<4>(inp, c.data(), rf, rp); break;
        case 5: r = poseidon29_hash<5>(inp, c.data(), rf, rp); break;
        case 6: r = poseidon29_hash<6>(inp, c.data(), rf, rp); break;
        case 7: r = poseidon29_hash<7>(inp, c.data(), rf, rp); break;
        case 8: r = poseidon29_hash<8>(inp, c.data(), rf, rp); break;
        default: return -1;
    }
    st<FrParams>(out, r);
    return 0;
}
int hc_fr_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) { return field_op<FrParams>(op, a, b, out); }
int hc_fp_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) { return field_op<FpParams>(op, a, b, out); }

// computes sum_i k_i * P_i with the XYZZ ops only (mixed add, add, dbl); pts affine 96 B, k u32
int hc_g1_lincomb(const uint8_t* pts, const uint32_t* k, int n, uint8_t* out97) {
    G1Xyzz acc = xyzz_identity<FpOps>();
    for (int i = 0; i < n; ++i) {
        G1Affine a = ld_g1(pts + 96 * i);
        G1Xyzz t = xyzz_identity<FpOps>();
        xyzz_add_mixed<FpOps>(t, a);
        G1Xyzz m = xyzz_mul_u32<FpOps>(t, k[i]);
        xyzz_add<FpOps>(acc, m);
    }
    st_g1(out97, acc);
    return 0;
}
int hc_g2_lincomb(const uint8_t* pts, const uint32_t* k, int n, uint8_t* out193) {
    G2Xyzz acc = xyzz_identity<Fp2Ops>();
    for (int i = 0; i < n; ++i) {
        G2Affine a = ld_g2(pts + 192 * i);
        G2Xyzz t = xyzz_identity<Fp2Ops>();
        xyzz_add_mixed<Fp2Ops>(t, a);
        G2Xyzz m = xyzz_mul_u32<Fp2Ops>(t, k[i]);
        xyzz_add<Fp2Ops>(acc, m);
    }
    st_g2(out193, acc);
    return 0;
}
// acc = sum of points via repeated mixed add (exercises add_mixed incl. doubling / cancellation)
int hc_g1_sum_mixed(const uint8_t* pts, int n, uint8_t* out97) {
    G1Xyzz acc = xyzz_identity<FpOps>();
    for (int i = 0; i < n; ++i) xyzz_add_mixed<FpOps>(acc, ld_g1(pts + 96 * i));
    st_g1(out97, acc);
    return 0;
}
int hc_g2_sum_mixed(const uint8_t* pts, int n, uint8_t* out193) {
    G2Xyzz acc = xyzz_identity<Fp2Ops>();
    for (int i = 0; i < n; ++i) xyzz_add_mixed<Fp2Ops>(acc, ld_g2(pts + 192 * i));
    st_g2(out193, acc);
    return 0;
}

// host_fp64.h: the 6 x 64-bit-limb host field of the Horner / to-affine / proof-assembly code (op codes of field_op)
int hc_hfp_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    HFp x, y = HFpOps::zero(), r;
    memcpy(x.l, a, 48);
    if (b) memcpy(y.l, b, 48);
    switch (op) {
        case 0: r = HFpOps::add(x, y); break;
        case 1: r = HFpOps::sub(x, y); break;
        case 2: r = HFpOps::mul(x, y); break;
        case 3: r = HFpOps::inv(x); break;
        case 6: r = HFpOps::neg(x); break;
        case 7: r = HFpOps::dbl(x); break;
        case 8: r = HFpOps::sqr(x); break;
        default: return -1;
    }
    memcpy(out, r.l, 48);
    return 0;
}
// k * p (+ q at the end) by double-and-add with the XYZZ formulas over the fast host fields, packed like libbzk packs a result
int hc_hfp_g1_mul_add(const uint8_t* p96, const uint32_t* k8, const uint8_t* q96, uint8_t* out97) {
    typedef XyzzT<HFpOps> H;
    const H P = to_host_fast<FpOps>(xyzz_from_affine<FpOps>(ld_g1(p96)));
    H r = xyzz_identity<HFpOps>();
    for (int i = 255; i >= 0; --i) {
        r = xyzz_dbl<HFpOps>(r);
        if ((k8[i >> 5] >> (i & 31)) & 1) xyzz_add<HFpOps>(r, P);
    }
    if (q96) xyzz_add<HFpOps>(r, to_host_fast<FpOps>(xyzz_from_affine<FpOps>(ld_g1(q96))));
    AffineT<HFpOps> a;
    const bool fin = xyzz_to_affine<HFpOps>(r, a);
    memcpy(out97, a.x.l, 48);
    memcpy(out97 + 48, a.y.l, 48);
    out97[96] = fin ? 0 : 1;
    return 0;
}
int hc_hfp_g2_mul_add(const uint8_t* p192, const uint32_t* k8, const uint8_t* q192, uint8_t* out193) {
    typedef XyzzT<HFp2Ops> H;
    const H P = to_host_fast<Fp2Ops>(xyzz_from_affine<Fp2Ops>(ld_g2(p192)));
    H r = xyzz_identity<HFp2Ops>();
    for (int i = 255; i >= 0; --i) {
        r = xyzz_dbl<HFp2Ops>(r);
        if ((k8[i >> 5] >> (i & 31)) & 1) xyzz_add<HFp2Ops>(r, P);
    }
    if (q192) xyzz_add<HFp2Ops>(r, to_host_fast<Fp2Ops>(xyzz_from_affine<Fp2Ops>(ld_g2(q192))));
    AffineT<HFp2Ops> a;
    const bool fin = xyzz_to_affine<HFp2Ops>(r, a);
    memcpy(out193, a.x.c0.l, 48);
    memcpy(out193 + 48, a.x.c1.l, 48);
    memcpy(out193 + 96, a.y.c0.l, 48);
    memcpy(out193 + 144, a.y.c1.l, 48);
    out193[192] = fin ? 0 : 1;
    return 0;
}

// host_fr64.h: the witness generator's Fr product and its dot product with one reduction
int hc_hfr_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) {
    st<FrParams>(out, hfr::mul(ld<FrParams>(a), ld<FrParams>(b)));
    return 0;
}
int hc_hfr_dot(const uint8_t* a, const uint8_t* b, int n, uint8_t* out) {
    Fr x[32], y[32];
    if (n < 0 || n > 32) return -1;
    for (int k = 0; k < n; ++k) {
        x[k] = ld<FrParams>(a + 32 * k);
        y[k] = ld<FrParams>(b + 32 * k);
    }
    st<FrParams>(out, hfr::dot(x, y, n));
    return 0;
}
int hc_hfr_inv(const uint8_t* a, uint8_t* out) {
    st<FrParams>(out, hfr::inv(ld<FrParams>(a)));
    return 0;
}
int hc_host_ifma_available() { return hfr::ifma_available() ? 1 : 0; }
// host_fr_ifma.h: first row k < n with a_k b_k != c_k, or -1 (the witness's satisfaction scan)
long hc_products_first_mismatch(const uint8_t* a, const uint8_t* b, const uint8_t* c, int n) {
    std::vector<Fr> x((size_t)n), y((size_t)n), z((size_t)n);
    for (int k = 0; k < n; ++k) {
        x[k] = ld<FrParams>(a + 32 * k);
        y[k] = ld<FrParams>(b + 32 * k);
        z[k] = ld<FrParams>(c + 32 * k);
    }
    return hfr::products_first_mismatch(x.data(), y.data(), z.data(), 0, (size_t)n);
}
// host_fr_ifma.h: out = M s for a t x t matrix; mode 0 = the dispatching entry (AVX-512 IFMA where the CPU has it), 1 = one hfr::dot per
// row.  Returns 1 when the IFMA path ran, 0 when the scalar one did.
int hc_mds_mul(int t, const uint8_t* m, const uint8_t* s, int mode, uint8_t* out) {
    if (t < 1 || t > 17) return -1;
    static Fr M[17 * 17];
    static hfr::MdsTable T;
    Fr x[17], y[17];
    for (int i = 0; i < t * t; ++i) M[i] = ld<FrParams>(m + 32 * i);
    for (int k = 0; k < t; ++k) x[k] = ld<FrParams>(s + 32 * k);
    int used = 0;
    if (mode == 0) {
        hfr::mds_table_build(T, M, t);
        hfr::mds_mul(T, x, y);
        used = T.ifma ? 1 : 0;
    } else {
        for (int j = 0; j < t; ++j) y[j] = hfr::dot(M + (size_t)j * t, x, t);
    }
    for (int j = 0; j < t; ++j) st<FrParams>(out + 32 * j, y[j]);
    return used;
}

// ---- the host pairing (host_pairing.h), piece by piece against slower forms of the same thing.  Random elements come from a seed.
static uint64_t hc_rng(uint64_t& st) { st += 0x9e3779b97f4a7c15ull; uint64_t z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static HFp hc_rand_fp(uint64_t& st) {
    HFp a;
    for (int i = 0; i < 6; ++i) a.l[i] = hc_rng(st);
    a.l[5] &= 0x0fffffffffffffffull;  // < 2^380 < p
    return a;
}
static hp::E12 hc_rand_e12(uint64_t& st) {
    hp::E12 f;
    HFp* v = (HFp*)&f;
    for (int i = 0; i < 12; ++i) v[i] = hc_rand_fp(st);
    return f;
}
static hp::E12 hc_e12_pow(const hp::E12& a, const uint64_t* e, int limbs) {
    hp::E12 r = hp::e12_one();
    for (int i = 64 * limbs - 1; i >= 0; --i) {
        r = hp::e12_sqr(r);
        if ((e[i >> 6] >> (i & 63)) & 1) r = hp::e12_mul(r, a);
    }
    return r;
}
static hp::E6 hc_e6_mul_schoolbook(const hp::E6& a, const hp::E6& b) {
    using namespace hp;
    const E2 t0 = F2::mul(a.c0, b.c0), t1 = F2::mul(a.c1, b.c1), t2 = F2::mul(a.c2, b.c2);
    E6 r;
    r.c0 = F2::add(t0, e2_mul_xi(F2::add(F2::mul(a.c1, b.c2), F2::mul(a.c2, b.c1))));
    r.c1 = F2::add(F2::add(F2::mul(a.c0, b.c1), F2::mul(a.c1, b.c0)), e2_mul_xi(t2));
    r.c2 = F2::add(F2::add(F2::mul(a.c0, b.c2), F2::mul(a.c2, b.c0)), t1);
    return r;
}
// bit i of the result set = check i failed:
//  0 Karatsuba Fp6 product == schoolbook          1 Fp12 squaring == product with itself        2 f * f^-1 == 1
//  3 Frobenius == f^p by square-and-multiply      4 sparse line product == full product          5 easy part lands in the cyclotomic subgroup (g conj(g) == 1)
//  6 Granger-Scott squaring == plain squaring there   7 cyclotomic exponentiation by x == plain exponentiation by |x|, conjugated
//  8 fast final exponentiation == (plain final exponentiation)^3                                 9 Frobenius^12 == identity
// 10 sparse product at tower positions 0, 1, 4 == full product
int hc_pairing_pieces(uint64_t seed) {
    using namespace hp;
    uint64_t st = seed;
    int bad = 0;
    const E12 f = hc_rand_e12(st), g = hc_rand_e12(st);
    if (!e6_eq(e6_mul(f.a0, g.a1), hc_e6_mul_schoolbook(f.a0, g.a1))) bad |= 1 << 0;
    if (!e12_eq(e12_sqr(f), e12_mul(f, f))) bad |= 1 << 1;
    if (!e12_is_one(e12_mul(f, e12_inv(f)))) bad |= 1 << 2;
    if (!e12_eq(e12_frob(f), hc_e12_pow(f, hfp::consts().p, 6))) bad |= 1 << 3;
    {
        const E2 l00 = {hc_rand_fp(st), hc_rand_fp(st)}, l01 = {hc_rand_fp(st), hc_rand_fp(st)};
        const HFp l11 = hc_rand_fp(st);
        E12 l;
        l.a0 = {l00, l01, F2::zero()};
        l.a1 = {F2::zero(), {l11, F1::zero()}, F2::zero()};
        if (!e12_eq(e12_mul_by_line(f, l00, l01, l11), e12_mul(f, l))) bad |= 1 << 4;
    }
    const E12 m = final_exp_easy(f);
    if (!e12_is_one(e12_mul(m, e12_conj(m)))) bad |= 1 << 5;
    if (!e12_eq(e12_cyc_sqr(m), e12_sqr(m))) bad |= 1 << 6;
    {
        const uint64_t x[1] = {X_ABS};
        if (!e12_eq(e12_cyc_exp_x(m), e12_conj(hc_e12_pow(m, x, 1)))) bad |= 1 << 7;
    }
    {
        const E12 slow = final_exp_plain(f);
        if (!e12_eq(final_exp(f), e12_mul(e12_sqr(slow), slow))) bad |= 1 << 8;
    }
    {
        E12 t = g;
        for (int i = 0; i < 12; ++i) t = e12_frob(t);
        if (!e12_eq(t, g)) bad |= 1 << 9;
    }
    {
        const E2 c0 = {hc_rand_fp(st), hc_rand_fp(st)}, c1 = {hc_rand_fp(st), hc_rand_fp(st)}, c4 = {hc_rand_fp(st), hc_rand_fp(st)};
        if (!e12_eq(e12_mul_by_014(f, c0, c1, c4), e12_mul(f, E12{{c0, c1, F2::zero()}, {F2::zero(), c4, F2::zero()}}))) bad |= 1 << 10;
    }
    return bad;
}
// prod_k e(P_k, Q_k) == 1 ?  (n <= 4 pairs, packed 96-byte G1 / 192-byte G2 affine points in the library's Montgomery form, all finite and on
// their curves; mode 0: the shipped Miller loop + final exponentiation, 1: the same Miller loop + the plain final exponentiation, 2: the affine Miller loop; mode 3 asks something else: is the
// pairing value of the projective loop EQUAL to that of the affine loop).  1 yes, 0 no, -1 degenerate
int hc_pairing_product_is_one(const uint8_t* g1s, const uint8_t* g2s, int n, int mode) {
    using namespace hp;
    if (n < 1 || n > 4) return -2;
    G1A p[4];
    G2A q[4];
    for (int k = 0; k < n; ++k) {
        memcpy(p[k].x.l, g1s + 96 * k, 48); memcpy(p[k].y.l, g1s + 96 * k + 48, 48); p[k].inf = false;
        memcpy(q[k].x.c0.l, g2s + 192 * k, 48); memcpy(q[k].x.c1.l, g2s + 192 * k + 48, 48);
        memcpy(q[k].y.c0.l, g2s + 192 * k + 96, 48); memcpy(q[k].y.c1.l, g2s + 192 * k + 144, 48); q[k].inf = false;
    }
    bool degenerate = false;
    const E12 f = mode == 2 ? multi_miller_affine(p, q, n, &degenerate) : multi_miller(p, q, n, &degenerate);
    if (degenerate) return -1;
    if (mode == 3) {  // the projective and the affine loop give the SAME pairing value (their lines differ by Fp2 factors only)
        bool d2 = false;
        const E12 g = multi_miller_affine(p, q, n, &d2);
        if (d2) return -1;
        return e12_eq(final_exp(f), final_exp(g)) ? 1 : 0;
    }
    return e12_is_one(mode == 1 ? final_exp_plain(f) : final_exp(f)) ? 1 : 0;
}
}
