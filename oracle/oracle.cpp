// CPU ORACLE - TEST INFRASTRUCTURE ONLY.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
// only as the checker / reported CPU baseline.  The product (libbzk.so) never links or calls it.
//
// CPU restatement of the Groth16 hot path of Bazuka's MPN rollup.  The reference cannot be built in
// this environment (Rust; bellman 0.14 / bls12_381 0.8 / ff 0.13 are un-vendored crates), so this is
// a "port"-kind baseline: same algorithms as the reference stack, written from the public
// definitions.  What each part follows:
//   poseidon_*      /root/reference/src/zk/poseidon/mod.rs:24-84 (permutation, output lane 1),
//                   params/mod.rs:27-80 (parameter shape); constants re-derived by Grain LFSR
//   merkle4_root    /root/reference/src/zk/state/mod.rs:353-391 (node = H(c0..c3)), dense form;
//                   heap order (4^k-1)/3+i as at :355,382-383
//   ntt             bellman 0.14 domain.rs semantics (omega = 7^((r-1)/2^32)^(2^(32-log m)),
//                   coset shift 7), call sites /root/reference/src/mpn/circuits/test.rs:135,175,215
//   msm_*           bellman multiexp semantics: sum s_i * P_i over canonical 255-bit scalars
//   groth16_*       bellman 0.14 generate_parameters / create_proof layout
//   byte formats    /root/reference/src/zk/groth16/mod.rs:19-38 (Montgomery limbs, x|y|inf)
// Pinned by: 16 Poseidon KATs (poseidon/mod.rs:114-149), VK blobs (config/blockchain.rs:32-37),
// and cross-checked against oracle/pyref.py.  Groth16 proof bytes: "parity unpinned" (no reference
// vector exists; every reference prove call draws OsRng) - DESIGN.md.
#include "curve.hpp"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

namespace orc {

FieldConsts<4> FrTag::C = {{0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}, {}, {}, 0};
FieldConsts<6> FpTag::C = {{0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull,
                            0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull}, {}, {}, 0};

static std::once_flag g_init_flag;
void init_fields() {
    std::call_once(g_init_flag, [] {
        derive_consts<4>(FrTag::C);
        derive_consts<6>(FpTag::C);
    });
}

// ---------------------------------------------------------------- helpers
static inline Fr fr_load(const uint8_t* p) { Fr a; memcpy(a.v, p, 32); return a; }
static inline void fr_store(uint8_t* p, const Fr& a) { memcpy(p, a.v, 32); }
static inline Fp fp_load(const uint8_t* p) { Fp a; memcpy(a.v, p, 48); return a; }
static inline void fp_store(uint8_t* p, const Fp& a) { memcpy(p, a.v, 48); }

static G1Affine g1_load96(const uint8_t* p) { return {fp_load(p), fp_load(p + 48), false}; }
static G2Affine g2_load192(const uint8_t* p) {
    return {{fp_load(p), fp_load(p + 48)}, {fp_load(p + 96), fp_load(p + 144)}, false};
}
static void g1_store97(uint8_t* p, const G1Affine& a) {
    fp_store(p, a.x); fp_store(p + 48, a.y); p[96] = a.inf ? 1 : 0;
}
static void g2_store193(uint8_t* p, const G2Affine& a) {
    fp_store(p, a.x.c0); fp_store(p + 48, a.x.c1); fp_store(p + 96, a.y.c0); fp_store(p + 144, a.y.c1);
    p[192] = a.inf ? 1 : 0;
}
static G1Affine g1_load97(const uint8_t* p) { G1Affine a = g1_load96(p); a.inf = p[96] != 0; return a; }
static G2Affine g2_load193(const uint8_t* p) { G2Affine a = g2_load192(p); a.inf = p[192] != 0; return a; }

template <class Fn>
static void parallel_for(size_t n, int nthreads, Fn fn) {
    if (nthreads <= 1 || n < 2) { fn(0, n, 0); return; }
    std::vector<std::thread> th;
    size_t chunk = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        size_t lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
        if (lo >= hi) break;
        th.emplace_back([=] { fn(lo, hi, t); });
    }
    for (auto& x : th) x.join();
}

// ---------------------------------------------------------------- Poseidon
struct PoseidonParams {
    int t = 0, rf = 8, rp = 0;
    std::vector<Fr> rc;   // t*(rf+rp)
    std::vector<Fr> mds;  // t*t row-major
};

struct Grain {
    uint8_t s[80];
    int pos = 0;  // ring buffer head
    Grain(int t, int rf, int rp) {
        int k = 0;
        auto put = [&](int val, int width) { for (int i = width - 1; i >= 0; --i) s[k++] = (val >> i) & 1; };
        put(1, 2); put(0, 4); put(255, 12); put(t, 12); put(rf, 10); put(rp, 10);
        for (int i = 0; i < 30; ++i) s[k++] = 1;
        for (int i = 0; i < 160; ++i) step();
    }
    int step() {
        auto b = [&](int i) { return s[(pos + i) % 80]; };
        uint8_t nw = b(62) ^ b(51) ^ b(38) ^ b(23) ^ b(13) ^ b(0);
        s[pos] = nw;  // overwrite oldest, becomes newest
        pos = (pos + 1) % 80;
        return nw;
    }
    int bit() {
        for (;;) { int a = step(); int b = step(); if (a) return b; }
    }
    void raw255(uint64_t out[4]) {
        memset(out, 0, 32);
        for (int i = 254; i >= 0; --i) if (bit()) out[i / 64] |= 1ull << (i % 64);
    }
};

static std::mutex g_pos_mu;
static PoseidonParams g_pos[18];

static const PoseidonParams& poseidon_params(int t) {
    std::lock_guard<std::mutex> lk(g_pos_mu);
    PoseidonParams& P = g_pos[t];
    if (P.t) return P;
    int rf = 8, rp = t <= 5 ? 56 : 57;
    Grain g(t, rf, rp);
    P.rf = rf; P.rp = rp;
    while ((int)P.rc.size() < t * (rf + rp)) {
        uint64_t v[4];
        g.raw255(v);
        if (!limbs_geq<4>(v, FrTag::C.mod)) P.rc.push_back(Fr::from_canon(v));  // rejection
    }
    std::vector<Fr> xs, ys;
    for (int i = 0; i < 2 * t; ++i) {
        uint64_t v[4];
        g.raw255(v);
        if (limbs_geq<4>(v, FrTag::C.mod)) limbs_sub<4>(v, v, FrTag::C.mod);  // reduced, not rejected
        (i < t ? xs : ys).push_back(Fr::from_canon(v));
    }
    P.mds.resize(t * t);
    for (int i = 0; i < t; ++i)
        for (int j = 0; j < t; ++j) P.mds[i * t + j] = xs[i].add(ys[j]).inv();
    P.t = t;
    return P;
}

static inline Fr sbox5(const Fr& x) { Fr x2 = x.sqr(); return x2.sqr().mul(x); }

static Fr poseidon_hash(const Fr* vals, int arity) {
    const int t = arity + 1;
    const PoseidonParams& P = poseidon_params(t);
    Fr st[17], tmp[17];
    st[0] = Fr::zero();
    for (int i = 0; i < arity; ++i) st[i + 1] = vals[i];
    int off = 0;
    for (int rnd = 0; rnd < P.rf + P.rp; ++rnd) {
        for (int i = 0; i < t; ++i) st[i] = st[i].add(P.rc[off + i]);
        off += t;
        bool full = rnd < P.rf / 2 || rnd >= P.rf / 2 + P.rp;
        if (full) for (int i = 0; i < t; ++i) st[i] = sbox5(st[i]);
        else st[0] = sbox5(st[0]);
        for (int j = 0; j < t; ++j) {
            Fr acc = Fr::zero();
            for (int k = 0; k < t; ++k) acc = acc.add(P.mds[j * t + k].mul(st[k]));
            tmp[j] = acc;
        }
        memcpy(st, tmp, sizeof(Fr) * t);
    }
    return st[1];
}

// ---------------------------------------------------------------- NTT
static Fr fr_root_of_unity() {  // 7^((r-1)/2^32)
    uint64_t e[4];
    uint64_t one[4] = {1, 0, 0, 0};
    limbs_sub<4>(e, FrTag::C.mod, one);
    // shift right by 32
    for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 32) | (i < 3 ? e[i + 1] << 32 : 0);
    return Fr::from_u64(7).pow(e, 4);
}
static Fr fr_omega(int log_n) {
    Fr w = fr_root_of_unity();
    for (int i = log_n; i < 32; ++i) w = w.sqr();
    return w;
}
static Fr fr_pow_u64(Fr b, uint64_t e) { return b.pow(&e, 1); }

static void ntt_inplace(std::vector<Fr>& a, int log_n, bool inverse, bool coset, int nthreads) {
    const size_t n = (size_t)1 << log_n;
    Fr w = fr_omega(log_n);
    if (inverse) w = w.inv();
    if (coset && !inverse) {
        Fr g = Fr::from_u64(7);
        parallel_for(n, nthreads, [&](size_t lo, size_t hi, int) {
            Fr x = fr_pow_u64(g, lo);
            for (size_t j = lo; j < hi; ++j) { a[j] = a[j].mul(x); x = x.mul(g); }
        });
    }
    // bit reversal
    for (size_t i = 0; i < n; ++i) {
        size_t r = 0;
        for (int b = 0; b < log_n; ++b) r |= ((i >> b) & 1) << (log_n - 1 - b);
        if (i < r) std::swap(a[i], a[r]);
    }
    std::vector<Fr> tw(n / 2 ? n / 2 : 1);
    tw[0] = Fr::one();
    for (size_t i = 1; i < n / 2; ++i) tw[i] = tw[i - 1].mul(w);
    for (size_t m = 1; m < n; m <<= 1) {
        size_t stride = n / (2 * m);
        parallel_for(n / 2, nthreads, [&](size_t lo, size_t hi, int) {
            for (size_t idx = lo; idx < hi; ++idx) {
                size_t k = (idx / m) * 2 * m, j = idx % m;
                Fr u = a[k + j], v = a[k + j + m].mul(tw[j * stride]);
                a[k + j] = u.add(v);
                a[k + j + m] = u.sub(v);
            }
        });
    }
    if (inverse) {
        Fr ninv = Fr::from_u64(n).inv();
        Fr gi = Fr::from_u64(7).inv();
        parallel_for(n, nthreads, [&](size_t lo, size_t hi, int) {
            Fr x = coset ? fr_pow_u64(gi, lo) : Fr::one();
            for (size_t j = lo; j < hi; ++j) {
                a[j] = a[j].mul(ninv);
                if (coset) { a[j] = a[j].mul(x); x = x.mul(gi); }
            }
        });
    }
}

// ---------------------------------------------------------------- MSM (Pippenger, window per task)
template <class F>
static Jac<F> msm_pippenger(const Affine<F>* bases, const uint64_t* scalars /*canonical, 4 limbs each*/,
                            size_t n, int nthreads) {
    if (n == 0) return Jac<F>::identity();
    int c = n < 32 ? 3 : (int)std::ceil(std::log((double)n));
    if (c > 20) c = 20;
    const int W = (255 + c - 1) / c;
    std::vector<Jac<F>> wsum(W, Jac<F>::identity());
    std::atomic<int> next(0);
    auto worker = [&] {
        std::vector<Jac<F>> buckets((size_t)1 << c);
        for (;;) {
            int w = next.fetch_add(1);
            if (w >= W) break;
            for (auto& b : buckets) b = Jac<F>::identity();
            const int bit0 = w * c;
            for (size_t i = 0; i < n; ++i) {
                const uint64_t* s = scalars + 4 * i;
                int limb = bit0 / 64, sh = bit0 % 64;
                uint64_t d = s[limb] >> sh;
                if (sh + c > 64 && limb + 1 < 4) d |= s[limb + 1] << (64 - sh);
                d &= ((uint64_t)1 << c) - 1;
                if (d) buckets[d] = buckets[d].add_mixed(bases[i]);
            }
            Jac<F> run = Jac<F>::identity(), acc = Jac<F>::identity();
            for (size_t b = ((size_t)1 << c) - 1; b >= 1; --b) {
                run = run.add(buckets[b]);
                acc = acc.add(run);
            }
            wsum[w] = acc;
        }
    };
    int nt = nthreads < 1 ? 1 : nthreads;
    if (nt > W) nt = W;
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(worker);
    worker();
    for (auto& x : th) x.join();
    Jac<F> total = Jac<F>::identity();
    for (int w = W - 1; w >= 0; --w) {
        for (int k = 0; k < c; ++k) total = total.dbl();
        total = total.add(wsum[w]);
    }
    return total;
}

template <class F>
static Jac<F> msm_naive(const Affine<F>* bases, const uint64_t* scalars, size_t n) {
    Jac<F> acc = Jac<F>::identity();
    for (size_t i = 0; i < n; ++i) acc = acc.add(Jac<F>::from_affine(bases[i]).mul(scalars + 4 * i, 4));
    return acc;
}

static void scalars_to_canon(const uint8_t* in, size_t n, int mont, std::vector<uint64_t>& out, int nthreads) {
    out.resize(4 * n);
    parallel_for(n, nthreads, [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) {
            if (mont) fr_load(in + 32 * i).to_canon(&out[4 * i]);
            else memcpy(&out[4 * i], in + 32 * i, 32);
        }
    });
}

// fixed-base windowed multiplication table (for CRS generation)
template <class F>
struct FixedBase {
    static constexpr int WB = 8;
    std::vector<Affine<F>> tab;  // [32 windows][255 entries]
    explicit FixedBase(const Jac<F>& g) {
        const int nw = 32;
        std::vector<Jac<F>> j((size_t)nw * 255);
        Jac<F> base = g;
        for (int w = 0; w < nw; ++w) {
            Jac<F> acc = base;
            for (int k = 1; k <= 255; ++k) {
                j[(size_t)w * 255 + k - 1] = acc;
                acc = acc.add(base);
            }
            base = acc;  // 256 * base
        }
        tab.resize(j.size());
        for (size_t i = 0; i < j.size(); ++i) tab[i] = j[i].to_affine();
    }
    Jac<F> mul(const Fr& k_mont) const {
        uint64_t k[4];
        k_mont.to_canon(k);
        Jac<F> r = Jac<F>::identity();
        for (int w = 0; w < 32; ++w) {
            unsigned d = (k[w / 8] >> ((w % 8) * 8)) & 0xff;
            if (d) r = r.add_mixed(tab[(size_t)w * 255 + d - 1]);
        }
        return r;
    }
};

static const uint8_t G1_GEN_X_BE[48] = {0x17,0xf1,0xd3,0xa7,0x31,0x97,0xd7,0x94,0x26,0x95,0x63,0x8c,0x4f,0xa9,0xac,0x0f,0xc3,0x68,0x8c,0x4f,0x97,0x74,0xb9,0x05,0xa1,0x4e,0x3a,0x3f,0x17,0x1b,0xac,0x58,0x6c,0x55,0xe8,0x3f,0xf9,0x7a,0x1a,0xef,0xfb,0x3a,0xf0,0x0a,0xdb,0x22,0xc6,0xbb};
static const uint8_t G1_GEN_Y_BE[48] = {0x08,0xb3,0xf4,0x81,0xe3,0xaa,0xa0,0xf1,0xa0,0x9e,0x30,0xed,0x74,0x1d,0x8a,0xe4,0xfc,0xf5,0xe0,0x95,0xd5,0xd0,0x0a,0xf6,0x00,0xdb,0x18,0xcb,0x2c,0x04,0xb3,0xed,0xd0,0x3c,0xc7,0x44,0xa2,0x88,0x8a,0xe4,0x0c,0xaa,0x23,0x29,0x46,0xc5,0xe7,0xe1};
static const uint8_t G2_GEN_X0_BE[48] = {0x02,0x4a,0xa2,0xb2,0xf0,0x8f,0x0a,0x91,0x26,0x08,0x05,0x27,0x2d,0xc5,0x10,0x51,0xc6,0xe4,0x7a,0xd4,0xfa,0x40,0x3b,0x02,0xb4,0x51,0x0b,0x64,0x7a,0xe3,0xd1,0x77,0x0b,0xac,0x03,0x26,0xa8,0x05,0xbb,0xef,0xd4,0x80,0x56,0xc8,0xc1,0x21,0xbd,0xb8};
static const uint8_t G2_GEN_X1_BE[48] = {0x13,0xe0,0x2b,0x60,0x52,0x71,0x9f,0x60,0x7d,0xac,0xd3,0xa0,0x88,0x27,0x4f,0x65,0x59,0x6b,0xd0,0xd0,0x99,0x20,0xb6,0x1a,0xb5,0xda,0x61,0xbb,0xdc,0x7f,0x50,0x49,0x33,0x4c,0xf1,0x12,0x13,0x94,0x5d,0x57,0xe5,0xac,0x7d,0x05,0x5d,0x04,0x2b,0x7e};
static const uint8_t G2_GEN_Y0_BE[48] = {0x0c,0xe5,0xd5,0x27,0x72,0x7d,0x6e,0x11,0x8c,0xc9,0xcd,0xc6,0xda,0x2e,0x35,0x1a,0xad,0xfd,0x9b,0xaa,0x8c,0xbd,0xd3,0xa7,0x6d,0x42,0x9a,0x69,0x51,0x60,0xd1,0x2c,0x92,0x3a,0xc9,0xcc,0x3b,0xac,0xa2,0x89,0xe1,0x93,0x54,0x86,0x08,0xb8,0x28,0x01};
static const uint8_t G2_GEN_Y1_BE[48] = {0x06,0x06,0xc4,0xa0,0x2e,0xa7,0x34,0xcc,0x32,0xac,0xd2,0xb0,0x2b,0xc2,0x8b,0x99,0xcb,0x3e,0x28,0x7e,0x85,0xa7,0x63,0xaf,0x26,0x74,0x92,0xab,0x57,0x2e,0x99,0xab,0x3f,0x37,0x0d,0x27,0x5c,0xec,0x1d,0xa1,0xaa,0xa9,0x07,0x5f,0xf0,0x5f,0x79,0xbe};

static Fp fp_from_be(const uint8_t* be) {
    uint64_t c[6];
    for (int i = 0; i < 6; ++i) {
        uint64_t w = 0;
        for (int j = 0; j < 8; ++j) w = (w << 8) | be[(5 - i) * 8 + j];
        c[i] = w;
    }
    return Fp::from_canon(c);
}
static G1Affine g1_generator() { return {fp_from_be(G1_GEN_X_BE), fp_from_be(G1_GEN_Y_BE), false}; }
static G2Affine g2_generator() {
    return {{fp_from_be(G2_GEN_X0_BE), fp_from_be(G2_GEN_X1_BE)}, {fp_from_be(G2_GEN_Y0_BE), fp_from_be(G2_GEN_Y1_BE)}, false};
}

struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};

}  // namespace orc

using namespace orc;

// =================================================================================================
// extern "C" surface (ctypes-friendly; all field/point data are byte buffers in the ABI formats)
// =================================================================================================
extern "C" {

void orc_init() { init_fields(); }

// ---- field ops (op: 0 add, 1 sub, 2 mul, 3 inv(a), 4 to_canon(a), 5 from_canon(a), 6 neg)
int orc_fr_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    init_fields();
    Fr x = fr_load(a), y = b ? fr_load(b) : Fr::zero(), r;
    switch (op) {
        case 0: r = x.add(y); break;
        case 1: r = x.sub(y); break;
        case 2: r = x.mul(y); break;
        case 3: r = x.inv(); break;
        case 4: x.to_canon(r.v); break;
        case 5: r = Fr::from_canon(x.v); break;
        case 6: r = x.neg(); break;
        default: return -1;
    }
    fr_store(out, r);
    return 0;
}
int orc_fp_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    init_fields();
    Fp x = fp_load(a), y = b ? fp_load(b) : Fp::zero(), r;
    switch (op) {
        case 0: r = x.add(y); break;
        case 1: r = x.sub(y); break;
        case 2: r = x.mul(y); break;
        case 3: r = x.inv(); break;
        case 4: x.to_canon(r.v); break;
        case 5: r = Fp::from_canon(x.v); break;
        case 6: r = x.neg(); break;
        default: return -1;
    }
    fp_store(out, r);
    return 0;
}

// ---- Poseidon
int orc_poseidon_batch(const uint8_t* in, uint32_t arity, uint64_t n, uint8_t* out, int nthreads) {
    init_fields();
    if (arity < 1 || arity > 16) return -1;
    poseidon_params(arity + 1);
    parallel_for(n, nthreads, [&](size_t lo, size_t hi, int) {
        Fr v[16];
        for (size_t i = lo; i < hi; ++i) {
            for (uint32_t k = 0; k < arity; ++k) v[k] = fr_load(in + (i * arity + k) * 32);
            fr_store(out + i * 32, poseidon_hash(v, arity));
        }
    });
    return 0;
}

// constants for width t, Montgomery form: rc[t*(rf+rp)] then mds[t*t]; returns count of elements
int orc_poseidon_params(uint32_t t, uint8_t* out, uint64_t cap_elems) {
    init_fields();
    if (t < 2 || t > 17) return -1;
    const PoseidonParams& P = poseidon_params(t);
    size_t tot = P.rc.size() + P.mds.size();
    if (!out) return (int)tot;
    if (cap_elems < tot) return -2;
    for (size_t i = 0; i < P.rc.size(); ++i) fr_store(out + 32 * i, P.rc[i]);
    for (size_t i = 0; i < P.mds.size(); ++i) fr_store(out + 32 * (P.rc.size() + i), P.mds[i]);
    return (int)tot;
}

// ---- dense 4-ary tree; nodes_opt (if non-NULL) gets ((4^log4 - 1)/3) * 32 bytes in heap order
int orc_merkle4_root(const uint8_t* leaves, uint32_t log4, uint8_t* root, uint8_t* nodes_opt, int nthreads) {
    init_fields();
    poseidon_params(5);
    size_t n = (size_t)1 << (2 * log4);
    if (log4 == 0) { memcpy(root, leaves, 32); return 0; }
    std::vector<Fr> cur(n), nxt;
    for (size_t i = 0; i < n; ++i) cur[i] = fr_load(leaves + 32 * i);
    for (int depth = (int)log4 - 1; depth >= 0; --depth) {
        size_t cnt = (size_t)1 << (2 * depth);
        nxt.resize(cnt);
        parallel_for(cnt, nthreads, [&](size_t lo, size_t hi, int) {
            for (size_t i = lo; i < hi; ++i) nxt[i] = poseidon_hash(&cur[4 * i], 4);
        });
        if (nodes_opt) {
            size_t base = (cnt - 1) / 3;  // (4^depth - 1)/3
            for (size_t i = 0; i < cnt; ++i) fr_store(nodes_opt + 32 * (base + i), nxt[i]);
        }
        cur.swap(nxt);
    }
    fr_store(root, cur[0]);
    return 0;
}

// ---- NTT in place, Montgomery elements
int orc_ntt(uint8_t* data, uint32_t log_n, int inverse, int coset, int nthreads) {
    init_fields();
    if (log_n > 32) return -1;
    size_t n = (size_t)1 << log_n;
    std::vector<Fr> a(n);
    memcpy(a.data(), data, n * 32);
    ntt_inplace(a, log_n, inverse != 0, coset != 0, nthreads);
    memcpy(data, a.data(), n * 32);
    return 0;
}

// ---- MSM
int orc_msm_g1(const uint8_t* bases, const uint8_t* scalars, uint64_t n, int scalars_mont, uint8_t* out97,
               int nthreads, int naive) {
    init_fields();
    std::vector<G1Affine> b(n);
    for (size_t i = 0; i < n; ++i) b[i] = g1_load96(bases + 96 * i);
    std::vector<uint64_t> s;
    scalars_to_canon(scalars, n, scalars_mont, s, nthreads);
    G1 r = naive ? msm_naive<Fp>(b.data(), s.data(), n) : msm_pippenger<Fp>(b.data(), s.data(), n, nthreads);
    g1_store97(out97, r.to_affine());
    return 0;
}
int orc_msm_g2(const uint8_t* bases, const uint8_t* scalars, uint64_t n, int scalars_mont, uint8_t* out193,
               int nthreads, int naive) {
    init_fields();
    std::vector<G2Affine> b(n);
    for (size_t i = 0; i < n; ++i) b[i] = g2_load192(bases + 192 * i);
    std::vector<uint64_t> s;
    scalars_to_canon(scalars, n, scalars_mont, s, nthreads);
    G2 r = naive ? msm_naive<Fp2>(b.data(), s.data(), n) : msm_pippenger<Fp2>(b.data(), s.data(), n, nthreads);
    g2_store193(out193, r.to_affine());
    return 0;
}

// ---- point utilities
// out[i] = k_i * G, k_i = 64-bit SplitMix64(seed) stream value | 1 ... (i-th draw); affine 96/192 B
int orc_g1_bases(uint64_t seed, uint64_t start, uint64_t n, uint8_t* out, int nthreads) {
    init_fields();
    G1 g = G1::from_affine(g1_generator());
    parallel_for(n, nthreads, [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) {
            SplitMix64 r{seed + 0x632BE59BD9B4E019ull * (start + i)};
            uint64_t k = r.next() | 1;
            G1Affine a = g.mul(&k, 1).to_affine();
            fp_store(out + 96 * i, a.x);
            fp_store(out + 96 * i + 48, a.y);
        }
    });
    return 0;
}
int orc_g2_bases(uint64_t seed, uint64_t start, uint64_t n, uint8_t* out, int nthreads) {
    init_fields();
    G2 g = G2::from_affine(g2_generator());
    parallel_for(n, nthreads, [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) {
            SplitMix64 r{seed + 0x632BE59BD9B4E019ull * (start + i)};
            uint64_t k = r.next() | 1;
            G2Affine a = g.mul(&k, 1).to_affine();
            uint8_t tmp[193];
            g2_store193(tmp, a);
            memcpy(out + 192 * i, tmp, 192);
        }
    });
    return 0;
}
int orc_g1_generator(uint8_t* out97) { init_fields(); g1_store97(out97, g1_generator()); return 0; }
int orc_g2_generator(uint8_t* out193) { init_fields(); g2_store193(out193, g2_generator()); return 0; }
// k canonical 32 B LE
int orc_g1_mul(const uint8_t* p97, const uint8_t* k, uint8_t* out97) {
    init_fields();
    uint64_t kk[4];
    memcpy(kk, k, 32);
    g1_store97(out97, G1::from_affine(g1_load97(p97)).mul(kk, 4).to_affine());
    return 0;
}
int orc_g2_mul(const uint8_t* p193, const uint8_t* k, uint8_t* out193) {
    init_fields();
    uint64_t kk[4];
    memcpy(kk, k, 32);
    g2_store193(out193, G2::from_affine(g2_load193(p193)).mul(kk, 4).to_affine());
    return 0;
}
int orc_g1_add(const uint8_t* a, const uint8_t* b, uint8_t* out) {
    init_fields();
    g1_store97(out, G1::from_affine(g1_load97(a)).add(G1::from_affine(g1_load97(b))).to_affine());
    return 0;
}
int orc_g2_add(const uint8_t* a, const uint8_t* b, uint8_t* out) {
    init_fields();
    g2_store193(out, G2::from_affine(g2_load193(a)).add(G2::from_affine(g2_load193(b))).to_affine());
    return 0;
}
int orc_g1_on_curve(const uint8_t* p97) {
    init_fields();
    G1Affine a = g1_load97(p97);
    if (a.inf) return 1;
    Fp four = Fp::from_u64(4);
    return a.y.sqr() == a.x.sqr().mul(a.x).add(four) ? 1 : 0;
}
int orc_g2_on_curve(const uint8_t* p193) {
    init_fields();
    G2Affine a = g2_load193(p193);
    if (a.inf) return 1;
    Fp2 b = {Fp::from_u64(4), Fp::from_u64(4)};
    return a.y.sqr() == a.x.sqr().mul(a.x).add(b) ? 1 : 0;
}

// =================================================================================================
// R1CS (CSR) + Groth16.  Variables: inputs 0..n_in-1 (0 = ONE), aux n_in..n_in+n_aux-1.  The CSR
// rows already contain bellman's trailing `input_i * 0 = 0` constraints (host generator appends).
// =================================================================================================
struct orc_csr {
    uint64_t n_rows;
    const uint32_t* row_ptr;  // n_rows + 1
    const uint32_t* col;      // nnz
    const uint8_t* val;       // nnz * 32 (Montgomery)
};

static void csr_eval(const orc_csr* m, const std::vector<Fr>& z, uint8_t* out, int nthreads) {
    parallel_for(m->n_rows, nthreads, [&](size_t lo, size_t hi, int) {
        for (size_t r = lo; r < hi; ++r) {
            Fr acc = Fr::zero();
            for (uint32_t k = m->row_ptr[r]; k < m->row_ptr[r + 1]; ++k)
                acc = acc.add(fr_load(m->val + 32 * (size_t)k).mul(z[m->col[k]]));
            fr_store(out + 32 * r, acc);
        }
    });
}

// z = inputs || aux (Montgomery).  Outputs az,bz,cz (n_rows*32 each).
int orc_r1cs_eval(const orc_csr* A, const orc_csr* B, const orc_csr* C, const uint8_t* z, uint64_t n_vars,
                  uint8_t* az, uint8_t* bz, uint8_t* cz, int nthreads) {
    init_fields();
    std::vector<Fr> zz(n_vars);
    memcpy(zz.data(), z, 32 * n_vars);
    csr_eval(A, zz, az, nthreads);
    csr_eval(B, zz, bz, nthreads);
    csr_eval(C, zz, cz, nthreads);
    return 0;
}

// density[v] = 1 iff variable v appears in any row of the matrix
int orc_r1cs_density(const orc_csr* M, uint64_t n_vars, uint8_t* density) {
    memset(density, 0, n_vars);
    for (uint64_t k = 0; k < M->row_ptr[M->n_rows]; ++k) density[M->col[k]] = 1;
    return 0;
}

// h coefficients (m-1) from evaluation vectors (n_rows each), as bellman create_proof does
static std::vector<Fr> h_coeffs(const uint8_t* az, const uint8_t* bz, const uint8_t* cz, uint64_t n_rows,
                                int log_m, int nthreads) {
    size_t m = (size_t)1 << log_m;
    std::vector<Fr> a(m, Fr::zero()), b(m, Fr::zero()), c(m, Fr::zero());
    memcpy(a.data(), az, 32 * n_rows);
    memcpy(b.data(), bz, 32 * n_rows);
    memcpy(c.data(), cz, 32 * n_rows);
    for (auto* v : {&a, &b, &c}) {
        ntt_inplace(*v, log_m, true, false, nthreads);
        ntt_inplace(*v, log_m, false, true, nthreads);
    }
    Fr zinv = fr_pow_u64(Fr::from_u64(7), m).sub(Fr::one()).inv();
    parallel_for(m, nthreads, [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) a[i] = a[i].mul(b[i]).sub(c[i]).mul(zinv);
    });
    ntt_inplace(a, log_m, true, true, nthreads);
    a.resize(m - 1);
    return a;
}

int orc_groth16_h(const uint8_t* az, const uint8_t* bz, const uint8_t* cz, uint64_t n_rows, uint32_t log_m,
                  uint8_t* h_out /*(m-1)*32*/, int nthreads) {
    init_fields();
    std::vector<Fr> h = h_coeffs(az, bz, cz, n_rows, log_m, nthreads);
    memcpy(h_out, h.data(), 32 * h.size());
    return 0;
}

struct orc_params {
    uint32_t n_in, n_aux, log_m;
    uint32_t n_a, n_b;         // lengths of a / b_g1 / b_g2 (dense variables only)
    const uint8_t* vk;          // alpha_g1|beta_g1|beta_g2|gamma_g2|delta_g1|delta_g2 (97,97,193,193,97,193)
    const uint8_t* h;           // (m-1)*96
    const uint8_t* l;           // n_aux*96
    const uint8_t* a;           // n_a*96
    const uint8_t* b_g1;        // n_b*96
    const uint8_t* b_g2;        // n_b*192
    const uint8_t* a_density;   // n_in+n_aux bytes
    const uint8_t* b_density;   // n_in+n_aux bytes
};

int orc_groth16_prove(const orc_params* P, const uint8_t* z /*(n_in+n_aux)*32 Montgomery*/,
                      const uint8_t* az, const uint8_t* bz, const uint8_t* cz, uint64_t n_rows,
                      const uint8_t* r32 /*Montgomery*/, const uint8_t* s32, uint8_t* proof387, int nthreads) {
    init_fields();
    const size_t nv = (size_t)P->n_in + P->n_aux, m = (size_t)1 << P->log_m;
    std::vector<Fr> h = h_coeffs(az, bz, cz, n_rows, P->log_m, nthreads);
    auto canon = [&](const Fr* v, size_t n) {
        std::vector<uint64_t> o(4 * n);
        for (size_t i = 0; i < n; ++i) v[i].to_canon(&o[4 * i]);
        return o;
    };
    auto load_g1 = [&](const uint8_t* p, size_t n) {
        std::vector<G1Affine> v(n);
        for (size_t i = 0; i < n; ++i) v[i] = g1_load96(p + 96 * i);
        return v;
    };
    std::vector<Fr> zz(nv);
    memcpy(zz.data(), z, 32 * nv);
    std::vector<Fr> za, zb;
    for (size_t v = 0; v < nv; ++v) {
        if (P->a_density[v]) za.push_back(zz[v]);
        if (P->b_density[v]) zb.push_back(zz[v]);
    }
    if (za.size() != P->n_a || zb.size() != P->n_b) return -2;
    auto hb = load_g1(P->h, m - 1), lb = load_g1(P->l, P->n_aux), ab = load_g1(P->a, P->n_a), b1 = load_g1(P->b_g1, P->n_b);
    std::vector<G2Affine> b2(P->n_b);
    for (size_t i = 0; i < P->n_b; ++i) b2[i] = g2_load192(P->b_g2 + 192 * i);
    auto hs = canon(h.data(), h.size()), ls = canon(zz.data() + P->n_in, P->n_aux), as = canon(za.data(), za.size()),
         bs = canon(zb.data(), zb.size());
    G1 H = msm_pippenger<Fp>(hb.data(), hs.data(), hb.size(), nthreads);
    G1 L = msm_pippenger<Fp>(lb.data(), ls.data(), lb.size(), nthreads);
    G1 A = msm_pippenger<Fp>(ab.data(), as.data(), ab.size(), nthreads);
    G1 B1 = msm_pippenger<Fp>(b1.data(), bs.data(), b1.size(), nthreads);
    G2 B2 = msm_pippenger<Fp2>(b2.data(), bs.data(), b2.size(), nthreads);
    const uint8_t* vk = P->vk;
    G1 alpha = G1::from_affine(g1_load97(vk)), beta1 = G1::from_affine(g1_load97(vk + 97));
    G2 beta2 = G2::from_affine(g2_load193(vk + 194));
    G1 delta1 = G1::from_affine(g1_load97(vk + 194 + 386));
    G2 delta2 = G2::from_affine(g2_load193(vk + 194 + 386 + 97));
    Fr r = fr_load(r32), s = fr_load(s32), rs = r.mul(s);
    uint64_t rc[4], sc[4], rsc[4];
    r.to_canon(rc); s.to_canon(sc); rs.to_canon(rsc);
    G1 ga = delta1.mul(rc, 4).add(alpha).add(A);
    G2 gb = delta2.mul(sc, 4).add(beta2).add(B2);
    G1 gc = delta1.mul(rsc, 4).add(alpha.mul(sc, 4)).add(beta1.mul(rc, 4)).add(A.mul(sc, 4)).add(B1.mul(rc, 4)).add(H).add(L);
    g1_store97(proof387, ga.to_affine());
    g2_store193(proof387 + 97, gb.to_affine());
    g1_store97(proof387 + 290, gc.to_affine());
    return 0;
}

// CRS generation (bellman generate_parameters layout).  toxic = tau, alpha, beta, gamma, delta as
// 5 x 32 B Montgomery.  Output buffers sized by the caller: vk 870 B; ic n_in*97; h (m-1)*96;
// l n_aux*96; a n_a*96; b_g1 n_b*96; b_g2 n_b*192 where n_a/n_b come from the densities.
int orc_groth16_setup(const orc_csr* A, const orc_csr* B, const orc_csr* C, uint32_t n_in, uint32_t n_aux,
                      uint32_t log_m, const uint8_t* toxic, const uint8_t* a_density, const uint8_t* b_density,
                      uint8_t* vk870, uint8_t* ic, uint8_t* h, uint8_t* l, uint8_t* a, uint8_t* b_g1, uint8_t* b_g2,
                      int nthreads) {
    init_fields();
    const size_t nv = (size_t)n_in + n_aux, m = (size_t)1 << log_m;
    if (A->n_rows > m) return -1;
    Fr tau = fr_load(toxic), alpha = fr_load(toxic + 32), beta = fr_load(toxic + 64), gamma = fr_load(toxic + 96),
       delta = fr_load(toxic + 128);
    std::vector<Fr> lag(m);
    lag[0] = Fr::one();
    for (size_t i = 1; i < m; ++i) lag[i] = lag[i - 1].mul(tau);
    Fr tau_m = lag[m - 1].mul(tau);
    ntt_inplace(lag, log_m, true, false, nthreads);  // L_j(tau)
    std::vector<Fr> at(nv, Fr::zero()), bt(nv, Fr::zero()), ct(nv, Fr::zero());
    auto accum = [&](const orc_csr* M, std::vector<Fr>& dst) {
        for (size_t r = 0; r < M->n_rows; ++r)
            for (uint32_t k = M->row_ptr[r]; k < M->row_ptr[r + 1]; ++k)
                dst[M->col[k]] = dst[M->col[k]].add(fr_load(M->val + 32 * (size_t)k).mul(lag[r]));
    };
    accum(A, at); accum(B, bt); accum(C, ct);
    Fr zt = tau_m.sub(Fr::one()), dinv = delta.inv(), ginv = gamma.inv();
    FixedBase<Fp> T1(G1::from_affine(g1_generator()));
    FixedBase<Fp2> T2(G2::from_affine(g2_generator()));
    g1_store97(vk870, T1.mul(alpha).to_affine());
    g1_store97(vk870 + 97, T1.mul(beta).to_affine());
    g2_store193(vk870 + 194, T2.mul(beta).to_affine());
    g2_store193(vk870 + 387, T2.mul(gamma).to_affine());
    g1_store97(vk870 + 580, T1.mul(delta).to_affine());
    g2_store193(vk870 + 677, T2.mul(delta).to_affine());
    std::atomic<int> bad(0);
    auto put96 = [&](uint8_t* dst, const G1& p) {
        G1Affine q = p.to_affine();
        if (q.inf) bad = 1;
        fp_store(dst, q.x); fp_store(dst + 48, q.y);
    };
    for (uint32_t v = 0; v < n_in; ++v)
        g1_store97(ic + 97 * v, T1.mul(beta.mul(at[v]).add(alpha.mul(bt[v])).add(ct[v]).mul(ginv)).to_affine());
    Fr coeff = zt.mul(dinv);
    std::vector<Fr> hp(m - 1);
    { Fr x = coeff; for (size_t i = 0; i + 1 < m; ++i) { hp[i] = x; x = x.mul(tau); } }
    parallel_for(m - 1, nthreads, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; ++i) put96(h + 96 * i, T1.mul(hp[i])); });
    parallel_for(n_aux, nthreads, [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) {
            size_t v = n_in + i;
            put96(l + 96 * i, T1.mul(beta.mul(at[v]).add(alpha.mul(bt[v])).add(ct[v]).mul(dinv)));
        }
    });
    std::vector<uint32_t> ia, ib;
    for (size_t v = 0; v < nv; ++v) { if (a_density[v]) ia.push_back(v); if (b_density[v]) ib.push_back(v); }
    parallel_for(ia.size(), nthreads, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; ++i) put96(a + 96 * i, T1.mul(at[ia[i]])); });
    parallel_for(ib.size(), nthreads, [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) {
            put96(b_g1 + 96 * i, T1.mul(bt[ib[i]]));
            G2Affine q = T2.mul(bt[ib[i]]).to_affine();
            if (q.inf) bad = 1;
            uint8_t tmp[193];
            g2_store193(tmp, q);
            memcpy(b_g2 + 192 * i, tmp, 192);
        }
    });
    return bad ? -3 : 0;
}

}  // extern "C"
