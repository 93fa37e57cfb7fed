// Host-side ZkScalar helpers, Poseidon, SHA3-256 and Jubjub/EdDSA (see host_zk.h for the mapping to
// the reference's src/zk and src/crypto/jubjub).
#include "host_fr64.h"
#include "host_fr_ifma.h"
#include "host_zk.h"

#include <atomic>
#include <cmath>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bzk_poseidon_opt.h"

#include <mutex>

namespace bzk {

// ------------------------------------------------------------------------------------------------
// ZkScalar
// ------------------------------------------------------------------------------------------------
ZkScalar ZkScalar::from_le_bytes_mod(const uint8_t* b, size_t len) {
    const ZkScalar k256 = ZkScalar::from_u64(256);
    ZkScalar acc;
    for (size_t i = len; i-- > 0;) acc = acc * k256 + ZkScalar::from_u64(b[i]);
    return acc;
}

ZkScalar ZkScalar::from_dec(const char* s) {
    const ZkScalar ten = ZkScalar::from_u64(10);
    ZkScalar acc;
    for (; *s; ++s) acc = acc * ten + ZkScalar::from_u64((uint64_t)(*s - '0'));
    return acc;
}

ZkScalar ZkScalar::pow(const uint32_t* e, int nlimbs) const {
    ZkScalar r = ZkScalar::one();
    for (int i = nlimbs * 32 - 1; i >= 0; --i) {
        r = r.square();
        if ((e[i >> 5] >> (i & 31)) & 1) r = r * *this;
    }
    return r;
}

// Tonelli-Shanks with r - 1 = 2^32 * t, non-residue generator 7 (src/zk/mod.rs:204)
bool ZkScalar::sqrt(ZkScalar* out) const {
    if (is_zero()) {
        *out = ZkScalar::zero();
        return true;
    }
    uint32_t rm1[8];
    {
        uint64_t borrow = 1;
        for (int i = 0; i < 8; ++i) {
            uint64_t d = (uint64_t)FrParams::MOD[i] - borrow;
            rm1[i] = (uint32_t)d;
            borrow = (d >> 63) & 1;
        }
    }
    uint32_t t[8] = {0}, th[8] = {0};  // t = (r-1) >> 32 ; th = (t+1)/2 = (t >> 1) + 1 (t odd)
    for (int i = 0; i < 7; ++i) t[i] = rm1[i + 1];
    for (int i = 0; i < 8; ++i) th[i] = (t[i] >> 1) | (i < 7 ? t[i + 1] << 31 : 0);
    {
        uint64_t c = 1;
        for (int i = 0; i < 8; ++i) {
            c += th[i];
            th[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    ZkScalar z = ZkScalar::from_u64(7).pow(t, 8);  // 2^32-th root of unity
    ZkScalar x = pow(th, 8);                       // a^((t+1)/2)
    ZkScalar b = pow(t, 8);                        // a^t
    int m = 32;
    while (b != ZkScalar::one()) {
        int k = 0;
        ZkScalar b2 = b;
        while (b2 != ZkScalar::one()) {
            b2 = b2.square();
            if (++k >= m) return false;  // non-residue
        }
        ZkScalar w = z;
        for (int i = 0; i < m - k - 1; ++i) w = w.square();
        z = w.square();
        b = b * z;
        x = x * w;
        m = k;
    }
    *out = x;
    return true;
}

// ------------------------------------------------------------------------------------------------
// Poseidon (host)
// ------------------------------------------------------------------------------------------------
// Evaluated in the sparse-partial-round form (bzk_poseidon_opt.h: same function, 2T-1 instead of T^2 products in each
// of the 56/57 partial rounds); the constants are derived once per width and checked against the plain round function
// below on a probe vector.  `poseidon_hash_plain` stays: it IS the reference's algorithm (src/zk/poseidon/mod.rs:24-84)
// and what the circuit gadget's witness follows round by round.
ZkScalar poseidon_hash_plain(const ZkScalar* vals, int arity) {
    const int t = arity + 1;
    PoseidonHostParams P = poseidon_host_params(t);
    Fr st[17], nw[17];
    st[0] = Fr::zero();
    for (int i = 0; i < arity; ++i) st[i + 1] = vals[i].v;
    int off = 0;
    for (int rnd = 0; rnd < P.rf + P.rp; ++rnd) {
        for (int i = 0; i < t; ++i) st[i] = fe_add<FrParams>(st[i], P.rc[off + i]);
        off += t;
        const bool full = rnd < P.rf / 2 || rnd >= P.rf / 2 + P.rp;
        const int ns = full ? t : 1;
        for (int i = 0; i < ns; ++i) {
            Fr x2 = hfr::sqr(st[i]);
            st[i] = hfr::mul(hfr::sqr(x2), st[i]);
        }
        for (int j = 0; j < t; ++j) {
            Fr acc = Fr::zero();
            for (int k = 0; k < t; ++k) acc = fe_add<FrParams>(acc, hfr::mul(P.mds[j * t + k], st[k]));
            nw[j] = acc;
        }
        for (int i = 0; i < t; ++i) st[i] = nw[i];
    }
    return ZkScalar(st[1]);
}

namespace {
struct SparseConsts {
    bool ready = false, usable = false;
    int rf = 0, rp = 0;
    std::vector<Fr> flat;  // layout of poseidon_optimize()
    hfr::MdsTable mds_tab, dmat_tab;  // the two dense blocks in the lane layout of host_fr_ifma.h
};
std::mutex g_sparse_mu;
SparseConsts g_sparse[18];

inline Fr sbox5(const Fr& x) {  // 64-bit-limb products (host_fr64.h): canonical, the values of fe_sqr / fe_mul
    const Fr x2 = hfr::mul(x, x);
    return hfr::mul(hfr::mul(x2, x2), x);
}
Fr hash_sparse(const SparseConsts& S, int t, const Fr* in) {
    const int half = S.rf / 2;
    const Fr* rc1 = S.flat.data();
    const Fr* pre = rc1 + (size_t)half * t;
    const Fr* part = pre + t;
    const Fr* dmat = part + (size_t)S.rp * 2 * t;
    const Fr* rc2 = dmat + (size_t)(t - 1) * (t - 1);
    const Fr* mds = rc2 + (size_t)half * t;
    Fr st[17], nw[17];
    st[0] = Fr::zero();
    for (int i = 1; i < t; ++i) st[i] = in[i - 1];
    auto full_round = [&](const Fr* rc) {
        for (int i = 0; i < t; ++i) st[i] = sbox5(fe_add<FrParams>(st[i], rc[i]));
        hfr::mds_mul(S.mds_tab, st, nw);  // t rows side by side on AVX-512 IFMA where the CPU has it, else one dot product per row
        for (int i = 0; i < t; ++i) st[i] = nw[i];
    };
    (void)mds;
    for (int r = 0; r < half; ++r) full_round(rc1 + (size_t)r * t);
    for (int i = 0; i < t; ++i) st[i] = fe_add<FrParams>(st[i], pre[i]);
    for (int i = 0; i < S.rp; ++i) {
        const Fr* c = part + (size_t)i * 2 * t;  // s_i, row0[t], what[t-1]
        st[0] = fe_add<FrParams>(sbox5(st[0]), c[0]);
        const Fr n0 = hfr::dot(c + 1, st, t);
        for (int j = 1; j < t; ++j) st[j] = fe_add<FrParams>(st[j], hfr::mul(c[t + j], st[0]));
        st[0] = n0;
    }
    if (t > 2) {
        hfr::mds_mul(S.dmat_tab, st + 1, nw);
    } else {
        nw[0] = hfr::dot(dmat, st + 1, 1);
    }
    for (int j = 1; j < t; ++j) st[j] = nw[j - 1];
    for (int r = 0; r < half; ++r) full_round(rc2 + (size_t)r * t);
    return st[1];
}
std::atomic<bool> g_sparse_ready[18];
const SparseConsts& sparse_consts(int t) {
    SparseConsts& S = g_sparse[t];
    if (g_sparse_ready[t].load(std::memory_order_acquire)) return S;  // no lock on the path every host hash takes
    std::lock_guard<std::mutex> lk(g_sparse_mu);
    if (S.ready) return S;
    PoseidonHostParams P = poseidon_host_params(t);
    S.rf = P.rf;
    S.rp = P.rp;
    std::vector<Fr> rc(P.rc, P.rc + (size_t)t * (P.rf + P.rp)), mds(P.mds, P.mds + (size_t)t * t);
    S.usable = poseidon_optimize(t, P.rf, P.rp, rc, mds, S.flat);
    if (S.usable) {
        const int half = S.rf / 2;
        const Fr* dmat = S.flat.data() + (size_t)half * t + t + (size_t)S.rp * 2 * t;
        const Fr* mds_flat = dmat + (size_t)(t - 1) * (t - 1) + (size_t)half * t;
        hfr::mds_table_build(S.mds_tab, mds_flat, t);
        if (t > 2) hfr::mds_table_build(S.dmat_tab, dmat, t - 1);
    }
    if (S.usable) {  // probe: the derived constants must reproduce the plain function
        ZkScalar probe[16];
        for (int k = 0; k < t - 1; ++k) probe[k] = ZkScalar(hfr::mul(P.mds[k % (t * t)], P.rc[k]));
        Fr in[16];
        for (int k = 0; k < t - 1; ++k) in[k] = probe[k].v;
        S.usable = hash_sparse(S, t, in).equals(poseidon_hash_plain(probe, t - 1).v);
    }
    S.ready = true;
    g_sparse_ready[t].store(true, std::memory_order_release);
    return S;
}
std::mutex g_mds_mu;
hfr::MdsTable g_mds_tab[18];
std::atomic<bool> g_mds_ready[18];
}  // namespace

int host_default_threads() {
    static const int n = [] {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        double quota = 0;
        {
            std::ifstream f("/sys/fs/cgroup/cpu.max");
            std::string q;
            double per = 0;
            if (f >> q >> per) {
                if (q != "max" && per > 0) quota = atof(q.c_str()) / per;
            } else {
                std::ifstream a("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), b("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
                double qq = 0;
                if ((a >> qq) && (b >> per) && qq > 0 && per > 0) quota = qq / per;
            }
        }
        if (const char* e = getenv("BZK_HOST_THREADS"))  // explicit override (A/B; hosts whose quota is not visible in the cgroup files)
            if (atoi(e) > 0) return atoi(e);
        if (quota > 0) return (int)std::max(1.0, std::min((double)hw, std::ceil(quota)));
        return (int)hw;
    }();
    return n;
}

// poseidon_host_params takes a lock on every call (poseidon.hip); the witness generator asks once per hash from several threads
PoseidonHostParams poseidon_host_params_cached(int t) {
    static PoseidonHostParams cache[18];
    static std::atomic<bool> ready[18];
    if (ready[t].load(std::memory_order_acquire)) return cache[t];
    std::lock_guard<std::mutex> lk(g_mds_mu);
    if (!ready[t].load(std::memory_order_relaxed)) {
        cache[t] = poseidon_host_params(t);
        ready[t].store(true, std::memory_order_release);
    }
    return cache[t];
}

const hfr::MdsTable& poseidon_mds_table(int t) {
    if (g_mds_ready[t].load(std::memory_order_acquire)) return g_mds_tab[t];
    std::lock_guard<std::mutex> lk(g_mds_mu);
    if (!g_mds_ready[t].load(std::memory_order_relaxed)) {
        hfr::mds_table_build(g_mds_tab[t], poseidon_host_params(t).mds, t);
        g_mds_ready[t].store(true, std::memory_order_release);
    }
    return g_mds_tab[t];
}

ZkScalar poseidon_hash(const ZkScalar* vals, int arity) {
    const int t = arity + 1;
    const SparseConsts& S = sparse_consts(t);
    if (!S.usable) return poseidon_hash_plain(vals, arity);  // the plain form is the definition; never hit for the reference's parameters
    Fr in[16];
    for (int i = 0; i < arity; ++i) in[i] = vals[i].v;
    return ZkScalar(hash_sparse(S, t, in));
}

// ------------------------------------------------------------------------------------------------
// SHA3-256 (FIPS 202)
// ------------------------------------------------------------------------------------------------
static void keccak_f(uint64_t s[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
        0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
        0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
        0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    auto rol = [](uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; };
    for (int rnd = 0; rnd < 24; ++rnd) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) s[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(s[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= RC[rnd];
    }
}

void sha3_256(const uint8_t* data, size_t len, uint8_t out[32]) {
    const size_t rate = 136;
    uint64_t s[25] = {0};
    uint8_t block[136];
    size_t off = 0;
    auto absorb = [&](const uint8_t* blk) {
        for (size_t i = 0; i < rate / 8; ++i) {
            uint64_t w = 0;
            for (int j = 7; j >= 0; --j) w = (w << 8) | blk[8 * i + j];
            s[i] ^= w;
        }
        keccak_f(s);
    };
    while (len - off >= rate) {
        absorb(data + off);
        off += rate;
    }
    memset(block, 0, rate);
    if (len - off) memcpy(block, data + off, len - off);  // data may be null for the empty message
    block[len - off] ^= 0x06;
    block[rate - 1] ^= 0x80;
    absorb(block);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(s[i] >> (8 * j));
}

// ------------------------------------------------------------------------------------------------
// Jubjub
// ------------------------------------------------------------------------------------------------
const ZkScalar& jubjub_d() {
    static const ZkScalar d = ZkScalar::from_dec("19257038036680949359750312669786877991949435402254120286184196891950884077233");
    return d;
}
const PointAffine& jubjub_base() {
    static const PointAffine b = {ZkScalar::from_dec("28867639725710769449342053336011988556061781325688749245863888315629457631946"),
                                  ZkScalar::from_u64(18)};
    return b;
}
const PointAffine& jubjub_base_cofactor() {
    static const PointAffine b = jubjub_base().multiply(ZkScalar::from_u64(8));
    return b;
}

PointAffine jubjub_decompress(const ZkScalar& x, bool y_is_odd) {
    // y = sqrt((1 - A x^2) / (1 - D x^2)) with A = -1, sign chosen by parity
    ZkScalar xx = x.square();
    ZkScalar y;
    ((ZkScalar::one() - jubjub_d() * xx).invert() * (ZkScalar::one() + xx)).sqrt(&y);
    if (y.is_odd() != y_is_odd) y = -y;
    return {x, y};
}
const PointAffine& jubjub_default_pubkey() {
    static const PointAffine p = jubjub_decompress(ZkScalar::zero(), false);
    return p;
}

bool PointAffine::is_on_curve() const {
    ZkScalar xx = x.square(), yy = y.square();
    return yy - xx == ZkScalar::one() + jubjub_d() * xx * yy;
}

PointAffine PointAffine::dbl() const {  // curve.rs:48-57 (A = -1)
    ZkScalar xx = x.square(), yy = y.square();
    // (A x^2 + y^2)^-1 and (2 - A x^2 - y^2)^-1 from ONE inversion of their product
    ZkScalar d1 = yy - xx, d2 = ZkScalar::one() + ZkScalar::one() + xx - yy;
    ZkScalar pi = (d1 * d2).invert();
    ZkScalar dx = pi * d2, dy = pi * d1;
    return {((x * y) * dx).dbl(), (yy + xx) * dy};
}

void PointAffine::add_assign(const PointAffine& o) {  // curve.rs:19-36
    if (*this == o) {
        *this = dbl();
        return;
    }
    ZkScalar dxy = jubjub_d() * x * o.x * y * o.y;
    ZkScalar e1 = ZkScalar::one() + dxy, e2 = ZkScalar::one() - dxy;
    ZkScalar pi = (e1 * e2).invert();  // one inversion for both denominators
    ZkScalar xi = pi * e2, yi = pi * e1;
    ZkScalar nx = (x * o.y + y * o.x) * xi, ny = (y * o.y + x * o.x) * yi;
    x = nx;
    y = ny;
}

namespace {
struct Proj {
    ZkScalar X, Y, Z;
    static Proj zero() { return {ZkScalar::zero(), ZkScalar::one(), ZkScalar::zero()}; }
    bool is_zero() const { return Z.is_zero(); }
    Proj dbl() const {  // curve.rs:127-138
        if (is_zero()) return zero();
        ZkScalar b = (X + Y).square(), c = X.square(), d = Y.square();
        ZkScalar e = -c, f = e + d, h = Z.square(), j = f - h.dbl();
        return {(b - c - d) * j, f * (e - d), f * j};
    }
    void add_assign(const Proj& o) {  // curve.rs:91-117; the unified formula is complete on Jubjub
        if (is_zero()) { *this = o; return; }   // (d is a non-square), so the reference's
        if (o.is_zero()) return;                 // equal-points detour to double() is not needed
        ZkScalar a = Z * o.Z, b = a.square(), c = X * o.X, d = Y * o.Y;
        ZkScalar e = jubjub_d() * c * d, f = b - e, g = b + e;
        ZkScalar nx = a * f * ((X + Y) * (o.X + o.Y) - c - d), ny = a * g * (d + c);
        X = nx; Y = ny; Z = f * g;
    }
    PointAffine to_affine() const {
        if (is_zero()) return PointAffine::zero();
        ZkScalar zi = Z.invert();
        return {X * zi, Y * zi};
    }
};
}  // namespace

PointAffine PointAffine::multiply(const ZkScalar& k) const {  // curve.rs:58-68
    Proj r = Proj::zero(), self = {x, y, ZkScalar::one()};
    uint32_t c[8];
    k.to_canonical(c);
    for (int i = 255; i >= 0; --i) {
        r = r.dbl();
        if ((c[i >> 5] >> (i & 31)) & 1) r.add_assign(self);
    }
    return r.to_affine();
}

// All intermediate points of the circuit's double-and-add ladder (eddsa/mod.rs:174-236), MSB first:
//   R_0 = bit_0 ? base : (0, 1);   D_i = 2 R_{i-1};   A_i = D_i + base;   R_i = bit_i ? A_i : D_i
// computed in projective coordinates with ONE field inversion for the whole ladder (Montgomery's trick) - the
// per-step affine formulas of the reference cost two inversions each.
void jubjub_ladder(const PointAffine& base, const std::vector<bool>& bits, std::vector<PointAffine>& dbls, std::vector<PointAffine>& adds) {
    const size_t n = bits.size();
    dbls.assign(n, PointAffine::zero());
    adds.assign(n, PointAffine::zero());
    if (n < 2) return;
    const Proj pb = {base.x, base.y, ZkScalar::one()};
    std::vector<Proj> pts(2 * (n - 1));
    Proj r = bits[0] ? pb : Proj{ZkScalar::zero(), ZkScalar::one(), ZkScalar::one()};
    for (size_t i = 1; i < n; ++i) {
        Proj d = r.dbl();
        Proj a = d;
        a.add_assign(pb);
        pts[2 * (i - 1)] = d;
        pts[2 * (i - 1) + 1] = a;
        r = bits[i] ? a : d;
    }
    // batch inversion of all Z
    std::vector<ZkScalar> pref(pts.size());
    ZkScalar acc = ZkScalar::one();
    for (size_t k = 0; k < pts.size(); ++k) {
        pref[k] = acc;
        acc = acc * pts[k].Z;
    }
    ZkScalar inv = acc.invert();
    for (size_t k = pts.size(); k-- > 0;) {
        ZkScalar zi = inv * pref[k];
        inv = inv * pts[k].Z;
        PointAffine p = {pts[k].X * zi, pts[k].Y * zi};
        if (k & 1) adds[k / 2 + 1] = p;
        else dbls[k / 2 + 1] = p;
    }
}

// ---- EdDSA (src/crypto/jubjub/mod.rs:112-167)
static void order_limbs(uint32_t out[8]) {
    static uint32_t ord[8];
    static std::once_flag f;
    std::call_once(f, [] { ZkScalar::from_dec("6554484396890773809930967563523245729705921265872317281365359162392183254199").to_canonical(ord); });
    memcpy(out, ord, 32);
}

JubjubPrivateKey jubjub_generate_keys(const uint8_t* seed, size_t len) {
    JubjubPrivateKey k;
    k.randomness = hash_to_scalar(seed, len);
    uint8_t repr[32];
    uint32_t c[8];
    k.randomness.to_canonical(c);
    memcpy(repr, c, 32);  // to_repr() = canonical little-endian bytes
    k.scalar = hash_to_scalar(repr, 32);
    k.public_key = jubjub_base().multiply(k.scalar);
    return k;
}

JubjubSignature jubjub_sign(const JubjubPrivateKey& sk, const ZkScalar& msg) {
    ZkScalar rr[2] = {sk.randomness, msg};
    ZkScalar r = poseidon_hash(rr, 2);
    PointAffine R = jubjub_base().multiply(r);
    ZkScalar hin[5] = {R.x, R.y, sk.public_key.x, sk.public_key.y, msg};
    ZkScalar h = poseidon_hash(hin, 5);
    // s = (r + h * a) mod ORDER over the integers
    uint32_t rc[8], hc[8], ac[8], ord[8];
    r.to_canonical(rc);
    h.to_canonical(hc);
    sk.scalar.to_canonical(ac);
    order_limbs(ord);
    uint32_t prod[17] = {0};
    for (int i = 0; i < 8; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < 8; ++j) {
            uint64_t t = (uint64_t)hc[i] * ac[j] + prod[i + j] + carry;
            prod[i + j] = (uint32_t)t;
            carry = t >> 32;
        }
        prod[i + 8] = (uint32_t)carry;
    }
    uint64_t carry = 0;
    for (int i = 0; i < 17; ++i) {
        uint64_t t = (uint64_t)prod[i] + (i < 8 ? rc[i] : 0) + carry;
        prod[i] = (uint32_t)t;
        carry = t >> 32;
    }
    // binary long division: rem = prod mod ORDER
    uint32_t rem[9] = {0};
    for (int bit = 17 * 32 - 1; bit >= 0; --bit) {
        for (int i = 8; i > 0; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 31);
        rem[0] = (rem[0] << 1) | ((prod[bit >> 5] >> (bit & 31)) & 1);
        // if rem >= ord: rem -= ord
        bool ge = rem[8] != 0;
        if (!ge) {
            ge = true;
            for (int i = 7; i >= 0; --i) {
                if (rem[i] != ord[i]) { ge = rem[i] > ord[i]; break; }
            }
        }
        if (ge) {
            uint64_t borrow = 0;
            for (int i = 0; i < 9; ++i) {
                uint64_t d = (uint64_t)rem[i] - (i < 8 ? ord[i] : 0) - borrow;
                rem[i] = (uint32_t)d;
                borrow = (d >> 63) & 1;
            }
        }
    }
    return {R, ZkScalar::from_canonical_limbs(rem)};
}

bool jubjub_verify(const PointAffine& pk, const ZkScalar& msg, const JubjubSignature& sig) {
    if (!pk.is_on_curve() || !sig.r.is_on_curve()) return false;
    ZkScalar hin[5] = {sig.r.x, sig.r.y, pk.x, pk.y, msg};
    ZkScalar h = poseidon_hash(hin, 5);
    PointAffine sb = jubjub_base().multiply(sig.s);
    PointAffine q = pk.multiply(h);
    q.add_assign(sig.r);
    return q == sb;
}

}  // namespace bzk
