// Host-side (CPU, C++) mirror of the reference's `src/zk` scalar / hasher / signature API, used by the
// witness generator.  This is product code (the host half of the drop-in), not the oracle: it shares
// no source with oracle/ and is what a Rust host would otherwise do with `ZkScalar`, `ZkHasher`,
// `JubJub` (reference: src/zk/mod.rs:152-155,202-324; src/crypto/jubjub/*).
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "bzk_field.cuh"
#include "host_fr64.h"

namespace bzk {

// ---- ZkScalar: thin value type over the Montgomery limbs (src/zk/mod.rs:202-206)
struct ZkScalar {
    Fr v;
    ZkScalar() : v(Fr::zero()) {}
    explicit ZkScalar(const Fr& f) : v(f) {}
    static ZkScalar zero() { return ZkScalar(); }
    static ZkScalar one() { return ZkScalar(Fr::one()); }
    static ZkScalar from_u64(uint64_t x) {  // `ZkScalar::from(u64)`
        Fr c = Fr::zero();
        c.l[0] = (uint32_t)x;
        c.l[1] = (uint32_t)(x >> 32);
        return ZkScalar(fe_to_mont<FrParams>(c));
    }
    // `ZkScalar::new(bytes)` = little-endian integer mod r (src/zk/mod.rs:262-271); len <= 64
    static ZkScalar from_le_bytes_mod(const uint8_t* b, size_t len);
    static ZkScalar from_canonical_limbs(const uint32_t* l8) {  // must be < r
        Fr c;
        memcpy(c.l, l8, 32);
        return ZkScalar(fe_to_mont<FrParams>(c));
    }
    static ZkScalar from_dec(const char* s);  // `from_str_vartime`
    void to_canonical(uint32_t out[8]) const {
        Fr c = fe_from_mont<FrParams>(v);
        memcpy(out, c.l, 32);
    }
    bool bit(int i) const {  // canonical little-endian bit i (`to_le_bits`)
        uint32_t c[8];
        to_canonical(c);
        return (c[i >> 5] >> (i & 31)) & 1;
    }
    bool is_zero() const { return v.is_zero(); }
    bool is_odd() const { return bit(0); }
    bool operator==(const ZkScalar& o) const { return v.equals(o.v); }
    bool operator!=(const ZkScalar& o) const { return !v.equals(o.v); }
    ZkScalar operator+(const ZkScalar& o) const { return ZkScalar(fe_add<FrParams>(v, o.v)); }
    ZkScalar operator-(const ZkScalar& o) const { return ZkScalar(fe_sub<FrParams>(v, o.v)); }
    ZkScalar operator*(const ZkScalar& o) const { return ZkScalar(hfr::mul(v, o.v)); }  // 64-bit-limb product (host_fr64.h): same canonical value
    ZkScalar operator-() const { return ZkScalar(fe_neg<FrParams>(v)); }
    ZkScalar square() const { return ZkScalar(hfr::mul(v, v)); }
    ZkScalar dbl() const { return ZkScalar(fe_dbl<FrParams>(v)); }
    ZkScalar invert() const { return ZkScalar(hfr::inv(v)); }  // 0 -> 0 (callers check)
    ZkScalar pow(const uint32_t* e, int nlimbs) const;
    bool sqrt(ZkScalar* out) const;  // Tonelli-Shanks (2-adicity 32); false if non-residue
    void to_bytes(uint8_t out[32]) const { memcpy(out, v.l, 32); }  // Montgomery limbs = wire form
    static ZkScalar from_bytes(const uint8_t in[32]) {
        ZkScalar s;
        memcpy(s.v.l, in, 32);
        return s;
    }
};

// Worker threads the host generator starts when the caller does not say: the CPUs this process may actually USE - the visible ones capped by the
// container's CPU quota (cgroup v2 cpu.max / v1 cfs quota).  The GPU pool's boxes show 256 CPUs under a 16-CPU quota: one thread per visible CPU
// there means 256 threads time-slicing 16 cores for a 256-transition witness (round 5, run 22: the deferred generator's bodies are short enough
// for that overhead to show in the CPU seconds).  Read once.
int host_default_threads();  // host_zk.hip

// ---- Poseidon on the host (same parameters as the device kernel, src/zk/poseidon/mod.rs:24-84)
struct PoseidonHostParams {
    int t, rf, rp;
    const Fr* rc;   // t * (rf + rp)
    const Fr* mds;  // t * t row-major
};
PoseidonHostParams poseidon_host_params(int t);              // poseidon.hip
PoseidonHostParams poseidon_host_params_cached(int t);       // host_zk.hip: the same, without a lock after the first call per width
namespace hfr { struct MdsTable; }
const hfr::MdsTable& poseidon_mds_table(int t);              // host_zk.hip: the dense MDS of width t in the lane layout of host_fr_ifma.h
ZkScalar poseidon_hash(const ZkScalar* vals, int arity);      // `ZkHasher::hash` (sparse-partial-round evaluation)
ZkScalar poseidon_hash_plain(const ZkScalar* vals, int arity);  // the reference's round function, literally
inline ZkScalar poseidon_hash(const std::vector<ZkScalar>& v) { return poseidon_hash(v.data(), (int)v.size()); }

// ---- SHA3-256 (`hash_to_scalar`, src/zk/mod.rs:218-220)
void sha3_256(const uint8_t* data, size_t len, uint8_t out[32]);
inline ZkScalar hash_to_scalar(const uint8_t* data, size_t len) {
    uint8_t h[32];
    sha3_256(data, len, h);
    return ZkScalar::from_le_bytes_mod(h, 32);
}

// ---- Jubjub (src/crypto/jubjub/curve.rs) : -x^2 + y^2 = 1 + d x^2 y^2
struct PointAffine {
    ZkScalar x, y;  // default (0, 0) like the reference's `Default` (NOT the neutral element)
    bool operator==(const PointAffine& o) const { return x == o.x && y == o.y; }
    static PointAffine zero() { return {ZkScalar::zero(), ZkScalar::one()}; }
    bool is_on_curve() const;
    PointAffine dbl() const;
    void add_assign(const PointAffine& o);
    PointAffine multiply(const ZkScalar& k) const;
};
void jubjub_ladder(const PointAffine& base, const std::vector<bool>& bits_msb_first, std::vector<PointAffine>& dbls,
                   std::vector<PointAffine>& adds);  // dbls[i], adds[i] for i >= 1
PointAffine jubjub_decompress(const ZkScalar& x, bool y_is_odd);  // PointCompressed::decompress (curve.rs:78-88)
const PointAffine& jubjub_default_pubkey();  // `PublicKey::default().decompress()` = (0, r - 1)
const ZkScalar& jubjub_d();
const PointAffine& jubjub_base();
const PointAffine& jubjub_base_cofactor();  // 8 * BASE
struct JubjubPrivateKey {
    PointAffine public_key;
    ZkScalar randomness, scalar;
};
struct JubjubSignature {
    PointAffine r;
    ZkScalar s;
};
JubjubPrivateKey jubjub_generate_keys(const uint8_t* seed, size_t len);  // mod.rs:112-124
JubjubSignature jubjub_sign(const JubjubPrivateKey& sk, const ZkScalar& msg);  // mod.rs:125-150
bool jubjub_verify(const PointAffine& pk, const ZkScalar& msg, const JubjubSignature& sig);  // mod.rs:151-167

}  // namespace bzk
