// f-2: the prover's wire format.  `MpnWork` (src/mpn/mod.rs:263-270) is what a proving worker receives
// from `GET /bincode/mpn/work` (src/node/mod.rs:393-398, src/client/messages.rs:368-376) and `ZkProof`
// (src/zk/mod.rs:646-651) is what it posts back; both travel as bincode 1.3.3 with the default options the reference
// uses everywhere (`bincode::serialize` / `deserialize`): little-endian fixed-width integers, u64 sequence / string /
// map lengths, u32 enum variant indices, `Option` as one tag byte, tuples / fixed arrays / `PhantomData` without any
// framing.  The layouts below restate the reference's `#[derive(Serialize)]` type definitions field by field:
//
//   MpnWork            { config, public_inputs, data, new_root, reward }                      src/mpn/mod.rs:263-270
//   MpnConfig          { 5 x u8 log4 sizes, mpn_contract_id, 3 x usize batch counts, 3 x ZkVerifierKey }   :202-216
//   ZkPublicInputs     { height u64, state, aux_data, next_state }                                          :250-256
//   MpnWorkData        enum { Deposit(Vec<..>) = 0, Withdraw(Vec<..>) = 1, Update(Vec<..>) = 2 }             :243-248
//   {Deposit,Withdraw,Update}Transition                                                                     :426-511
//   MpnAccount         { tx_nonce u32, withdraw_nonce u32, address PointAffine, tokens HashMap<u64, Money> } src/zk/mod.rs:59-65
//   MpnTransaction     { nonce u32, src_pub_key, dst_pub_key (PointCompressed = ZkScalar + bool), amount, fee, sig }  :584-593
//   ZkCompressedState  { state_hash, state_size u64 }                                                       :542-546
//   ZkVerifierKey      enum { Groth16(Box<Groth16VerifyingKey>) = 0 }   (the 1460-byte blobs of src/config/blockchain.rs:32-37)
//   Money { token_id: ContractId, amount: Amount(u64) }; ContractId enum { Null = 0, Ziesha = 1, Custom(ZkScalar) = 2 }
//                                                                                      src/core/transaction.rs:60-81
//   MpnDeposit { mpn_address, payment: ContractDeposit }, MpnWithdraw { mpn_address, mpn_withdraw_nonce, mpn_sig, payment }
//   ContractDeposit / ContractWithdraw (L1 payments)                                                        :136-174
//   ZkScalar = 4 x u64 Montgomery limbs (`#[derive(Serialize)] struct ZkScalar([u64; 4])`, src/zk/mod.rs:202-206)
//
// Two leaf types come from crates that are not vendored (Cargo.toml:31 `ed25519-dalek = "1"`, no lockfile) [recalled]:
// `ed25519_dalek::PublicKey` serialises as a byte string (u64 length 32 + 32 bytes) and `ed25519::Signature` as a
// 64-element tuple (64 bytes, no length; releases before ed25519 1.3 wrote a length-prefixed byte string - selectable
// with BZK_WORK_SIG_LEN_PREFIXED).  They only occur inside the L1 payments of deposit / withdraw works, which the
// circuits read three fields of; the payment is otherwise carried as an opaque byte range.
//
// The reference holds no golden `MpnWork` payloads (parity unpinned for this format too); what IS pinned
// is the `ZkVerifierKey` encoding (the hard-coded verifying keys), which this codec round-trips byte for byte
// (tests/test_work_codec_cpu.py), and every scalar type it is built from (Poseidon KATs on the Montgomery limbs).
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "host_mpn_types.h"

namespace bzk {

#ifndef BZK_WORK_SIG_LEN_PREFIXED
#define BZK_WORK_SIG_LEN_PREFIXED 1u  // include/bzk.h
#endif

struct BinReader {
    const uint8_t* p;
    size_t n, pos = 0;
    bool ok = true;
    std::string err;
    BinReader(const uint8_t* d, size_t len) : p(d), n(len) {}
    bool fail(const char* what) {
        if (ok) {
            ok = false;
            err = std::string(what) + " at byte " + std::to_string(pos);
        }
        return false;
    }
    bool need(size_t k, const char* what) { return (ok && n - pos >= k) ? true : fail(what); }
    uint8_t u8(const char* what = "u8") {
        if (!need(1, what)) return 0;
        return p[pos++];
    }
    bool boolean(const char* what = "bool") {
        const uint8_t v = u8(what);
        if (ok && v > 1) fail("bool is neither 0 nor 1");  // bincode rejects other values
        return v == 1;
    }
    uint32_t u32(const char* what = "u32") {
        if (!need(4, what)) return 0;
        uint32_t v;
        memcpy(&v, p + pos, 4);
        pos += 4;
        return v;
    }
    uint64_t u64(const char* what = "u64") {
        if (!need(8, what)) return 0;
        uint64_t v;
        memcpy(&v, p + pos, 8);
        pos += 8;
        return v;
    }
    const uint8_t* bytes(size_t k, const char* what = "bytes") {
        if (!need(k, what)) return nullptr;
        const uint8_t* r = p + pos;
        pos += k;
        return r;
    }
    ZkScalar scalar(const char* what = "ZkScalar") {
        const uint8_t* b = bytes(32, what);
        return b ? ZkScalar::from_bytes(b) : ZkScalar();
    }
    uint64_t len(size_t min_elem_bytes, const char* what) {  // a sequence length that the remaining input can hold
        const uint64_t v = u64(what);
        if (ok && min_elem_bytes && v > (n - pos) / min_elem_bytes) fail(what);
        return ok ? v : 0;
    }
};

struct BinWriter {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    void raw(const void* d, size_t k) {
        const uint8_t* s = (const uint8_t*)d;
        b.insert(b.end(), s, s + k);
    }
    void scalar(const ZkScalar& s) {
        uint8_t t[32];
        s.to_bytes(t);
        raw(t, 32);
    }
};

// ---- scalars and small composites ---------------------------------------------------------------
// ContractId <-> the scalar the circuits use (`impl From<ContractId> for ZkScalar`, src/zk/mod.rs:280-288)
inline ZkScalar rd_contract_id(BinReader& r) {
    const uint32_t tag = r.u32("ContractId tag");
    if (tag == 0) return ZkScalar::zero();
    if (tag == 1) return ZkScalar::one();
    if (tag == 2) return r.scalar("ContractId::Custom");
    r.fail("ContractId variant");
    return ZkScalar();
}
inline void wr_contract_id(BinWriter& w, const ZkScalar& id) {  // `impl From<ZkScalar> for ContractId`, transaction.rs:97-107
    if (id.is_zero()) return w.u32(0);
    if (id == ZkScalar::one()) return w.u32(1);
    w.u32(2);
    w.scalar(id);
}
inline Money rd_money(BinReader& r) {
    Money m;
    m.token_id = rd_contract_id(r);
    m.amount = r.u64("Amount");
    return m;
}
inline void wr_money(BinWriter& w, const Money& m) {
    wr_contract_id(w, m.token_id);
    w.u64(m.amount);
}
inline PointAffine rd_point(BinReader& r) {
    PointAffine p;
    p.x = r.scalar("PointAffine.0");
    p.y = r.scalar("PointAffine.1");
    return p;
}
inline void wr_point(BinWriter& w, const PointAffine& p) {
    w.scalar(p.x);
    w.scalar(p.y);
}
// jubjub::PublicKey(PointCompressed(x, y_is_odd)); the circuits allocate its decompression (curve.rs:78-88, which
// `unwrap`s the square root: an x that is not on the curve is rejected here instead of panicking)
inline PointAffine rd_pubkey(BinReader& r) {
    const ZkScalar x = r.scalar("PointCompressed.0");
    const bool odd = r.boolean("PointCompressed.1");
    if (!r.ok) return PointAffine();
    PointAffine p = jubjub_decompress(x, odd);
    if (!p.is_on_curve()) r.fail("PointCompressed does not decompress");
    return p;
}
inline void wr_pubkey(BinWriter& w, const PointAffine& p) {  // PointAffine::compress (curve.rs:70-74)
    w.scalar(p.x);
    w.u8(p.y.is_odd() ? 1 : 0);
}
inline JubjubSignature rd_zksig(BinReader& r) {
    JubjubSignature s;
    s.r = rd_point(r);
    s.s = r.scalar("Signature.s");
    return s;
}
inline void wr_zksig(BinWriter& w, const JubjubSignature& s) {
    wr_point(w, s.r);
    w.scalar(s.s);
}
inline Proof4 rd_proof(BinReader& r, const char* what) {
    const uint64_t k = r.len(96, what);
    Proof4 p((size_t)k);
    for (uint64_t i = 0; i < k && r.ok; ++i)
        for (int j = 0; j < 3; ++j) p[(size_t)i][j] = r.scalar(what);
    return p;
}
inline void wr_proof(BinWriter& w, const Proof4& p) {
    w.u64(p.size());
    for (auto& t : p)
        for (int j = 0; j < 3; ++j) w.scalar(t[j]);
}
inline MpnAccount rd_account(BinReader& r) {
    MpnAccount a;
    a.tx_nonce = r.u32("MpnAccount.tx_nonce");
    a.withdraw_nonce = r.u32("MpnAccount.withdraw_nonce");
    a.address = rd_point(r);
    const uint64_t k = r.len(20, "MpnAccount.tokens");
    for (uint64_t i = 0; i < k && r.ok; ++i) {
        const uint64_t key = r.u64("token index");
        a.tokens[key] = rd_money(r);
    }
    return a;
}
inline void wr_account(BinWriter& w, const MpnAccount& a) {  // HashMap order is unspecified on the wire: ascending here
    w.u32(a.tx_nonce);
    w.u32(a.withdraw_nonce);
    wr_point(w, a.address);
    w.u64(a.tokens.size());
    for (auto& kv : a.tokens) {
        w.u64(kv.first);
        wr_money(w, kv.second);
    }
}
inline MpnTx rd_mpn_tx(BinReader& r) {
    MpnTx t;
    t.nonce = r.u32("MpnTransaction.nonce");
    t.src_pub = rd_pubkey(r);
    t.dst_pub = rd_pubkey(r);
    t.amount = rd_money(r);
    t.fee = rd_money(r);
    t.sig = rd_zksig(r);
    return t;
}
inline void wr_mpn_tx(BinWriter& w, const MpnTx& t) {
    w.u32(t.nonce);
    wr_pubkey(w, t.src_pub);
    wr_pubkey(w, t.dst_pub);
    wr_money(w, t.amount);
    wr_money(w, t.fee);
    wr_zksig(w, t.sig);
}

// ---- L1 payments (opaque except for amount / fee and, for withdrawals, the fingerprint) ---------------------------
inline void skip_string(BinReader& r) {
    const uint64_t k = r.len(1, "String length");
    r.bytes((size_t)k, "String");
}
inline void skip_l1_pub(BinReader& r) {  // ed25519_dalek::PublicKey: byte string of 32
    if (r.u64("ed25519 public key length") != 32) r.fail("ed25519 public key length");
    r.bytes(32, "ed25519 public key");
}
// ContractDeposit { memo, contract_id, deposit_circuit_id, calldata, src, amount, fee, nonce, sig: Option<Sig> }
inline void rd_contract_deposit(BinReader& r, uint32_t flags, DepositTx& tx) {
    const size_t p0 = r.pos;
    skip_string(r);
    rd_contract_id(r);
    r.u32("deposit_circuit_id");
    r.scalar("calldata");
    skip_l1_pub(r);
    tx.amount = rd_money(r);
    rd_money(r);  // fee: paid on L1, not seen by the circuit
    r.u32("nonce");
    const uint8_t some = r.u8("Option<Signature> tag");
    if (r.ok && some > 1) r.fail("Option tag");
    if (r.ok && some) {
        if (flags & BZK_WORK_SIG_LEN_PREFIXED)
            if (r.u64("ed25519 signature length") != 64) r.fail("ed25519 signature length");
        r.bytes(64, "ed25519 signature");
    }
    if (r.ok) tx.payment.assign(r.p + p0, r.p + r.pos);
}
// ContractWithdraw { memo, contract_id, withdraw_circuit_id, calldata, dst, amount, fee }
// fingerprint = ZkScalar::new(sha3(bincode(payment with calldata := 0)))  (transaction.rs:204-211)
inline void rd_contract_withdraw(BinReader& r, WithdrawTx& tx) {
    const size_t p0 = r.pos;
    skip_string(r);
    rd_contract_id(r);
    r.u32("withdraw_circuit_id");
    const size_t calldata_at = r.pos;
    r.scalar("calldata");
    skip_l1_pub(r);
    tx.amount = rd_money(r);
    tx.fee = rd_money(r);
    if (!r.ok) return;
    tx.payment.assign(r.p + p0, r.p + r.pos);
    std::vector<uint8_t> unsigned_bin = tx.payment;
    memset(unsigned_bin.data() + (calldata_at - p0), 0, 32);
    tx.fingerprint = hash_to_scalar(unsigned_bin.data(), unsigned_bin.size());
}
// the payment a synthetic world attaches to a queued deposit / withdrawal (tests, benches): empty memo, circuit 0,
// zero calldata, an all-zero L1 key, no L1 signature
inline std::vector<uint8_t> default_contract_deposit(const ZkScalar& contract_id, const Money& amount) {
    BinWriter w;
    w.u64(0);
    wr_contract_id(w, contract_id);
    w.u32(0);
    w.scalar(ZkScalar());
    w.u64(32);
    const uint8_t z[32] = {0};
    w.raw(z, 32);
    wr_money(w, amount);
    wr_money(w, Money{ZkScalar::one(), 0});
    w.u32(0);
    w.u8(0);
    return w.b;
}
inline std::vector<uint8_t> default_contract_withdraw(const ZkScalar& contract_id, const Money& amount, const Money& fee,
                                                      const ZkScalar& calldata) {
    BinWriter w;
    w.u64(0);
    wr_contract_id(w, contract_id);
    w.u32(0);
    w.scalar(calldata);
    w.u64(32);
    const uint8_t z[32] = {0};
    w.raw(z, 32);
    wr_money(w, amount);
    wr_money(w, fee);
    return w.b;
}
inline ZkScalar contract_withdraw_fingerprint(const std::vector<uint8_t>& payment) {  // calldata sits after memo + id + u32
    BinReader r(payment.data(), payment.size());
    skip_string(r);
    rd_contract_id(r);
    r.u32();
    if (!r.ok || payment.size() < r.pos + 32) return ZkScalar();
    std::vector<uint8_t> u = payment;
    memset(u.data() + r.pos, 0, 32);
    return hash_to_scalar(u.data(), u.size());
}

// ---- transitions ------------------------------------------------------------------------------------------------
inline DepositTransition rd_deposit_transition(BinReader& r, uint32_t flags) {
    DepositTransition t;
    t.enabled = r.boolean("DepositTransition.enabled");
    t.tx.mpn_address = rd_pubkey(r);
    rd_contract_deposit(r, flags, t.tx);
    t.before = rd_account(r);
    t.before_balances_hash = r.scalar("before_balances_hash");
    t.before_balance = rd_money(r);
    t.proof = rd_proof(r, "proof");
    t.account_index = r.u64("account_index");
    t.token_index = r.u64("token_index");
    t.balance_proof = rd_proof(r, "balance_proof");
    return t;
}
inline void wr_deposit_transition(BinWriter& w, const DepositTransition& t, const ZkScalar& contract_id) {
    w.u8(t.enabled ? 1 : 0);
    wr_pubkey(w, t.tx.mpn_address);
    const std::vector<uint8_t> pay = t.tx.payment.empty() ? default_contract_deposit(contract_id, t.tx.amount) : t.tx.payment;
    w.raw(pay.data(), pay.size());
    wr_account(w, t.before);
    w.scalar(t.before_balances_hash);
    wr_money(w, t.before_balance);
    wr_proof(w, t.proof);
    w.u64(t.account_index);
    w.u64(t.token_index);
    wr_proof(w, t.balance_proof);
}
inline WithdrawTransition rd_withdraw_transition(BinReader& r) {
    WithdrawTransition t;
    t.enabled = r.boolean("WithdrawTransition.enabled");
    t.tx.mpn_address = rd_pubkey(r);
    t.tx.nonce = r.u32("mpn_withdraw_nonce");
    t.tx.sig = rd_zksig(r);
    rd_contract_withdraw(r, t.tx);
    t.before = rd_account(r);
    t.before_token_balance = rd_money(r);
    t.before_fee_balance = rd_money(r);
    t.proof = rd_proof(r, "proof");
    t.account_index = r.u64("account_index");
    t.token_index = r.u64("token_index");
    t.token_balance_proof = rd_proof(r, "token_balance_proof");
    t.before_token_hash = r.scalar("before_token_hash");
    t.fee_token_index = r.u64("fee_token_index");
    t.fee_balance_proof = rd_proof(r, "fee_balance_proof");
    return t;
}
inline void wr_withdraw_transition(BinWriter& w, const WithdrawTransition& t) {
    w.u8(t.enabled ? 1 : 0);
    wr_pubkey(w, t.tx.mpn_address);
    w.u32(t.tx.nonce);
    wr_zksig(w, t.tx.sig);
    w.raw(t.tx.payment.data(), t.tx.payment.size());  // never empty: see bzk_mpn_push_withdraw / the decoder
    wr_account(w, t.before);
    wr_money(w, t.before_token_balance);
    wr_money(w, t.before_fee_balance);
    wr_proof(w, t.proof);
    w.u64(t.account_index);
    w.u64(t.token_index);
    wr_proof(w, t.token_balance_proof);
    w.scalar(t.before_token_hash);
    w.u64(t.fee_token_index);
    wr_proof(w, t.fee_balance_proof);
}
inline UpdateTransition rd_update_transition(BinReader& r) {
    UpdateTransition t;
    t.enabled = r.boolean("UpdateTransition.enabled");
    t.tx = rd_mpn_tx(r);
    t.src_before = rd_account(r);
    t.src_before_balances_hash = r.scalar("src_before_balances_hash");
    t.src_before_balance = rd_money(r);
    t.src_before_fee_balance = rd_money(r);
    t.src_proof = rd_proof(r, "src_proof");
    t.src_index = r.u64("src_index");
    t.src_token_index = r.u64("src_token_index");
    t.src_balance_proof = rd_proof(r, "src_balance_proof");
    t.src_fee_token_index = r.u64("src_fee_token_index");
    t.src_fee_balance_proof = rd_proof(r, "src_fee_balance_proof");
    t.dst_before = rd_account(r);
    t.dst_before_balances_hash = r.scalar("dst_before_balances_hash");
    t.dst_before_balance = rd_money(r);
    t.dst_proof = rd_proof(r, "dst_proof");
    t.dst_index = r.u64("dst_index");
    t.dst_token_index = r.u64("dst_token_index");
    t.dst_balance_proof = rd_proof(r, "dst_balance_proof");
    return t;
}
inline void wr_update_transition(BinWriter& w, const UpdateTransition& t) {
    w.u8(t.enabled ? 1 : 0);
    wr_mpn_tx(w, t.tx);
    wr_account(w, t.src_before);
    w.scalar(t.src_before_balances_hash);
    wr_money(w, t.src_before_balance);
    wr_money(w, t.src_before_fee_balance);
    wr_proof(w, t.src_proof);
    w.u64(t.src_index);
    w.u64(t.src_token_index);
    wr_proof(w, t.src_balance_proof);
    w.u64(t.src_fee_token_index);
    wr_proof(w, t.src_fee_balance_proof);
    wr_account(w, t.dst_before);
    w.scalar(t.dst_before_balances_hash);
    wr_money(w, t.dst_before_balance);
    wr_proof(w, t.dst_proof);
    w.u64(t.dst_index);
    w.u64(t.dst_token_index);
    wr_proof(w, t.dst_balance_proof);
}

// ---- MpnConfig / MpnWork ------------------------------------------------------------------------------------------
// Groth16VerifyingKey: alpha_g1, beta_g1 (97 B each), beta_g2, gamma_g2 (193), delta_g1 (97), delta_g2 (193),
// ic: Vec<97 B>  (src/zk/groth16/mod.rs:22-31; 870 + 8 + 97 * ic.len() bytes, 1460 for the MPN circuits' 6 entries)
inline std::vector<uint8_t> rd_verifier_key(BinReader& r) {
    if (r.u32("ZkVerifierKey tag") != 0) r.fail("ZkVerifierKey variant (only Groth16 = 0 exists outside cfg(test))");
    const size_t p0 = r.pos;
    r.bytes(870, "Groth16VerifyingKey points");
    const uint64_t k = r.len(97, "Groth16VerifyingKey.ic");
    r.bytes((size_t)k * 97, "Groth16VerifyingKey.ic");
    return r.ok ? std::vector<uint8_t>(r.p + p0, r.p + r.pos) : std::vector<uint8_t>();
}
inline bool verifier_key_well_formed(const std::vector<uint8_t>& vk) {
    if (vk.size() < 878) return false;
    uint64_t k;
    memcpy(&k, vk.data() + 870, 8);
    return k <= (vk.size() - 878) / 97 && vk.size() == 878 + 97 * (size_t)k;
}
struct MpnWorkConfig {
    uint8_t log4_tree = 0, log4_token_tree = 0, log4_deposit_batch = 0, log4_withdraw_batch = 0, log4_update_batch = 0;
    ZkScalar mpn_contract_id;
    uint64_t num_update_batches = 0, num_deposit_batches = 0, num_withdraw_batches = 0;
    std::vector<uint8_t> deposit_vk, withdraw_vk, update_vk;  // bincode(Groth16VerifyingKey), without the enum tag
};
struct MpnWork {
    MpnWorkConfig config;
    uint64_t height = 0;
    ZkScalar state, aux_data, next_state;
    int kind = 2;  // MpnWorkData variant index: 0 Deposit, 1 Withdraw, 2 Update
    std::vector<DepositTransition> deposits;
    std::vector<WithdrawTransition> withdraws;
    std::vector<UpdateTransition> updates;
    ZkScalar new_root_hash;
    uint64_t new_root_size = 0;
    uint64_t reward = 0;
    size_t n_transitions() const { return kind == 0 ? deposits.size() : kind == 1 ? withdraws.size() : updates.size(); }
    int log4_batch() const {
        return kind == 0 ? config.log4_deposit_batch : kind == 1 ? config.log4_withdraw_batch : config.log4_update_batch;
    }
    const std::vector<uint8_t>& vk() const {  // MpnWork::vk (src/mpn/mod.rs:273-280)
        return kind == 0 ? config.deposit_vk : kind == 1 ? config.withdraw_vk : config.update_vk;
    }
};

inline bool mpn_work_decode(BinReader& r, uint32_t flags, MpnWork& w) {
    MpnWorkConfig& c = w.config;
    c.log4_tree = r.u8("log4_tree_size");
    c.log4_token_tree = r.u8("log4_token_tree_size");
    c.log4_deposit_batch = r.u8("log4_deposit_batch_size");
    c.log4_withdraw_batch = r.u8("log4_withdraw_batch_size");
    c.log4_update_batch = r.u8("log4_update_batch_size");
    c.mpn_contract_id = rd_contract_id(r);
    c.num_update_batches = r.u64("mpn_num_update_batches");
    c.num_deposit_batches = r.u64("mpn_num_deposit_batches");
    c.num_withdraw_batches = r.u64("mpn_num_withdraw_batches");
    c.deposit_vk = rd_verifier_key(r);
    c.withdraw_vk = rd_verifier_key(r);
    c.update_vk = rd_verifier_key(r);
    w.height = r.u64("public_inputs.height");
    w.state = r.scalar("public_inputs.state");
    w.aux_data = r.scalar("public_inputs.aux_data");
    w.next_state = r.scalar("public_inputs.next_state");
    const uint32_t tag = r.u32("MpnWorkData tag");
    if (r.ok && tag > 2) r.fail("MpnWorkData variant");
    w.kind = (int)tag;
    const uint64_t k = r.len(1, "transitions");
    for (uint64_t i = 0; i < k && r.ok; ++i) {
        if (tag == 0) w.deposits.push_back(rd_deposit_transition(r, flags));
        else if (tag == 1) w.withdraws.push_back(rd_withdraw_transition(r));
        else w.updates.push_back(rd_update_transition(r));
    }
    w.new_root_hash = r.scalar("new_root.state_hash");
    w.new_root_size = r.u64("new_root.state_size");
    w.reward = r.u64("reward");
    return r.ok;
}
inline void mpn_work_encode(BinWriter& o, const MpnWork& w) {
    const MpnWorkConfig& c = w.config;
    o.u8(c.log4_tree);
    o.u8(c.log4_token_tree);
    o.u8(c.log4_deposit_batch);
    o.u8(c.log4_withdraw_batch);
    o.u8(c.log4_update_batch);
    wr_contract_id(o, c.mpn_contract_id);
    o.u64(c.num_update_batches);
    o.u64(c.num_deposit_batches);
    o.u64(c.num_withdraw_batches);
    for (const std::vector<uint8_t>* vk : {&c.deposit_vk, &c.withdraw_vk, &c.update_vk}) {
        o.u32(0);
        o.raw(vk->data(), vk->size());
    }
    o.u64(w.height);
    o.scalar(w.state);
    o.scalar(w.aux_data);
    o.scalar(w.next_state);
    o.u32((uint32_t)w.kind);
    o.u64(w.n_transitions());
    for (auto& t : w.deposits) wr_deposit_transition(o, t, c.mpn_contract_id);
    for (auto& t : w.withdraws) wr_withdraw_transition(o, t);
    for (auto& t : w.updates) wr_update_transition(o, t);
    o.scalar(w.new_root_hash);
    o.u64(w.new_root_size);
    o.u64(w.reward);
}

// commitment of a solution: ZkScalar::new(sha3_256(bincode((prover, reward))))  (MpnWork::verify, src/mpn/mod.rs:281-295)
// prover = the worker's L1 address (ed25519 public key: byte string of 32), reward = Amount(u64)
inline ZkScalar mpn_work_commitment(const uint8_t prover_pub[32], uint64_t reward) {
    BinWriter w;
    w.u64(32);
    w.raw(prover_pub, 32);
    w.u64(reward);
    return hash_to_scalar(w.b.data(), w.b.size());
}

}  // namespace bzk
