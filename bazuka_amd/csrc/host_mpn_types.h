// Host-side value types of the MPN path: the reference's transition / transaction structs as the witness builders and
// the circuits use them (src/mpn/mod.rs:426-537, src/zk/mod.rs:59-118,584-627, src/core/transaction.rs:77-174).
// Shared by the generator (mpn.hip) and the bincode codec of `MpnWork` (host_bincode.h).
#pragma once
#include <array>
#include <map>
#include <vector>

#include "host_zk.h"

namespace bzk {

struct Money {
    ZkScalar token_id;  // ContractId as scalar: Null -> 0, Ziesha -> 1, Custom(x) -> x (src/zk/mod.rs:280-288)
    uint64_t amount = 0;
};

struct MpnAccount {
    uint32_t tx_nonce = 0, withdraw_nonce = 0;
    PointAffine address;  // default (0, 0) = empty slot
    std::map<uint64_t, Money> tokens;
    long find_token_index(int log4_cap, const ZkScalar& token, bool empty_allowed) const {
        for (auto& kv : tokens)
            if (kv.second.token_id == token) return (long)kv.first;
        if (empty_allowed)
            for (uint64_t i = 0; i < ((uint64_t)1 << (2 * log4_cap)); ++i)
                if (!tokens.count(i)) return (long)i;
        return -1;
    }
};

static ZkScalar token_leaf(const Money& m) {
    ZkScalar v[2] = {m.token_id, ZkScalar::from_u64(m.amount)};
    return poseidon_hash(v, 2);
}

struct MpnTx {  // MpnTransaction with decompressed keys
    uint32_t nonce = 0;
    PointAffine src_pub, dst_pub;
    Money amount, fee;
    JubjubSignature sig;
    ZkScalar hash() const {
        ZkScalar v[7] = {ZkScalar::from_u64(nonce), dst_pub.x, dst_pub.y, amount.token_id, ZkScalar::from_u64(amount.amount),
                         fee.token_id, ZkScalar::from_u64(fee.amount)};
        return poseidon_hash(v, 7);
    }
};

typedef std::vector<std::array<ZkScalar, 3>> Proof4;

struct UpdateTransition {
    bool enabled = false;
    MpnTx tx;
    MpnAccount src_before, dst_before;
    ZkScalar src_before_balances_hash, dst_before_balances_hash;
    Money src_before_balance, src_before_fee_balance, dst_before_balance;
    Proof4 src_proof, src_balance_proof, src_fee_balance_proof, dst_proof, dst_balance_proof;
    uint64_t src_index = 0, src_token_index = 0, src_fee_token_index = 0, dst_index = 0, dst_token_index = 0;
    ZkScalar state_after;  // account-tree root once this transition is applied (not a reference field: scheduling aid)
    static UpdateTransition null(int L, int T) {
        UpdateTransition t;
        std::array<ZkScalar, 3> z = {ZkScalar(), ZkScalar(), ZkScalar()};
        t.src_proof.assign(L, z);
        t.dst_proof.assign(L, z);
        t.src_balance_proof.assign(T, z);
        t.src_fee_balance_proof.assign(T, z);
        t.dst_balance_proof.assign(T, z);
        // `tx: Default::default()`: the circuit allocates `tx.dst_pub_key.0.decompress()`, and the default
        // compressed key (x = 0, even y) decompresses to (0, r - 1), not to (0, 0)
        t.tx.dst_pub = jubjub_default_pubkey();
        t.tx.src_pub = jubjub_default_pubkey();
        return t;
    }
};

// ---- deposits / withdrawals (src/mpn/mod.rs:426-489; MpnDeposit / MpnWithdraw src/core/transaction.rs:163-174)
struct DepositTx {
    PointAffine mpn_address = jubjub_default_pubkey();
    Money amount;                  // payment.amount: the one part of the L1 payment the circuit reads
    std::vector<uint8_t> payment;  // bincode(ContractDeposit) as received (empty: a synthetic deposit, see host_bincode.h)
};
struct DepositTransition {
    bool enabled = false;
    DepositTx tx;
    MpnAccount before;
    ZkScalar before_balances_hash;
    Money before_balance;
    Proof4 proof, balance_proof;
    uint64_t account_index = 0, token_index = 0;
    ZkScalar state_after;
    static DepositTransition null(int L, int T) {
        DepositTransition t;
        std::array<ZkScalar, 3> z = {ZkScalar(), ZkScalar(), ZkScalar()};
        t.proof.assign(L, z);
        t.balance_proof.assign(T, z);
        return t;
    }
};
struct WithdrawTx {
    PointAffine mpn_address = jubjub_default_pubkey();
    uint32_t nonce = 0;
    JubjubSignature sig;
    Money amount, fee;
    ZkScalar fingerprint;          // ContractWithdraw::fingerprint() (src/core/transaction.rs:204-211)
    std::vector<uint8_t> payment;  // bincode(ContractWithdraw) the fingerprint was taken from (empty: opaque fingerprint)
    ZkScalar sign_message() const {
        ZkScalar v[2] = {fingerprint, ZkScalar::from_u64(nonce)};
        return poseidon_hash(v, 2);
    }
    ZkScalar calldata() const {
        ZkScalar v[6] = {mpn_address.x, mpn_address.y, ZkScalar::from_u64(nonce), sig.r.x, sig.r.y, sig.s};
        return poseidon_hash(v, 6);
    }
};
struct WithdrawTransition {
    bool enabled = false;
    WithdrawTx tx;
    MpnAccount before;
    Money before_token_balance, before_fee_balance;
    Proof4 proof, token_balance_proof, fee_balance_proof;
    uint64_t account_index = 0, token_index = 0, fee_token_index = 0;
    ZkScalar before_token_hash, state_after;
    static WithdrawTransition null(int L, int T) {
        WithdrawTransition t;
        std::array<ZkScalar, 3> z = {ZkScalar(), ZkScalar(), ZkScalar()};
        t.proof.assign(L, z);
        t.token_balance_proof.assign(T, z);
        t.fee_balance_proof.assign(T, z);
        return t;
    }
};


}  // namespace bzk
