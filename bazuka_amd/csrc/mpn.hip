// a3 / a9: host-side MPN witness + R1CS generator (C++), the CPU half of the prover.
//
// Restates, for the Update function of the Main Payment Network:
//   * the sparse 4-ary account / token state (KvStoreStateManager semantics)   src/zk/state/mod.rs:218-420,
//     MpnConfig::state_model                                                   src/mpn/mod.rs:218-241,
//     MpnAccount::{tokens_hash, find_token_index}                              src/zk/mod.rs:68-118
//   * the witness builder `update::update`                                     src/mpn/update.rs:8-299
//   * `UpdateTransition` / `UpdateTransition::null`                            src/mpn/mod.rs:491-537
//   * `impl Circuit for UpdateCircuit` (allocation + constraint order)         src/mpn/circuits/update_circuit.rs:49-494
//   * MpnTransaction::{hash, sign}                                             src/zk/mod.rs:609-627
//   * wallet `create_mpn_transaction` semantics for the synthetic batch        src/wallet/tx_builder.rs:287-306
// Output: the Groth16 assignment (z, A.z, B.z, C.z, densities) and, on request, the CSR matrices for
// CRS generation.  The storage engine (KvStore / LevelDB keys) is out of scope: state lives in RAM.
#include <array>
#include <atomic>
#include <chrono>
#include <map>
#include <thread>
#include <tuple>
#include <memory>
#include <mutex>
#include <unordered_map>

#include "bzk_internal.h"
#include "host_r1cs.h"

namespace bzk {

// ------------------------------------------------------------------------------------------------
// sparse 4-ary Poseidon tree
// ------------------------------------------------------------------------------------------------
struct SparseTree4 {
    int depth;
    std::vector<ZkScalar> defaults;                              // [0] = leaf default ... [depth] = empty root
    std::vector<std::unordered_map<uint64_t, ZkScalar>> level;   // level[0] = leaves
    SparseTree4(int d, const ZkScalar& leaf_default) : depth(d), level(d + 1) {
        defaults.push_back(leaf_default);
        for (int i = 0; i < d; ++i) {
            ZkScalar c[4] = {defaults.back(), defaults.back(), defaults.back(), defaults.back()};
            defaults.push_back(poseidon_hash(c, 4));
        }
    }
    ZkScalar get(int lv, uint64_t i) const {
        auto it = level[lv].find(i);
        return it == level[lv].end() ? defaults[lv] : it->second;
    }
    ZkScalar root() const { return get(depth, 0); }
    void set_leaf(uint64_t i, const ZkScalar& v) {
        level[0][i] = v;
        for (int lv = 0; lv < depth; ++lv) {
            const uint64_t base = i & ~(uint64_t)3;
            ZkScalar c[4] = {get(lv, base), get(lv, base + 1), get(lv, base + 2), get(lv, base + 3)};
            i >>= 2;
            level[lv + 1][i] = poseidon_hash(c, 4);
        }
    }
    // sibling triples, leaf level first (src/zk/state/mod.rs:218-264)
    std::vector<std::array<ZkScalar, 3>> prove(uint64_t i) const {
        std::vector<std::array<ZkScalar, 3>> out;
        for (int lv = 0; lv < depth; ++lv) {
            const uint64_t base = i & ~(uint64_t)3;
            std::array<ZkScalar, 3> t;
            int k = 0;
            for (uint64_t j = 0; j < 4; ++j)
                if (base + j != i) t[k++] = get(lv, base + j);
            out.push_back(t);
            i >>= 2;
        }
        return out;
    }
};

}  // namespace bzk

#include "host_mpn_types.h"  // Money, MpnAccount, MpnTx, {Update,Deposit,Withdraw}Transition
#include "host_bincode.h"    // MpnWork <-> bincode bytes (f-2: the prover's wire format)

using namespace bzk;

// ------------------------------------------------------------------------------------------------
// the MPN world (RAM state) - opaque handle of the C ABI
// ------------------------------------------------------------------------------------------------
struct bzk_mpn {
    int L, T;
    ZkScalar token_default, tokens_tree_default, account_default;
    std::unique_ptr<SparseTree4> accounts, empty_tokens;
    std::map<uint64_t, MpnAccount> acct;
    std::map<uint64_t, JubjubPrivateKey> keys;
    std::vector<MpnTx> mempool;
    std::vector<DepositTx> deposit_queue;
    std::vector<WithdrawTx> withdraw_queue;
    uint64_t height = 0;
    ZkScalar contract_id = ZkScalar::from_u64(0x4D504E);  // ContractId::Custom of the MPN contract (payments of synthetic txs)
    int threads = host_default_threads();  // the CPUs this process may use (visible ones capped by the cgroup quota); bzk_mpn_set_threads overrides
    bzk_ctx* dev = nullptr;  // bzk_mpn_set_device: the witness builders hash their Merkle updates in batches on this context
    bool defer = false;      // bzk_mpn_set_defer: witness-only Update instances leave the hash-dependent values to the device (host_r1cs.h DeferProgram)
    std::string dev_error;

    bzk_mpn(int l, int t) : L(l), T(t) {
        token_default = token_leaf(Money());
        empty_tokens.reset(new SparseTree4(T, token_default));
        tokens_tree_default = empty_tokens->root();
        account_default = account_hash(MpnAccount());
        accounts.reset(new SparseTree4(L, account_default));
    }
    SparseTree4 tokens_tree(const MpnAccount& a) const {
        SparseTree4 t = *empty_tokens;  // copy of the empty tree (defaults computed once)
        for (auto& kv : a.tokens) t.set_leaf(kv.first, token_leaf(kv.second));
        return t;
    }
    void set_with_tokens_root(uint64_t i, const MpnAccount& a, const ZkScalar& tokens_root) {
        acct[i] = a;
        ZkScalar v[5] = {ZkScalar::from_u64(a.tx_nonce), ZkScalar::from_u64(a.withdraw_nonce), a.address.x, a.address.y, tokens_root};
        accounts->set_leaf(i, poseidon_hash(v, 5));
    }
    ZkScalar tokens_hash(const MpnAccount& a) const { return a.tokens.empty() ? tokens_tree_default : tokens_tree(a).root(); }
    ZkScalar account_hash(const MpnAccount& a) const {
        ZkScalar v[5] = {ZkScalar::from_u64(a.tx_nonce), ZkScalar::from_u64(a.withdraw_nonce), a.address.x, a.address.y, tokens_hash(a)};
        return poseidon_hash(v, 5);
    }
    MpnAccount get(uint64_t i) const {
        auto it = acct.find(i);
        return it == acct.end() ? MpnAccount() : it->second;
    }
    void set(uint64_t i, const MpnAccount& a) {
        acct[i] = a;
        accounts->set_leaf(i, account_hash(a));
    }
};

struct bzk_r1cs {
    ConstraintSystem cs;
    std::vector<uint8_t, bzk::PinnedPoolAlloc<uint8_t>> z_bytes;
    std::vector<uint8_t> a_density, b_density;
    std::vector<uint32_t> colA, colB, colC;  // flat variable indices (after finalize)
    uint64_t accepted = 0, rejected = 0;
    std::unique_ptr<bzk::DeferData> defer;  // set: the arrays have holes a DeferProgram fills (device: bzk_groth16_prove_r1cs, host: bzk_r1cs_fill_host)
    explicit bzk_r1cs(bool rec) : cs(rec) {}
};

namespace bzk {

// ---- pinned host pool behind PinnedPoolAlloc (host_r1cs.h)
namespace {
struct PinnedPool {
    static constexpr size_t MIN_PINNED = (size_t)1 << 22;  // smaller blocks: malloc
    static constexpr size_t MAX_POOLED = (size_t)16 << 30;
    std::mutex mu;
    std::multimap<size_t, void*> free_blocks;     // capacity -> block
    std::map<void*, std::pair<size_t, bool>> live;  // block -> (capacity, pinned)
    std::map<void*, bool> pinned_of;                  // pooled block -> pinned?
    size_t pooled = 0;
    int have_device = -1;
    void* take(size_t bytes) {
        if (bytes < MIN_PINNED) return malloc(bytes ? bytes : 1);
        std::lock_guard<std::mutex> g(mu);
        auto it = free_blocks.lower_bound(bytes);
        if (it != free_blocks.end() && it->first <= bytes + (bytes >> 1)) {
            void* p = it->second;
            pooled -= it->first;
            const auto f = pinned_of.find(p);
            live[p] = {it->first, f == pinned_of.end() ? true : f->second};
            if (f != pinned_of.end()) pinned_of.erase(f);
            free_blocks.erase(it);
            return p;
        }
        if (have_device < 0) {
            int n = 0;
            have_device = (hipGetDeviceCount(&n) == hipSuccess && n > 0) ? 1 : 0;
            (void)hipGetLastError();
        }
        static const bool dbg = env_on("BZK_POOL_DEBUG");
        void* p = nullptr;
        if (have_device && hipHostMalloc(&p, bytes, hipHostMallocPortable) == hipSuccess && p) {
            if (dbg) fprintf(stderr, "[bzk] pool: hipHostMalloc %zu MB (pooled %zu MB in %zu blocks)\n", bytes >> 20, pooled >> 20, free_blocks.size());
            live[p] = {bytes, true};
            return p;
        }
        (void)hipGetLastError();
        p = malloc(bytes);
        if (p) live[p] = {bytes, false};
        return p;
    }
    void give(void* p, size_t bytes) {
        if (!p) return;
        if (bytes < MIN_PINNED) return free(p);
        std::lock_guard<std::mutex> g(mu);
        auto it = live.find(p);
        if (it == live.end()) return free(p);
        const size_t cap = it->second.first;
        const bool pinned = it->second.second;
        live.erase(it);
        // blocks obtained with malloc (no device: a validator-side host) are pooled like the pinned ones - without the pool every
        // witness of a 2^20-class circuit page-faults 145 MB in again (measured: 290 ms per synthesis against 75 ms)
        if (pooled + cap > MAX_POOLED) {
            if (pinned) (void)hipHostFree(p);
            else free(p);
            return;
        }
        pinned_of[p] = pinned;
        pooled += cap;
        free_blocks.emplace(cap, p);
    }
};
PinnedPool& pinned_pool() {
    static PinnedPool* pool = new PinnedPool();  // never destroyed: blocks may be released during static teardown
    return *pool;
}
}  // namespace
void* pinned_pool_take(size_t bytes) { return pinned_pool().take(bytes); }
void pinned_pool_give(void* p, size_t bytes) { pinned_pool().give(p, bytes); }

// update::update for the queued transactions (src/mpn/update.rs:8-299); mutates the world.
static void build_transitions(bzk_mpn& w, int log4_batch, const ZkScalar& fee_token, std::vector<UpdateTransition>& out,
                              uint64_t& fee_sum, uint64_t& rejected) {
    const size_t cap = (size_t)1 << (2 * log4_batch);
    fee_sum = 0;
    rejected = 0;
    std::vector<MpnTx> rest;
    for (const MpnTx& tx : w.mempool) {
        if (out.size() == cap) {
            rest.push_back(tx);
            continue;
        }
        if (tx.fee.token_id != fee_token || !tx.src_pub.is_on_curve() || !tx.dst_pub.is_on_curve()) { ++rejected; continue; }
        long src_index = -1, dst_index = -1;
        for (auto& kv : w.acct) {
            if (kv.second.address == tx.src_pub && src_index < 0) src_index = (long)kv.first;
            if (kv.second.address == tx.dst_pub && dst_index < 0) dst_index = (long)kv.first;
        }
        if (src_index < 0) { ++rejected; continue; }
        if (dst_index < 0) dst_index = w.acct.empty() ? 0 : (long)(w.acct.rbegin()->first + 1);  // next free slot
        MpnAccount src_before = w.get(src_index), dst_before0 = w.get(dst_index);
        // (the reference passes the ACCOUNT-tree log4 size here; harmless, kept)
        long sti = src_before.find_token_index(w.L, tx.amount.token_id, false);
        long dti = dst_before0.find_token_index(w.L, tx.amount.token_id, true);
        long sfi = src_before.find_token_index(w.L, tx.fee.token_id, false);
        if (sti < 0 || dti < 0 || sfi < 0) { ++rejected; continue; }
        Money src_token = src_before.tokens[sti];
        const bool dst_has = dst_before0.tokens.count(dti) != 0;
        if (tx.nonce != src_before.tx_nonce + 1 || !(src_before.address == tx.src_pub) ||
            (dst_before0.address.is_on_curve() && !(dst_before0.address == tx.dst_pub)) ||
            (dst_has && src_token.token_id != dst_before0.tokens[dti].token_id) || src_token.token_id != tx.amount.token_id ||
            src_token.amount < tx.amount.amount) {
            ++rejected;
            continue;
        }
        // All rejection tests of update.rs that depend only on account DATA are decided before the state is
        // touched (the reference works on an isolated mirror and discards it on reject; the effect is the same).
        MpnAccount src_after = src_before;
        src_after.tx_nonce += 1;
        src_after.tokens[sti].amount -= tx.amount.amount;
        if (!src_after.tokens.count(sfi)) { ++rejected; continue; }
        const Money src_fee_token = src_after.tokens[sfi];
        if (src_fee_token.token_id != tx.fee.token_id || src_fee_token.amount < tx.fee.amount) { ++rejected; continue; }

        UpdateTransition t;
        t.enabled = true;
        t.tx = tx;
        t.src_index = src_index; t.dst_index = dst_index;
        t.src_token_index = sti; t.dst_token_index = dti; t.src_fee_token_index = sfi;
        t.src_before = src_before;
        t.src_before_balance = src_token;
        // token tree of the sender, updated in place step by step (one root path per update, as set_data does)
        SparseTree4 stree = w.tokens_tree(src_before);
        t.src_before_balances_hash = stree.root();
        t.src_proof = w.accounts->prove(src_index);
        t.src_balance_proof = stree.prove(sti);
        stree.set_leaf(sti, token_leaf(src_after.tokens[sti]));
        t.src_before_fee_balance = src_fee_token;
        t.src_fee_balance_proof = stree.prove(sfi);
        src_after.tokens[sfi].amount -= tx.fee.amount;
        stree.set_leaf(sfi, token_leaf(src_after.tokens[sfi]));
        w.set_with_tokens_root(src_index, src_after, stree.root());
        t.dst_proof = w.accounts->prove(dst_index);
        MpnAccount dst_before = w.get(dst_index);  // read AFTER the sender update (matters when src == dst)
        SparseTree4 dtree = w.tokens_tree(dst_before);
        t.dst_balance_proof = dtree.prove(dti);
        t.dst_before = dst_before;
        t.dst_before_balances_hash = dtree.root();
        t.dst_before_balance = dst_before.tokens.count(dti) ? dst_before.tokens[dti] : Money();
        MpnAccount dst_after = dst_before;
        dst_after.address = tx.dst_pub;
        if (!dst_after.tokens.count(dti)) dst_after.tokens[dti] = Money{tx.amount.token_id, 0};
        dst_after.tokens[dti].amount += tx.amount.amount;
        dtree.set_leaf(dti, token_leaf(dst_after.tokens[dti]));
        w.set_with_tokens_root(dst_index, dst_after, dtree.root());
        fee_sum += tx.fee.amount;
        t.state_after = w.accounts->root();
        out.push_back(std::move(t));
    }
    w.mempool.swap(rest);
}

// ------------------------------------------------------------------------------------------------
// Device path of the witness builders ("replacing the per-tx KV walk in prepare_works", src/mpn/mod.rs:353-414).
//
// The host builders above walk the account tree once per transaction: prove -> set -> prove -> set, ~50 sequential Poseidon hashes per
// update transaction (1.06 ms on a host core at L = 15, T = 3: 0.27 s of a 256-tx batch).  Inside a batch every proof is taken against
// the state the previous transition left, so the transactions cannot simply be hashed side by side - but the hashes can be grouped BY
// LEVEL: every transaction's ACCEPTANCE depends on account data only (balances, nonces, keys), so a first pass decides the batch and
// lists, in time order, the leaf writes it causes ("events": token slot writes in the accounts' token trees, then account-leaf writes).
// For one tree level, event e re-hashes exactly one node from its four children, each of which is either the value event e itself
// just produced one level below, or the LATEST earlier event's value for that child, or the value the tree held before the batch.
// That is known on the host without hashing anything, so each level of each tree is ONE batched Poseidon launch over all events
// (hash_plan_run: T + 1 launches for the token forest, 1 for the account leaves, L for the account tree), and the three sibling
// inputs of event e's hash at each level ARE its Merkle proof.  No conflict-free waves are needed: transactions may share accounts.
// Result: the same transitions, byte for byte (tests/test_gpu_mpn_prove.py), and the host-side sparse tree is brought up to date
// from the returned node values.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t VH = 0x80000000u;  // value-id tag: output `pos` of hash group `g`: VH | g << 20 | pos
struct VPlan {
    std::vector<uint8_t> up;           // uploaded scalars
    std::vector<HashGroup> groups;     // inputs hold SYMBOLIC ids until resolve()
    uint32_t upload(const ZkScalar& v) {
        const uint32_t id = (uint32_t)(up.size() / 32);
        up.resize(up.size() + 32);
        v.to_bytes(up.data() + 32 * (size_t)id);
        return id;
    }
    uint32_t new_group(uint32_t arity) {
        groups.emplace_back();
        groups.back().arity = arity;
        return (uint32_t)groups.size() - 1;
    }
    uint32_t add(uint32_t g, const uint32_t* in) {
        HashGroup& G = groups[g];
        const uint32_t pos = G.count();
        G.in.insert(G.in.end(), in, in + G.arity);
        return VH | (g << 20) | pos;
    }
    std::vector<uint32_t> base;        // first output id of each group (after resolve)
    std::vector<uint8_t> hashed;
    bool sizes_ok() const {
        if (groups.size() >= 2048) return false;
        for (auto& g : groups)
            if (g.count() >= (1u << 20)) return false;
        return up.size() / 32 < (1u << 30);
    }
    uint32_t final_id(uint32_t id) const { return (id & VH) ? base[(id >> 20) & 2047] + (id & 0xfffffu) : id; }
    int32_t run(bzk_ctx* ctx) {
        if (!sizes_ok()) return BZK_E_ARG;
        base.assign(groups.size(), 0);
        uint32_t next = (uint32_t)(up.size() / 32);
        for (size_t g = 0; g < groups.size(); ++g) { base[g] = next; next += groups[g].count(); }
        for (auto& g : groups)
            for (uint32_t& id : g.in) id = final_id(id);
        return hash_plan_run(ctx, up.data(), up.size() / 32, groups, hashed);
    }
    ZkScalar value(uint32_t id) const {
        const uint32_t f = final_id(id), n_up = (uint32_t)(up.size() / 32);
        return ZkScalar::from_bytes(f < n_up ? up.data() + 32 * (size_t)f : hashed.data() + 32 * (size_t)(f - n_up));
    }
};

// one write of a leaf in one tree of a forest of equal-depth 4-ary trees
struct VEvent {
    uint64_t tree = 0, leaf = 0;
    uint32_t leaf_val = 0;                       // value id of the new leaf
    std::vector<std::array<uint32_t, 3>> sib;    // per level, leaf level first: the value ids of the three siblings BEFORE this write
    std::vector<uint32_t> node_val;              // per level 1..depth: value id of the re-hashed ancestor
    uint32_t root_before = 0, root_after = 0;
};
// plans the re-hash of every event's path, level by level (one hash group per level); initial(lv, tree, idx) = value id of a node
// no event of this batch has written yet (lv 0 = leaves)
template <class Initial>
void vforest_plan(int depth, std::vector<VEvent>& evs, VPlan& P, Initial&& initial) {
    const size_t E = evs.size();
    std::vector<uint32_t> val(E);
    std::vector<uint64_t> idx(E);
    for (size_t e = 0; e < E; ++e) {
        val[e] = evs[e].leaf_val;
        idx[e] = evs[e].leaf;
        evs[e].sib.assign(depth, {0, 0, 0});
        evs[e].node_val.assign(depth, 0);
    }
    auto key = [](uint64_t tree, uint64_t i) { return (tree << 34) ^ i; };  // i < 4^15 = 2^30, tree < 2^30
    for (int lv = 0; lv < depth; ++lv) {
        const uint32_t g = P.new_group(4);
        std::unordered_map<uint64_t, uint32_t> latest;
        latest.reserve(E * 2);
        for (size_t e = 0; e < E; ++e) {
            const uint64_t own = idx[e], b = own & ~(uint64_t)3;
            uint32_t in[4];
            int k = 0;
            for (uint64_t j = 0; j < 4; ++j) {
                if (b + j == own) { in[j] = val[e]; continue; }
                auto it = latest.find(key(evs[e].tree, b + j));
                in[j] = it != latest.end() ? it->second : initial(lv, evs[e].tree, b + j);
                evs[e].sib[lv][k++] = in[j];
            }
            latest[key(evs[e].tree, own)] = val[e];
            val[e] = P.add(g, in);
            evs[e].node_val[lv] = val[e];
            idx[e] = own >> 2;
        }
    }
    std::unordered_map<uint64_t, uint32_t> root;
    for (size_t e = 0; e < E; ++e) {
        auto it = root.find(evs[e].tree);
        evs[e].root_before = it != root.end() ? it->second : initial(depth, evs[e].tree, 0);
        evs[e].root_after = depth ? val[e] : evs[e].leaf_val;
        root[evs[e].tree] = evs[e].root_after;
    }
}

// the token forest + account tree of one batch
struct DevBatch {
    bzk_mpn& w;
    VPlan P;
    std::vector<VEvent> tok, acc;           // events in time order
    uint32_t g_h2, g_h5;
    std::vector<uint32_t> tok_default;      // value ids of the empty token tree's levels
    std::map<uint64_t, bool> seen;          // accounts whose token tree has been populated in this batch
    explicit DevBatch(bzk_mpn& world) : w(world) {
        g_h2 = P.new_group(2);
        for (int lv = 0; lv <= w.T; ++lv) tok_default.push_back(P.upload(w.empty_tokens->defaults[lv]));
    }
    uint32_t token_leaf_id(const Money& m) {
        const uint32_t in[2] = {P.upload(m.token_id), P.upload(ZkScalar::from_u64(m.amount))};
        return P.add(g_h2, in);
    }
    // first touch of an account in this batch: its token tree as it stands is built by the same machinery (writes of the populated
    // slots, before any real event of the account)
    void touch(uint64_t index, const MpnAccount& as_it_stands) {
        if (seen.count(index)) return;
        seen[index] = true;
        for (auto& kv : as_it_stands.tokens) {
            VEvent e;
            e.tree = index;
            e.leaf = kv.first;
            e.leaf_val = token_leaf_id(kv.second);
            tok.push_back(e);
        }
    }
    size_t token_write(uint64_t index, uint64_t slot, const Money& m) {
        VEvent e;
        e.tree = index;
        e.leaf = slot;
        e.leaf_val = token_leaf_id(m);
        tok.push_back(e);
        return tok.size() - 1;
    }
    // account leaf H5(nonce, withdraw nonce, x, y, tokens root after token event `tok_ev`)
    struct PendingLeaf { uint64_t index; MpnAccount a; size_t tok_ev; };
    std::vector<PendingLeaf> leaves;
    size_t account_write(uint64_t index, const MpnAccount& a, size_t tok_ev) {
        leaves.push_back({index, a, tok_ev});
        return leaves.size() - 1;
    }
    int32_t run() {
        vforest_plan(w.T, tok, P, [&](int lv, uint64_t, uint64_t) { return tok_default[lv]; });
        g_h5 = P.new_group(5);
        for (auto& pl : leaves) {
            const uint32_t in[5] = {P.upload(ZkScalar::from_u64(pl.a.tx_nonce)), P.upload(ZkScalar::from_u64(pl.a.withdraw_nonce)),
                                    P.upload(pl.a.address.x), P.upload(pl.a.address.y), tok[pl.tok_ev].root_after};
            VEvent e;
            e.tree = 0;
            e.leaf = pl.index;
            e.leaf_val = P.add(g_h5, in);
            acc.push_back(e);
        }
        std::unordered_map<uint64_t, uint32_t> init_cache;
        vforest_plan(w.L, acc, P, [&](int lv, uint64_t, uint64_t i) {
            const uint64_t k = ((uint64_t)lv << 40) | i;
            auto it = init_cache.find(k);
            if (it != init_cache.end()) return it->second;
            return init_cache[k] = P.upload(w.accounts->get(lv, i));
        });
#ifdef BZK_TEST_HOOKS
        // fault injection, compiled into bazuka_amd/libbzk_testhooks.so only; read per call, so that a test can fail one
        // batch and run the next (tests/test_gpu_mpn_devtree.py)
        const char* fault = getenv("BZK_MPN_TEST_FAULT");
        const bool injected = fault && fault[0] == '1';
#else
        const bool injected = false;
#endif
        const int32_t st = injected ? BZK_E_DEVICE : P.run(w.dev);
        if (st != BZK_OK) {
            w.dev_error = injected ? "injected fault" : bzk_last_error(w.dev);
            return st;
        }
        // bring the host-side sparse account tree up to date: every event's leaf and ancestors, in time order (the last write wins)
        for (auto& e : acc) {
            uint64_t i = e.leaf;
            w.accounts->level[0][i] = P.value(e.leaf_val);
            for (int lv = 0; lv < w.L; ++lv) {
                i >>= 2;
                w.accounts->level[lv + 1][i] = P.value(e.node_val[lv]);
            }
        }
        return BZK_OK;
    }
    Proof4 proof(const VEvent& e) const {
        Proof4 p(e.sib.size());
        for (size_t lv = 0; lv < e.sib.size(); ++lv)
            for (int k = 0; k < 3; ++k) p[lv][k] = P.value(e.sib[lv][k]);
        return p;
    }
};
}  // namespace

// The device builders decide a batch on account data first (balances, nonces, queue) and hash afterwards in one batched step.  If that
// step fails (allocation, HIP error) the decision must not stay behind - the Merkle tree was not advanced, so every later root, proof
// or work of this handle would silently disagree with its accounts.  First-touch copies of the accounts; the caller keeps
// the old queue in `rest` after the swap.
struct AcctUndo {
    bzk_mpn& w;
    std::map<uint64_t, std::pair<bool, MpnAccount>> before;  // slot -> (existed, contents)
    explicit AcctUndo(bzk_mpn& w_) : w(w_) {}
    void note(uint64_t i) {
        if (before.count(i)) return;
        auto it = w.acct.find(i);
        before[i] = {it != w.acct.end(), it != w.acct.end() ? it->second : MpnAccount()};
    }
    void rollback() {
        for (auto& kv : before) {
            if (kv.second.first) w.acct[kv.first] = kv.second.second;
            else w.acct.erase(kv.first);
        }
    }
};

// update::update (src/mpn/update.rs:8-299) with the Merkle work batched on the device; same acceptance rules, same transitions
static int32_t build_transitions_dev(bzk_mpn& w, int log4_batch, const ZkScalar& fee_token, std::vector<UpdateTransition>& out,
                                     uint64_t& fee_sum, uint64_t& rejected) {
    const size_t cap = (size_t)1 << (2 * log4_batch);
    fee_sum = 0;
    rejected = 0;
    DevBatch B(w);
    AcctUndo undo(w);
    struct Rec { size_t ev_sti, ev_sfi, ev_dti, leaf_src, leaf_dst; };
    std::vector<Rec> recs;
    std::vector<MpnTx> rest;
    for (const MpnTx& tx : w.mempool) {
        if (out.size() == cap) {
            rest.push_back(tx);
            continue;
        }
        if (tx.fee.token_id != fee_token || !tx.src_pub.is_on_curve() || !tx.dst_pub.is_on_curve()) { ++rejected; continue; }
        long src_index = -1, dst_index = -1;
        for (auto& kv : w.acct) {
            if (kv.second.address == tx.src_pub && src_index < 0) src_index = (long)kv.first;
            if (kv.second.address == tx.dst_pub && dst_index < 0) dst_index = (long)kv.first;
        }
        if (src_index < 0) { ++rejected; continue; }
        if (dst_index < 0) dst_index = w.acct.empty() ? 0 : (long)(w.acct.rbegin()->first + 1);
        MpnAccount src_before = w.get(src_index), dst_before0 = w.get(dst_index);
        long sti = src_before.find_token_index(w.L, tx.amount.token_id, false);
        long dti = dst_before0.find_token_index(w.L, tx.amount.token_id, true);
        long sfi = src_before.find_token_index(w.L, tx.fee.token_id, false);
        if (sti < 0 || dti < 0 || sfi < 0) { ++rejected; continue; }
        Money src_token = src_before.tokens[sti];
        const bool dst_has = dst_before0.tokens.count(dti) != 0;
        if (tx.nonce != src_before.tx_nonce + 1 || !(src_before.address == tx.src_pub) ||
            (dst_before0.address.is_on_curve() && !(dst_before0.address == tx.dst_pub)) ||
            (dst_has && src_token.token_id != dst_before0.tokens[dti].token_id) || src_token.token_id != tx.amount.token_id ||
            src_token.amount < tx.amount.amount) {
            ++rejected;
            continue;
        }
        MpnAccount src_after = src_before;
        src_after.tx_nonce += 1;
        src_after.tokens[sti].amount -= tx.amount.amount;
        if (!src_after.tokens.count(sfi)) { ++rejected; continue; }
        const Money src_fee_token = src_after.tokens[sfi];
        if (src_fee_token.token_id != tx.fee.token_id || src_fee_token.amount < tx.fee.amount) { ++rejected; continue; }

        UpdateTransition t;
        t.enabled = true;
        t.tx = tx;
        t.src_index = src_index; t.dst_index = dst_index;
        t.src_token_index = sti; t.dst_token_index = dti; t.src_fee_token_index = sfi;
        t.src_before = src_before;
        t.src_before_balance = src_token;
        t.src_before_fee_balance = src_fee_token;
        Rec r;
        B.touch(src_index, src_before);
        r.ev_sti = B.token_write(src_index, sti, src_after.tokens[sti]);
        src_after.tokens[sfi].amount -= tx.fee.amount;
        r.ev_sfi = B.token_write(src_index, sfi, src_after.tokens[sfi]);
        undo.note(src_index);
        w.acct[src_index] = src_after;
        r.leaf_src = B.account_write(src_index, src_after, r.ev_sfi);
        MpnAccount dst_before = w.get(dst_index);  // read AFTER the sender update (matters when src == dst)
        t.dst_before = dst_before;
        t.dst_before_balance = dst_before.tokens.count(dti) ? dst_before.tokens[dti] : Money();
        MpnAccount dst_after = dst_before;
        dst_after.address = tx.dst_pub;
        if (!dst_after.tokens.count(dti)) dst_after.tokens[dti] = Money{tx.amount.token_id, 0};
        dst_after.tokens[dti].amount += tx.amount.amount;
        B.touch(dst_index, dst_before);
        r.ev_dti = B.token_write(dst_index, dti, dst_after.tokens[dti]);
        undo.note(dst_index);
        w.acct[dst_index] = dst_after;
        r.leaf_dst = B.account_write(dst_index, dst_after, r.ev_dti);
        fee_sum += tx.fee.amount;
        out.push_back(std::move(t));
        recs.push_back(r);
    }
    w.mempool.swap(rest);
    if (recs.empty()) return BZK_OK;
    if (const int32_t st = B.run(); st != BZK_OK) {  // nothing of the batch stays: accounts, queue and (untouched) tree agree again
        undo.rollback();
        w.mempool.swap(rest);
        out.resize(out.size() - recs.size());
        return st;
    }
    const size_t first = out.size() - recs.size();
    for (size_t k = 0; k < recs.size(); ++k) {
        UpdateTransition& t = out[first + k];
        const Rec& r = recs[k];
        t.src_before_balances_hash = B.P.value(B.tok[r.ev_sti].root_before);
        t.src_balance_proof = B.proof(B.tok[r.ev_sti]);
        t.src_fee_balance_proof = B.proof(B.tok[r.ev_sfi]);
        t.dst_before_balances_hash = B.P.value(B.tok[r.ev_dti].root_before);
        t.dst_balance_proof = B.proof(B.tok[r.ev_dti]);
        t.src_proof = B.proof(B.acc[r.leaf_src]);
        t.dst_proof = B.proof(B.acc[r.leaf_dst]);
        t.state_after = B.P.value(B.acc[r.leaf_dst].root_after);
    }
    return BZK_OK;
}

struct LcModeGuard {
    bool prev;
    explicit LcModeGuard(bool on) : prev(lc_tracking()) { lc_tracking() = on; }
    ~LcModeGuard() { lc_tracking() = prev; }
};

static MerkleProofWit alloc_proof(ConstraintSystem& cs, const Proof4& p) {
    MerkleProofWit w;
    for (auto& tr : p)
        for (int k = 0; k < 3; ++k) w.sib.push_back(num_alloc(cs, tr[k].v));
    return w;
}

// One transition of `impl Circuit for UpdateCircuit` (src/mpn/circuits/update_circuit.rs:81-469); steps = SURVEY App. F.
// Depends on the rest of the circuit only through `accepted_fee_token` and the running `state_wit`, and allocates
// a fixed number of variables / constraints - which is what lets transitions be synthesized on worker threads.
struct TxOut {
    Num state_out, final_fee;
};
static TxOut synth_tx(ConstraintSystem& cs, int L, int T, const Num& accepted_fee_token, const Num& state_wit,
                      const UpdateTransition& tr) {
    Bool enabled = Bool::is(bit_alloc(cs, tr.enabled));                                                  // 1
    UInt src_token_index = UInt::alloc(cs, fr_from_u64(tr.src_token_index), 2 * T);                      // 2
    UInt src_fee_token_index = UInt::alloc(cs, fr_from_u64(tr.src_fee_token_index), 2 * T);
    UInt dst_token_index = UInt::alloc(cs, fr_from_u64(tr.dst_token_index), 2 * T);
    Num src_tx_nonce = num_alloc(cs, fr_from_u64(tr.src_before.tx_nonce));                               // 3
    Num src_withdraw_nonce = num_alloc(cs, fr_from_u64(tr.src_before.withdraw_nonce));
    APoint src_addr = APoint::alloc(cs, tr.src_before.address);
    src_addr.assert_on_curve(cs, enabled);
    Num src_before_bh = num_alloc(cs, tr.src_before_balances_hash.v);                                    // 4
    Num dst_before_bh = num_alloc(cs, tr.dst_before_balances_hash.v);
    Num src_token_id = num_alloc(cs, tr.src_before_balance.token_id.v);                                  // 5
    UInt src_balance = UInt::alloc_64(cs, tr.src_before_balance.amount);
    Number src_token_balance_hash = g_poseidon(cs, {Number::from(src_token_id), src_balance.num});
    Num src_fee_token_id = num_alloc(cs, tr.src_before_fee_balance.token_id.v);                          // 6
    UInt src_fee_balance = UInt::alloc_64(cs, tr.src_before_fee_balance.amount);
    Number src_fee_token_balance_hash = g_poseidon(cs, {Number::from(src_fee_token_id), src_fee_balance.num});
    MerkleProofWit src_balance_proof = alloc_proof(cs, tr.src_balance_proof);                            // 7
    g_check_proof4(cs, enabled, src_token_index, src_token_balance_hash, src_balance_proof, Number::from(src_before_bh));
    UInt tx_amount = UInt::alloc_64(cs, tr.tx.amount.amount);                                            // 8
    UInt tx_fee = UInt::alloc_64(cs, tr.tx.fee.amount);
    Number new_token_balance_hash = g_poseidon(cs, {Number::from(src_token_id), src_balance.num.minus(tx_amount.num)});  // 9
    Number balance_middle_root = g_calc_root4(cs, src_token_index, new_token_balance_hash, src_balance_proof);
    MerkleProofWit src_fee_balance_proof = alloc_proof(cs, tr.src_fee_balance_proof);                    // 10
    g_check_proof4(cs, enabled, src_fee_token_index, src_fee_token_balance_hash, src_fee_balance_proof, balance_middle_root);
    Number new_fee_token_balance_hash =
        g_poseidon(cs, {Number::from(src_fee_token_id), src_fee_balance.num.minus(tx_fee.num)});        // 11
    Number src_balance_final_root = g_calc_root4(cs, src_fee_token_index, new_fee_token_balance_hash, src_fee_balance_proof);
    Num tx_nonce = num_alloc(cs, fr_from_u64(tr.tx.nonce));                                              // 12
    UInt tx_src_index = UInt::alloc(cs, fr_from_u64(tr.src_index), 2 * L);
    Num tx_amount_token_id = num_alloc(cs, tr.tx.amount.token_id.v);
    Num tx_fee_token_id = num_alloc(cs, tr.tx.fee.token_id.v);
    Number::from(accepted_fee_token).assert_equal_if_enabled(cs, enabled, Number::from(tx_fee_token_id));  // 13
    Number::from(src_token_id).assert_equal(cs, Number::from(tx_amount_token_id));
    Number::from(src_fee_token_id).assert_equal(cs, Number::from(tx_fee_token_id));
    Number src_hash = g_poseidon(cs, {Number::from(src_tx_nonce), Number::from(src_withdraw_nonce), Number::from(src_addr.x),
                                      Number::from(src_addr.y), Number::from(src_before_bh)});         // 14
    Num dst_token_id = num_alloc(cs, tr.dst_before_balance.token_id.v);                                  // 15
    Num dst_balance = num_alloc(cs, fr_from_u64(tr.dst_before_balance.amount));
    Number dst_token_balance_hash = g_poseidon(cs, {Number::from(dst_token_id), Number::from(dst_balance)});
    Number new_dst_token_balance_hash =
        g_poseidon(cs, {Number::from(tx_amount_token_id), Number::from(dst_balance).plus(tx_amount.num)});
    MerkleProofWit dst_balance_proof = alloc_proof(cs, tr.dst_balance_proof);                            // 16
    g_check_proof4(cs, enabled, dst_token_index, dst_token_balance_hash, dst_balance_proof, Number::from(dst_before_bh));
    Number dst_balance_final_root = g_calc_root4(cs, dst_token_index, new_dst_token_balance_hash, dst_balance_proof);
    MerkleProofWit src_proof = alloc_proof(cs, tr.src_proof);                                            // 17
    g_check_proof4(cs, enabled, tx_src_index, src_hash, src_proof, Number::from(state_wit));
    Number new_src_tx_nonce = Number::from(src_tx_nonce).plus(Number::constant(Fr::one()));              // 18
    Number new_src_hash = g_poseidon(cs, {new_src_tx_nonce, Number::from(src_withdraw_nonce), Number::from(src_addr.x),
                                          Number::from(src_addr.y), src_balance_final_root});
    Number middle_root = g_calc_root4(cs, tx_src_index, new_src_hash, src_proof);
    APoint tx_dst_addr = APoint::alloc(cs, tr.tx.dst_pub);                                               // 19
    tx_dst_addr.assert_on_curve(cs, enabled);
    UInt tx_dst_index = UInt::alloc(cs, fr_from_u64(tr.dst_index), 2 * L);
    Num dst_tx_nonce = num_alloc(cs, fr_from_u64(tr.dst_before.tx_nonce));
    Num dst_withdraw_nonce = num_alloc(cs, fr_from_u64(tr.dst_before.withdraw_nonce));
    APoint dst_addr = APoint::alloc(cs, tr.dst_before.address);
    Number dst_hash = g_poseidon(cs, {Number::from(dst_tx_nonce), Number::from(dst_withdraw_nonce), Number::from(dst_addr.x),
                                      Number::from(dst_addr.y), Number::from(dst_before_bh)});         // 20
    MerkleProofWit dst_proof = alloc_proof(cs, tr.dst_proof);
    Bool is_dst_null = dst_addr.is_null(cs);
    Bool is_dst_eq = dst_addr.is_equal(cs, tx_dst_addr);
    Bool addr_valid = boolean_or(cs, is_dst_null, is_dst_eq);
    assert_true(cs, addr_valid);
    g_check_proof4(cs, enabled, tx_dst_index, dst_hash, dst_proof, middle_root);                         // 21
    Number new_dst_hash = g_poseidon(cs, {Number::from(dst_tx_nonce), Number::from(dst_withdraw_nonce), Number::from(tx_dst_addr.x),
                                          Number::from(tx_dst_addr.y), dst_balance_final_root});
    Number next_state_wit = g_calc_root4(cs, tx_dst_index, new_dst_hash, dst_proof);
    Num state_out = mux(cs, enabled, Number::from(state_wit), next_state_wit);                           // 22
    UInt amount_plus_fee = UInt::constrain(cs, tx_amount.num.plus(tx_fee.num), 64);                      // 23
    Bool is_lte = amount_plus_fee.lte(cs, src_balance);
    assert_true(cs, is_lte);
    Number::from(tx_nonce).assert_equal_if_enabled(cs, enabled, Number::from(src_tx_nonce).plus(Number::constant(Fr::one())));  // 24
    Num final_fee = mux(cs, enabled, Number::zero(), tx_fee.num);
    Number tx_hash = g_poseidon(cs, {Number::from(tx_nonce), Number::from(tx_dst_addr.x), Number::from(tx_dst_addr.y),
                                     Number::from(tx_amount_token_id), tx_amount.num, Number::from(tx_fee_token_id), tx_fee.num},
                                true);  // 25 (its VALUE feeds the signature gadget's bit decomposition: hashed on the host even when values are deferred)
    APoint sig_r = APoint::alloc(cs, tr.tx.sig.r);
    sig_r.assert_on_curve(cs, enabled);
    Num sig_s = num_alloc(cs, tr.tx.sig.s.v);
    g_verify_eddsa(cs, enabled, src_addr, tx_hash, sig_r, sig_s);
        return {state_out, final_fee};
}

// impl Circuit for UpdateCircuit (src/mpn/circuits/update_circuit.rs:49-494)
static void synthesize_update(ConstraintSystem& cs, int L, int T, const ZkScalar& commitment, uint64_t height, const ZkScalar& state,
                              const ZkScalar& aux_data, const ZkScalar& next_state, const ZkScalar& fee_token,
                              const std::vector<UpdateTransition>& transitions, int nthreads, DeferData* dd = nullptr) {
    Num commitment_wit = num_alloc(cs, commitment.v);
    num_inputize(cs, commitment_wit);
    Num height_wit = num_alloc(cs, fr_from_u64(height));
    num_inputize(cs, height_wit);
    Num state_wit = num_alloc(cs, state.v);
    num_inputize(cs, state_wit);
    Num accepted_fee_token = num_alloc(cs, fee_token.v);
    Num aux_wit = num_alloc(cs, aux_data.v);
    num_inputize(cs, aux_wit);
    Num claimed_next = num_alloc(cs, next_state.v);
    num_inputize(cs, claimed_next);
    Number fee_sum = Number::zero();

    bool done = false;
    if (lc_tracking()) dd = nullptr;  // deferral is a witness-only form
    if (!lc_tracking() && ((nthreads > 1 && transitions.size() > 1) || (dd && !transitions.empty()))) {
        // Witness-only mode, parallel over transitions.  Variable identities do not matter here (no LCs), only the
        // ORDER of the emitted values, so each worker fills a private system and the pieces are concatenated.
        // The state entering transition t is predicted from the witness builder (`state_after` chain) and checked
        // against what the circuit computes; on any disagreement (an invalid witness) fall back to the sequential
        // walk so that the reported unsatisfied constraint is exactly the sequential one.
        const size_t n = transitions.size();
        size_t size_hint_aux = 0, size_hint_cons = 0;
        {
            // every transition of a circuit allocates the same number of variables and constraints: measured once per
            // (L, T) on a disabled slot
            static std::mutex mu;
            static std::map<std::pair<int, int>, std::pair<size_t, size_t>> shape;
            std::lock_guard<std::mutex> lk(mu);
            auto it = shape.find({L, T});
            if (it == shape.end()) {
                ConstraintSystem probe(false);
                LcModeGuard g(false);
                Num st0 = {VAR_ONE, state.v};
                synth_tx(probe, L, T, accepted_fee_token, st0, UpdateTransition::null(L, T));
                it = shape.emplace(std::make_pair(L, T), std::make_pair(probe.aux.size(), probe.az.size())).first;
            }
            size_hint_aux = it->second.first;
            size_hint_cons = it->second.second;
        }
        // deferred values: the program of one transition, recorded once per (L, T) on a null transition (every transition runs the same
        // gadget calls); the chain check against the builder's predicted state is its last op
        const DeferProgram* prog = nullptr;
        auto chain_check = [](ConstraintSystem& part, Defer& D, const Num& state_out, const Fr& predicted) {
            wf::Op f{};
            f.kind = wf::F_CHECK_EQ; f.out = -1;
            f.aux_off = (uint32_t)part.win_n_aux; f.con_off = (uint32_t)part.win_n_con;
            f.in[0] = state_out.ref >= 0 ? state_out.ref : D.in_val(state_out.val);
            f.in[1] = D.in_val(predicted);
            D.emit(f);
        };
        if (dd) {
            static std::mutex mu;
            static std::map<std::pair<int, int>, std::unique_ptr<DeferProgram>> progs;
            std::lock_guard<std::mutex> lk(mu);
            auto it = progs.find({L, T});
            if (it == progs.end()) {
                std::unique_ptr<DeferProgram> P(new DeferProgram());
                std::vector<Fr> sa(size_hint_aux), sx(size_hint_cons), sy(size_hint_cons), sz(size_hint_cons), sin(1 << 14);
                ConstraintSystem plan(false);
                plan.set_window(sa.data(), size_hint_aux, sx.data(), sy.data(), sz.data(), size_hint_cons);
                Defer D{P.get(), true, sin.data(), (uint32_t)sin.size()};
                plan.defer = &D;
                LcModeGuard g(false);
                Num st0 = {VAR_ONE, state.v};
                TxOut o = synth_tx(plan, L, T, accepted_fee_token, st0, UpdateTransition::null(L, T));
                chain_check(plan, D, o.state_out, state.v);
                if (D.overflow || plan.win_overflow || plan.win_n_aux != size_hint_aux || plan.win_n_con != size_hint_cons)
                    throw std::logic_error("deferred synthesis changed the shape of a transition");
                P->n_regs = (uint32_t)D.n_regs;
                P->n_inputs = D.n_in;
                P->n_aux = size_hint_aux;
                P->n_con = size_hint_cons;
                P->finalize();
                it = progs.emplace(std::make_pair(L, T), std::move(P)).first;
            }
            prog = it->second.get();
            dd->prog = prog;
            dd->n_tx = n;
            dd->inputs.resize(n * (size_t)prog->n_inputs);
        }
        // workers write straight into their slice of the final (pinned) arrays: no per-transition buffers, no merge copy
        const auto ta0 = std::chrono::steady_clock::now();
        const size_t base_aux = cs.aux.size(), base_con = cs.az.size();
        cs.aux.reserve(base_aux + n * size_hint_aux + 4096);
        cs.az.reserve(base_con + n * size_hint_cons + 4096);
        cs.bz.reserve(base_con + n * size_hint_cons + 4096);
        cs.cz.reserve(base_con + n * size_hint_cons + 4096);
        cs.aux.resize(base_aux + n * size_hint_aux);
        cs.az.resize(base_con + n * size_hint_cons);
        cs.bz.resize(base_con + n * size_hint_cons);
        cs.cz.resize(base_con + n * size_hint_cons);
        if (getenv("BZK_DEBUG"))
            fprintf(stderr, "[bzk] witness arrays sized: %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - ta0).count());
        std::vector<Fr> state_in(n + 1);
        state_in[0] = state.v;
        for (size_t t = 0; t < n; ++t) state_in[t + 1] = transitions[t].enabled ? transitions[t].state_after.v : state_in[t];
        std::vector<TxOut> outs(n);
        std::vector<uint8_t> shape_ok(n, 0);
        std::atomic<size_t> next(0);
        auto worker = [&] {
            LcModeGuard g(false);
            for (;;) {
                const size_t t = next.fetch_add(1);
                if (t >= n) break;
                ConstraintSystem part(false);
                part.set_window(cs.aux.data() + base_aux + t * size_hint_aux, size_hint_aux, cs.az.data() + base_con + t * size_hint_cons,
                                cs.bz.data() + base_con + t * size_hint_cons, cs.cz.data() + base_con + t * size_hint_cons, size_hint_cons);
                Num st_in = {VAR_ONE, state_in[t]};
                const auto q0 = std::chrono::steady_clock::now();
                Defer D{nullptr, false, dd ? dd->inputs.data() + t * (size_t)prog->n_inputs : nullptr, dd ? prog->n_inputs : 0u};
                if (dd) part.defer = &D;
                outs[t] = synth_tx(part, L, T, accepted_fee_token, st_in, transitions[t]);
                if (dd) chain_check(part, D, outs[t].state_out, state_in[t + 1]);
                shape_ok[t] = !part.win_overflow && part.win_n_aux == size_hint_aux && part.win_n_con == size_hint_cons &&
                              (!dd || (!D.overflow && D.n_in == prog->n_inputs && (uint32_t)D.n_regs == prog->n_regs));
                if (getenv("BZK_DEBUG") && t < 3)
                    fprintf(stderr, "[bzk]   tx %zu: %.3f s\n", t, std::chrono::duration<double>(std::chrono::steady_clock::now() - q0).count());
            }
        };
        std::vector<std::thread> th;
        const int nt = (int)std::min<size_t>((size_t)nthreads, n);
        const auto tp0 = std::chrono::steady_clock::now();
        for (int i = 1; i < nt; ++i) th.emplace_back(worker);
        worker();
        for (auto& x : th) x.join();
        if (getenv("BZK_DEBUG"))
            fprintf(stderr, "[bzk] %d workers: %.3f s\n", nt, std::chrono::duration<double>(std::chrono::steady_clock::now() - tp0).count());
        bool chain_ok = true;
        for (size_t t = 0; t < n; ++t) {
            // (deferred: the computed state only exists on the device, which raises wf::FLAG_CHAIN where it differs from the prediction)
            if (!shape_ok[t] || (!dd && !outs[t].state_out.val.equals(state_in[t + 1]))) {
                if (getenv("BZK_DEBUG")) fprintf(stderr, "[bzk] state chain / shape mismatch at transition %zu (enabled %d)\n", t, (int)transitions[t].enabled);
                chain_ok = false;
            }
        }
        if (chain_ok) {
            for (size_t t = 0; t < n; ++t) fee_sum.add_num(Fr::one(), outs[t].final_fee);
            state_wit = {VAR_ONE, state_in[n]};
            done = true;
            if (dd) {
                dd->base_aux = base_aux;
                dd->base_con = base_con;
                dd->stride_aux = size_hint_aux;
                dd->stride_con = size_hint_cons;
            }
        } else {  // back to the state before the workers ran; the sequential walk below redoes everything
            cs.aux.resize(base_aux);
            cs.az.resize(base_con);
            cs.bz.resize(base_con);
            cs.cz.resize(base_con);
        }
    }
    if (!done && dd) {  // nothing was deferred after all (the sequential walk below computes every value)
        dd->prog = nullptr;
        dd->n_tx = 0;
    }
    if (!done) {
        for (const UpdateTransition& tr : transitions) {
            TxOut o = synth_tx(cs, L, T, accepted_fee_token, state_wit, tr);
            state_wit = o.state_out;
            fee_sum.add_num(Fr::one(), o.final_fee);
        }
    }
    Number fee_hash = g_poseidon(cs, {Number::from(accepted_fee_token), fee_sum});
    cs.enforce(LC::of(aux_wit.var), aux_wit.val, LC::one(), Fr::one(), fee_hash.lc, fee_hash.val);
    cs.enforce(LC::of(state_wit.var), state_wit.val, LC::one(), Fr::one(), LC::of(claimed_next.var), claimed_next.val);
    cs.finalize();
}

// ------------------------------------------------------------------------------------------------
// Per-transaction bodies of a circuit's main loop, run either sequentially on `cs` (matrix-recording mode, one thread,
// or a witness that does not follow the predicted state chain) or - witness-only mode - by worker threads that write
// straight into their slice of the final arrays (ConstraintSystem::set_window).  Every body allocates the same number
// of variables and constraints (`shape`, measured once per circuit shape by the caller); the state entering
// transaction i is predicted from the witness builder's `state_after` chain and checked against what the body
// computes, so a bad witness falls back to the sequential walk and fails at exactly the sequential constraint.
//   body(cs, i, state_in) -> state_out
// ------------------------------------------------------------------------------------------------
// shape (variables, constraints) of one per-transaction body, measured once per (circuit kind, L, T) on a scratch system
template <class Probe>
static std::pair<size_t, size_t> body_shape(int kind, int L, int T, Probe probe) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, std::pair<size_t, size_t>> cache;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(std::make_tuple(kind, L, T));
        if (it != cache.end()) return it->second;
    }
    ConstraintSystem scratch(false);
    const std::pair<size_t, size_t> sh = probe(scratch);
    std::lock_guard<std::mutex> lk(mu);
    cache[std::make_tuple(kind, L, T)] = sh;
    return sh;
}
// predicted state entering each transaction (and leaving the last): the witness builder's `state_after` chain
template <class Tr>
static std::vector<Fr> chain_states(const ZkScalar& state, const std::vector<Tr>& trs) {
    std::vector<Fr> st(trs.size() + 1);
    st[0] = state.v;
    for (size_t t = 0; t < trs.size(); ++t) st[t + 1] = trs[t].enabled ? trs[t].state_after.v : st[t];
    return st;
}

// the last op of every transition's deferred-value program: the state it computes against the state the witness builder predicted for it
static void defer_chain_check(ConstraintSystem& part, Defer& D, const Num& state_out, const Fr& predicted) {
    wf::Op f{};
    f.kind = wf::F_CHECK_EQ; f.out = -1;
    f.aux_off = (uint32_t)part.win_n_aux; f.con_off = (uint32_t)part.win_n_con;
    f.in[0] = state_out.ref >= 0 ? state_out.ref : D.in_val(state_out.val);
    f.in[1] = D.in_val(predicted);
    D.emit(f);
}
// dd (optional, witness-only mode): the hash-dependent values of every body are deferred (host_r1cs.h DeferProgram); prog_key names the circuit
// shape (kind, L, T) whose program this is - recorded once, on body 0 run against a scratch window
template <class Body>
static Num run_tx_bodies(ConstraintSystem& cs, int nthreads, size_t n, std::pair<size_t, size_t> shape, const Num& state0,
                         const std::vector<Fr>& state_in, Body body, DeferData* dd = nullptr, uint64_t prog_key = 0) {
    if (lc_tracking()) dd = nullptr;
    if (!lc_tracking() && ((nthreads > 1 && n > 1) || (dd && n >= 1)) && shape.first && shape.second) {
        const DeferProgram* prog = nullptr;
        if (dd) {
            static std::mutex mu;
            static std::map<uint64_t, std::unique_ptr<DeferProgram>> progs;
            std::lock_guard<std::mutex> lk(mu);
            auto it = progs.find(prog_key);
            if (it == progs.end()) {
                std::unique_ptr<DeferProgram> P(new DeferProgram());
                std::vector<Fr> sa(shape.first), sx(shape.second), sy(shape.second), sz(shape.second), sin(1 << 14);
                ConstraintSystem plan(false);
                plan.set_window(sa.data(), shape.first, sx.data(), sy.data(), sz.data(), shape.second);
                Defer D{P.get(), true, sin.data(), (uint32_t)sin.size()};
                plan.defer = &D;
                LcModeGuard g(false);
                const Num st_in = {VAR_ONE, state_in[0]};
                const Num st_out = body(plan, 0, st_in);
                defer_chain_check(plan, D, st_out, state_in[1]);
                if (D.overflow || plan.win_overflow || plan.win_n_aux != shape.first || plan.win_n_con != shape.second)
                    throw std::logic_error("deferred synthesis changed the shape of a transition");
                P->n_regs = (uint32_t)D.n_regs;
                P->n_inputs = D.n_in;
                P->n_aux = shape.first;
                P->n_con = shape.second;
                P->finalize();
                it = progs.emplace(prog_key, std::move(P)).first;
            }
            prog = it->second.get();
            dd->prog = prog;
            dd->n_tx = n;
            dd->inputs.resize(n * (size_t)prog->n_inputs);
        }
        const size_t base_aux = cs.aux.size(), base_con = cs.az.size();
        cs.aux.resize(base_aux + n * shape.first);
        cs.az.resize(base_con + n * shape.second);
        cs.bz.resize(base_con + n * shape.second);
        cs.cz.resize(base_con + n * shape.second);
        std::vector<uint8_t> ok(n, 0);
        std::atomic<size_t> next(0);
        auto worker = [&] {
            LcModeGuard g(false);
            for (;;) {
                const size_t t = next.fetch_add(1);
                if (t >= n) break;
                ConstraintSystem part(false);
                part.set_window(cs.aux.data() + base_aux + t * shape.first, shape.first, cs.az.data() + base_con + t * shape.second,
                                cs.bz.data() + base_con + t * shape.second, cs.cz.data() + base_con + t * shape.second, shape.second);
                const Num st_in = {VAR_ONE, state_in[t]};
                Defer D{nullptr, false, dd ? dd->inputs.data() + t * (size_t)prog->n_inputs : nullptr, dd ? prog->n_inputs : 0u};
                if (dd) part.defer = &D;
                const Num st_out = body(part, t, st_in);
                if (dd) defer_chain_check(part, D, st_out, state_in[t + 1]);
                // (deferred: the computed state only exists on the device, which raises wf::FLAG_CHAIN where it differs from the prediction)
                ok[t] = !part.win_overflow && part.win_n_aux == shape.first && part.win_n_con == shape.second &&
                        (dd ? (!D.overflow && D.n_in == prog->n_inputs && (uint32_t)D.n_regs == prog->n_regs) : st_out.val.equals(state_in[t + 1]));
            }
        };
        std::vector<std::thread> th;
        const int nt = (int)std::min<size_t>((size_t)nthreads, n);
        for (int i = 1; i < nt; ++i) th.emplace_back(worker);
        worker();
        for (auto& x : th) x.join();
        bool all_ok = true;
        for (size_t t = 0; t < n; ++t) all_ok = all_ok && ok[t];
        if (getenv("BZK_DEBUG")) fprintf(stderr, "[bzk] run_tx_bodies: %zu bodies on %d threads, shape (%zu, %zu): %s\n", n, nt, shape.first, shape.second, all_ok ? "ok" : "MISMATCH -> sequential");
        if (all_ok) {
            if (dd) {
                dd->base_aux = base_aux;
                dd->base_con = base_con;
                dd->stride_aux = shape.first;
                dd->stride_con = shape.second;
            }
            return {VAR_ONE, state_in[n]};
        }
        if (dd) {  // nothing is deferred after all: the sequential walk below computes every value
            dd->prog = nullptr;
            dd->n_tx = 0;
        }
        cs.aux.resize(base_aux);
        cs.az.resize(base_con);
        cs.bz.resize(base_con);
        cs.cz.resize(base_con);
    }
    Num st = state0;
    for (size_t i = 0; i < n; ++i) st = body(cs, i, st);
    return st;
}

// ------------------------------------------------------------------------------------------------
// reveal gadget (src/zk/groth16/gadgets/reveal/mod.rs:13-61) for the one shape the MPN circuits use:
// List{log4 = B, Struct[k scalars]}: a Poseidon of the k fields per item, then a 4-ary Poseidon tree
// ------------------------------------------------------------------------------------------------
static Number g_reveal_batch(ConstraintSystem& cs, const std::vector<std::vector<Number>>& items) {
    std::vector<Number> leaves;
    for (auto& fields : items) leaves.push_back(g_poseidon(cs, fields));
    while (leaves.size() != 1) {
        std::vector<Number> next;
        for (size_t i = 0; i < leaves.size(); i += 4) next.push_back(g_poseidon(cs, {leaves[i], leaves[i + 1], leaves[i + 2], leaves[i + 3]}));
        leaves.swap(next);
    }
    return leaves[0];
}
// native counterpart: ZkStateBuilder::compress of the same model with only the first `n_set` items present
static ZkScalar native_batch_root(const std::vector<std::vector<ZkScalar>>& items, int log4_batch, int n_fields) {
    std::vector<ZkScalar> zero_fields((size_t)n_fields);
    const ZkScalar def = poseidon_hash(zero_fields);  // compress_default of the Struct
    std::vector<ZkScalar> leaves((size_t)1 << (2 * log4_batch), def);
    for (size_t i = 0; i < items.size(); ++i) leaves[i] = poseidon_hash(items[i]);
    while (leaves.size() != 1) {
        std::vector<ZkScalar> next;
        for (size_t i = 0; i < leaves.size(); i += 4) next.push_back(poseidon_hash(&leaves[i], 4));
        leaves.swap(next);
    }
    return leaves[0];
}

// ------------------------------------------------------------------------------------------------
// deposit::deposit (src/mpn/deposit.rs:11-233) and DepositCircuit (src/mpn/circuits/deposit_circuit.rs:47-293)
// ------------------------------------------------------------------------------------------------
static void build_deposits(bzk_mpn& w, int log4_batch, std::vector<DepositTransition>& out, uint64_t& rejected) {
    const size_t cap = (size_t)1 << (2 * log4_batch);
    rejected = 0;
    std::vector<DepositTx> rest;
    for (const DepositTx& tx : w.deposit_queue) {
        if (out.size() == cap) { rest.push_back(tx); continue; }
        long index = -1;
        for (auto& kv : w.acct)
            if (kv.second.address == tx.mpn_address) { index = (long)kv.first; break; }
        if (index < 0) index = w.acct.empty() ? 0 : (long)(w.acct.rbegin()->first + 1);
        MpnAccount acc = w.get(index);
        long ti = acc.find_token_index(w.L, tx.amount.token_id, true);
        if (ti < 0) { ++rejected; continue; }
        const bool has = acc.tokens.count(ti) != 0;
        if ((!(acc.address == PointAffine()) && !(tx.mpn_address == acc.address)) || (has && acc.tokens[ti].token_id != tx.amount.token_id)) {
            ++rejected;
            continue;
        }
        DepositTransition t;
        t.enabled = true;
        t.tx = tx;
        t.account_index = index;
        t.token_index = ti;
        t.before = acc;
        SparseTree4 tree = w.tokens_tree(acc);
        t.before_balances_hash = tree.root();
        t.before_balance = has ? acc.tokens[ti] : Money();
        t.balance_proof = tree.prove(ti);
        t.proof = w.accounts->prove(index);
        MpnAccount upd = acc;
        upd.address = tx.mpn_address;
        if (!has) upd.tokens[ti] = Money{tx.amount.token_id, 0};
        upd.tokens[ti].amount += tx.amount.amount;
        tree.set_leaf(ti, token_leaf(upd.tokens[ti]));
        w.set_with_tokens_root(index, upd, tree.root());
        t.state_after = w.accounts->root();
        out.push_back(std::move(t));
    }
    w.deposit_queue.swap(rest);
}

// deposit::deposit (src/mpn/deposit.rs:11-233) with the Merkle work batched on the device (see DevBatch)
static int32_t build_deposits_dev(bzk_mpn& w, int log4_batch, std::vector<DepositTransition>& out, uint64_t& rejected) {
    const size_t cap = (size_t)1 << (2 * log4_batch);
    rejected = 0;
    DevBatch B(w);
    AcctUndo undo(w);
    struct Rec { size_t ev_tok, leaf; bool had_tokens; };
    std::vector<Rec> recs;
    std::vector<DepositTx> rest;
    for (const DepositTx& tx : w.deposit_queue) {
        if (out.size() == cap) { rest.push_back(tx); continue; }
        long index = -1;
        for (auto& kv : w.acct)
            if (kv.second.address == tx.mpn_address) { index = (long)kv.first; break; }
        if (index < 0) index = w.acct.empty() ? 0 : (long)(w.acct.rbegin()->first + 1);
        MpnAccount acc = w.get(index);
        long ti = acc.find_token_index(w.L, tx.amount.token_id, true);
        if (ti < 0) { ++rejected; continue; }
        const bool has = acc.tokens.count(ti) != 0;
        if ((!(acc.address == PointAffine()) && !(tx.mpn_address == acc.address)) || (has && acc.tokens[ti].token_id != tx.amount.token_id)) {
            ++rejected;
            continue;
        }
        DepositTransition t;
        t.enabled = true;
        t.tx = tx;
        t.account_index = index;
        t.token_index = ti;
        t.before = acc;
        t.before_balance = has ? acc.tokens[ti] : Money();
        MpnAccount upd = acc;
        upd.address = tx.mpn_address;
        if (!has) upd.tokens[ti] = Money{tx.amount.token_id, 0};
        upd.tokens[ti].amount += tx.amount.amount;
        Rec r;
        B.touch(index, acc);
        r.ev_tok = B.token_write(index, ti, upd.tokens[ti]);
        undo.note(index);
        w.acct[index] = upd;
        r.leaf = B.account_write(index, upd, r.ev_tok);
        out.push_back(std::move(t));
        recs.push_back(r);
    }
    w.deposit_queue.swap(rest);
    if (recs.empty()) return BZK_OK;
    if (const int32_t st = B.run(); st != BZK_OK) {  // see AcctUndo: nothing of the batch stays behind
        undo.rollback();
        w.deposit_queue.swap(rest);
        out.resize(out.size() - recs.size());
        return st;
    }
    const size_t first = out.size() - recs.size();
    for (size_t k = 0; k < recs.size(); ++k) {
        DepositTransition& t = out[first + k];
        t.before_balances_hash = B.P.value(B.tok[recs[k].ev_tok].root_before);
        t.balance_proof = B.proof(B.tok[recs[k].ev_tok]);
        t.proof = B.proof(B.acc[recs[k].leaf]);
        t.state_after = B.P.value(B.acc[recs[k].leaf].root_after);
    }
    return BZK_OK;
}

static ZkScalar deposit_aux(const std::vector<DepositTransition>& trs, int log4_batch) {
    std::vector<std::vector<ZkScalar>> items;
    for (auto& t : trs) {
        if (!t.enabled) break;
        ZkScalar pk[2] = {t.tx.mpn_address.x, t.tx.mpn_address.y};
        items.push_back({ZkScalar::from_u64(1), t.tx.amount.token_id, ZkScalar::from_u64(t.tx.amount.amount), poseidon_hash(pk, 2)});
    }
    return native_batch_root(items, log4_batch, 4);
}

static void synthesize_deposit(ConstraintSystem& cs, int L, int T, const ZkScalar& commitment, uint64_t height, const ZkScalar& state,
                               const ZkScalar& aux_data, const ZkScalar& next_state, const std::vector<DepositTransition>& trs,
                               int nthreads = 1, DeferData* dd = nullptr) {
    Num commitment_wit = num_alloc(cs, commitment.v);
    num_inputize(cs, commitment_wit);
    Num height_wit = num_alloc(cs, fr_from_u64(height));
    num_inputize(cs, height_wit);
    Num state_wit = num_alloc(cs, state.v);
    num_inputize(cs, state_wit);
    Num aux_wit = num_alloc(cs, aux_data.v);
    num_inputize(cs, aux_wit);
    Num claimed_next = num_alloc(cs, next_state.v);
    num_inputize(cs, claimed_next);
    struct TxWit { Bool enabled; Num token_id; UInt amount; APoint pub_key; };
    std::vector<TxWit> wits;
    std::vector<std::vector<Number>> children;
    for (const DepositTransition& tr : trs) {
        Bit enabled = bit_alloc(cs, tr.enabled);
        Num token_id = num_alloc(cs, tr.tx.amount.token_id.v);
        UInt amount = UInt::alloc_64(cs, tr.tx.amount.amount);
        APoint pub_key = APoint::alloc(cs, tr.tx.mpn_address);
        wits.push_back({Bool::is(enabled), token_id, amount, pub_key});
        Number pkh = g_poseidon(cs, {Number::from(pub_key.x), Number::from(pub_key.y)});
        Num calldata = mux(cs, Bool::is(enabled), Number::zero(), pkh);
        children.push_back({Number::from(enabled), Number::from(token_id), amount.num, Number::from(calldata)});
    }
    Number tx_root = g_reveal_batch(cs, children);
    cs.enforce(LC::of(aux_wit.var), aux_wit.val, LC::one(), Fr::one(), tx_root.lc, tx_root.val);
    auto body = [&](ConstraintSystem& cs, size_t i, const Num& state_wit) -> Num {
        const DepositTransition& tr = trs[i];
        const TxWit& tw = wits[i];
        UInt tx_index = UInt::alloc(cs, fr_from_u64(tr.account_index), 2 * L);
        UInt tx_token_index = UInt::alloc(cs, fr_from_u64(tr.token_index), 2 * T);
        tw.pub_key.assert_on_curve(cs, tw.enabled);
        Num src_tx_nonce = num_alloc(cs, fr_from_u64(tr.before.tx_nonce));
        Num src_withdraw_nonce = num_alloc(cs, fr_from_u64(tr.before.withdraw_nonce));
        APoint src_addr = APoint::alloc(cs, tr.before.address);
        Num src_balances_hash = num_alloc(cs, tr.before_balances_hash.v);
        Num src_token_id = num_alloc(cs, tr.before_balance.token_id.v);
        Num src_balance = num_alloc(cs, fr_from_u64(tr.before_balance.amount));
        Number src_token_balance_hash = g_poseidon(cs, {Number::from(src_token_id), Number::from(src_balance)});
        MerkleProofWit balance_proof = alloc_proof(cs, tr.balance_proof);
        g_check_proof4(cs, tw.enabled, tx_token_index, src_token_balance_hash, balance_proof, Number::from(src_balances_hash));
        Number src_hash = g_poseidon(cs, {Number::from(src_tx_nonce), Number::from(src_withdraw_nonce), Number::from(src_addr.x),
                                          Number::from(src_addr.y), Number::from(src_balances_hash)});
        MerkleProofWit proof = alloc_proof(cs, tr.proof);
        Bool tok_null = Number::from(src_token_id).is_zero(cs);
        Bool tok_eq = Number::from(src_token_id).is_equal(cs, Number::from(tw.token_id));
        assert_true(cs, boolean_or(cs, tok_null, tok_eq));
        Bool addr_null = src_addr.is_null(cs);
        Bool addr_eq = src_addr.is_equal(cs, tw.pub_key);
        assert_true(cs, boolean_or(cs, addr_null, addr_eq));
        g_check_proof4(cs, tw.enabled, tx_index, src_hash, proof, Number::from(state_wit));
        Number new_leaf = g_poseidon(cs, {Number::from(tw.token_id), Number::from(src_balance).plus(tw.amount.num)});
        Number new_balances_hash = g_calc_root4(cs, tx_token_index, new_leaf, balance_proof);
        Number new_hash = g_poseidon(cs, {Number::from(src_tx_nonce), Number::from(src_withdraw_nonce), Number::from(tw.pub_key.x),
                                          Number::from(tw.pub_key.y), new_balances_hash});
        Number next_state_wit = g_calc_root4(cs, tx_index, new_hash, proof);
        return mux(cs, tw.enabled, Number::from(state_wit), next_state_wit);
    };
    state_wit = run_tx_bodies(cs, nthreads, trs.size(), body_shape(0, L, T, [&](ConstraintSystem& pcs) {
                                  LcModeGuard g(false);
                                  const Num st0 = {VAR_ONE, state.v};
                                  const size_t a0 = pcs.aux.size(), c0 = pcs.az.size();
                                  body(pcs, 0, st0);
                                  return std::make_pair(pcs.aux.size() - a0, pcs.az.size() - c0);
                              }),
                              state_wit, chain_states(state, trs), body, dd, ((uint64_t)0 << 32) | ((uint64_t)L << 16) | (uint64_t)T);
    cs.enforce(LC::of(state_wit.var), state_wit.val, LC::one(), Fr::one(), LC::of(claimed_next.var), claimed_next.val);
    cs.finalize();
}

// ------------------------------------------------------------------------------------------------
// withdraw::withdraw (src/mpn/withdraw.rs:10-259) and WithdrawCircuit (src/mpn/circuits/withdraw_circuit.rs:50-413)
// ------------------------------------------------------------------------------------------------
static void build_withdraws(bzk_mpn& w, int log4_batch, std::vector<WithdrawTransition>& out, uint64_t& rejected) {
    const size_t cap = (size_t)1 << (2 * log4_batch);
    rejected = 0;
    std::vector<WithdrawTx> rest;
    for (const WithdrawTx& tx : w.withdraw_queue) {
        if (out.size() == cap) { rest.push_back(tx); continue; }
        long index = -1;
        for (auto& kv : w.acct)
            if (kv.second.address == tx.mpn_address) { index = (long)kv.first; break; }
        if (index < 0) { ++rejected; continue; }
        MpnAccount acc = w.get(index);
        long ti = acc.find_token_index(w.L, tx.amount.token_id, false), fi = acc.find_token_index(w.L, tx.fee.token_id, false);
        if (ti < 0 || fi < 0 || !acc.tokens.count(ti)) { ++rejected; continue; }
        const Money acc_token = acc.tokens[ti];
        if ((!(acc.address == PointAffine()) && !(tx.mpn_address == acc.address)) ||
            !jubjub_verify(tx.mpn_address, tx.sign_message(), tx.sig) || tx.nonce != acc.withdraw_nonce + 1 ||
            tx.amount.token_id != acc_token.token_id || tx.amount.amount > acc_token.amount) {
            ++rejected;
            continue;
        }
        MpnAccount upd = acc;
        upd.address = tx.mpn_address;
        upd.withdraw_nonce += 1;
        upd.tokens[ti].amount -= tx.amount.amount;
        if (!upd.tokens.count(fi)) { ++rejected; continue; }
        const Money acc_fee = upd.tokens[fi];
        if (tx.fee.token_id != acc_fee.token_id || tx.fee.amount > acc_fee.amount) { ++rejected; continue; }
        WithdrawTransition t;
        t.enabled = true;
        t.tx = tx;
        t.account_index = index; t.token_index = ti; t.fee_token_index = fi;
        t.before = acc;
        t.before_token_balance = acc_token;
        t.before_fee_balance = acc_fee;
        SparseTree4 tree = w.tokens_tree(acc);
        t.before_token_hash = tree.root();
        t.token_balance_proof = tree.prove(ti);
        tree.set_leaf(ti, token_leaf(upd.tokens[ti]));
        t.fee_balance_proof = tree.prove(fi);
        upd.tokens[fi].amount -= tx.fee.amount;
        tree.set_leaf(fi, token_leaf(upd.tokens[fi]));
        t.proof = w.accounts->prove(index);  // siblings of the account leaf (independent of the leaf itself)
        w.set_with_tokens_root(index, upd, tree.root());
        t.state_after = w.accounts->root();
        out.push_back(std::move(t));
    }
    w.withdraw_queue.swap(rest);
}

// withdraw::withdraw (src/mpn/withdraw.rs:10-259) with the Merkle work batched on the device (see DevBatch)
static int32_t build_withdraws_dev(bzk_mpn& w, int log4_batch, std::vector<WithdrawTransition>& out, uint64_t& rejected) {
    const size_t cap = (size_t)1 << (2 * log4_batch);
    rejected = 0;
    DevBatch B(w);
    AcctUndo undo(w);
    struct Rec { size_t ev_ti, ev_fi, leaf; };
    std::vector<Rec> recs;
    std::vector<WithdrawTx> rest;
    for (const WithdrawTx& tx : w.withdraw_queue) {
        if (out.size() == cap) { rest.push_back(tx); continue; }
        long index = -1;
        for (auto& kv : w.acct)
            if (kv.second.address == tx.mpn_address) { index = (long)kv.first; break; }
        if (index < 0) { ++rejected; continue; }
        MpnAccount acc = w.get(index);
        long ti = acc.find_token_index(w.L, tx.amount.token_id, false), fi = acc.find_token_index(w.L, tx.fee.token_id, false);
        if (ti < 0 || fi < 0 || !acc.tokens.count(ti)) { ++rejected; continue; }
        const Money acc_token = acc.tokens[ti];
        if ((!(acc.address == PointAffine()) && !(tx.mpn_address == acc.address)) ||
            !jubjub_verify(tx.mpn_address, tx.sign_message(), tx.sig) || tx.nonce != acc.withdraw_nonce + 1 ||
            tx.amount.token_id != acc_token.token_id || tx.amount.amount > acc_token.amount) {
            ++rejected;
            continue;
        }
        MpnAccount upd = acc;
        upd.address = tx.mpn_address;
        upd.withdraw_nonce += 1;
        upd.tokens[ti].amount -= tx.amount.amount;
        if (!upd.tokens.count(fi)) { ++rejected; continue; }
        const Money acc_fee = upd.tokens[fi];
        if (tx.fee.token_id != acc_fee.token_id || tx.fee.amount > acc_fee.amount) { ++rejected; continue; }
        WithdrawTransition t;
        t.enabled = true;
        t.tx = tx;
        t.account_index = index; t.token_index = ti; t.fee_token_index = fi;
        t.before = acc;
        t.before_token_balance = acc_token;
        t.before_fee_balance = acc_fee;
        Rec r;
        B.touch(index, acc);
        r.ev_ti = B.token_write(index, ti, upd.tokens[ti]);
        upd.tokens[fi].amount -= tx.fee.amount;
        r.ev_fi = B.token_write(index, fi, upd.tokens[fi]);
        undo.note(index);
        w.acct[index] = upd;
        r.leaf = B.account_write(index, upd, r.ev_fi);
        out.push_back(std::move(t));
        recs.push_back(r);
    }
    w.withdraw_queue.swap(rest);
    if (recs.empty()) return BZK_OK;
    if (const int32_t st = B.run(); st != BZK_OK) {  // see AcctUndo: nothing of the batch stays behind
        undo.rollback();
        w.withdraw_queue.swap(rest);
        out.resize(out.size() - recs.size());
        return st;
    }
    const size_t first = out.size() - recs.size();
    for (size_t k = 0; k < recs.size(); ++k) {
        WithdrawTransition& t = out[first + k];
        t.before_token_hash = B.P.value(B.tok[recs[k].ev_ti].root_before);
        t.token_balance_proof = B.proof(B.tok[recs[k].ev_ti]);
        t.fee_balance_proof = B.proof(B.tok[recs[k].ev_fi]);
        t.proof = B.proof(B.acc[recs[k].leaf]);
        t.state_after = B.P.value(B.acc[recs[k].leaf].root_after);
    }
    return BZK_OK;
}

static ZkScalar withdraw_aux(const std::vector<WithdrawTransition>& trs, int log4_batch) {
    std::vector<std::vector<ZkScalar>> items;
    for (auto& t : trs) {
        if (!t.enabled) break;
        items.push_back({ZkScalar::from_u64(1), t.tx.amount.token_id, ZkScalar::from_u64(t.tx.amount.amount), t.tx.fee.token_id,
                         ZkScalar::from_u64(t.tx.fee.amount), t.tx.fingerprint, t.tx.calldata()});
    }
    return native_batch_root(items, log4_batch, 7);
}

static void synthesize_withdraw(ConstraintSystem& cs, int L, int T, const ZkScalar& commitment, uint64_t height, const ZkScalar& state,
                                const ZkScalar& aux_data, const ZkScalar& next_state, const std::vector<WithdrawTransition>& trs,
                                int nthreads = 1, DeferData* dd = nullptr) {
    Num commitment_wit = num_alloc(cs, commitment.v);
    num_inputize(cs, commitment_wit);
    Num height_wit = num_alloc(cs, fr_from_u64(height));
    num_inputize(cs, height_wit);
    Num state_wit = num_alloc(cs, state.v);
    num_inputize(cs, state_wit);
    Num aux_wit = num_alloc(cs, aux_data.v);
    num_inputize(cs, aux_wit);
    Num claimed_next = num_alloc(cs, next_state.v);
    num_inputize(cs, claimed_next);
    struct TxWit { Bool enabled; Num amount_token_id; UInt amount; Num fee_token_id; UInt fee; Num fingerprint; APoint pub_key; Num nonce; APoint sig_r; Num sig_s; };
    std::vector<TxWit> wits;
    std::vector<std::vector<Number>> children;
    for (const WithdrawTransition& tr : trs) {
        Bit enabled = bit_alloc(cs, tr.enabled);
        Num amount_token_id = num_alloc(cs, tr.tx.amount.token_id.v);
        UInt amount = UInt::alloc_64(cs, tr.tx.amount.amount);
        Num fee_token_id = num_alloc(cs, tr.tx.fee.token_id.v);
        UInt fee = UInt::alloc_64(cs, tr.tx.fee.amount);
        Num fingerprint = num_alloc(cs, tr.enabled ? tr.tx.fingerprint.v : Fr::zero());
        APoint pub_key = APoint::alloc(cs, tr.tx.mpn_address);
        Num nonce = num_alloc(cs, fr_from_u64(tr.tx.nonce));
        APoint sig_r = APoint::alloc(cs, tr.tx.sig.r);
        Num sig_s = num_alloc(cs, tr.tx.sig.s.v);
        wits.push_back({Bool::is(enabled), amount_token_id, amount, fee_token_id, fee, fingerprint, pub_key, nonce, sig_r, sig_s});
        Number cdh = g_poseidon(cs, {Number::from(pub_key.x), Number::from(pub_key.y), Number::from(nonce), Number::from(sig_r.x),
                                     Number::from(sig_r.y), Number::from(sig_s)});
        Num calldata = mux(cs, Bool::is(enabled), Number::zero(), cdh);
        children.push_back({Number::from(enabled), Number::from(amount_token_id), amount.num, Number::from(fee_token_id), fee.num,
                            Number::from(fingerprint), Number::from(calldata)});
    }
    Number tx_root = g_reveal_batch(cs, children);
    cs.enforce(LC::of(aux_wit.var), aux_wit.val, LC::one(), Fr::one(), tx_root.lc, tx_root.val);
    auto body = [&](ConstraintSystem& cs, size_t i, const Num& state_wit) -> Num {
        const WithdrawTransition& tr = trs[i];
        const TxWit& tw = wits[i];
        UInt tx_index = UInt::alloc(cs, fr_from_u64(tr.account_index), 2 * L);
        UInt tx_token_index = UInt::alloc(cs, fr_from_u64(tr.token_index), 2 * T);
        UInt tx_fee_token_index = UInt::alloc(cs, fr_from_u64(tr.fee_token_index), 2 * T);
        tw.pub_key.assert_on_curve(cs, tw.enabled);
        Number tx_hash = g_poseidon(cs, {Number::from(tw.fingerprint), Number::from(tw.nonce)}, true);  // its value feeds the signature gadget
        tw.sig_r.assert_on_curve(cs, tw.enabled);
        g_verify_eddsa(cs, tw.enabled, tw.pub_key, tx_hash, tw.sig_r, tw.sig_s);
        Num src_tx_nonce = num_alloc(cs, fr_from_u64(tr.before.tx_nonce));
        Num src_withdraw_nonce = num_alloc(cs, fr_from_u64(tr.before.withdraw_nonce));
        APoint src_addr = APoint::alloc(cs, tr.before.address);
        src_addr.assert_on_curve(cs, tw.enabled);
        Num before_token_hash = num_alloc(cs, tr.before_token_hash.v);
        Num src_token_id = num_alloc(cs, tr.before_token_balance.token_id.v);
        Number::from(src_token_id).assert_equal(cs, Number::from(tw.amount_token_id));
        Num src_balance = num_alloc(cs, fr_from_u64(tr.before_token_balance.amount));
        Number src_token_balance_hash = g_poseidon(cs, {Number::from(src_token_id), Number::from(src_balance)});
        MerkleProofWit token_proof = alloc_proof(cs, tr.token_balance_proof);
        g_check_proof4(cs, tw.enabled, tx_token_index, src_token_balance_hash, token_proof, Number::from(before_token_hash));
        Number new_token_leaf = g_poseidon(cs, {Number::from(src_token_id), Number::from(src_balance).minus(tw.amount.num)});
        Number balance_middle_root = g_calc_root4(cs, tx_token_index, new_token_leaf, token_proof);
        Num src_fee_token_id = num_alloc(cs, tr.before_fee_balance.token_id.v);
        Number::from(src_fee_token_id).assert_equal(cs, Number::from(tw.fee_token_id));
        Num src_fee_balance = num_alloc(cs, fr_from_u64(tr.before_fee_balance.amount));
        Number src_fee_leaf = g_poseidon(cs, {Number::from(src_fee_token_id), Number::from(src_fee_balance)});
        MerkleProofWit fee_proof = alloc_proof(cs, tr.fee_balance_proof);
        g_check_proof4(cs, tw.enabled, tx_fee_token_index, src_fee_leaf, fee_proof, balance_middle_root);
        Number new_fee_leaf = g_poseidon(cs, {Number::from(src_fee_token_id), Number::from(src_fee_balance).minus(tw.fee.num)});
        Number src_hash = g_poseidon(cs, {Number::from(src_tx_nonce), Number::from(src_withdraw_nonce), Number::from(src_addr.x),
                                          Number::from(src_addr.y), Number::from(before_token_hash)});
        MerkleProofWit proof = alloc_proof(cs, tr.proof);
        g_check_proof4(cs, tw.enabled, tx_index, src_hash, proof, Number::from(state_wit));
        Number::from(tw.nonce).assert_equal_if_enabled(cs, tw.enabled, Number::from(src_withdraw_nonce).plus(Number::constant(Fr::one())));
        Number balance_final_root = g_calc_root4(cs, tx_fee_token_index, new_fee_leaf, fee_proof);
        Number new_hash = g_poseidon(cs, {Number::from(src_tx_nonce), Number::from(src_withdraw_nonce).plus(Number::constant(Fr::one())),
                                          Number::from(tw.pub_key.x), Number::from(tw.pub_key.y), balance_final_root});
        Number next_state_wit = g_calc_root4(cs, tx_index, new_hash, proof);
        return mux(cs, tw.enabled, Number::from(state_wit), next_state_wit);
    };
    state_wit = run_tx_bodies(cs, nthreads, trs.size(), body_shape(1, L, T, [&](ConstraintSystem& pcs) {
                                  LcModeGuard g(false);
                                  const Num st0 = {VAR_ONE, state.v};
                                  const size_t a0 = pcs.aux.size(), c0 = pcs.az.size();
                                  body(pcs, 0, st0);
                                  return std::make_pair(pcs.aux.size() - a0, pcs.az.size() - c0);
                              }),
                              state_wit, chain_states(state, trs), body, dd, ((uint64_t)1 << 32) | ((uint64_t)L << 16) | (uint64_t)T);
    cs.enforce(LC::of(state_wit.var), state_wit.val, LC::one(), Fr::one(), LC::of(claimed_next.var), claimed_next.val);
    cs.finalize();
}

static void finish_r1cs(bzk_r1cs* r) {
    ConstraintSystem& cs = r->cs;
    const size_t n_in = cs.inputs.size(), n_aux = cs.aux.size();
    r->z_bytes.resize((n_in + n_aux) * 32);
    for (size_t i = 0; i < n_in; ++i) memcpy(&r->z_bytes[32 * i], cs.inputs[i].l, 32);
    {
        // z = inputs | aux: Fr is 32 contiguous bytes, so the aux part is one block copy - split over a few threads once
        // it is large (1.85 GB for the 1024-tx circuit)
        uint8_t* dst = r->z_bytes.data() + 32 * n_in;
        const uint8_t* src = (const uint8_t*)cs.aux.data();
        const size_t bytes = n_aux * 32;
        const size_t nt = bytes < ((size_t)64 << 20) ? 1 : std::min<size_t>(8, (size_t)host_default_threads());
        if (nt <= 1) {
            if (bytes) memcpy(dst, src, bytes);
        } else {
            std::vector<std::thread> th;
            const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
            for (size_t t = 0; t < nt; ++t) {
                const size_t lo = t * per, hi = std::min(bytes, lo + per);
                if (lo < hi) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
            }
            for (auto& x : th) x.join();
        }
    }
    auto dens = [&](const std::vector<uint8_t>& din, const std::vector<uint8_t>& daux, std::vector<uint8_t>& out) {
        out.assign(n_in + n_aux, 0);
        for (size_t i = 0; i < din.size() && i < n_in; ++i) out[i] = din[i];
        for (size_t i = 0; i < daux.size() && i < n_aux; ++i) out[n_in + i] = daux[i];
    };
    dens(cs.a_in_d, cs.a_aux_d, r->a_density);
    dens(cs.b_in_d, cs.b_aux_d, r->b_density);
    if (cs.record_matrices) {
        auto flat = [&](const CsrBuilder& m, std::vector<uint32_t>& out) {
            out.resize(m.col.size());
            for (size_t k = 0; k < m.col.size(); ++k) out[k] = cs.flat_index(m.col[k]);
        };
        flat(cs.A, r->colA);
        flat(cs.B, r->colB);
        flat(cs.C, r->colC);
    }
}

// groth16.hip (bzk_groth16_prove_r1cs): the instance's arrays as an assignment, and what is deferred in them (nullptr: nothing, or already filled in)
void r1cs_assignment(const bzk_r1cs* r, bzk_assignment* a, const DeferData** dd, uint64_t* n_in) {
    const ConstraintSystem& cs = r->cs;
    if (n_in) *n_in = cs.inputs.size();
    a->z = r->z_bytes.data();
    a->az = (const uint8_t*)cs.az.data();
    a->bz = (const uint8_t*)cs.bz.data();
    a->cz = (const uint8_t*)cs.cz.data();
    a->n_rows = cs.az.size();
    a->n_vars = cs.inputs.size() + cs.aux.size();
    *dd = (r->defer && !r->defer->filled) ? r->defer.get() : nullptr;
}

}  // namespace bzk

extern "C" {

int32_t bzk_mpn_create(uint32_t log4_tree, uint32_t log4_token_tree, bzk_mpn** out) {
    if (!out || log4_tree == 0 || log4_tree > 30 || log4_token_tree == 0 || log4_token_tree > 8) return BZK_E_ARG;
    *out = new (std::nothrow) bzk_mpn((int)log4_tree, (int)log4_token_tree);
    return *out ? BZK_OK : BZK_E_ALLOC;
}

void bzk_mpn_destroy(bzk_mpn* w) { delete w; }

int32_t bzk_mpn_set_height(bzk_mpn* w, uint64_t height) {
    if (!w) return BZK_E_ARG;
    w->height = height;
    return BZK_OK;
}

// ctx non-NULL: the witness builders of this world (bzk_mpn_{update,deposit,withdraw}_synthesize, bzk_mpn_make_work) batch their
// Merkle re-hashing on that context (DevBatch above) instead of hashing transaction by transaction on the host; NULL: host path.
// The caller keeps the context alive and does not use it concurrently.  Same transitions and works, byte for byte.
int32_t bzk_mpn_set_device(bzk_mpn* w, bzk_ctx* ctx) {
    if (!w) return BZK_E_ARG;
    w->dev = ctx;
    return BZK_OK;
}

int32_t bzk_host_default_threads(void) { return host_default_threads(); }
int32_t bzk_mpn_set_threads(bzk_mpn* w, int32_t n) {
    if (!w || n < 1) return BZK_E_ARG;
    w->threads = n;
    return BZK_OK;
}

int32_t bzk_mpn_add_account(bzk_mpn* w, uint64_t index, const uint8_t* seed, uint32_t seed_len, const uint8_t token_id[32],
                            uint64_t balance, uint8_t pub_xy_out[64]) {
    if (!w || !seed || !token_id || index >= ((uint64_t)1 << (2 * w->L))) return BZK_E_ARG;
    JubjubPrivateKey k = jubjub_generate_keys(seed, seed_len);
    MpnAccount a;
    a.address = k.public_key;
    a.tokens[0] = Money{ZkScalar::from_bytes(token_id), balance};
    w->keys[index] = k;
    w->set(index, a);
    if (pub_xy_out) {
        k.public_key.x.to_bytes(pub_xy_out);
        k.public_key.y.to_bytes(pub_xy_out + 32);
    }
    return BZK_OK;
}

int32_t bzk_mpn_root(bzk_mpn* w, uint8_t root[32]) {
    if (!w || !root) return BZK_E_ARG;
    w->accounts->root().to_bytes(root);
    return BZK_OK;
}

// create_mpn_transaction semantics (src/wallet/tx_builder.rs:287-306): nonce = sender nonce + 1 (+ txs
// already queued from that sender), signed with the sender's key.  dst may be a not-yet-existing
// account index whose key seed was registered with bzk_mpn_add_key.
int32_t bzk_mpn_push_tx(bzk_mpn* w, uint64_t src_index, uint64_t dst_index, const uint8_t token_id[32], uint64_t amount,
                        const uint8_t fee_token[32], uint64_t fee) {
    if (!w || !token_id || !fee_token) return BZK_E_ARG;
    if (!w->keys.count(src_index) || !w->keys.count(dst_index)) return BZK_E_ARG;
    MpnTx tx;
    tx.src_pub = w->keys[src_index].public_key;
    tx.dst_pub = w->keys[dst_index].public_key;
    uint32_t queued = 0;
    for (auto& q : w->mempool)
        if (q.src_pub == tx.src_pub) ++queued;
    tx.nonce = w->get(src_index).tx_nonce + 1 + queued;
    tx.amount = Money{ZkScalar::from_bytes(token_id), amount};
    tx.fee = Money{ZkScalar::from_bytes(fee_token), fee};
    tx.sig = jubjub_sign(w->keys[src_index], tx.hash());
    w->mempool.push_back(tx);
    return BZK_OK;
}

int32_t bzk_mpn_add_key(bzk_mpn* w, uint64_t index, const uint8_t* seed, uint32_t seed_len) {
    if (!w || !seed) return BZK_E_ARG;
    w->keys[index] = jubjub_generate_keys(seed, seed_len);
    return BZK_OK;
}

void bzk_r1cs_free(bzk_r1cs* r) { delete r; }

int32_t bzk_mpn_update_synthesize(bzk_mpn* w, uint32_t log4_batch, const uint8_t commitment[32], const uint8_t fee_token[32],
                                  int32_t record_matrices, bzk_r1cs** out) {
    if (!w || !commitment || !fee_token || !out || log4_batch > 6) return BZK_E_ARG;
    *out = nullptr;
    try {
        const bool dbg = getenv("BZK_DEBUG") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t0 = now();
        const ZkScalar ft = ZkScalar::from_bytes(fee_token);
        const ZkScalar state = w->accounts->root();
        std::vector<UpdateTransition> trs;
        uint64_t fee_sum = 0, rejected = 0;
        if (w->dev) { const int32_t ds = build_transitions_dev(*w, (int)log4_batch, ft, trs, fee_sum, rejected); if (ds != BZK_OK) return ds; }
        else build_transitions(*w, (int)log4_batch, ft, trs, fee_sum, rejected);
        double t1 = now();
        const uint64_t accepted = trs.size();
        while (trs.size() < ((size_t)1 << (2 * log4_batch))) trs.push_back(UpdateTransition::null(w->L, w->T));
        ZkScalar auxin[2] = {ft, ZkScalar::from_u64(fee_sum)};
        const ZkScalar aux = poseidon_hash(auxin, 2);
        const ZkScalar next_state = w->accounts->root();
        std::unique_ptr<bzk_r1cs> r(new bzk_r1cs(record_matrices != 0));
        // KeypairAssembly role (matrices + densities, LCs re-evaluated against the supplied values as a self-check)
        // or pure ProvingAssignment role (values only: no linear-combination bookkeeping at all)
        LcModeGuard guard(record_matrices != 0);
        r->cs.self_check = record_matrices != 0;
        if (w->defer && !record_matrices) r->defer.reset(new DeferData());
        synthesize_update(r->cs, w->L, w->T, ZkScalar::from_bytes(commitment), w->height, state, aux, next_state, ft, trs, w->threads, r->defer.get());
        if (r->defer && !r->defer->prog) r->defer.reset();
        if (r->cs.check_failed_at >= 0) return BZK_E_INTERNAL;  // a gadget supplied a value that is not <LC, z>
        double t2 = now();
        r->accepted = accepted;
        r->rejected = rejected;
        finish_r1cs(r.get());
        if (dbg) fprintf(stderr, "[bzk] update_synthesize: transitions %.3f s, circuit %.3f s, finish %.3f s\n", t1 - t0, t2 - t1, now() - t2);
        *out = r.release();
        return BZK_OK;
    } catch (const std::bad_alloc&) {
        return BZK_E_ALLOC;
    } catch (const std::exception&) {
        return BZK_E_INTERNAL;
    }
}

// The all-disabled instance `MpnCircuit::empty(L, T, B)` with explicit public inputs - the circuit the
// reference uses for setup (src/config/blockchain.rs:373-399) and in its own test
// (src/mpn/circuits/test.rs:117-149).
int32_t bzk_mpn_update_empty(uint32_t log4_tree, uint32_t log4_token_tree, uint32_t log4_batch, const uint8_t commitment[32],
                             uint64_t height, const uint8_t state[32], const uint8_t aux_data[32], const uint8_t next_state[32],
                             const uint8_t fee_token[32], int32_t record_matrices, bzk_r1cs** out) {
    if (!commitment || !state || !aux_data || !next_state || !fee_token || !out || log4_batch > 6 || log4_tree == 0 ||
        log4_tree > 30 || log4_token_tree == 0 || log4_token_tree > 8)
        return BZK_E_ARG;
    *out = nullptr;
    try {
        std::vector<UpdateTransition> trs((size_t)1 << (2 * log4_batch), UpdateTransition::null((int)log4_tree, (int)log4_token_tree));
        std::unique_ptr<bzk_r1cs> r(new bzk_r1cs(record_matrices != 0));
        LcModeGuard guard(record_matrices != 0);
        r->cs.self_check = record_matrices != 0;
        synthesize_update(r->cs, (int)log4_tree, (int)log4_token_tree, ZkScalar::from_bytes(commitment), height,
                          ZkScalar::from_bytes(state), ZkScalar::from_bytes(aux_data), ZkScalar::from_bytes(next_state),
                          ZkScalar::from_bytes(fee_token), trs, 1);
        finish_r1cs(r.get());
        *out = r.release();
        return BZK_OK;
    } catch (const std::bad_alloc&) {
        return BZK_E_ALLOC;
    } catch (const std::exception&) {
        return BZK_E_INTERNAL;
    }
}

// ---- deposits and withdrawals: the other two MpnWorkData variants (src/mpn/mod.rs:243-248)
int32_t bzk_mpn_push_deposit(bzk_mpn* w, uint64_t key_index, const uint8_t token_id[32], uint64_t amount) {
    if (!w || !token_id || !w->keys.count(key_index)) return BZK_E_ARG;
    DepositTx tx;
    tx.mpn_address = w->keys[key_index].public_key;
    tx.amount = Money{ZkScalar::from_bytes(token_id), amount};
    w->deposit_queue.push_back(tx);
    return BZK_OK;
}

// signed as the wallet does (src/wallet/tx_builder.rs:376-425): sig over H2(fingerprint, nonce), nonce = account's
// withdraw nonce + 1 (+ already queued withdrawals of that account)
int32_t bzk_mpn_push_withdraw(bzk_mpn* w, uint64_t account_index, const uint8_t token_id[32], uint64_t amount,
                              const uint8_t fee_token[32], uint64_t fee, const uint8_t fingerprint[32]) {
    if (!w || !token_id || !fee_token || !w->keys.count(account_index)) return BZK_E_ARG;
    WithdrawTx tx;
    tx.mpn_address = w->keys[account_index].public_key;
    uint32_t queued = 0;
    for (auto& q : w->withdraw_queue)
        if (q.mpn_address == tx.mpn_address) ++queued;
    tx.nonce = w->get(account_index).withdraw_nonce + 1 + queued;
    tx.amount = Money{ZkScalar::from_bytes(token_id), amount};
    tx.fee = Money{ZkScalar::from_bytes(fee_token), fee};
    if (fingerprint) {
        tx.fingerprint = ZkScalar::from_bytes(fingerprint);  // opaque: such a withdrawal cannot be put on the wire
        tx.sig = jubjub_sign(w->keys[account_index], tx.sign_message());
    } else {
        // as the wallet does (src/wallet/tx_builder.rs:376-425): fingerprint of the payment with zero calldata, signature
        // over H2(fingerprint, nonce), then calldata = H6(address, nonce, signature) written into the payment
        tx.payment = default_contract_withdraw(w->contract_id, tx.amount, tx.fee, ZkScalar());
        tx.fingerprint = contract_withdraw_fingerprint(tx.payment);
        tx.sig = jubjub_sign(w->keys[account_index], tx.sign_message());
        tx.payment = default_contract_withdraw(w->contract_id, tx.amount, tx.fee, tx.calldata());
    }
    w->withdraw_queue.push_back(tx);
    return BZK_OK;
}

int32_t bzk_mpn_deposit_synthesize(bzk_mpn* w, uint32_t log4_batch, const uint8_t commitment[32], int32_t record_matrices,
                                   bzk_r1cs** out) {
    if (!w || !commitment || !out || log4_batch > 6) return BZK_E_ARG;
    *out = nullptr;
    try {
        const ZkScalar state = w->accounts->root();
        std::vector<DepositTransition> trs;
        uint64_t rejected = 0;
        if (w->dev) { const int32_t ds = build_deposits_dev(*w, (int)log4_batch, trs, rejected); if (ds != BZK_OK) return ds; }
        else build_deposits(*w, (int)log4_batch, trs, rejected);
        const uint64_t accepted = trs.size();
        const ZkScalar aux = deposit_aux(trs, (int)log4_batch);
        while (trs.size() < ((size_t)1 << (2 * log4_batch))) trs.push_back(DepositTransition::null(w->L, w->T));
        std::unique_ptr<bzk_r1cs> r(new bzk_r1cs(record_matrices != 0));
        LcModeGuard guard(record_matrices != 0);
        r->cs.self_check = record_matrices != 0;
        if (w->defer && !record_matrices) r->defer.reset(new DeferData());
        synthesize_deposit(r->cs, w->L, w->T, ZkScalar::from_bytes(commitment), w->height, state, aux, w->accounts->root(), trs, w->threads, r->defer.get());
        if (r->defer && !r->defer->prog) r->defer.reset();
        if (r->cs.check_failed_at >= 0) return BZK_E_INTERNAL;
        r->accepted = accepted;
        r->rejected = rejected;
        finish_r1cs(r.get());
        *out = r.release();
        return BZK_OK;
    } catch (const std::bad_alloc&) {
        return BZK_E_ALLOC;
    } catch (const std::exception&) {
        return BZK_E_INTERNAL;
    }
}

int32_t bzk_mpn_withdraw_synthesize(bzk_mpn* w, uint32_t log4_batch, const uint8_t commitment[32], int32_t record_matrices,
                                    bzk_r1cs** out) {
    if (!w || !commitment || !out || log4_batch > 6) return BZK_E_ARG;
    *out = nullptr;
    try {
        const ZkScalar state = w->accounts->root();
        std::vector<WithdrawTransition> trs;
        uint64_t rejected = 0;
        if (w->dev) { const int32_t ds = build_withdraws_dev(*w, (int)log4_batch, trs, rejected); if (ds != BZK_OK) return ds; }
        else build_withdraws(*w, (int)log4_batch, trs, rejected);
        const uint64_t accepted = trs.size();
        const ZkScalar aux = withdraw_aux(trs, (int)log4_batch);
        while (trs.size() < ((size_t)1 << (2 * log4_batch))) trs.push_back(WithdrawTransition::null(w->L, w->T));
        std::unique_ptr<bzk_r1cs> r(new bzk_r1cs(record_matrices != 0));
        LcModeGuard guard(record_matrices != 0);
        r->cs.self_check = record_matrices != 0;
        if (w->defer && !record_matrices) r->defer.reset(new DeferData());
        synthesize_withdraw(r->cs, w->L, w->T, ZkScalar::from_bytes(commitment), w->height, state, aux, w->accounts->root(), trs, w->threads, r->defer.get());
        if (r->defer && !r->defer->prog) r->defer.reset();
        if (r->cs.check_failed_at >= 0) return BZK_E_INTERNAL;
        r->accepted = accepted;
        r->rejected = rejected;
        finish_r1cs(r.get());
        *out = r.release();
        return BZK_OK;
    } catch (const std::bad_alloc&) {
        return BZK_E_ALLOC;
    } catch (const std::exception&) {
        return BZK_E_INTERNAL;
    }
}

// `MpnCircuit::empty` for the other two circuits (src/mpn/circuits/test.rs:151-229): kind 0 = deposit, 1 = withdraw
int32_t bzk_mpn_circuit_empty(int32_t kind, uint32_t log4_tree, uint32_t log4_token_tree, uint32_t log4_batch,
                              const uint8_t commitment[32], uint64_t height, const uint8_t state[32], const uint8_t aux_data[32],
                              const uint8_t next_state[32], int32_t record_matrices, bzk_r1cs** out) {
    if (!commitment || !state || !aux_data || !next_state || !out || log4_batch > 6 || log4_tree == 0 || log4_tree > 30 ||
        log4_token_tree == 0 || log4_token_tree > 8 || kind < 0 || kind > 1)
        return BZK_E_ARG;
    *out = nullptr;
    try {
        std::unique_ptr<bzk_r1cs> r(new bzk_r1cs(record_matrices != 0));
        LcModeGuard guard(record_matrices != 0);
        r->cs.self_check = record_matrices != 0;
        const size_t n = (size_t)1 << (2 * log4_batch);
        if (kind == 0) {
            std::vector<DepositTransition> trs(n, DepositTransition::null((int)log4_tree, (int)log4_token_tree));
            synthesize_deposit(r->cs, (int)log4_tree, (int)log4_token_tree, ZkScalar::from_bytes(commitment), height,
                               ZkScalar::from_bytes(state), ZkScalar::from_bytes(aux_data), ZkScalar::from_bytes(next_state), trs);
        } else {
            std::vector<WithdrawTransition> trs(n, WithdrawTransition::null((int)log4_tree, (int)log4_token_tree));
            synthesize_withdraw(r->cs, (int)log4_tree, (int)log4_token_tree, ZkScalar::from_bytes(commitment), height,
                                ZkScalar::from_bytes(state), ZkScalar::from_bytes(aux_data), ZkScalar::from_bytes(next_state), trs);
        }
        if (r->cs.check_failed_at >= 0) return BZK_E_INTERNAL;
        finish_r1cs(r.get());
        *out = r.release();
        return BZK_OK;
    } catch (const std::bad_alloc&) {
        return BZK_E_ALLOC;
    } catch (const std::exception&) {
        return BZK_E_INTERNAL;
    }
}

// info[0..8) = n_in, n_aux, n_constraints, nnz(A), nnz(B), nnz(C), first unsatisfied constraint + 1 (0 = ok),
//              accepted txs, rejected txs  (9 entries)
int32_t bzk_r1cs_info(const bzk_r1cs* r, uint64_t info[9]) {
    if (!r || !info) return BZK_E_ARG;
    const ConstraintSystem& cs = r->cs;
    info[0] = cs.inputs.size();
    info[1] = cs.aux.size();
    info[2] = cs.num_constraints();
    info[3] = cs.A.col.size();
    info[4] = cs.B.col.size();
    info[5] = cs.C.col.size();
    {   // a_k * b_k == c_k for every row: independent rows, checked by a few threads (0.9 M products for a 16-tx batch)
        const size_t nrows = cs.num_constraints();
        const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(8, nrows / 65536));
        std::vector<long> first((size_t)nt, -1);
        // rows a DeferProgram fills on the device are not the host's to judge (they hold stale bytes until then): the scan covers the gaps
        // between them; the device reports its own rows through wf::FLAG_UNSATISFIED
        std::vector<std::pair<size_t, size_t>> ranges;
        if (r->defer && !r->defer->filled) {
            const DeferData& dd = *r->defer;
            size_t pos = 0;
            for (size_t t = 0; t < dd.n_tx; ++t) {
                const size_t base = dd.base_con + t * dd.stride_con;
                for (auto& h : dd.prog->con_holes) {
                    if (base + h.first > pos) ranges.push_back({pos, base + h.first});
                    pos = base + h.first + h.second;
                }
            }
            if (pos < nrows) ranges.push_back({pos, nrows});
        } else {
            ranges.push_back({0, nrows});
        }
        auto scan = [&](unsigned t) {
            const size_t lo = nrows * t / nt, hi = nrows * (t + 1) / nt;
            long bad = -1;
            for (auto& rg : ranges) {
                const size_t a = std::max(lo, rg.first), b = std::min(hi, rg.second);
                if (a >= b) continue;
                // eight rows per step on AVX-512 IFMA where the CPU has it, the 64-bit-limb product otherwise (host_fr_ifma.h)
                bad = hfr::products_first_mismatch(cs.az.data(), cs.bz.data(), cs.cz.data(), a, b);
                if (bad >= 0) break;
            }
            first[t] = bad;
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(scan, t);
        scan(0);
        for (auto& x : th) x.join();
        long bad = -1;
        for (unsigned t = 0; t < nt && bad < 0; ++t) bad = first[t];
        info[6] = (uint64_t)(bad + 1);
    }
    info[7] = r->accepted;
    info[8] = r->rejected;
    return BZK_OK;
}

// which: 0 z, 1 az, 2 bz, 3 cz, 4 a_density, 5 b_density, 6/7/8 val of A/B/C, 9/10/11 col of A/B/C (u32, flat variable
// index), 12/13/14 row_ptr of A/B/C (u32).  Returns a borrowed pointer valid until bzk_r1cs_free; *bytes = length.
const void* bzk_r1cs_data(const bzk_r1cs* r, int32_t which, uint64_t* bytes) {
    if (!r || !bytes) return nullptr;
    const ConstraintSystem& cs = r->cs;
    auto ret = [&](const void* p, size_t n) { *bytes = n; return p; };
    switch (which) {
        case 0: return ret(r->z_bytes.data(), r->z_bytes.size());
        case 1: return ret(cs.az.data(), cs.az.size() * 32);
        case 2: return ret(cs.bz.data(), cs.bz.size() * 32);
        case 3: return ret(cs.cz.data(), cs.cz.size() * 32);
        case 4: return ret(r->a_density.data(), r->a_density.size());
        case 5: return ret(r->b_density.data(), r->b_density.size());
        case 6: return ret(cs.A.val.data(), cs.A.val.size() * 32);
        case 7: return ret(cs.B.val.data(), cs.B.val.size() * 32);
        case 8: return ret(cs.C.val.data(), cs.C.val.size() * 32);
        case 9: return ret(r->colA.data(), r->colA.size() * 4);
        case 10: return ret(r->colB.data(), r->colB.size() * 4);
        case 11: return ret(r->colC.data(), r->colC.size() * 4);
        case 12: return ret(cs.A.row_ptr.data(), cs.A.row_ptr.size() * 4);
        case 13: return ret(cs.B.row_ptr.data(), cs.B.row_ptr.size() * 4);
        case 14: return ret(cs.C.row_ptr.data(), cs.C.row_ptr.size() * 4);
        default: *bytes = 0; return nullptr;
    }
}

// ---- deferred witness values (host_r1cs.h DeferProgram, bzk_witfill.cuh)
int32_t bzk_mpn_set_defer(bzk_mpn* w, int32_t on) {
    if (!w) return BZK_E_ARG;
    w->defer = on != 0;
    return BZK_OK;
}
// info: 0 deferred (1 / 0), 1 transitions, 2 ops, 3 registers, 4 inputs per transition, 5 levels of pass 1, 6 / 7 variable / constraint
// slots per transition left to the device, 8 filled on the host (1 / 0), 9 wf::FLAG_* of that fill
int32_t bzk_r1cs_defer_info(const bzk_r1cs* r, uint64_t info[10]) {
    if (!r || !info) return BZK_E_ARG;
    for (int i = 0; i < 10; ++i) info[i] = 0;
    if (!r->defer) return BZK_OK;
    const DeferData& dd = *r->defer;
    info[0] = 1; info[1] = dd.n_tx; info[2] = dd.prog->ops.size(); info[3] = dd.prog->n_regs; info[4] = dd.prog->n_inputs;
    info[5] = dd.prog->n_levels; info[6] = dd.prog->hole_aux; info[7] = dd.prog->hole_con; info[8] = dd.filled ? 1 : 0; info[9] = dd.flags;
    return BZK_OK;
}
// the schedule the one-launch device kernel would run for this instance's program, checked on the host: info = {stages, segments, hash ops covered,
// fill ops covered, largest segment, violations (must be 0)}
int32_t bzk_r1cs_defer_schedule_info(const bzk_r1cs* r, uint64_t info[6]) {
    if (!r || !info) return BZK_E_ARG;
    for (int i = 0; i < 6; ++i) info[i] = 0;
    if (!r->defer) return BZK_OK;
    witfill_schedule_info(*r->defer->prog, info);
    return BZK_OK;
}
// completes the host arrays with the same ops the device runs (CPU consumers of bzk_r1cs_data; the CPU suite)
int32_t bzk_r1cs_fill_host(bzk_r1cs* r) {
    if (!r) return BZK_E_ARG;
    if (!r->defer || r->defer->filled) return BZK_OK;
    try {
        ConstraintSystem& cs = r->cs;
        wf::Arrays A{(Fr*)(r->z_bytes.data() + 32 * cs.inputs.size()), cs.az.data(), cs.bz.data(), cs.cz.data()};
        r->defer->flags = witfill_run_host(*r->defer, A);
        memcpy((void*)cs.aux.data(), A.z_aux, cs.aux.size() * 32);  // the generator's own copy of the aux part
        r->defer->filled = true;
        return BZK_OK;
    } catch (const std::exception&) {
        return BZK_E_INTERNAL;
    }
}

// host-side mirrors of the reference's ZkHasher / jubjub entry points (CPU; for wallets and tests)
// ------------------------------------------------------------------------------------------------
// f-2: MpnWork on the wire (host_bincode.h).  Worker side: decode -> synthesize (-> bzk_groth16_prove) -> ZkProof bytes;
// validator side (`prepare_works`, src/mpn/mod.rs:298-424): the queued transactions of a world -> MpnWork bytes.
// ------------------------------------------------------------------------------------------------
}  // extern "C"

struct bzk_mpn_work {
    MpnWork w;
};

namespace {
thread_local std::string g_work_error;

// root of the 4-ary tree after the leaf at `index` is replaced (calc_root of the merkle gadget, natively)
ZkScalar root_from_path(const Proof4& proof, uint64_t index, ZkScalar node) {
    for (const auto& sib : proof) {
        ZkScalar c[4];
        const int pos = (int)(index & 3);
        for (int j = 0, k = 0; j < 4; ++j) c[j] = j == pos ? node : sib[k++];
        node = poseidon_hash(c, 4);
        index >>= 2;
    }
    return node;
}
ZkScalar h2(const ZkScalar& a, const ZkScalar& b) {
    ZkScalar v[2] = {a, b};
    return poseidon_hash(v, 2);
}
ZkScalar h5(const ZkScalar& a, const ZkScalar& b, const ZkScalar& c, const ZkScalar& d, const ZkScalar& e) {
    ZkScalar v[5] = {a, b, c, d, e};
    return poseidon_hash(v, 5);
}
// The account-tree root each enabled transition leaves behind, computed as the circuits compute `next_state_wit`
// (update_circuit.rs:420-436, deposit_circuit.rs:262-283, withdraw_circuit.rs:380-401): what the witness builder's
// worker threads use as the state entering the next transition.  A transition whose witness is inconsistent merely
// mispredicts; the generator then falls back to the sequential walk (run_tx_bodies).
void fill_state_after(MpnWork& w) {
    for (UpdateTransition& t : w.updates) {
        const ZkScalar leaf = h2(t.tx.amount.token_id, ZkScalar::from_u64(t.dst_before_balance.amount) + ZkScalar::from_u64(t.tx.amount.amount));
        const ZkScalar bal = root_from_path(t.dst_balance_proof, t.dst_token_index, leaf);
        const ZkScalar acc = h5(ZkScalar::from_u64(t.dst_before.tx_nonce), ZkScalar::from_u64(t.dst_before.withdraw_nonce), t.tx.dst_pub.x,
                                t.tx.dst_pub.y, bal);
        t.state_after = root_from_path(t.dst_proof, t.dst_index, acc);
    }
    for (DepositTransition& t : w.deposits) {
        const ZkScalar leaf = h2(t.tx.amount.token_id, ZkScalar::from_u64(t.before_balance.amount) + ZkScalar::from_u64(t.tx.amount.amount));
        const ZkScalar bal = root_from_path(t.balance_proof, t.token_index, leaf);
        const ZkScalar acc = h5(ZkScalar::from_u64(t.before.tx_nonce), ZkScalar::from_u64(t.before.withdraw_nonce), t.tx.mpn_address.x,
                                t.tx.mpn_address.y, bal);
        t.state_after = root_from_path(t.proof, t.account_index, acc);
    }
    for (WithdrawTransition& t : w.withdraws) {
        const ZkScalar leaf = h2(t.before_fee_balance.token_id, ZkScalar::from_u64(t.before_fee_balance.amount) - ZkScalar::from_u64(t.tx.fee.amount));
        const ZkScalar bal = root_from_path(t.fee_balance_proof, t.fee_token_index, leaf);
        const ZkScalar acc = h5(ZkScalar::from_u64(t.before.tx_nonce), ZkScalar::from_u64(t.before.withdraw_nonce) + ZkScalar::one(),
                                t.tx.mpn_address.x, t.tx.mpn_address.y, bal);
        t.state_after = root_from_path(t.proof, t.account_index, acc);
    }
}
// a work whose shape the circuits cannot take is refused at decode time (the reference would panic or build a
// circuit of a different size than its parameters)
bool work_shape_ok(const MpnWork& w, std::string& why) {
    const MpnWorkConfig& c = w.config;
    if (c.log4_tree == 0 || c.log4_tree > 30 || c.log4_token_tree == 0 || c.log4_token_tree > 8) { why = "tree sizes out of range"; return false; }
    if (w.log4_batch() > 6) { why = "batch size out of range"; return false; }
    if (w.n_transitions() > ((size_t)1 << (2 * w.log4_batch()))) { why = "more transitions than the batch holds"; return false; }
    const size_t L = c.log4_tree, T = c.log4_token_tree;
    for (auto& t : w.updates)
        if (t.src_proof.size() != L || t.dst_proof.size() != L || t.src_balance_proof.size() != T || t.src_fee_balance_proof.size() != T ||
            t.dst_balance_proof.size() != T) { why = "update transition: proof depth"; return false; }
    for (auto& t : w.deposits)
        if (t.proof.size() != L || t.balance_proof.size() != T) { why = "deposit transition: proof depth"; return false; }
    for (auto& t : w.withdraws)
        if (t.proof.size() != L || t.token_balance_proof.size() != T || t.fee_balance_proof.size() != T) { why = "withdraw transition: proof depth"; return false; }
    if (!verifier_key_well_formed(c.deposit_vk) || !verifier_key_well_formed(c.withdraw_vk) || !verifier_key_well_formed(c.update_vk)) {
        why = "verifying key";
        return false;
    }
    return true;
}
}  // namespace

extern "C" {

const char* bzk_mpn_work_last_error(void) { return g_work_error.c_str(); }

int32_t bzk_mpn_work_decode(const uint8_t* bytes, uint64_t len, uint32_t flags, bzk_mpn_work** out, uint64_t* consumed) {
    if (!bytes || !out) return BZK_E_ARG;
    *out = nullptr;
    try {
        std::unique_ptr<bzk_mpn_work> h(new bzk_mpn_work());
        BinReader r(bytes, (size_t)len);
        if (!mpn_work_decode(r, flags, h->w)) {
            g_work_error = r.err;
            return BZK_E_ARG;
        }
        std::string why;
        if (!work_shape_ok(h->w, why)) {
            g_work_error = why;
            return BZK_E_ARG;
        }
        fill_state_after(h->w);
        if (consumed) *consumed = r.pos;
        *out = h.release();
        return BZK_OK;
    } catch (const std::bad_alloc&) {
        return BZK_E_ALLOC;
    } catch (const std::exception& e) {
        g_work_error = e.what();
        return BZK_E_INTERNAL;
    }
}

void bzk_mpn_work_free(bzk_mpn_work* h) { delete h; }

// info[0..12) = kind (0 deposit, 1 withdraw, 2 update), log4_tree, log4_token_tree, log4 batch size of this kind, transitions
//               on the wire (<= 4^batch; the rest is padded with null transitions), height, reward, new_root.state_size,
//               mpn_num_update_batches, mpn_num_deposit_batches, mpn_num_withdraw_batches, byte length of this work's verifying key
int32_t bzk_mpn_work_info(const bzk_mpn_work* h, uint64_t info[12]) {
    if (!h || !info) return BZK_E_ARG;
    const MpnWork& w = h->w;
    info[0] = (uint64_t)w.kind;
    info[1] = w.config.log4_tree;
    info[2] = w.config.log4_token_tree;
    info[3] = (uint64_t)w.log4_batch();
    info[4] = w.n_transitions();
    info[5] = w.height;
    info[6] = w.reward;
    info[7] = w.new_root_size;
    info[8] = w.config.num_update_batches;
    info[9] = w.config.num_deposit_batches;
    info[10] = w.config.num_withdraw_batches;
    info[11] = w.vk().size();
    return BZK_OK;
}

// out = state | aux_data | next_state | new_root.state_hash | mpn_contract_id (as the circuits' scalar), 32 B each
int32_t bzk_mpn_work_scalars(const bzk_mpn_work* h, uint8_t out[160]) {
    if (!h || !out) return BZK_E_ARG;
    h->w.state.to_bytes(out);
    h->w.aux_data.to_bytes(out + 32);
    h->w.next_state.to_bytes(out + 64);
    h->w.new_root_hash.to_bytes(out + 96);
    h->w.config.mpn_contract_id.to_bytes(out + 128);
    return BZK_OK;
}

// which: -1 = the key this work is verified with (MpnWork::vk), 0 / 1 / 2 = deposit / withdraw / update key
int32_t bzk_mpn_work_vk(const bzk_mpn_work* h, int32_t which, uint8_t* out, uint64_t cap, uint64_t* len) {
    if (!h || which < -1 || which > 2) return BZK_E_ARG;
    const MpnWorkConfig& c = h->w.config;
    const std::vector<uint8_t>& vk = which < 0 ? h->w.vk() : which == 0 ? c.deposit_vk : which == 1 ? c.withdraw_vk : c.update_vk;
    if (len) *len = vk.size();
    if (out) {
        if (cap < vk.size()) return BZK_E_ARG;
        memcpy(out, vk.data(), vk.size());
    }
    return BZK_OK;
}

int32_t bzk_mpn_work_commitment(const bzk_mpn_work* h, const uint8_t prover_pub[32], uint8_t out[32]) {
    if (!h || !prover_pub || !out) return BZK_E_ARG;
    mpn_work_commitment(prover_pub, h->w.reward).to_bytes(out);
    return BZK_OK;
}

// `MpnWork::verify(prover, proof)` (src/mpn/mod.rs:281-295): groth16_verify under the work's key with the public inputs
// [H(prover, reward), height, state, aux_data, next_state].  1 = accepted, 0 = refused, negative = bad arguments.  Host code.
int32_t bzk_mpn_work_verify(const bzk_mpn_work* h, const uint8_t prover_pub[32], const uint8_t proof[387]) {
    if (!h || !prover_pub || !proof) return BZK_E_ARG;
    uint8_t in[160];
    mpn_work_commitment(prover_pub, h->w.reward).to_bytes(in);
    ZkScalar::from_u64(h->w.height).to_bytes(in + 32);
    h->w.state.to_bytes(in + 64);
    h->w.aux_data.to_bytes(in + 96);
    h->w.next_state.to_bytes(in + 128);
    const std::vector<uint8_t>& vk = h->w.vk();
    return bzk_groth16_verify(vk.data(), vk.size(), in, 5, proof);
}

int32_t bzk_mpn_work_encode(const bzk_mpn_work* h, uint8_t* out, uint64_t cap, uint64_t* len) {
    if (!h) return BZK_E_ARG;
    try {
        for (auto& t : h->w.withdraws)
            if (t.tx.payment.empty()) return BZK_E_ARG;  // opaque-fingerprint withdrawals have no wire form
        BinWriter o;
        mpn_work_encode(o, h->w);
        if (len) *len = o.b.size();
        if (out) {
            if (cap < o.b.size()) return BZK_E_ARG;
            memcpy(out, o.b.data(), o.b.size());
        }
        return BZK_OK;
    } catch (const std::bad_alloc&) {
        return BZK_E_ALLOC;
    }
}

// The circuit instance a worker proves for this work: commitment = H(prover, reward), public inputs from the work,
// transitions padded with `::null` to 4^batch (what the reference's provers do before `create_random_proof`).
// fee_token: the update circuit's private `fee_token` (NULL = Ziesha, what prepare_works uses - src/mpn/mod.rs:400).
int32_t bzk_mpn_work_synthesize(const bzk_mpn_work* h, const uint8_t prover_pub[32], const uint8_t fee_token[32], int32_t threads,
                                int32_t record_matrices, bzk_r1cs** out) {
    if (!h || !prover_pub || !out) return BZK_E_ARG;
    *out = nullptr;
    // 0 witness only, 1 with the CSR matrices, BZK_SYNTH_DEFER witness only with deferred values: anything else is a caller's mistake, not "non-zero = matrices"
    if (record_matrices != 0 && record_matrices != 1 && record_matrices != (int32_t)BZK_SYNTH_DEFER) return BZK_E_ARG;
    try {
        const MpnWork& w = h->w;
        const int L = w.config.log4_tree, T = w.config.log4_token_tree, B = w.log4_batch();
        const size_t cap = (size_t)1 << (2 * B);
        const int nt = threads > 0 ? threads : host_default_threads();
        const ZkScalar commitment = mpn_work_commitment(prover_pub, w.reward);
        const bool rec = record_matrices == 1, defer = record_matrices == BZK_SYNTH_DEFER;  // 0: witness only
        std::unique_ptr<bzk_r1cs> r(new bzk_r1cs(rec));
        LcModeGuard guard(rec);
        r->cs.self_check = rec;
        if (w.kind == 2) {
            std::vector<UpdateTransition> trs = w.updates;
            while (trs.size() < cap) trs.push_back(UpdateTransition::null(L, T));
            const ZkScalar ft = fee_token ? ZkScalar::from_bytes(fee_token) : ZkScalar::one();
            if (defer) r->defer.reset(new DeferData());
            synthesize_update(r->cs, L, T, commitment, w.height, w.state, w.aux_data, w.next_state, ft, trs, nt, r->defer.get());
            if (r->defer && !r->defer->prog) r->defer.reset();
        } else if (w.kind == 0) {
            std::vector<DepositTransition> trs = w.deposits;
            while (trs.size() < cap) trs.push_back(DepositTransition::null(L, T));
            if (defer) r->defer.reset(new DeferData());
            synthesize_deposit(r->cs, L, T, commitment, w.height, w.state, w.aux_data, w.next_state, trs, nt, r->defer.get());
            if (r->defer && !r->defer->prog) r->defer.reset();
        } else {
            std::vector<WithdrawTransition> trs = w.withdraws;
            while (trs.size() < cap) trs.push_back(WithdrawTransition::null(L, T));
            if (defer) r->defer.reset(new DeferData());
            synthesize_withdraw(r->cs, L, T, commitment, w.height, w.state, w.aux_data, w.next_state, trs, nt, r->defer.get());
            if (r->defer && !r->defer->prog) r->defer.reset();
        }
        if (r->cs.check_failed_at >= 0) return BZK_E_INTERNAL;
        r->accepted = w.n_transitions();
        finish_r1cs(r.get());
        *out = r.release();
        return BZK_OK;
    } catch (const std::bad_alloc&) {
        return BZK_E_ALLOC;
    } catch (const std::exception& e) {
        g_work_error = e.what();
        return BZK_E_INTERNAL;
    }
}

// Validator side: one work of `prepare_works` (src/mpn/mod.rs:352-414) from the transactions queued in the world: runs the
// witness builder of that kind (the world moves to the next state, as the reference's mirror does), fills the public
// inputs [height, state, aux_data, next_state] and new_root.  kind: 0 deposit, 1 withdraw, 2 update.
int32_t bzk_mpn_make_work(bzk_mpn* w, int32_t kind, const bzk_mpn_work_config* cfg, uint64_t reward, bzk_mpn_work** out) {
    if (!w || !cfg || !out || kind < 0 || kind > 2) return BZK_E_ARG;
    *out = nullptr;
    if (cfg->log4_deposit_batch > 6 || cfg->log4_withdraw_batch > 6 || cfg->log4_update_batch > 6) return BZK_E_ARG;
    if (!cfg->deposit_vk || !cfg->withdraw_vk || !cfg->update_vk) return BZK_E_ARG;
    try {
        std::unique_ptr<bzk_mpn_work> h(new bzk_mpn_work());
        MpnWork& o = h->w;
        MpnWorkConfig& c = o.config;
        c.log4_tree = (uint8_t)w->L;
        c.log4_token_tree = (uint8_t)w->T;
        c.log4_deposit_batch = cfg->log4_deposit_batch;
        c.log4_withdraw_batch = cfg->log4_withdraw_batch;
        c.log4_update_batch = cfg->log4_update_batch;
        c.mpn_contract_id = w->contract_id;
        c.num_update_batches = cfg->num_update_batches;
        c.num_deposit_batches = cfg->num_deposit_batches;
        c.num_withdraw_batches = cfg->num_withdraw_batches;
        c.deposit_vk.assign(cfg->deposit_vk, cfg->deposit_vk + cfg->deposit_vk_len);
        c.withdraw_vk.assign(cfg->withdraw_vk, cfg->withdraw_vk + cfg->withdraw_vk_len);
        c.update_vk.assign(cfg->update_vk, cfg->update_vk + cfg->update_vk_len);
        if (!verifier_key_well_formed(c.deposit_vk) || !verifier_key_well_formed(c.withdraw_vk) || !verifier_key_well_formed(c.update_vk))
            return BZK_E_ARG;
        o.kind = kind;
        o.height = w->height;
        o.state = w->accounts->root();
        o.reward = reward;
        uint64_t rejected = 0;
        if (kind == 2) {
            uint64_t fee_sum = 0;
            if (w->dev) { const int32_t ds = build_transitions_dev(*w, c.log4_update_batch, ZkScalar::one(), o.updates, fee_sum, rejected); if (ds != BZK_OK) return ds; }
            else build_transitions(*w, c.log4_update_batch, ZkScalar::one(), o.updates, fee_sum, rejected);
            o.aux_data = h2(ZkScalar::one(), ZkScalar::from_u64(fee_sum));
        } else if (kind == 0) {
            if (w->dev) { const int32_t ds = build_deposits_dev(*w, c.log4_deposit_batch, o.deposits, rejected); if (ds != BZK_OK) return ds; }
            else build_deposits(*w, c.log4_deposit_batch, o.deposits, rejected);
            o.aux_data = deposit_aux(o.deposits, c.log4_deposit_batch);
        } else {
            for (auto& tx : w->withdraw_queue)
                if (tx.payment.empty()) return BZK_E_ARG;  // queued with an opaque fingerprint: no wire form
            if (w->dev) { const int32_t ds = build_withdraws_dev(*w, c.log4_withdraw_batch, o.withdraws, rejected); if (ds != BZK_OK) return ds; }
            else build_withdraws(*w, c.log4_withdraw_batch, o.withdraws, rejected);
            o.aux_data = withdraw_aux(o.withdraws, c.log4_withdraw_batch);
        }
        o.next_state = w->accounts->root();
        o.new_root_hash = o.next_state;
        o.new_root_size = cfg->new_root_state_size;
        *out = h.release();
        return BZK_OK;
    } catch (const std::bad_alloc&) {
        return BZK_E_ALLOC;
    } catch (const std::exception&) {
        return BZK_E_INTERNAL;
    }
}

// ZkProof::Groth16(Box<Groth16Proof>) (src/zk/mod.rs:646-651): u32 variant 0 + the 387 proof bytes bzk_groth16_prove writes
int32_t bzk_zkproof_encode(const uint8_t proof[387], uint8_t out[391]) {
    if (!proof || !out) return BZK_E_ARG;
    const uint32_t tag = 0;
    memcpy(out, &tag, 4);
    memcpy(out + 4, proof, 387);
    return BZK_OK;
}
int32_t bzk_zkproof_decode(const uint8_t* in, uint64_t len, uint8_t proof[387]) {
    if (!in || !proof || len < 391) return BZK_E_ARG;
    uint32_t tag;
    memcpy(&tag, in, 4);
    if (tag != 0) return BZK_E_ARG;
    for (int off : {4 + 96, 4 + 97 + 192, 4 + 97 + 193 + 96})
        if (in[off] > 1) return BZK_E_ARG;  // the three `bool` infinity flags
    memcpy(proof, in + 4, 387);
    return BZK_OK;
}

int32_t bzk_host_poseidon(const uint8_t* in, uint32_t arity, uint8_t out[32]) {
    if (!in || !out || arity < 1 || arity > 16) return BZK_E_ARG;
    ZkScalar v[16];
    for (uint32_t i = 0; i < arity; ++i) v[i] = ZkScalar::from_bytes(in + 32 * i);
    poseidon_hash(v, (int)arity).to_bytes(out);
    return BZK_OK;
}
// `ZkScalar::new(bytes)` (src/zk/mod.rs:262-271): the little-endian integer mod r, as Montgomery limbs; len <= 64
int32_t bzk_host_scalar_new(const uint8_t* in, uint32_t len, uint8_t out[32]) {
    if ((!in && len) || !out || len > 64) return BZK_E_ARG;
    ZkScalar::from_le_bytes_mod(in, len).to_bytes(out);
    return BZK_OK;
}
int32_t bzk_host_sha3_256(const uint8_t* in, uint64_t len, uint8_t out[32]) {
    if ((!in && len) || !out) return BZK_E_ARG;
    sha3_256(in, len, out);
    return BZK_OK;
}
// keys from seed: out = pub.x | pub.y | randomness | scalar (4 x 32 B Montgomery)
int32_t bzk_host_jubjub_keys(const uint8_t* seed, uint32_t len, uint8_t out[128]) {
    if (!seed || !out) return BZK_E_ARG;
    JubjubPrivateKey k = jubjub_generate_keys(seed, len);
    k.public_key.x.to_bytes(out);
    k.public_key.y.to_bytes(out + 32);
    k.randomness.to_bytes(out + 64);
    k.scalar.to_bytes(out + 96);
    return BZK_OK;
}
// sig = r.x | r.y | s
int32_t bzk_host_jubjub_sign(const uint8_t key[128], const uint8_t msg[32], uint8_t sig_out[96]) {
    if (!key || !msg || !sig_out) return BZK_E_ARG;
    JubjubPrivateKey k;
    k.public_key = {ZkScalar::from_bytes(key), ZkScalar::from_bytes(key + 32)};
    k.randomness = ZkScalar::from_bytes(key + 64);
    k.scalar = ZkScalar::from_bytes(key + 96);
    JubjubSignature s = jubjub_sign(k, ZkScalar::from_bytes(msg));
    s.r.x.to_bytes(sig_out);
    s.r.y.to_bytes(sig_out + 32);
    s.s.to_bytes(sig_out + 64);
    return BZK_OK;
}
// returns 1 valid, 0 invalid, negative on bad arguments
int32_t bzk_host_jubjub_verify(const uint8_t pub_xy[64], const uint8_t msg[32], const uint8_t sig[96]) {
    if (!pub_xy || !msg || !sig) return BZK_E_ARG;
    PointAffine pk = {ZkScalar::from_bytes(pub_xy), ZkScalar::from_bytes(pub_xy + 32)};
    JubjubSignature s = {{ZkScalar::from_bytes(sig), ZkScalar::from_bytes(sig + 32)}, ZkScalar::from_bytes(sig + 64)};
    return jubjub_verify(pk, ZkScalar::from_bytes(msg), s) ? 1 : 0;
}

}  // extern "C"
