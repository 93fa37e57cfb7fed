// Row (b), seam "state compress": `ZkStateModel::compress<H>(&self, data: &ZkDataPairs) -> Result<ZkCompressedState, _>`
// (/root/reference/src/zk/mod.rs:392-399) for ANY state model - arbitrary nestings of Scalar / Struct{field_types} /
// List{log4_size, item_type} (src/zk/mod.rs:332-345) - over SPARSE (ZkDataLocator, ZkScalar) pairs.
//
// The reference builds a RAM KV store and replays `KvStoreStateManager::set_data` once per pair (src/zk/state/mod.rs:66-90, 310-420):
// depth-many hashes per pair, one at a time.  The value it ends with does not depend on the order of the pairs:
//   value(Scalar)  = the stored scalar, 0 when absent
//   value(Struct)  = H(value(field_0), ..., value(field_{k-1}))
//   value(List)    = root of the 4-ary tree over the 4^log4 items (node = H(c0, c1, c2, c3); log4 = 0: the item itself)
//   an untouched sub-tree has value compress_default(type) (src/zk/mod.rs:401-423)
//   state_size     = number of non-zero scalars stored (set_data's size_diff bookkeeping over an initially empty state)
// Here the touched part of that tree is laid out on the host (sorted locators -> nodes with their input lists, defaults as shared
// constants), nodes are grouped by (height, arity), and each group is ONE batched Poseidon launch on the device (K1): a node is hashed
// exactly once, level by level from the scalars up.  Dense shapes have dedicated, faster entries (bzk_merkle4_root: List{Scalar};
// bzk_mpn_state_compress_dev / bzk_mpn_tree_*: the MPN account model); this entry is the general seam.
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <unordered_map>

#include "bzk_internal.h"
#include "host_zk.h"

namespace bzk {
int32_t poseidon_launch(bzk_ctx* ctx, const void* in_dev, uint32_t arity, uint64_t n, void* out_dev);  // poseidon.hip
}

namespace {
using namespace bzk;

struct ModelNode {
    int kind = 0;  // 0 Scalar, 1 Struct, 2 List
    std::vector<int> fields;
    int log4 = 0, item = -1;
    ZkScalar dflt;                    // compress_default of this type
    std::vector<ZkScalar> list_dflt;  // List: default value of a node at depth k of its tree (k = log4: the item default, k = 0: = dflt)
};
struct Model {
    std::vector<ModelNode> nodes;
    int root = -1;
};

// bincode 1.3 (fixint, little endian): enum tag u32 - 0 Scalar, 1 Struct { field_types: Vec (u64 length) }, 2 List { log4_size: u8,
// item_type: Box } (declaration order of src/zk/mod.rs:332-345)
struct Rd {
    const uint8_t* p;
    uint64_t len, off = 0;
    bool ok = true;
    uint64_t u(int bytes) {
        if (!ok || off + bytes > len) { ok = false; return 0; }
        uint64_t v = 0;
        for (int i = 0; i < bytes; ++i) v |= (uint64_t)p[off + i] << (8 * i);
        off += bytes;
        return v;
    }
};
int parse_model(Rd& r, Model& M, int depth) {
    if (depth > 32 || M.nodes.size() > 100000) { r.ok = false; return -1; }
    const uint64_t tag = r.u(4);
    if (!r.ok) return -1;
    const int id = (int)M.nodes.size();
    M.nodes.emplace_back();
    if (tag == 0) {
        M.nodes[id].kind = 0;
    } else if (tag == 1) {
        M.nodes[id].kind = 1;
        const uint64_t k = r.u(8);
        // `is_valid`: at most MAX_ARITY fields (src/zk/mod.rs:355-367); an empty struct would hash zero values, for which no Poseidon
        // instance exists (the reference unwraps a missing parameter set)
        if (!r.ok || k == 0 || k > 16) { r.ok = false; return -1; }
        std::vector<int> f;
        for (uint64_t i = 0; i < k; ++i) {
            const int c = parse_model(r, M, depth + 1);
            if (c < 0) return -1;
            f.push_back(c);
        }
        M.nodes[id].fields = f;
    } else if (tag == 2) {
        M.nodes[id].kind = 2;
        const uint64_t lg = r.u(1);
        if (!r.ok || lg > 31) { r.ok = false; return -1; }  // indices are u64: 4^31 items at most
        const int c = parse_model(r, M, depth + 1);
        if (c < 0) return -1;
        M.nodes[id].log4 = (int)lg;
        M.nodes[id].item = c;
    } else {
        r.ok = false;
        return -1;
    }
    return id;
}
void model_defaults(Model& M, int id) {  // `compress_default` (src/zk/mod.rs:401-423), children first
    ModelNode& n = M.nodes[id];
    if (n.kind == 0) {
        n.dflt = ZkScalar::zero();
    } else if (n.kind == 1) {
        std::vector<ZkScalar> v;
        for (int f : n.fields) {
            model_defaults(M, f);
            v.push_back(M.nodes[f].dflt);
        }
        M.nodes[id].dflt = poseidon_hash(v);
    } else {
        model_defaults(M, n.item);
        ModelNode& nn = M.nodes[id];
        nn.list_dflt.assign((size_t)nn.log4 + 1, ZkScalar::zero());
        nn.list_dflt[nn.log4] = M.nodes[nn.item].dflt;
        for (int k = nn.log4 - 1; k >= 0; --k) {
            ZkScalar c[4] = {nn.list_dflt[k + 1], nn.list_dflt[k + 1], nn.list_dflt[k + 1], nn.list_dflt[k + 1]};
            nn.list_dflt[k] = poseidon_hash(c, 4);
        }
        nn.dflt = nn.list_dflt[0];
    }
}
int32_t load_model(const uint8_t* bytes, uint64_t len, Model& M) {
    if (!bytes) return BZK_E_ARG;
    Rd r{bytes, len};
    M.root = parse_model(r, M, 0);
    if (!r.ok || M.root < 0 || r.off != len) return BZK_E_ARG;
    model_defaults(M, M.root);
    return BZK_OK;
}

// the touched part of the state as a hash schedule
struct Plan {
    // value ids: [0, consts.size()) shared default constants, then the n pair values, then the hashed nodes
    std::vector<ZkScalar> consts;
    std::map<std::pair<int, int>, uint32_t> const_id;  // (model node, tree depth or -1) -> id
    struct Node { uint32_t first_in, arity, level; };
    std::vector<Node> nodes;      // hashed nodes in creation order
    std::vector<uint32_t> inputs;  // their input ids, back to back (node ids are offset by `hash_base` once all leaves are known)
    uint64_t n_pairs = 0;
    std::string err;
    int32_t code = 0;  // BZK_REFUSE_* of `err`
};
constexpr uint32_t HASHED = 0x80000000u;  // input id tag: index into Plan::nodes (resolved after levelling)

struct Builder {
    const Model& M;
    Plan& P;
    const uint64_t* loc_off;
    const uint64_t* loc;
    const std::vector<uint64_t>& order;  // pair indices, sorted by locator
    uint32_t cst(int model, int depth, const ZkScalar& v) {
        auto key = std::make_pair(model, depth);
        auto it = P.const_id.find(key);
        if (it != P.const_id.end()) return it->second;
        const uint32_t id = (uint32_t)P.consts.size();
        P.consts.push_back(v);
        P.const_id[key] = id;
        return id;
    }
    uint64_t at(uint64_t pos, uint64_t d) const { return loc[loc_off[order[pos]] + d]; }
    uint64_t len(uint64_t pos) const { return loc_off[order[pos] + 1] - loc_off[order[pos]]; }
    uint32_t level_of(uint32_t id) const { return (id & HASHED) ? P.nodes[id & ~HASHED].level : 0; }
    uint32_t hashed(const std::vector<uint32_t>& in) {
        uint32_t lv = 0;
        for (uint32_t x : in) lv = std::max(lv, level_of(x));
        if (P.nodes.size() >= 0x7ffffff0u) { P.err = "too many nodes"; P.code = BZK_REFUSE_OTHER; return 0; }
        P.nodes.push_back({(uint32_t)P.inputs.size(), (uint32_t)in.size(), lv + 1});
        P.inputs.insert(P.inputs.end(), in.begin(), in.end());
        return HASHED | (uint32_t)(P.nodes.size() - 1);
    }
    // value id of the sub-state of type `m` under the common prefix of length d shared by the sorted pairs [lo, hi) (non-empty)
    uint32_t build(int m, uint64_t d, uint64_t lo, uint64_t hi) {
        const ModelNode& n = M.nodes[m];
        if (!P.err.empty()) return 0;
        if (n.kind == 0) {
            // `set_data`: the locator must end exactly here (NonScalarLocatorError / LocatorError otherwise); a HashMap holds a key once
            if (hi - lo != 1) { P.err = "duplicate locator"; P.code = BZK_REFUSE_DUPLICATE_LOCATOR; return 0; }
            if (len(lo) != d) { P.err = "locator points below a scalar (ZkLocatorError::InvalidLocator)"; P.code = BZK_REFUSE_INVALID_LOCATOR; return 0; }
            return 0x40000000u | (uint32_t)order[lo];  // pair value, resolved to consts.size() + pair index
        }
        for (uint64_t i = lo; i < hi; ++i)
            if (len(i) <= d) { P.err = "locator does not reach a scalar (StateManagerError::NonScalarLocatorError)"; P.code = BZK_REFUSE_NON_SCALAR_LOCATOR; return 0; }
        if (n.kind == 1) {
            std::vector<uint32_t> in(n.fields.size());
            uint64_t i = lo;
            for (size_t f = 0; f < n.fields.size(); ++f) {
                uint64_t j = i;
                while (j < hi && at(j, d) == f) ++j;
                in[f] = j > i ? build(n.fields[f], d + 1, i, j) : cst(n.fields[f], -1, M.nodes[n.fields[f]].dflt);
                i = j;
            }
            if (i != hi) { P.err = "struct field index out of range (the reference indexes field_types out of bounds)"; P.code = BZK_REFUSE_INVALID_LOCATOR; return 0; }
            return hashed(in);
        }
        // List: items first, then the sparse 4-ary tree over them, level by level
        const uint64_t size = n.log4 >= 32 ? ~0ull : ((uint64_t)1 << (2 * n.log4));
        std::vector<std::pair<uint64_t, uint32_t>> cur;  // (index at the current depth, value id), ascending
        for (uint64_t i = lo; i < hi;) {
            const uint64_t idx = at(i, d);
            if (idx >= size) { P.err = "list index out of range (ZkLocatorError::InvalidLocator)"; P.code = BZK_REFUSE_INVALID_LOCATOR; return 0; }
            uint64_t j = i;
            while (j < hi && at(j, d) == idx) ++j;
            cur.push_back({idx, build(n.item, d + 1, i, j)});
            if (!P.err.empty()) return 0;
            i = j;
        }
        for (int k = n.log4; k > 0; --k) {  // children at depth k -> parents at depth k - 1
            const uint32_t dk = cst(m, k, n.list_dflt[k]);
            std::vector<std::pair<uint64_t, uint32_t>> up;
            for (size_t i = 0; i < cur.size();) {
                const uint64_t parent = cur[i].first >> 2;
                std::vector<uint32_t> in(4, dk);
                while (i < cur.size() && (cur[i].first >> 2) == parent) {
                    in[cur[i].first & 3] = cur[i].second;
                    ++i;
                }
                up.push_back({parent, hashed(in)});
            }
            cur.swap(up);
        }
        return cur[0].second;
    }
};

__global__ void __launch_bounds__(256) state_gather_kernel(const Fr* __restrict__ vals, const uint32_t* __restrict__ idx, uint64_t count,
                                                           Fr* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = vals[idx[i]];
}

int32_t compress_core(bzk_ctx* ctx, const Model& M, const uint64_t* loc_off, const uint64_t* loc, const uint8_t* values, uint64_t n,
                      uint8_t state_hash[32], uint64_t* state_size) {
    const ModelNode& root = M.nodes[M.root];
    uint64_t nz = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t* v = values + 32 * i;
        bool any = false;
        for (int b = 0; b < 32; ++b) any |= v[b] != 0;
        nz += any;
        // Montgomery limbs of a residue: < r as an integer (a `ZkScalar` cannot hold anything else)
        Fr f;
        memcpy(f.l, v, 32);
        Fr g = f;
        fe_reduce_once<FrParams>(g);
        if (!g.equals(f)) { ctx->last_error = "state_compress: a value is not a canonical field element"; ctx->last_refusal = BZK_REFUSE_NON_CANONICAL_VALUE; return BZK_E_ARG; }
    }
    if (state_size) *state_size = nz;
    if (n == 0) {
        root.dflt.to_bytes(state_hash);
        return BZK_OK;
    }
    if (n >= 0x3fffffffull) return BZK_E_ARG;
    // sort the pairs by locator (lexicographic; a proper prefix sorts first and is then reported as an error by the builder)
    std::vector<uint64_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    for (uint64_t i = 0; i < n; ++i)
        if (loc_off[i] > loc_off[i + 1]) return BZK_E_ARG;
    std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
        return std::lexicographical_compare(loc + loc_off[a], loc + loc_off[a + 1], loc + loc_off[b], loc + loc_off[b + 1]);
    });
    Plan P;
    P.n_pairs = n;
    Builder B{M, P, loc_off, loc, order};
    const uint32_t top = B.build(M.root, 0, 0, n);
    if (!P.err.empty()) {
        ctx->last_error = "state_compress: " + P.err;
        ctx->last_refusal = P.code;
        return BZK_E_ARG;
    }
    const uint32_t n_const = (uint32_t)P.consts.size();
    const uint32_t hash_base = n_const + (uint32_t)n;
    if (!(top & HASHED)) {  // the model is a bare Scalar (or a chain of log4 = 0 lists down to one): the value itself
        memcpy(state_hash, values + 32 * (size_t)(top & 0x3fffffffu), 32);
        return BZK_OK;
    }
    // group the hashed nodes by (level, arity): final id = hash_base + rank in that order
    const size_t nn = P.nodes.size();
    std::vector<uint32_t> by(nn), rank(nn);
    std::iota(by.begin(), by.end(), 0u);
    std::stable_sort(by.begin(), by.end(), [&](uint32_t a, uint32_t b) {
        if (P.nodes[a].level != P.nodes[b].level) return P.nodes[a].level < P.nodes[b].level;
        return P.nodes[a].arity < P.nodes[b].arity;
    });
    for (size_t r = 0; r < nn; ++r) rank[by[r]] = (uint32_t)r;
    auto resolve = [&](uint32_t id) -> uint32_t {
        if (id & HASHED) return hash_base + rank[id & ~HASHED];
        if (id & 0x40000000u) return n_const + (id & 0x3fffffffu);
        return id;
    };
    std::vector<uint32_t> gidx;  // gather indices of all groups, back to back in group order
    gidx.reserve(P.inputs.size());
    struct Group { uint32_t arity; uint64_t count, in_off, first_rank; };
    std::vector<Group> groups;
    for (size_t r = 0; r < nn;) {
        const Plan::Node& a = P.nodes[by[r]];
        size_t e = r;
        while (e < nn && P.nodes[by[e]].level == a.level && P.nodes[by[e]].arity == a.arity) ++e;
        groups.push_back({a.arity, (uint64_t)(e - r), (uint64_t)gidx.size(), (uint64_t)r});
        for (size_t q = r; q < e; ++q) {
            const Plan::Node& x = P.nodes[by[q]];
            for (uint32_t k = 0; k < x.arity; ++k) gidx.push_back(resolve(P.inputs[x.first_in + k]));
        }
        r = e;
    }
    // device: values array (constants | pair values | hashed nodes), gather indices, one staging buffer for the widest group
    uint64_t widest = 0;
    for (auto& g : groups) widest = std::max<uint64_t>(widest, g.count * g.arity);
    const size_t vals_bytes = ((size_t)hash_base + nn) * 32;
    const size_t total = ws_pad(vals_bytes) + ws_pad(gidx.size() * 4) + ws_pad(widest * 32) + 4096;
    (void)hipSetDevice(ctx->device);
    BZK_TRY(ws_reserve(ctx, total));
    WsCursor cur(ctx->ws);
    Fr* d_vals = cur.take<Fr>((size_t)hash_base + nn);
    uint32_t* d_idx = cur.take<uint32_t>(gidx.size());
    Fr* d_in = cur.take<Fr>(widest);
    BZK_HIP(ctx, hipMemcpyAsync(d_vals, P.consts.data(), (size_t)n_const * 32, hipMemcpyHostToDevice, ctx->stream));
    BZK_HIP(ctx, hipMemcpyAsync(d_vals + n_const, values, (size_t)n * 32, hipMemcpyHostToDevice, ctx->stream));
    BZK_HIP(ctx, hipMemcpyAsync(d_idx, gidx.data(), gidx.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    for (auto& g : groups) {
        const uint64_t cnt = g.count * g.arity;
        BZK_LAUNCH(ctx, "state_gather", state_gather_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, (const Fr*)d_vals,
                   (const uint32_t*)(d_idx + g.in_off), cnt, d_in);
        BZK_TRY(poseidon_launch(ctx, d_in, g.arity, g.count, d_vals + hash_base + g.first_rank));
    }
    BZK_TRY(pinned_reserve(ctx, 64));
    BZK_HIP(ctx, hipMemcpyAsync(ctx->pinned, d_vals + resolve(top), 32, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // also: the host vectors above outlive every copy out of them
    memcpy(state_hash, ctx->pinned, 32);
    return BZK_OK;
}


// ------------------------------------------------------------------------------------------------------------------------------
// PERSISTENT device state of one contract, any model: `KvStoreStateManager::{update_contract, set_data, get_data, prove, root}`
// (/root/reference/src/zk/state/mod.rs:218-438) with the VALUES resident in HBM.
//
// The reference keeps every scalar and every non-default node in a KV store and walks root-wards once per written scalar
// (depth-many hashes, one at a time, each sibling a KV read).  Here the store is one device array of 32-byte slots: [0, n_const)
// the defaults of the model (compress_default per type and per tree depth), then one slot per scalar / struct / list / inner tree
// node that was ever touched.  The host keeps only the INDEX (key -> slot) and one bit per scalar (zero / non-zero, for
// state_size).  One `update_contract` = one plan: the union of the root-ward paths of all written scalars, every node once,
// grouped by (height, arity) -> one gather + one batched Poseidon launch (K1) + one scatter per group, in place.  Untouched
// siblings are read from their slots on the device; nothing but the new scalars goes up and nothing but the root comes back.
//   key of a value at a locator            = the locator
//   key of an inner node of a list's tree  = locator of the list ++ {AUX | depth, index at that depth}   (0 < depth < log4_size)
// Nodes that return to their default stay in the index (the reference removes them): same values either way.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr uint64_t AUX = (uint64_t)1 << 63;  // no list index reaches it: 4^31 items at most
typedef std::vector<uint64_t> Key;
struct KeyHash {
    size_t operator()(const Key& k) const {
        uint64_t h = 0x9e3779b97f4a7c15ull ^ k.size();
        for (uint64_t x : k) {
            h ^= x + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
            h *= 0xff51afd7ed558ccdull;
            h ^= h >> 33;
        }
        return (size_t)h;
    }
};

__global__ void __launch_bounds__(256) state_scatter_kernel(const Fr* __restrict__ in, const uint32_t* __restrict__ dst, uint64_t count,
                                                            Fr* __restrict__ vals) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) vals[dst[i]] = in[i];
}

}  // namespace

struct bzk_state {
    bzk_ctx* ctx = nullptr;
    int device = 0;
    Model M;
    std::vector<uint32_t> dflt_slot;                // per model node: slot of compress_default(type)
    std::vector<std::vector<uint32_t>> depth_slot;  // lists: slot of the default of a node at depth k of the tree (k = 0 .. log4)
    uint32_t n_const = 0;
    Fr* d_vals = nullptr;
    size_t cap = 0, used = 0;
    std::unordered_map<Key, uint32_t, KeyHash> slot_of;
    std::vector<uint8_t> nonzero;  // per slot; meaningful for scalar slots
    uint64_t size = 0, height = 0;
    uint8_t root_hash[32];
    bool poisoned = false;  // a device error in the middle of an update: the slots no longer describe one state
    std::mutex m;
    // every call that used the store has synchronised its stream before returning, and the context is not touched here: a host that
    // tears its handles down in arbitrary order (a garbage collector) may already have destroyed it
    ~bzk_state() {
        if (d_vals) {
            (void)hipSetDevice(device);
            (void)hipFree(d_vals);
        }
    }
};

namespace {

int32_t state_grow(bzk_state* S, size_t need) {
    if (need <= S->cap) return BZK_OK;
    bzk_ctx* ctx = S->ctx;
    size_t cap = std::max<size_t>(S->cap * 2, 4096);
    while (cap < need) cap *= 2;
    if (cap >= 0x7fffffffull) cap = 0x7fffffffull;
    if (need > cap) { ctx->last_error = "state: more than 2^31 slots"; return BZK_E_ALLOC; }
    Fr* nv = nullptr;
    if (hipMalloc(&nv, cap * sizeof(Fr)) != hipSuccess) {
        (void)hipGetLastError();
        ctx->last_error = "state: out of device memory growing the value store";
        return BZK_E_ALLOC;
    }
    if (S->used) {
        hipError_t e = hipMemcpyAsync(nv, S->d_vals, S->used * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(nv);
            ctx->last_error = std::string("state: copying the value store: ") + hipGetErrorString(e);
            return BZK_E_DEVICE;
        }
    }
    if (S->d_vals) (void)hipFree(S->d_vals);
    S->d_vals = nv;
    S->cap = cap;
    return BZK_OK;
}

// `ZkStateModel::locate` (src/zk/mod.rs:367-390): the model node a locator names, -1 where the reference errors (or indexes
// field_types out of bounds)
int state_locate(const Model& M, const uint64_t* loc, uint64_t len) {
    int m = M.root;
    for (uint64_t i = 0; i < len; ++i) {
        const ModelNode& n = M.nodes[m];
        if (n.kind == 1) {
            if (loc[i] >= n.fields.size()) return -1;
            m = n.fields[loc[i]];
        } else if (n.kind == 2) {
            if (n.log4 < 32 && loc[i] >= ((uint64_t)1 << (2 * n.log4))) return -1;
            m = n.item;
        } else {
            return -1;
        }
    }
    return m;
}

// gathers `idx.size()` slots and brings them to the host
int32_t state_read_slots(bzk_state* S, const std::vector<uint32_t>& idx, uint8_t* out) {
    bzk_ctx* ctx = S->ctx;
    if (idx.empty()) return BZK_OK;
    (void)hipSetDevice(ctx->device);
    BZK_TRY(ws_reserve(ctx, ws_pad(idx.size() * 4) + ws_pad(idx.size() * 32) + 1024));
    WsCursor cur(ctx->ws);
    uint32_t* d_idx = cur.take<uint32_t>(idx.size());
    Fr* d_out = cur.take<Fr>(idx.size());
    BZK_HIP(ctx, hipMemcpyAsync(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    BZK_LAUNCH(ctx, "state_gather", state_gather_kernel, dim3((unsigned)((idx.size() + 255) / 256)), dim3(256), 0, (const Fr*)S->d_vals,
               (const uint32_t*)d_idx, (uint64_t)idx.size(), d_out);
    BZK_HIP(ctx, hipMemcpyAsync(out, d_out, idx.size() * 32, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

// the plan of one update_contract
struct UpPlan {
    struct Node { uint32_t dst, first_in, arity, level; };
    std::vector<Node> nodes;
    std::vector<uint32_t> inputs;                    // slots
    std::vector<std::pair<Key, uint32_t>> new_keys;  // index entries to add once the update has run
    std::vector<uint32_t> scalar_dst;                // per pair: its slot
    uint32_t next_slot = 0;
    std::string err;
    int32_t code = 0;  // BZK_REFUSE_* of `err`
};
struct Updater {
    bzk_state& S;
    UpPlan& P;
    const uint64_t* loc_off;
    const uint64_t* loc;
    const std::vector<uint64_t>& order;
    Key prefix;
    struct Val { uint32_t slot, level; };
    uint64_t at(uint64_t pos, uint64_t d) const { return loc[loc_off[order[pos]] + d]; }
    uint64_t len(uint64_t pos) const { return loc_off[order[pos] + 1] - loc_off[order[pos]]; }
    uint32_t slot_for(const Key& k) {  // the slot of a key that is being written: its own, or a fresh one
        auto it = S.slot_of.find(k);
        if (it != S.slot_of.end()) return it->second;
        if (P.next_slot >= 0x7ffffff0u) { P.err = "too many nodes"; P.code = BZK_REFUSE_OTHER; return 0; }
        P.new_keys.push_back({k, P.next_slot});
        return P.next_slot++;
    }
    uint32_t read_slot(const Key& k, uint32_t dflt) const {  // an untouched sibling: its slot, or the default of its kind
        auto it = S.slot_of.find(k);
        return it != S.slot_of.end() ? it->second : dflt;
    }
    Val hashed(const Key& dst_key, const std::vector<uint32_t>& in, uint32_t level) {
        const uint32_t dst = slot_for(dst_key);
        P.nodes.push_back({dst, (uint32_t)P.inputs.size(), (uint32_t)in.size(), level});
        P.inputs.insert(P.inputs.end(), in.begin(), in.end());
        return {dst, level};
    }
    // the value at `prefix` (of model type m, prefix.size() == d) given the sorted pairs [lo, hi) below it (non-empty)
    Val build(int m, uint64_t d, uint64_t lo, uint64_t hi) {
        const ModelNode& n = S.M.nodes[m];
        if (!P.err.empty()) return {0, 0};
        if (n.kind == 0) {
            if (hi - lo != 1) { P.err = "duplicate locator"; P.code = BZK_REFUSE_DUPLICATE_LOCATOR; return {0, 0}; }
            if (len(lo) != d) { P.err = "locator points below a scalar (ZkLocatorError::InvalidLocator)"; P.code = BZK_REFUSE_INVALID_LOCATOR; return {0, 0}; }
            const uint32_t s = slot_for(prefix);
            P.scalar_dst[order[lo]] = s;
            return {s, 0};
        }
        for (uint64_t i = lo; i < hi; ++i)
            if (len(i) <= d) { P.err = "locator does not reach a scalar (StateManagerError::NonScalarLocatorError)"; P.code = BZK_REFUSE_NON_SCALAR_LOCATOR; return {0, 0}; }
        if (n.kind == 1) {
            std::vector<uint32_t> in(n.fields.size());
            uint32_t lv = 0;
            uint64_t i = lo;
            for (size_t f = 0; f < n.fields.size(); ++f) {
                uint64_t j = i;
                while (j < hi && at(j, d) == f) ++j;
                prefix.push_back(f);
                if (j > i) {
                    const Val v = build(n.fields[f], d + 1, i, j);
                    in[f] = v.slot;
                    lv = std::max(lv, v.level);
                } else {
                    in[f] = read_slot(prefix, S.dflt_slot[n.fields[f]]);
                }
                prefix.pop_back();
                i = j;
            }
            if (i != hi) { P.err = "struct field index out of range (the reference indexes field_types out of bounds)"; P.code = BZK_REFUSE_INVALID_LOCATOR; return {0, 0}; }
            if (!P.err.empty()) return {0, 0};
            return hashed(prefix, in, lv + 1);
        }
        const uint64_t size = n.log4 >= 32 ? ~0ull : ((uint64_t)1 << (2 * n.log4));
        struct Cur { uint64_t idx; Val v; };
        std::vector<Cur> cur;
        for (uint64_t i = lo; i < hi;) {
            const uint64_t idx = at(i, d);
            if (idx >= size) { P.err = "list index out of range (ZkLocatorError::InvalidLocator)"; P.code = BZK_REFUSE_INVALID_LOCATOR; return {0, 0}; }
            uint64_t j = i;
            while (j < hi && at(j, d) == idx) ++j;
            prefix.push_back(idx);
            cur.push_back({idx, build(n.item, d + 1, i, j)});
            prefix.pop_back();
            if (!P.err.empty()) return {0, 0};
            i = j;
        }
        if (n.log4 == 0) {
            // `set_data`'s level loop does not run: the list's value IS its only item; both locators name one slot
            // (this prefix is visited once per plan, so the index alone says whether the alias exists)
            if (S.slot_of.find(prefix) == S.slot_of.end()) P.new_keys.push_back({prefix, cur[0].v.slot});
            return cur[0].v;
        }
        for (int k = n.log4; k > 0; --k) {  // children at depth k -> parents at depth k - 1
            std::vector<Cur> up;
            for (size_t i = 0; i < cur.size();) {
                const uint64_t parent = cur[i].idx >> 2;
                std::vector<uint32_t> in(4);
                uint32_t lv = 0;
                for (uint64_t j = 0; j < 4; ++j) {
                    const uint64_t child = 4 * parent + j;
                    if (i < cur.size() && cur[i].idx == child) {
                        in[j] = cur[i].v.slot;
                        lv = std::max(lv, cur[i].v.level);
                        ++i;
                    } else {
                        if (k == n.log4) {
                            prefix.push_back(child);
                        } else {
                            prefix.push_back(AUX | (uint64_t)k);
                            prefix.push_back(child);
                        }
                        in[j] = read_slot(prefix, S.depth_slot[m][k]);
                        prefix.resize(d);
                    }
                }
                if (k > 1) {
                    prefix.push_back(AUX | (uint64_t)(k - 1));
                    prefix.push_back(parent);
                }
                const Val v = hashed(prefix, in, lv + 1);
                prefix.resize(d);
                if (!P.err.empty()) return {0, 0};
                up.push_back({parent, v});
            }
            cur.swap(up);
        }
        return cur[0].v;
    }
};

int32_t state_update_impl(bzk_state* S, const uint64_t* loc_off, const uint64_t* loc, const uint8_t* values, uint64_t n, uint64_t target_height,
                          uint8_t* prev_values_out) {
    bzk_ctx* ctx = S->ctx;
    if (S->poisoned) { ctx->last_error = "state: an earlier update failed on the device; this state is unusable"; return BZK_E_DEVICE; }
    if (n >= 0x3fffffffull) return BZK_E_ARG;
    for (uint64_t i = 0; i < n; ++i) {
        if (loc_off[i] > loc_off[i + 1]) return BZK_E_ARG;
        Fr f;
        memcpy(f.l, values + 32 * i, 32);
        Fr g = f;
        fe_reduce_once<FrParams>(g);
        if (!g.equals(f)) { ctx->last_error = "state_update: a value is not a canonical field element"; ctx->last_refusal = BZK_REFUSE_NON_CANONICAL_VALUE; return BZK_E_ARG; }
    }
    if (n == 0) {
        S->height = target_height;
        return BZK_OK;
    }
    std::vector<uint64_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
        return std::lexicographical_compare(loc + loc_off[a], loc + loc_off[a + 1], loc + loc_off[b], loc + loc_off[b + 1]);
    });
    UpPlan P;
    P.next_slot = (uint32_t)S->used;
    P.scalar_dst.assign(n, 0);
    Updater U{*S, P, loc_off, loc, order, {}};
    const Updater::Val top = U.build(S->M.root, 0, 0, n);
    if (!P.err.empty()) {
        ctx->last_error = "state_update: " + P.err;
        ctx->last_refusal = P.code;
        return BZK_E_ARG;
    }
    (void)hipSetDevice(ctx->device);
    BZK_TRY(state_grow(S, P.next_slot));
    // the rollback of this delta (`ZkState::push_delta`, src/zk/mod.rs:521-530): what the named scalars held before; 0 = nothing
    if (prev_values_out) {
        std::vector<uint32_t> idx;
        std::vector<uint64_t> which;
        for (uint64_t i = 0; i < n; ++i) {
            if (P.scalar_dst[i] < S->used) {
                idx.push_back(P.scalar_dst[i]);
                which.push_back(i);
            } else {
                memset(prev_values_out + 32 * i, 0, 32);
            }
        }
        std::vector<uint8_t> got(idx.size() * 32);
        BZK_TRY(state_read_slots(S, idx, got.data()));
        for (size_t q = 0; q < which.size(); ++q) memcpy(prev_values_out + 32 * which[q], got.data() + 32 * q, 32);
    }
    // groups of equal (level, arity), lowest level first
    const size_t nn = P.nodes.size();
    std::vector<uint32_t> by(nn);
    std::iota(by.begin(), by.end(), 0u);
    std::stable_sort(by.begin(), by.end(), [&](uint32_t a, uint32_t b) {
        if (P.nodes[a].level != P.nodes[b].level) return P.nodes[a].level < P.nodes[b].level;
        return P.nodes[a].arity < P.nodes[b].arity;
    });
    struct Group { uint32_t arity; uint64_t count, in_off, dst_off; };
    std::vector<Group> groups;
    std::vector<uint32_t> gidx, gdst;
    gidx.reserve(P.inputs.size());
    gdst.reserve(nn);
    uint64_t widest = n;
    for (size_t r = 0; r < nn;) {
        const UpPlan::Node& a = P.nodes[by[r]];
        size_t e = r;
        while (e < nn && P.nodes[by[e]].level == a.level && P.nodes[by[e]].arity == a.arity) ++e;
        groups.push_back({a.arity, (uint64_t)(e - r), (uint64_t)gidx.size(), (uint64_t)gdst.size()});
        for (size_t q = r; q < e; ++q) {
            const UpPlan::Node& x = P.nodes[by[q]];
            for (uint32_t k = 0; k < x.arity; ++k) gidx.push_back(P.inputs[x.first_in + k]);
            gdst.push_back(x.dst);
        }
        widest = std::max<uint64_t>(widest, (uint64_t)(e - r) * a.arity);
        r = e;
    }
    BZK_TRY(ws_reserve(ctx, ws_pad(n * 32) + ws_pad(n * 4) + ws_pad(gidx.size() * 4) + ws_pad(gdst.size() * 4) + 2 * ws_pad(widest * 32) + 4096));
    WsCursor cur(ctx->ws);
    Fr* d_new = cur.take<Fr>(n);
    uint32_t* d_sdst = cur.take<uint32_t>(n);
    uint32_t* d_idx = cur.take<uint32_t>(gidx.size());
    uint32_t* d_dst = cur.take<uint32_t>(gdst.size());
    Fr* d_in = cur.take<Fr>(widest);
    Fr* d_out = cur.take<Fr>(widest);
    // the sources of these copies are pageable vectors of this frame and the caller's `values`: no early return may leave one of
    // them in flight - a failed upload waits for the stream before it reports
    auto upload = [&]() -> int32_t {
        BZK_HIP(ctx, hipMemcpyAsync(d_new, values, (size_t)n * 32, hipMemcpyHostToDevice, ctx->stream));
        BZK_HIP(ctx, hipMemcpyAsync(d_sdst, P.scalar_dst.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!gidx.empty()) BZK_HIP(ctx, hipMemcpyAsync(d_idx, gidx.data(), gidx.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!gdst.empty()) BZK_HIP(ctx, hipMemcpyAsync(d_dst, gdst.data(), gdst.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        return BZK_OK;
    };
    if (const int32_t up = upload(); up != BZK_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return up;  // no slot was written yet: the state is still the old one
    }
    // from here on the slots change: a failure leaves them half-written
    auto run = [&]() -> int32_t {
        BZK_LAUNCH(ctx, "state_scatter", state_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (const Fr*)d_new, (const uint32_t*)d_sdst,
                   n, S->d_vals);
        for (auto& g : groups) {
            const uint64_t cnt = g.count * g.arity;
            BZK_LAUNCH(ctx, "state_gather", state_gather_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, (const Fr*)S->d_vals,
                       (const uint32_t*)(d_idx + g.in_off), cnt, d_in);
            BZK_TRY(poseidon_launch(ctx, d_in, g.arity, g.count, d_out));
            BZK_LAUNCH(ctx, "state_scatter", state_scatter_kernel, dim3((unsigned)((g.count + 255) / 256)), dim3(256), 0, (const Fr*)d_out,
                       (const uint32_t*)(d_dst + g.dst_off), g.count, S->d_vals);
        }
        BZK_TRY(pinned_reserve(ctx, 64));
        BZK_HIP(ctx, hipMemcpyAsync(ctx->pinned, S->d_vals + top.slot, 32, hipMemcpyDeviceToHost, ctx->stream));
        BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return BZK_OK;
    };
    const int32_t st = run();
    if (st != BZK_OK) {
        (void)hipStreamSynchronize(ctx->stream);  // whatever was queued reads this frame's vectors: let it finish (or fail) first
        S->poisoned = true;
        return st;
    }
    memcpy(S->root_hash, ctx->pinned, 32);
    S->used = P.next_slot;
    S->nonzero.resize(S->used, 0);
    for (auto& nk : P.new_keys) S->slot_of.emplace(std::move(nk.first), nk.second);
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t* v = values + 32 * i;
        bool now = false;
        for (int b = 0; b < 32; ++b) now |= v[b] != 0;
        uint8_t& was = S->nonzero[P.scalar_dst[i]];
        if (now && !was) ++S->size;
        if (!now && was) --S->size;
        was = now ? 1 : 0;
    }
    S->height = target_height;
    return BZK_OK;
}

}  // namespace

namespace bzk {
int32_t hash_plan_run(bzk_ctx* ctx, const uint8_t* uploaded, uint64_t n_up, const std::vector<HashGroup>& groups, std::vector<uint8_t>& hashed_out) {
    if (!ctx || (n_up && !uploaded)) return BZK_E_ARG;
    uint64_t n_hash = 0, n_idx = 0, widest = 0;
    for (auto& g : groups) {
        if (g.arity < 1 || g.arity > 16 || g.in.size() % g.arity) return BZK_E_ARG;
        for (uint32_t id : g.in)
            if (id >= n_up + n_hash) return BZK_E_INTERNAL;  // a group may only consume earlier values
        n_hash += g.count();
        n_idx += g.in.size();
        widest = std::max<uint64_t>(widest, g.in.size());
    }
    hashed_out.assign((size_t)n_hash * 32, 0);
    if (!n_hash) return BZK_OK;
    (void)hipSetDevice(ctx->device);
    BZK_TRY(ws_reserve(ctx, ws_pad((n_up + n_hash) * 32) + ws_pad(n_idx * 4) + ws_pad(widest * 32) + 4096));
    WsCursor cur(ctx->ws);
    Fr* d_vals = cur.take<Fr>(n_up + n_hash);
    uint32_t* d_idx = cur.take<uint32_t>(n_idx);
    Fr* d_in = cur.take<Fr>(widest);
    std::vector<uint32_t> all;
    all.reserve(n_idx);
    for (auto& g : groups) all.insert(all.end(), g.in.begin(), g.in.end());
    if (n_up) BZK_HIP(ctx, hipMemcpyAsync(d_vals, uploaded, (size_t)n_up * 32, hipMemcpyHostToDevice, ctx->stream));
    BZK_HIP(ctx, hipMemcpyAsync(d_idx, all.data(), all.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    uint64_t off = 0, out = n_up;
    for (auto& g : groups) {
        const uint64_t cnt = g.in.size();
        if (!cnt) continue;
        BZK_LAUNCH(ctx, "state_gather", state_gather_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, (const Fr*)d_vals,
                   (const uint32_t*)(d_idx + off), cnt, d_in);
        BZK_TRY(poseidon_launch(ctx, d_in, g.arity, g.count(), d_vals + out));
        off += cnt;
        out += g.count();
    }
    BZK_HIP(ctx, hipMemcpyAsync(hashed_out.data(), d_vals + n_up, (size_t)n_hash * 32, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}
}  // namespace bzk

// bzk_last_refusal (include/bzk.h): every public state call starts with "not refused"; a BZK_E_ARG that no site classified is OTHER
template <class Fn>
static int32_t refusal_scope(bzk_ctx* ctx, Fn&& fn) {
    if (ctx) ctx->last_refusal = BZK_REFUSE_NONE;
    const int32_t st = fn();
    if (ctx) {
        if (st != BZK_E_ARG) ctx->last_refusal = BZK_REFUSE_NONE;
        else if (ctx->last_refusal == BZK_REFUSE_NONE) ctx->last_refusal = BZK_REFUSE_OTHER;
    }
    return st;
}

extern "C" {

int32_t bzk_state_model_default(const uint8_t* model, uint64_t model_len, uint8_t out[32]) {
    if (!out) return BZK_E_ARG;
    Model M;
    BZK_TRY(load_model(model, model_len, M));
    M.nodes[M.root].dflt.to_bytes(out);
    return BZK_OK;
}

static int32_t state_entry_compress(bzk_ctx* ctx, const uint8_t* model, uint64_t model_len, const uint64_t* loc_off, const uint64_t* loc,
                           const uint8_t* values, uint64_t n, uint8_t state_hash[32], uint64_t* state_size) {
    if (!ctx || !state_hash || (n && (!loc_off || !values))) return BZK_E_ARG;
    if (n && loc_off[n] && !loc) return BZK_E_ARG;
    Model M;
    if (load_model(model, model_len, M) != BZK_OK) {
        ctx->last_error = "state_compress: not a bincode ZkStateModel (or a struct with 0 / more than 16 fields)";
        return BZK_E_ARG;
    }
    static const uint64_t zero_off[1] = {0};
    return compress_core(ctx, M, n ? loc_off : zero_off, loc, values, n, state_hash, state_size);
}

// pairs = bincode(ZkDataPairs) = HashMap<ZkDataLocator, ZkScalar>: u64 count, then per entry Vec<u64> (u64 length + items) and the
// scalar's four Montgomery limbs (src/zk/mod.rs:202-206, 425-426, 469-470).  out = bincode(ZkCompressedState) = state_hash | u64 state_size
static int32_t state_entry_compress_bincode(bzk_ctx* ctx, const uint8_t* model, uint64_t model_len, const uint8_t* pairs, uint64_t pairs_len,
                                   uint8_t compressed_out[40]) {
    if (!ctx || !pairs || !compressed_out) return BZK_E_ARG;
    Rd r{pairs, pairs_len};
    const uint64_t n = r.u(8);
    if (!r.ok || n > pairs_len / 40) return BZK_E_ARG;
    std::vector<uint64_t> off(1, 0), loc;
    std::vector<uint8_t> vals;
    vals.reserve((size_t)n * 32);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t k = r.u(8);
        if (!r.ok || k > 64) return BZK_E_ARG;
        for (uint64_t j = 0; j < k; ++j) loc.push_back(r.u(8));
        off.push_back(loc.size());
        if (!r.ok || r.off + 32 > pairs_len) return BZK_E_ARG;
        vals.insert(vals.end(), pairs + r.off, pairs + r.off + 32);
        r.off += 32;
    }
    if (!r.ok || r.off != pairs_len) return BZK_E_ARG;
    uint64_t size = 0;
    BZK_TRY(state_entry_compress(ctx, model, model_len, off.data(), loc.data(), vals.data(), n, compressed_out, &size));
    memcpy(compressed_out + 32, &size, 8);
    return BZK_OK;
}

int32_t bzk_state_create(bzk_ctx* ctx, const uint8_t* model, uint64_t model_len, bzk_state** out) {
    if (!ctx || !out) return BZK_E_ARG;
    *out = nullptr;
    std::unique_ptr<bzk_state> S(new (std::nothrow) bzk_state);
    if (!S) return BZK_E_ALLOC;
    S->ctx = ctx;
    S->device = ctx->device;
    if (load_model(model, model_len, S->M) != BZK_OK) {
        ctx->last_error = "state_create: not a bincode ZkStateModel (or a struct with 0 / more than 16 fields)";
        return BZK_E_ARG;
    }
    std::vector<ZkScalar> consts;
    S->dflt_slot.assign(S->M.nodes.size(), 0);
    S->depth_slot.assign(S->M.nodes.size(), {});
    for (size_t i = 0; i < S->M.nodes.size(); ++i) {
        const ModelNode& n = S->M.nodes[i];
        S->dflt_slot[i] = (uint32_t)consts.size();
        consts.push_back(n.dflt);
        if (n.kind == 2) {
            for (int k = 0; k <= n.log4; ++k) {
                S->depth_slot[i].push_back((uint32_t)consts.size());
                consts.push_back(n.list_dflt[k]);
            }
        }
    }
    S->n_const = (uint32_t)consts.size();
    (void)hipSetDevice(ctx->device);
    BZK_TRY(state_grow(S.get(), S->n_const));
    std::vector<uint8_t> bytes(consts.size() * 32);
    for (size_t i = 0; i < consts.size(); ++i) consts[i].to_bytes(bytes.data() + 32 * i);
    BZK_HIP(ctx, hipMemcpyAsync(S->d_vals, bytes.data(), bytes.size(), hipMemcpyHostToDevice, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    S->used = S->n_const;
    S->nonzero.assign(S->used, 0);
    S->M.nodes[S->M.root].dflt.to_bytes(S->root_hash);
    *out = S.release();
    return BZK_OK;
}

void bzk_state_free(bzk_state* st) {
    delete st;  // frees the value store (see ~bzk_state)
}

static int32_t state_entry_update(bzk_state* st, const uint64_t* loc_off, const uint64_t* loc, const uint8_t* values, uint64_t n, uint64_t target_height,
                         uint8_t state_hash[32], uint64_t* state_size, uint8_t* prev_values_out) {
    if (!st || (n && (!loc_off || !values))) return BZK_E_ARG;
    if (n && loc_off[n] && !loc) return BZK_E_ARG;
    std::lock_guard<std::mutex> g(st->m);
    static const uint64_t zero_off[1] = {0};
    BZK_TRY(state_update_impl(st, n ? loc_off : zero_off, loc, values, n, target_height, prev_values_out));
    if (state_hash) memcpy(state_hash, st->root_hash, 32);
    if (state_size) *state_size = st->size;
    return BZK_OK;
}

// delta = bincode(ZkDeltaPairs) = HashMap<ZkDataLocator, Option<ZkScalar>> (src/zk/mod.rs:473-474): u64 count; per entry Vec<u64>,
// the Option's u8 tag, the four Montgomery limbs when Some.  None writes zero, as `update_contract` does (`v.unwrap_or_default()`).
static int32_t state_entry_update_bincode(bzk_state* st, const uint8_t* delta, uint64_t delta_len, uint64_t target_height, uint8_t compressed_out[40]) {
    if (!st || !delta || !compressed_out) return BZK_E_ARG;
    Rd r{delta, delta_len};
    const uint64_t n = r.u(8);
    if (!r.ok || n > delta_len / 9) return BZK_E_ARG;
    std::vector<uint64_t> off(1, 0), loc;
    std::vector<uint8_t> vals((size_t)n * 32, 0);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t k = r.u(8);
        if (!r.ok || k > 64) return BZK_E_ARG;
        for (uint64_t j = 0; j < k; ++j) loc.push_back(r.u(8));
        off.push_back(loc.size());
        const uint64_t tag = r.u(1);
        if (!r.ok || tag > 1) return BZK_E_ARG;
        if (tag) {
            if (r.off + 32 > delta_len) return BZK_E_ARG;
            memcpy(vals.data() + 32 * i, delta + r.off, 32);
            r.off += 32;
        }
    }
    if (!r.ok || r.off != delta_len) return BZK_E_ARG;
    uint64_t size = 0;
    BZK_TRY(state_entry_update(st, off.data(), loc.data(), vals.data(), n, target_height, compressed_out, &size, nullptr));
    memcpy(compressed_out + 32, &size, 8);
    return BZK_OK;
}

int32_t bzk_state_root(bzk_state* st, uint8_t state_hash[32], uint64_t* state_size, uint64_t* height) {
    if (!st) return BZK_E_ARG;
    std::lock_guard<std::mutex> g(st->m);
    if (st->poisoned) return BZK_E_DEVICE;
    if (state_hash) memcpy(state_hash, st->root_hash, 32);
    if (state_size) *state_size = st->size;
    if (height) *height = st->height;
    return BZK_OK;
}

// `get_data` for n locators: the value at each (a scalar, or the hash of the struct / list it names), the type's default where
// nothing was ever written
static int32_t state_entry_get(bzk_state* st, const uint64_t* loc_off, const uint64_t* loc, uint64_t n, uint8_t* values_out) {
    if (!st || (n && (!loc_off || !values_out))) return BZK_E_ARG;
    if (n && loc_off[n] && !loc) return BZK_E_ARG;
    std::lock_guard<std::mutex> g(st->m);
    if (st->poisoned) return BZK_E_DEVICE;
    if (n >= 0x3fffffffull) return BZK_E_ARG;
    std::vector<uint32_t> idx(n);
    Key k;
    for (uint64_t i = 0; i < n; ++i) {
        if (loc_off[i] > loc_off[i + 1]) return BZK_E_ARG;
        const uint64_t* l = loc + loc_off[i];
        const uint64_t len = loc_off[i + 1] - loc_off[i];
        const int m = state_locate(st->M, l, len);
        if (m < 0) {
            st->ctx->last_error = "state_get: the locator names nothing in this model (ZkLocatorError::InvalidLocator)";
            st->ctx->last_refusal = BZK_REFUSE_INVALID_LOCATOR;
            return BZK_E_ARG;
        }
        k.assign(l, l + len);
        auto it = st->slot_of.find(k);
        idx[i] = it != st->slot_of.end() ? it->second : st->dflt_slot[m];
    }
    return state_read_slots(st, idx, values_out);
}

// `prove` (src/zk/state/mod.rs:218-264) for n indices of the list at `tree_loc`: per index log4_size x 3 scalars, leaf level first,
// the three siblings of each level in ascending position
static int32_t state_entry_prove(bzk_state* st, const uint64_t* tree_loc, uint64_t tree_loc_len, const uint64_t* indices, uint64_t n, uint8_t* proof_out,
                        uint32_t* log4_size) {
    if (!st || (tree_loc_len && !tree_loc) || (n && (!indices || !proof_out))) return BZK_E_ARG;
    std::lock_guard<std::mutex> g(st->m);
    if (st->poisoned) return BZK_E_DEVICE;
    const int m = state_locate(st->M, tree_loc, tree_loc_len);
    if (m < 0) {
        st->ctx->last_error = "state_prove: the locator names nothing in this model (ZkLocatorError::InvalidLocator)";
        st->ctx->last_refusal = BZK_REFUSE_INVALID_LOCATOR;
        return BZK_E_ARG;
    }
    const ModelNode& nd = st->M.nodes[m];
    if (nd.kind != 2) {
        st->ctx->last_error = "state_prove: not locating a tree (StateManagerError::NonTreeLocatorError)";
        st->ctx->last_refusal = BZK_REFUSE_NON_TREE_LOCATOR;
        return BZK_E_ARG;
    }
    if (log4_size) *log4_size = (uint32_t)nd.log4;
    if ((uint64_t)nd.log4 * 3 * n >= 0x3fffffffull) return BZK_E_ARG;
    std::vector<uint32_t> idx;
    idx.reserve((size_t)n * nd.log4 * 3);
    Key k(tree_loc, tree_loc + tree_loc_len);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t cur = indices[i];
        if (nd.log4 < 32 && cur >= ((uint64_t)1 << (2 * nd.log4))) {
            st->ctx->last_error = "state_prove: index beyond the list";
            st->ctx->last_refusal = BZK_REFUSE_INVALID_LOCATOR;
            return BZK_E_ARG;
        }
        for (int d = nd.log4; d > 0; --d) {  // siblings at depth d
            const uint64_t first = cur & ~(uint64_t)3;
            for (uint64_t j = first; j < first + 4; ++j) {
                if (j == cur) continue;
                if (d == nd.log4) {
                    k.push_back(j);
                } else {
                    k.push_back(AUX | (uint64_t)d);
                    k.push_back(j);
                }
                auto it = st->slot_of.find(k);
                idx.push_back(it != st->slot_of.end() ? it->second : st->depth_slot[m][d]);
                k.resize(tree_loc_len);
            }
            cur >>= 2;
        }
    }
    return state_read_slots(st, idx, proof_out);
}

int32_t bzk_state_stats(bzk_state* st, uint64_t* slots, uint64_t* device_bytes, uint64_t* keys) {
    if (!st) return BZK_E_ARG;
    std::lock_guard<std::mutex> g(st->m);
    if (slots) *slots = st->used;
    if (device_bytes) *device_bytes = st->cap * sizeof(Fr);
    if (keys) *keys = st->slot_of.size();
    return BZK_OK;
}

int32_t bzk_state_compress(bzk_ctx* ctx, const uint8_t* model, uint64_t model_len, const uint64_t* loc_off, const uint64_t* loc, const uint8_t* values, uint64_t n, uint8_t state_hash[32], uint64_t* state_size) {
    return refusal_scope(ctx, [&] { return state_entry_compress(ctx, model, model_len, loc_off, loc, values, n, state_hash, state_size); });
}
int32_t bzk_state_compress_bincode(bzk_ctx* ctx, const uint8_t* model, uint64_t model_len, const uint8_t* pairs, uint64_t pairs_len, uint8_t compressed_out[40]) {
    return refusal_scope(ctx, [&] { return state_entry_compress_bincode(ctx, model, model_len, pairs, pairs_len, compressed_out); });
}
int32_t bzk_state_update(bzk_state* st, const uint64_t* loc_off, const uint64_t* loc, const uint8_t* values, uint64_t n, uint64_t target_height, uint8_t state_hash[32], uint64_t* state_size, uint8_t* prev_values_out) {
    return refusal_scope((st ? st->ctx : nullptr), [&] { return state_entry_update(st, loc_off, loc, values, n, target_height, state_hash, state_size, prev_values_out); });
}
int32_t bzk_state_update_bincode(bzk_state* st, const uint8_t* delta, uint64_t delta_len, uint64_t target_height, uint8_t compressed_out[40]) {
    return refusal_scope((st ? st->ctx : nullptr), [&] { return state_entry_update_bincode(st, delta, delta_len, target_height, compressed_out); });
}
int32_t bzk_state_get(bzk_state* st, const uint64_t* loc_off, const uint64_t* loc, uint64_t n, uint8_t* values_out) {
    return refusal_scope((st ? st->ctx : nullptr), [&] { return state_entry_get(st, loc_off, loc, n, values_out); });
}
int32_t bzk_state_prove(bzk_state* st, const uint64_t* tree_loc, uint64_t tree_loc_len, const uint64_t* indices, uint64_t n, uint8_t* proof_out, uint32_t* log4_size) {
    return refusal_scope((st ? st->ctx : nullptr), [&] { return state_entry_prove(st, tree_loc, tree_loc_len, indices, n, proof_out, log4_size); });
}

}  // extern "C"
