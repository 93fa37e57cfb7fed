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
#include <numeric>

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
        if (P.nodes.size() >= 0x7ffffff0u) { P.err = "too many nodes"; return 0; }
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
            if (hi - lo != 1) { P.err = "duplicate locator"; return 0; }
            if (len(lo) != d) { P.err = "locator points below a scalar (ZkLocatorError::InvalidLocator)"; return 0; }
            return 0x40000000u | (uint32_t)order[lo];  // pair value, resolved to consts.size() + pair index
        }
        for (uint64_t i = lo; i < hi; ++i)
            if (len(i) <= d) { P.err = "locator does not reach a scalar (StateManagerError::NonScalarLocatorError)"; return 0; }
        if (n.kind == 1) {
            std::vector<uint32_t> in(n.fields.size());
            uint64_t i = lo;
            for (size_t f = 0; f < n.fields.size(); ++f) {
                uint64_t j = i;
                while (j < hi && at(j, d) == f) ++j;
                in[f] = j > i ? build(n.fields[f], d + 1, i, j) : cst(n.fields[f], -1, M.nodes[n.fields[f]].dflt);
                i = j;
            }
            if (i != hi) { P.err = "struct field index out of range (the reference indexes field_types out of bounds)"; return 0; }
            return hashed(in);
        }
        // List: items first, then the sparse 4-ary tree over them, level by level
        const uint64_t size = n.log4 >= 32 ? ~0ull : ((uint64_t)1 << (2 * n.log4));
        std::vector<std::pair<uint64_t, uint32_t>> cur;  // (index at the current depth, value id), ascending
        for (uint64_t i = lo; i < hi;) {
            const uint64_t idx = at(i, d);
            if (idx >= size) { P.err = "list index out of range (ZkLocatorError::InvalidLocator)"; return 0; }
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
        if (!g.equals(f)) { ctx->last_error = "state_compress: a value is not a canonical field element"; return BZK_E_ARG; }
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

extern "C" {

int32_t bzk_state_model_default(const uint8_t* model, uint64_t model_len, uint8_t out[32]) {
    if (!out) return BZK_E_ARG;
    Model M;
    BZK_TRY(load_model(model, model_len, M));
    M.nodes[M.root].dflt.to_bytes(out);
    return BZK_OK;
}

int32_t bzk_state_compress(bzk_ctx* ctx, const uint8_t* model, uint64_t model_len, const uint64_t* loc_off, const uint64_t* loc,
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
int32_t bzk_state_compress_bincode(bzk_ctx* ctx, const uint8_t* model, uint64_t model_len, const uint8_t* pairs, uint64_t pairs_len,
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
    BZK_TRY(bzk_state_compress(ctx, model, model_len, off.data(), loc.data(), vals.data(), n, compressed_out, &size));
    memcpy(compressed_out + 32, &size, 8);
    return BZK_OK;
}

}  // extern "C"
