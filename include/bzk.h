/* bzk.h - C ABI of the MI355X-native Groth16 hot path for Bazuka's MPN rollup (libbzk.so).
 *
 * This is the drop-in boundary a Rust host binds with `extern "C"`.  Each entry
 * point names the reference interface it replaces; paths are relative to ziesha-network/bazuka.
 *
 * Conventions
 *  - plain pointers and sizes only; the caller owns every buffer; nothing allocated by the library
 *    crosses the boundary except opaque handles (bzk_ctx, bzk_params) released by their destroy call;
 *  - every function returns an int32 status: 0 = BZK_OK, negative = BZK_E_*; no exceptions, no abort;
 *  - one ctx is bound to one GPU and one HIP stream and may be used from one thread at a time;
 *    distinct ctxs are independent (one process per GPU is the intended deployment);
 *  - scalars are 32-byte little-endian limbs.  Default = MONTGOMERY form, i.e. the in-memory
 *    `ZkScalar` / `bls12_381::Scalar` (src/zk/mod.rs:202-206; transmute-compatible per
 *    src/zk/groth16/mod.rs:7-17).  BZK_F_CANONICAL selects canonical integers (test drivers);
 *  - Fp elements are 48-byte little-endian Montgomery limbs (R = 2^384), as `Fp([u64;6])` in
 *    src/zk/groth16/mod.rs:19-20.  G1 affine = x|y (96 B, "raw", never the identity) or x|y|inf
 *    (97 B, "packed", as `(Fp, Fp, bool)` at :33-38).  G2 = x.c0|x.c1|y.c0|y.c1 (192 B raw) or +inf
 *    (193 B packed).  A Groth16 proof is a|b|c = 97+193+97 = 387 B (= bincode of `Groth16Proof`);
 *  - `*_dev` variants take DEVICE pointers (HBM-resident data; what bench.py times).  The plain
 *    variants take HOST pointers and stage through PCIe.
 *  - There is NO CPU fallback: without a usable gfx950 device bzk_ctx_create fails with
 *    BZK_E_DEVICE and nothing else can be called.
 */
#ifndef BZK_H
#define BZK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BZK_OK 0
#define BZK_E_ARG (-1)      /* bad argument (null pointer, size out of range, arity > 16 ...) */
#define BZK_E_ALLOC (-2)    /* device or host allocation failed */
#define BZK_E_DEVICE (-3)   /* HIP runtime / kernel launch error, or no gfx950 device */
#define BZK_E_UNSAT (-4)    /* witness does not satisfy the constraint system */
#define BZK_E_INTERNAL (-5) /* invariant violated (bug) */

#define BZK_F_CANONICAL 1u  /* scalars are canonical integers instead of Montgomery limbs */
#define BZK_F_DEDUP 2u      /* MSM entries: the scalar vector repeats itself (a Groth16 witness): bases of equal scalars
                             * are summed once before the bucket phase, zero scalars dropped.  Same result. */
#define BZK_F_THROUGHPUT 4u /* MSM entries: this call overlaps other device work (the five MSMs of a proof): prefer the
                             * forms that do less arithmetic over those with the shortest dependency chain (two-level
                             * bucket reduction).  Same result. */

typedef struct bzk_ctx bzk_ctx;

/* ---- context ------------------------------------------------------------------------------- */
/* device_id: HIP ordinal.  stream: a hipStream_t the caller owns (e.g. torch's current stream), or
 * NULL to let the ctx create its own non-blocking stream. */
int32_t bzk_ctx_create(int32_t device_id, void* stream, bzk_ctx** out);
void bzk_ctx_destroy(bzk_ctx* ctx);
int32_t bzk_sync(bzk_ctx* ctx); /* wait for the ctx stream */
const char* bzk_strerror(int32_t status);
const char* bzk_last_error(bzk_ctx* ctx); /* detail of the last BZK_E_DEVICE on this ctx */
/* Which of the reference's errors the last BZK_E_ARG of a bzk_state_* call on this ctx stands for, so that a Rust host can return
 * `StateManagerError::{LocatorError(InvalidLocator), NonScalarLocatorError, NonTreeLocatorError}` (src/zk/state/mod.rs:12-27,
 * src/zk/mod.rs:348-351) instead of parsing bzk_last_error; 0 after a state call that was not refused.  OTHER = malformed input the
 * reference cannot express (bad bincode, locator offsets out of order, counts beyond the ABI's limits). */
#define BZK_REFUSE_NONE 0
#define BZK_REFUSE_INVALID_LOCATOR 1     /* names nothing in the model, index beyond a list / struct, points below a scalar */
#define BZK_REFUSE_NON_SCALAR_LOCATOR 2  /* ends above a scalar where a scalar is written */
#define BZK_REFUSE_NON_TREE_LOCATOR 3    /* `prove` on something that is not a list */
#define BZK_REFUSE_DUPLICATE_LOCATOR 4   /* the same locator twice in one delta (a HashMap cannot hold that) */
#define BZK_REFUSE_NON_CANONICAL_VALUE 5 /* 32 bytes that are not the Montgomery limbs of a residue (a ZkScalar cannot hold that) */
#define BZK_REFUSE_OTHER 6
int32_t bzk_last_refusal(bzk_ctx* ctx);
uint32_t bzk_abi_version(void);

/* device-memory plumbing for hosts that want to keep inputs resident without a tensor library */
int32_t bzk_dev_alloc(bzk_ctx* ctx, uint64_t bytes, void** dptr);
int32_t bzk_dev_free(bzk_ctx* ctx, void* dptr);
/* The call workspace of a ctx is grow-only (one slab, re-used by every call: no allocation on the hot path).  After a one-off large
 * call - a 2^26-point MSM leaves ~24 GB behind - bzk_ctx_trim waits for the stream and hands the slab back to the device; the next
 * call allocates what it needs.  *released (may be NULL) = bytes freed.  (A context whose stand-alone G1 MSM calls run as two window ranges in
 * flight - 2^18 <= n < 2^20 points, DESIGN 3.2d - owns a child context with a workspace of its own and a buffer for converted raw bases: both are
 * released as well.) */
int32_t bzk_ctx_trim(bzk_ctx* ctx, uint64_t* released);
int32_t bzk_h2d(bzk_ctx* ctx, void* dst_dev, const void* src_host, uint64_t bytes);
int32_t bzk_d2h(bzk_ctx* ctx, void* dst_host, const void* src_dev, uint64_t bytes);

/* per-kernel timing with HIP events on the ctx stream (used by bench.py for `roofline.achieved`) */
int32_t bzk_prof_enable(bzk_ctx* ctx, int32_t on);
/* restrict the event pairs to launches whose label contains `substr` (NULL: every launch).  Timing every launch costs two event
 * creations per kernel (~0.25 ms per 2^20-point MSM with its ~25 launches); bench.py times only the dominant kernel inside its
 * timed region */
int32_t bzk_prof_filter(bzk_ctx* ctx, const char* substr);
int32_t bzk_prof_reset(bzk_ctx* ctx);
/* sums over launches whose kernel label == name; synchronises the stream */
int32_t bzk_prof_query(bzk_ctx* ctx, const char* name, uint64_t* launches, double* total_ms);
/* writes up to cap bytes of "name launches total_ms\n" lines */
int32_t bzk_prof_dump(bzk_ctx* ctx, char* buf, uint64_t cap);

/* ---- K1: batched Poseidon --------------------------------------------------------------------
 * Replaces `ZkHasher::hash` = `PoseidonHasher::hash` -> `poseidon::poseidon`
 * (src/zk/mod.rs:152-155, 496-511; src/zk/poseidon/mod.rs:24-84) for n independent inputs.
 * in: n*arity scalars (hash i uses in[i*arity .. (i+1)*arity)), out: n scalars.  1 <= arity <= 16
 * (MAX_ARITY, src/zk/poseidon/params/mod.rs:25); other arities -> BZK_E_ARG where the reference
 * panics.  Always Montgomery form. */
int32_t bzk_poseidon_batch(bzk_ctx* ctx, const uint8_t* in, uint32_t arity, uint64_t n, uint8_t* out);
int32_t bzk_poseidon_batch_dev(bzk_ctx* ctx, const void* in_dev, uint32_t arity, uint64_t n, void* out_dev);

/* ---- K2: dense 4-ary ZkState tree re-hash ----------------------------------------------------
 * Root of `ZkStateModel::List{log4_size, Scalar}` with every leaf present, as
 * `ZkStateBuilder::compress` / `KvStoreStateManager::root` would give (src/zk/state/mod.rs:66-90,
 * 274-283, 310-420: node = H(c0,c1,c2,c3)).  leaves: 4^log4 scalars.  nodes_opt (may be NULL):
 * receives all internal nodes in the reference's heap order, index (4^k-1)/3 + i for node i of
 * depth k (src/zk/state/mod.rs:355,382-383), root at index 0; (4^log4-1)/3 scalars. */
int32_t bzk_merkle4_root(bzk_ctx* ctx, const uint8_t* leaves, uint32_t log4_size, uint8_t root[32], uint8_t* nodes_opt);
int32_t bzk_merkle4_root_dev(bzk_ctx* ctx, const void* leaves_dev, uint32_t log4_size, uint8_t root[32], void* nodes_opt_dev);

/* Device-resident tree with batched updates and proofs: the level loop of
 * `KvStoreStateManager::set_data` (src/zk/state/mod.rs:310-420) and `prove` (218-264) for a dense
 * `List{log4_size, Scalar}` kept in HBM: (4^(log4+1)-1)/3 scalars, all levels in heap order (node i of depth k at
 * (4^k-1)/3 + i, leaves at depth log4).  log4 <= 15 (4^15 leaves = 46 GB: an account tree of the production size fits).
 *   create : from 4^log4 leaves in device memory, or (leaves_dev NULL) the empty tree of `default_leaf`
 *   update : n (index, leaf) pairs from host memory; re-hashes each affected parent once per level; a later entry
 *            for the same index wins
 *   prove  : for each of n indices log4 sibling triples, leaf level first, index order, self left out
 *            (= the `Vec<[ZkScalar; 3]>` the circuits take); out = n * log4 * 96 bytes */
typedef struct bzk_tree4 bzk_tree4;
int32_t bzk_tree4_create(bzk_ctx* ctx, uint32_t log4_size, const void* leaves_dev, const uint8_t default_leaf[32], bzk_tree4** out);
void    bzk_tree4_free(bzk_ctx* ctx, bzk_tree4* tree);
int32_t bzk_tree4_root(bzk_ctx* ctx, const bzk_tree4* tree, uint8_t root[32]);
int32_t bzk_tree4_update(bzk_ctx* ctx, bzk_tree4* tree, const uint64_t* indices, const uint8_t* leaves, uint64_t n);
int32_t bzk_tree4_prove(bzk_ctx* ctx, const bzk_tree4* tree, const uint64_t* indices, uint64_t n, uint8_t* out);
int32_t bzk_tree4_node(bzk_ctx* ctx, const bzk_tree4* tree, uint32_t depth, uint64_t index, uint8_t out[32]);

/* Device-resident MPN ACCOUNT STATE (second half): the production state model
 *   List{log4_tree, Struct{tx_nonce, withdraw_nonce, pub_x, pub_y, List{log4_token_tree, Struct{token_id, balance}}}}
 * (`MpnConfig::state_model`, src/mpn/mod.rs:218-241) with the batched forms of `KvStoreStateManager::set_mpn_account`
 * (src/zk/state/mod.rs:158-208), `get_mpn_account` (93-137) and `prove` (218-264).  The account level is a dense tree of leaf
 * hashes H5(nonce, wnonce, x, y, tokens_root) in HBM (46 GB at log4_tree = 15); account contents and token sub-trees exist for
 * populated accounts only (`capacity` of them; BZK_E_ALLOC beyond).  All scalars: 32-byte Montgomery.
 *   set_accounts : n DISTINCT accounts; cells = n x 4 scalars (nonce, withdraw nonce, x, y); account a's token updates are
 *                  entries tok_off[a] .. tok_off[a+1] of tok_index (slot < 4^log4_token_tree, distinct per account) and
 *                  tok_vals (token_id, balance); slots not named keep their contents, as in the reference
 *   get_accounts : per account 5 + 2 * 4^T scalars: the 4 cells, tokens_root (`MpnAccount::tokens_hash`), then every token slot
 *   prove        : log4_tree sibling triples per account (the transitions' `proof` / `src_proof` / `dst_proof`)
 *   prove_token  : log4_token_tree sibling triples per (account, token slot) (the `*_balance_proof`s) */
/* `ZkStateModel::compress` of a DENSE MPN-shaped state resident in HBM (BASELINE configs[4], secondary instance): cells_dev = 4^L x 4
 * scalars, tokens_dev = 4^L x 4^T x 2 scalars; one dense launch per level (H2 token slots, T x H4, H5 per account, L x H4). L <= 12. */
int32_t bzk_mpn_state_compress_dev(bzk_ctx* ctx, uint32_t log4_tree, uint32_t log4_token_tree, const void* cells_dev, const void* tokens_dev,
                                   uint8_t root[32]);
typedef struct bzk_mpn_tree bzk_mpn_tree;
int32_t bzk_mpn_tree_create(bzk_ctx* ctx, uint32_t log4_tree, uint32_t log4_token_tree, uint64_t capacity, bzk_mpn_tree** out);
void    bzk_mpn_tree_free(bzk_ctx* ctx, bzk_mpn_tree* tree);
int32_t bzk_mpn_tree_root(bzk_ctx* ctx, const bzk_mpn_tree* tree, uint8_t root[32]);
uint64_t bzk_mpn_tree_accounts(const bzk_mpn_tree* tree);
int32_t bzk_mpn_tree_set_accounts(bzk_ctx* ctx, bzk_mpn_tree* tree, const uint64_t* indices, const uint8_t* cells, const uint64_t* tok_off,
                                  const uint64_t* tok_index, const uint8_t* tok_vals, uint64_t n);
int32_t bzk_mpn_tree_get_accounts(bzk_ctx* ctx, const bzk_mpn_tree* tree, const uint64_t* indices, uint64_t n, uint8_t* out);
int32_t bzk_mpn_tree_prove(bzk_ctx* ctx, const bzk_mpn_tree* tree, const uint64_t* indices, uint64_t n, uint8_t* out);
int32_t bzk_mpn_tree_prove_token(bzk_ctx* ctx, const bzk_mpn_tree* tree, const uint64_t* account_indices, const uint64_t* token_indices,
                                 uint64_t n, uint8_t* out);

/* General seam `ZkStateModel::compress::<H>(&data)` (src/zk/mod.rs:392-399; `ZkStateBuilder::{batch_set, compress}`
 * src/zk/state/mod.rs:66-90 over `set_data` 310-420): ANY nesting of Scalar / Struct{field_types} / List{log4_size, item_type}
 * (src/zk/mod.rs:332-345) over SPARSE (ZkDataLocator, ZkScalar) pairs, untouched sub-trees at `compress_default` (401-423).  The
 * touched part of the state is laid out on the host, grouped by (height, arity), and every group is one batched Poseidon launch.
 *   model          bincode(ZkStateModel) (u32 tags 0 Scalar / 1 Struct + Vec / 2 List + u8 + Box)
 *   loc_off, loc   CSR of the n locators: pair i's indices are loc[loc_off[i] .. loc_off[i + 1])
 *   values         n x 32-byte Montgomery scalars
 *   state_hash, state_size = the two fields of `ZkCompressedState` (size = non-zero scalars, as set_data counts them)
 * BZK_E_ARG where the reference returns LocatorError / NonScalarLocatorError (or panics): a locator that does not end at a
 * scalar, an index >= 4^log4_size or >= the number of fields, a repeated locator, a struct with 0 or > 16 fields (`is_valid`).
 * _bincode: pairs = bincode(ZkDataPairs), out = bincode(ZkCompressedState) (40 B) - what a Rust host holds already.
 * bzk_state_model_default = `compress_default` (host). */
int32_t bzk_state_compress(bzk_ctx* ctx, const uint8_t* model, uint64_t model_len, const uint64_t* loc_off, const uint64_t* loc,
                           const uint8_t* values, uint64_t n, uint8_t state_hash[32], uint64_t* state_size);
int32_t bzk_state_compress_bincode(bzk_ctx* ctx, const uint8_t* model, uint64_t model_len, const uint8_t* pairs, uint64_t pairs_len,
                                   uint8_t compressed_out[40]);
int32_t bzk_state_model_default(const uint8_t* model, uint64_t model_len, uint8_t out[32]);

/* PERSISTENT device state of one contract, any model: `KvStoreStateManager::{update_contract, set_data, get_data, prove, root}`
 * (src/zk/state/mod.rs:286-308, 310-420, 422-438, 218-264, 274-284) with the values resident in HBM.  The reference walks
 * root-wards once per written scalar, one hash and four KV reads per level; here one update is ONE plan - the union of the
 * root-ward paths of every written scalar, each node once, grouped by (height, arity): per group one gather, one batched
 * Poseidon launch, one scatter, in place on the device.  Only the written scalars go up, only the root comes back.
 *   bzk_state_update   = `update_contract(db, id, &ZkDeltaPairs, target_height)`: n (locator, value) pairs, a zero value = the
 *                        reference's `None` (remove).  All or nothing: every locator is checked before a slot changes.
 *                        state_hash / state_size = the new `ZkCompressedState` (either may be NULL).
 *                        prev_values_out (n x 32 B, or NULL) = what the named scalars held before, zero where nothing was stored:
 *                        the rollback delta `ZkState::push_delta` records (src/zk/mod.rs:521-530); feeding it back undoes the update.
 *   _bincode           delta = bincode(ZkDeltaPairs) (HashMap<ZkDataLocator, Option<ZkScalar>>), out = bincode(ZkCompressedState)
 *   bzk_state_root     = `root` (+ `height_of`)
 *   bzk_state_get      = `get_data` for n locators: scalar, or the value of the struct / list named; defaults where untouched
 *   bzk_state_prove    = `prove(db, id, tree_loc, index)` for n indices of ONE list: per index log4_size x 3 scalars, leaf level
 *                        first, the three siblings of a level in ascending position (`Vec<[ZkScalar; 3]>`)
 * Errors as bzk_state_compress (BZK_E_ARG + bzk_last_error(ctx) naming the reference's error); BZK_E_DEVICE after a device
 * failure in the middle of an update (the handle is then unusable).  A handle is bound to its ctx; one thread at a time. */
typedef struct bzk_state bzk_state;
int32_t bzk_state_create(bzk_ctx* ctx, const uint8_t* model, uint64_t model_len, bzk_state** out);
void bzk_state_free(bzk_state* st);
int32_t bzk_state_update(bzk_state* st, const uint64_t* loc_off, const uint64_t* loc, const uint8_t* values, uint64_t n,
                         uint64_t target_height, uint8_t state_hash[32], uint64_t* state_size, uint8_t* prev_values_out);
int32_t bzk_state_update_bincode(bzk_state* st, const uint8_t* delta, uint64_t delta_len, uint64_t target_height,
                                 uint8_t compressed_out[40]);
int32_t bzk_state_root(bzk_state* st, uint8_t state_hash[32], uint64_t* state_size, uint64_t* height);
int32_t bzk_state_get(bzk_state* st, const uint64_t* loc_off, const uint64_t* loc, uint64_t n, uint8_t* values_out);
int32_t bzk_state_prove(bzk_state* st, const uint64_t* tree_loc, uint64_t tree_loc_len, const uint64_t* indices, uint64_t n,
                        uint8_t* proof_out, uint32_t* log4_size);
int32_t bzk_state_stats(bzk_state* st, uint64_t* slots, uint64_t* device_bytes, uint64_t* keys);

/* ---- K3: radix-2 NTT over Fr -----------------------------------------------------------------
 * bellman 0.14 `EvaluationDomain::{fft, ifft, coset_fft, icoset_fft}` (third-party crate; reached
 * from `create_random_proof`, src/mpn/circuits/test.rs:135,175,215).  In place, natural order in and
 * out.  omega = 7^((r-1)/2^32)^(2^(32-log_n)); coset shift g = 7.
 *   inverse=0,coset=0: a_k = sum_j a_j w^jk          inverse=1,coset=0: ... w^-jk / n
 *   inverse=0,coset=1: scale a_j by g^j, then fft     inverse=1,coset=1: ifft, then scale by g^-j */
int32_t bzk_ntt(bzk_ctx* ctx, uint8_t* data, uint32_t log_n, int32_t inverse, int32_t coset);
int32_t bzk_ntt_dev(bzk_ctx* ctx, void* data_dev, uint32_t log_n, int32_t inverse, int32_t coset);

/* ---- K4 / K5: Pippenger multi-scalar multiplication ------------------------------------------
 * bellman 0.14 `multiexp` over G1 / G2 (third-party; the `h`, `l`, `a`, `b_g1`, `b_g2` queries of
 * `create_proof`).  result = sum_i scalar_i * base_i, written packed (97 / 193 B).
 * bases: raw affine, never the identity (bellman drops identity points from the CRS). */
int32_t bzk_msm_g1(bzk_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, uint64_t n, uint32_t flags, uint8_t out[97]);
int32_t bzk_msm_g1_dev(bzk_ctx* ctx, const void* bases_dev, const void* scalars_dev, uint64_t n, uint32_t flags, uint8_t out[97]);
int32_t bzk_msm_g2(bzk_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, uint64_t n, uint32_t flags, uint8_t out[193]);
int32_t bzk_msm_g2_dev(bzk_ctx* ctx, const void* bases_dev, const void* scalars_dev, uint64_t n, uint32_t flags, uint8_t out[193]);

/* Sharded form for multi-GPU: this rank computes only windows
 * [w_begin, w_end) of the W = bzk_msm_window_count(n) signed c-bit windows and returns their partial
 * sum  sum_w 2^(c*w) * S_w  as a packed point; the partial sums of all ranks add up to the MSM.
 * Ranks exchange the 97/193-byte partials as raw bytes (RCCL all-gather) and fold them with
 * bzk_g1_sum / bzk_g2_sum - RCCL itself cannot add curve points. */
uint32_t bzk_msm_window_count(uint64_t n);
int32_t bzk_msm_g1_windows_dev(bzk_ctx* ctx, const void* bases_dev, const void* scalars_dev, uint64_t n, uint32_t flags,
                               uint32_t w_begin, uint32_t w_end, uint8_t out[97]);
int32_t bzk_msm_g2_windows_dev(bzk_ctx* ctx, const void* bases_dev, const void* scalars_dev, uint64_t n, uint32_t flags,
                               uint32_t w_begin, uint32_t w_end, uint8_t out[193]);
int32_t bzk_g1_sum(const uint8_t* packed_points, uint32_t count, uint8_t out[97]);   /* host, tiny */
int32_t bzk_g2_sum(const uint8_t* packed_points, uint32_t count, uint8_t out[193]);

/* ---- Groth16 prove (a4-a7) --------------------------------------------------------------------
 * bellman 0.14 `groth16::create_proof(circuit, params, r, s)` after synthesis (third-party; call
 * sites src/mpn/circuits/test.rs:135,175,215): 3 iNTT + 3 coset NTT + pointwise (a*b-c)/Z + 1 inverse
 * coset NTT -> h; MSMs over the h / l / a / b_g1 / b_g2 queries; assembly of (A, B, C).
 * Output = the 387 bytes of `Groth16Proof` (src/zk/groth16/mod.rs:33-38), i.e. the payload of
 * `ZkProof::Groth16` (src/zk/mod.rs:646-651) that `groth16_verify` (src/zk/groth16/mod.rs:67-121)
 * accepts.
 *
 * CRS layout (= bellman `Parameters`, identity points dropped): variables are numbered inputs first
 * (index 0 = ONE) then aux.  a / b_g1 / b_g2 hold ONLY the variables whose density flag is set, in
 * variable order (a_density[v] = v occurs in some A row; b_density likewise). */
typedef struct bzk_params bzk_params;
typedef struct {
    uint32_t n_in, n_aux, log_m; /* m = 2^log_m >= number of constraints */
    uint32_t n_a, n_b;           /* popcount of a_density / b_density */
    const uint8_t* vk;           /* 870 B: alpha_g1|beta_g1|beta_g2|gamma_g2|delta_g1|delta_g2, packed */
    const uint8_t* h;            /* (m-1) raw G1 */
    const uint8_t* l;            /* n_aux raw G1 */
    const uint8_t* a;            /* n_a raw G1 */
    const uint8_t* b_g1;         /* n_b raw G1 */
    const uint8_t* b_g2;         /* n_b raw G2 */
    const uint8_t* a_density;    /* n_in + n_aux bytes, 0/1 */
    const uint8_t* b_density;    /* n_in + n_aux bytes, 0/1 */
} bzk_params_desc;
typedef struct {
    const uint8_t* z;            /* (n_in + n_aux) scalars: inputs (z[0] = 1) then aux, Montgomery */
    const uint8_t* az;           /* n_rows scalars: <A_k, z> per constraint */
    const uint8_t* bz;
    const uint8_t* cz;
    uint64_t n_rows;             /* number of constraints incl. bellman's n_in trailing `input*0=0` rows */
    uint64_t n_vars;             /* scalars behind `z`: must equal the params' n_in + n_aux, else BZK_E_ARG (an assignment
                                    of another circuit shape is refused, never read out of bounds) */
} bzk_assignment;
/* uploads the CRS to HBM once (host pointers in the descriptor are not retained) */
int32_t bzk_params_load(bzk_ctx* ctx, const bzk_params_desc* desc, bzk_params** out);
void bzk_params_free(bzk_ctx* ctx, bzk_params* params);
/* A further prover SLOT over the same device-resident CRS: `bzk_groth16_prove` is not re-entrant per params handle, which owns its
 * per-proof device scratch,, so a prover that keeps several proofs in flight on one GPU (their latency-bound phases hide under
 * each other's accumulation) uses one ctx + one slot per thread.  Slots SHARE the uploaded CRS, its resident internal forms and the
 * h table (reference-counted; the last bzk_params_free releases them): n slots cost n x scratch, not n x CRS. */
int32_t bzk_params_slot(bzk_ctx* ctx, const bzk_params* params, bzk_params** out);
/* The first proof over a CRS builds, ONCE per device and only where hipMemGetInfo shows room for them beside a workspace reserve,
 * the resident internal forms of l / a / b_g1 / b_g2 (no base conversion inside a proof) and, for 2^16 .. 2^24 domains, a static
 * table of the h query (13 levels: 1.5 GB at 2^20, 24 GB at the production 2^24 - one per DEVICE, shared by the slots).  bzk_params_h_table(.., 1) builds that table now, for any domain size that
 * fits; (.., 0) drops it / keeps the first proof from building it (only while this handle is the CRS's sole slot).  Environment:
 * BZK_PROVE_H_TABLE=0, BZK_PROVE_H_TABLE_MAX_LOG, BZK_PROVE_RESIDENT_BASES=0. */
int32_t bzk_params_h_table(bzk_ctx* ctx, bzk_params* params, int32_t on);
/* r, s: Montgomery scalars (the prover's blinding factors; bellman draws them from the rng) */
int32_t bzk_groth16_prove(bzk_ctx* ctx, bzk_params* params, const bzk_assignment* asg, const uint8_t r[32],
                          const uint8_t s[32], uint8_t proof_out[387]);
/* bellman's `groth16::Parameters<Bls12>` file format (third-party crate; the proving keys a real network ships and its
 * external prover loads, README.md:26-28; dev networks generate theirs in memory, src/config/blockchain.rs:355-417):
 *   Parameters::write = VerifyingKey::write | u32-BE n, h.. | u32-BE n, l.. | u32-BE n, a.. | u32-BE n, b_g1.. | u32-BE n, b_g2..
 * with `to_uncompressed` points (48-byte big-endian canonical coordinates, G2 as c1 | c0, flag bits in byte 0).  Host code.
 *   info   : lengths of ic, h, l, a, b_g1, b_g2 and the bytes consumed
 *   decode : -> vk870 (packed alpha_g1 | beta_g1 | beta_g2 | gamma_g2 | delta_g1 | delta_g2), ic (packed 97 B each), raw
 *            Montgomery h / l / a / b_g1 (96 B) and b_g2 (192 B); canonical-range and on-curve checks, infinity refused in
 *            the queries as `Parameters::read` does; threads 0 = all cores
 *   encode : the inverse; out NULL = size query
 *   bzk_params_load_bellman : parse + upload for a prover.  The file does not say which variables the a / b queries belong
 *            to (bellman derives that from the circuit): pass the density maps of the circuit shape (bzk_r1cs_data views 4, 5
 *            of the empty circuit); every length is checked against them.  vk_out (optional): bincode(Groth16VerifyingKey). */
int32_t bzk_bellman_params_info(const uint8_t* bytes, uint64_t len, uint64_t info[7]);
int32_t bzk_bellman_params_decode(const uint8_t* bytes, uint64_t len, uint8_t* vk870, uint8_t* ic, uint8_t* h, uint8_t* l, uint8_t* a,
                                  uint8_t* b_g1, uint8_t* b_g2, int32_t threads);
int32_t bzk_bellman_params_encode(const uint8_t vk870[870], const uint8_t* ic, uint64_t n_ic, const uint8_t* h, uint64_t n_h, const uint8_t* l,
                                  uint64_t n_l, const uint8_t* a, uint64_t n_a, const uint8_t* b_g1, const uint8_t* b_g2, uint64_t n_b,
                                  uint8_t* out, uint64_t cap, uint64_t* size_out);
int32_t bzk_params_load_bellman(bzk_ctx* ctx, const uint8_t* bytes, uint64_t len, uint32_t n_in, uint32_t n_aux, const uint8_t* a_density,
                                const uint8_t* b_density, bzk_params** out, uint8_t* vk_out, uint64_t vk_cap);
/* `groth16_verify` (src/zk/groth16/mod.rs:67-121; bellman `verify_proof`): host code, no GPU.  vk = bincode(Groth16VerifyingKey)
 * (870 + 8 + 97 n bytes, n = n_inputs + 1); inputs = n_inputs Montgomery scalars (the reference passes commitment, height, state,
 * aux_data, next_state).  Returns 1 (verifies), 0 (does not - incl. malformed or off-curve points, for which the reference returns
 * false), negative on bad arguments.  ~20 ms on one core. */
int32_t bzk_groth16_verify(const uint8_t* vk, uint64_t vk_len, const uint8_t* inputs, uint32_t n_inputs, const uint8_t proof[387]);
/* reads a CRS component back (tests): which = 0 vk (870 B), 1 h, 2 l, 3 a, 4 b_g1, 5 b_g2 */
int32_t bzk_params_read(bzk_ctx* ctx, const bzk_params* params, int32_t which, uint8_t* out, uint64_t cap, uint64_t* size_out);

/* CRS generation on the GPU = bellman `generate_parameters(circuit, g1, g2, alpha, beta, gamma, delta, tau)` with
 * g1, g2 the standard generators (the reference's dev-mode CRS: src/config/blockchain.rs:355-417).  The R1CS comes
 * as CSR matrices over flat variable indices (inputs then aux), rows already including bellman's trailing
 * `input_i * 0 = 0` constraints (bzk_r1cs_data views 6-14 of a circuit synthesized with record_matrices).
 * toxic = tau | alpha | beta | gamma | delta (5 x 32 B Montgomery).  vk_out (optional) receives the bincode form of
 * `Groth16VerifyingKey` (src/zk/groth16/mod.rs:22-31): 870 + 8 + 97 * n_in bytes. */
typedef struct {
    uint64_t n_rows;
    const uint32_t* row_ptr; /* n_rows + 1 */
    const uint32_t* col;     /* nnz */
    const uint8_t* val;      /* nnz * 32, Montgomery */
} bzk_csr;
int32_t bzk_groth16_setup(bzk_ctx* ctx, const bzk_csr* A, const bzk_csr* B, const bzk_csr* C, uint32_t n_in, uint32_t n_aux,
                          const uint8_t toxic[160], bzk_params** out_params, uint8_t* vk_out, uint64_t vk_cap);

/* h-polynomial stage alone (7 NTTs + pointwise), device resident: a,b,c hold az,bz,cz zero-padded to
 * m scalars on entry; on return a[0 .. m-1) holds the h coefficients.  Exposed for parity tests. */
int32_t bzk_groth16_h_dev(bzk_ctx* ctx, void* a_dev, void* b_dev, void* c_dev, uint32_t log_m);

/* ---- a3 / a9: host-side MPN witness + R1CS generator (CPU, C++) --------------------------------
 * The part of the prover that bellman runs through `Circuit::synthesize`: restates
 * `impl Circuit for UpdateCircuit` (src/mpn/circuits/update_circuit.rs:49-494), the gadgets under
 * src/zk/groth16/gadgets/, the witness builder `update::update` (src/mpn/update.rs:8-299) over a
 * RAM-resident sparse account tree (src/zk/state/mod.rs semantics, MpnConfig::state_model
 * src/mpn/mod.rs:218-241) and the wallet's `create_mpn_transaction` (src/wallet/tx_builder.rs:287-306).
 * No GPU involved; the result feeds bzk_groth16_prove. */
typedef struct bzk_mpn bzk_mpn;   /* an MPN state: accounts, keys, mempool */
typedef struct bzk_r1cs bzk_r1cs; /* a synthesized circuit instance: assignment (+ matrices) */
int32_t bzk_mpn_create(uint32_t log4_tree, uint32_t log4_token_tree, bzk_mpn** out);
void bzk_mpn_destroy(bzk_mpn* w);
int32_t bzk_mpn_set_height(bzk_mpn* w, uint64_t height);
int32_t bzk_mpn_set_threads(bzk_mpn* w, int32_t n); /* worker threads of the witness generator (default: bzk_host_default_threads) */
/* the host generator's default thread count: the visible CPUs capped by the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota);
 * BZK_HOST_THREADS=n overrides.  Read once per process. */
int32_t bzk_host_default_threads(void);
/* SURVEY 8f-3 ("replacing the per-tx KV walk in prepare_works", src/mpn/mod.rs:353-414): with a context set, the builders below and
 * bzk_mpn_make_work decide a batch from account DATA first and then re-hash its Merkle paths LEVEL BY LEVEL on the device - one
 * batched Poseidon launch per tree level over all transactions of the batch (~L + T + 2 launches instead of ~50 sequential host
 * hashes per transaction); the sibling inputs of those hashes are the transitions' Merkle proofs.  Transactions may share accounts.
 * Same transitions / MpnWork bytes as the host path.  ctx NULL: back to the host path.  The ctx must outlive its use here. */
int32_t bzk_mpn_set_device(bzk_mpn* w, bzk_ctx* ctx);
/* account `index` := keys from `JubJub::generate_keys(seed)` (src/crypto/jubjub/mod.rs:112-124), token
 * slot 0 = (token_id, balance); pub_xy_out (optional) receives address x|y */
int32_t bzk_mpn_add_account(bzk_mpn* w, uint64_t index, const uint8_t* seed, uint32_t seed_len, const uint8_t token_id[32],
                            uint64_t balance, uint8_t pub_xy_out[64]);
int32_t bzk_mpn_add_key(bzk_mpn* w, uint64_t index, const uint8_t* seed, uint32_t seed_len); /* key for a future account */
int32_t bzk_mpn_root(bzk_mpn* w, uint8_t root[32]);
/* queue a signed MpnTransaction src -> dst (nonce = sender nonce + 1 + already queued from that sender) */
int32_t bzk_mpn_push_tx(bzk_mpn* w, uint64_t src_index, uint64_t dst_index, const uint8_t token_id[32], uint64_t amount,
                        const uint8_t fee_token[32], uint64_t fee);
/* applies up to 4^log4_batch queued txs (update::update), pads with UpdateTransition::null, synthesizes the
 * circuit.  Public inputs = [commitment, height, state, aux_data, next_state]. */
int32_t bzk_mpn_update_synthesize(bzk_mpn* w, uint32_t log4_batch, const uint8_t commitment[32], const uint8_t fee_token[32],
                                  int32_t record_matrices, bzk_r1cs** out);
/* Deposit / Withdraw (the other two `MpnWorkData` variants, src/mpn/mod.rs:243-248): DepositCircuit
 * (src/mpn/circuits/deposit_circuit.rs:47-293) with witness builder src/mpn/deposit.rs:11-233, WithdrawCircuit
 * (src/mpn/circuits/withdraw_circuit.rs:50-413) with src/mpn/withdraw.rs:10-259.  aux_data = root of the batch
 * tree (`reveal`, src/zk/groth16/gadgets/reveal/mod.rs:13-61).  `fingerprint` = ContractWithdraw::fingerprint()
 * (src/core/transaction.rs:204-211), taken as an opaque scalar: the L1 payment serialisation is out of scope. */
int32_t bzk_mpn_push_deposit(bzk_mpn* w, uint64_t key_index, const uint8_t token_id[32], uint64_t amount);
/* fingerprint NULL: the withdrawal carries a synthetic L1 payment and its fingerprint is derived from it as the wallet does
 * (src/wallet/tx_builder.rs:376-425); non-NULL: an opaque fingerprint (such a withdrawal cannot be put on the wire) */
int32_t bzk_mpn_push_withdraw(bzk_mpn* w, uint64_t account_index, const uint8_t token_id[32], uint64_t amount,
                              const uint8_t fee_token[32], uint64_t fee, const uint8_t fingerprint[32]);
int32_t bzk_mpn_deposit_synthesize(bzk_mpn* w, uint32_t log4_batch, const uint8_t commitment[32], int32_t record_matrices, bzk_r1cs** out);
int32_t bzk_mpn_withdraw_synthesize(bzk_mpn* w, uint32_t log4_batch, const uint8_t commitment[32], int32_t record_matrices, bzk_r1cs** out);
/* all-disabled instances (kind 0 = deposit, 1 = withdraw), src/mpn/circuits/test.rs:151-229 */
int32_t bzk_mpn_circuit_empty(int32_t kind, uint32_t log4_tree, uint32_t log4_token_tree, uint32_t log4_batch,
                              const uint8_t commitment[32], uint64_t height, const uint8_t state[32], const uint8_t aux_data[32],
                              const uint8_t next_state[32], int32_t record_matrices, bzk_r1cs** out);
/* `MpnCircuit::empty(L, T, B)` with explicit public inputs (src/mpn/circuits/test.rs:117-132) */
int32_t bzk_mpn_update_empty(uint32_t log4_tree, uint32_t log4_token_tree, uint32_t log4_batch, const uint8_t commitment[32],
                             uint64_t height, const uint8_t state[32], const uint8_t aux_data[32], const uint8_t next_state[32],
                             const uint8_t fee_token[32], int32_t record_matrices, bzk_r1cs** out);
/* info: n_in, n_aux, n_constraints, nnz(A), nnz(B), nnz(C), first unsatisfied constraint + 1 (0 = satisfied),
 * accepted txs, rejected txs */
int32_t bzk_r1cs_info(const bzk_r1cs* r, uint64_t info[9]);
/* borrowed views: 0 z, 1 az, 2 bz, 3 cz, 4 a_density, 5 b_density, 6-8 val(A/B/C), 9-11 col(A/B/C) u32 flat
 * variable index, 12-14 row_ptr(A/B/C) u32 */
const void* bzk_r1cs_data(const bzk_r1cs* r, int32_t which, uint64_t* bytes);
void bzk_r1cs_free(bzk_r1cs* r);
/* Witness value traces on the device.  With bzk_mpn_set_defer(w, 1) a witness-only bzk_mpn_{update,deposit,withdraw}_synthesize does NOT
 * evaluate what hangs off Poseidon outputs - the Poseidon gadget's S-box / idle-lane variables (src/zk/groth16/gadgets/poseidon/mod.rs:8-95: two
 * thirds of a transition's constraints), the Merkle gadget's muxes (gadgets/merkle/mod.rs:21-78, common/mux.rs:7-47), the equality checks against
 * computed roots (common/number.rs:121-177) - but records a small program per circuit shape and, per transition, the values the host does know;
 * those slots of z / A.z / B.z / C.z are left untouched.  bzk_groth16_prove_r1cs (below, beside bzk_groth16_prove) uploads the arrays as they are
 * and runs the program on the device before anything reads them; bzk_r1cs_fill_host runs the SAME ops on the CPU for consumers of the complete
 * arrays (bzk_r1cs_data) - the result is byte-identical to a synthesis without deferral.  bzk_r1cs_info's satisfied-check covers the rows the host
 * wrote; the deferred rows are judged where they are computed (BZK_E_UNSAT from the prove call, info[9] after a host fill).
 *   defer_info: 0 deferred (1 / 0), 1 transitions, 2 ops of the program, 3 registers, 4 host-known values per transition, 5 dependency levels
 *               (a Merkle path is a chain of hashes), 6 / 7 variable / constraint slots per transition left to the device, 8 filled on the
 *               host (1 / 0), 9 flags of that fill (1 a deferred constraint does not hold, 2 a computed state differs from the builder's) */
int32_t bzk_mpn_set_defer(bzk_mpn* w, int32_t on);
int32_t bzk_r1cs_defer_info(const bzk_r1cs* r, uint64_t info[10]);
int32_t bzk_r1cs_fill_host(bzk_r1cs* r);
/* the schedule of the instance's program for the one-launch device kernel, checked on the host (no device needed): stages, segments, hash ops
 * covered, fill ops covered, largest segment (<= 64), violations (0: every op exactly once, in a stage its operands are ready for) */
int32_t bzk_r1cs_defer_schedule_info(const bzk_r1cs* r, uint64_t info[6]);
/* bzk_groth16_prove over an instance of the host generator (z / A.z / B.z / C.z are the instance's own pinned arrays), completing deferred
 * witness values on the device first.  BZK_E_UNSAT: a deferred constraint does not hold or a transition's computed state differs from the
 * witness builder's prediction (synthesize again without deferral for the exact first unsatisfied row). */
int32_t bzk_groth16_prove_r1cs(bzk_ctx* ctx, bzk_params* params, const bzk_r1cs* r, const uint8_t r_blind[32], const uint8_t s_blind[32],
                               uint8_t proof_out[387]);
/* Staging - the uploads and the deferred-value program in the witness PRODUCER's pipeline instead of the prover slot's.  bzk_r1cs_stage puts the
 * instance's z | A.z | B.z | C.z into HBM on `ctx`'s stream (a context of the producer's), runs the deferred-value program behind the uploads if the instance
 * has one, and returns at once; the instance's host arrays must stay alive until bzk_staged_wait (or a prove call on the handle) has returned.
 * bzk_groth16_prove_staged (any context of the same device) waits for that work in stream order and copies device to device instead of uploading; same proof
 * bytes, BZK_E_UNSAT as bzk_groth16_prove_r1cs.  bzk_staged_free hands the buffers back to the staging context (any thread; after the last prove call on the
 * handle has returned, before that context is destroyed). */
typedef struct bzk_staged bzk_staged;
int32_t bzk_r1cs_stage(bzk_ctx* ctx, const bzk_r1cs* r, bzk_staged** out);
int32_t bzk_staged_wait(bzk_staged* staged);
void bzk_staged_free(bzk_staged* staged);
/* read-back of one staged array (which: 0 z, 1 A.z, 2 B.z, 3 C.z) once the staging work has finished - i.e. what the DEVICE-side fill produced; *size_out = the
 * array's size, at most `cap` bytes are copied.  The counterpart of bzk_r1cs_fill_host + bzk_r1cs_data for the device executor: byte-identical to a synthesis
 * without deferral (what bellman's `ProvingAssignment` holds after `Circuit::synthesize`, src/mpn/circuits/update_circuit.rs:49-494). */
int32_t bzk_staged_read(const bzk_staged* staged, int32_t which, uint8_t* out, uint64_t cap, uint64_t* size_out);
int32_t bzk_groth16_prove_staged(bzk_ctx* ctx, bzk_params* params, const bzk_staged* staged, const uint8_t r_blind[32], const uint8_t s_blind[32],
                                 uint8_t proof_out[387]);
/* CPU mirrors of `ZkHasher::hash`, `hash_to_scalar`'s SHA3 and `JubJub::{generate_keys, sign, verify}` */
/* ---- f-2: the proving worker's wire format ---------------------------------------------------------------------------
 * `MpnWork` (src/mpn/mod.rs:263-270) as `GET /bincode/mpn/work` delivers it (src/node/mod.rs:393-398,
 * src/client/messages.rs:368-376) and `ZkProof` (src/zk/mod.rs:646-651) as `POST /bincode/mpn/solution` takes it, both
 * bincode 1.3.3 with default options.  Layouts: bazuka_amd/csrc/host_bincode.h.  A Rust host that already holds an
 * `MpnWork` passes `bincode::serialize(&work)`; the worker loop of bazuka_amd/worker.py passes the HTTP body. */
#define BZK_WORK_SIG_LEN_PREFIXED 1u /* ed25519 signatures inside L1 payments carry a u64 length (ed25519 < 1.3) */
typedef struct bzk_mpn_work bzk_mpn_work;
/* decodes ONE MpnWork from the front of `bytes`; *consumed = its encoded length (a response holds several) */
int32_t bzk_mpn_work_decode(const uint8_t* bytes, uint64_t len, uint32_t flags, bzk_mpn_work** out, uint64_t* consumed);
const char* bzk_mpn_work_last_error(void); /* why the last decode on this thread failed */
void bzk_mpn_work_free(bzk_mpn_work* work);
/* info = kind (0 deposit, 1 withdraw, 2 update), log4_tree, log4_token_tree, log4 batch size of that kind, transitions on
 * the wire, height, reward, new_root.state_size, mpn_num_{update,deposit,withdraw}_batches, this work's VK byte length */
int32_t bzk_mpn_work_info(const bzk_mpn_work* work, uint64_t info[12]);
/* out = state | aux_data | next_state | new_root.state_hash | mpn_contract_id, 32 bytes each */
int32_t bzk_mpn_work_scalars(const bzk_mpn_work* work, uint8_t out[160]);
/* bincode(Groth16VerifyingKey) (src/zk/groth16/mod.rs:22-31); which = -1: `MpnWork::vk()`, 0/1/2: deposit/withdraw/update */
int32_t bzk_mpn_work_vk(const bzk_mpn_work* work, int32_t which, uint8_t* out, uint64_t cap, uint64_t* len);
/* ZkScalar::new(sha3_256(bincode((prover, reward)))) - the commitment `MpnWork::verify` binds a solution to
 * (src/mpn/mod.rs:281-295); prover_pub = the worker's 32-byte ed25519 address */
int32_t bzk_mpn_work_commitment(const bzk_mpn_work* work, const uint8_t prover_pub[32], uint8_t out[32]);
/* `MpnWork::verify(prover, proof)` (src/mpn/mod.rs:281-295) on the host: 1 accepted / 0 refused / negative bad arguments */
int32_t bzk_mpn_work_verify(const bzk_mpn_work* work, const uint8_t prover_pub[32], const uint8_t proof[387]);
/* the circuit instance to prove: transitions padded with null ones to 4^batch; fee_token NULL = Ziesha; threads 0 = all;
 * record_matrices: 0 witness only, 1 with the CSR matrices (setup), BZK_SYNTH_DEFER witness only with the hash-dependent values of the
 * work's transitions left to the device (see bzk_mpn_set_defer; all three kinds) */
#define BZK_SYNTH_DEFER 2
int32_t bzk_mpn_work_synthesize(const bzk_mpn_work* work, const uint8_t prover_pub[32], const uint8_t fee_token[32],
                                int32_t threads, int32_t record_matrices, bzk_r1cs** out);
int32_t bzk_mpn_work_encode(const bzk_mpn_work* work, uint8_t* out, uint64_t cap, uint64_t* len); /* out NULL: size query */
/* validator side (`prepare_works`, src/mpn/mod.rs:298-424): one work of the given kind from the world's queued
 * transactions; the world advances to the work's next_state */
typedef struct {
    uint8_t log4_deposit_batch, log4_withdraw_batch, log4_update_batch;
    uint64_t num_update_batches, num_deposit_batches, num_withdraw_batches;
    const uint8_t* deposit_vk;  uint64_t deposit_vk_len;  /* bincode(Groth16VerifyingKey) */
    const uint8_t* withdraw_vk; uint64_t withdraw_vk_len;
    const uint8_t* update_vk;   uint64_t update_vk_len;
    uint64_t new_root_state_size; /* ZkCompressedState::state_size: bookkeeping of the node's KV store, passed through */
} bzk_mpn_work_config;
int32_t bzk_mpn_make_work(bzk_mpn* w, int32_t kind, const bzk_mpn_work_config* cfg, uint64_t reward, bzk_mpn_work** out);
int32_t bzk_zkproof_encode(const uint8_t proof[387], uint8_t out[391]); /* ZkProof::Groth16(Box<Groth16Proof>) */
int32_t bzk_zkproof_decode(const uint8_t* in, uint64_t len, uint8_t proof[387]);

int32_t bzk_host_poseidon(const uint8_t* in, uint32_t arity, uint8_t out[32]);
int32_t bzk_host_scalar_new(const uint8_t* le_bytes, uint32_t len, uint8_t out[32]); /* ZkScalar::new: LE integer mod r; len <= 64 */
int32_t bzk_host_sha3_256(const uint8_t* in, uint64_t len, uint8_t out[32]);
int32_t bzk_host_jubjub_keys(const uint8_t* seed, uint32_t len, uint8_t out[128]); /* pub.x|pub.y|randomness|scalar */
int32_t bzk_host_jubjub_sign(const uint8_t key[128], const uint8_t msg[32], uint8_t sig_out[96]); /* r.x|r.y|s */
int32_t bzk_host_jubjub_verify(const uint8_t pub_xy[64], const uint8_t msg[32], const uint8_t sig[96]); /* 1 / 0 */

/* ---- static-base tables (the Groth16 CRS queries are fixed point sets) -------------------------------------------
 * build: tab[w][i] = 2^(c w) * base_i for every window, kept in HBM (W x the base memory, internal limb form).  With a
 * table every window feeds ONE bucket set: the bucket reduction shrinks from W windows to one and no host-side Horner
 * remains.  Same result bytes as bzk_msm_*_dev on the same inputs.  n scalars (n <= table size) use bases [0, n). */
typedef struct bzk_msm_table bzk_msm_table;
int32_t bzk_msm_g1_table_build(bzk_ctx* ctx, const void* bases_dev, uint64_t n, bzk_msm_table** out);
int32_t bzk_msm_g2_table_build(bzk_ctx* ctx, const void* bases_dev, uint64_t n, bzk_msm_table** out);
/* full table with an explicit window size c (4 .. 22): every window shares one set of 2^(c-1) buckets, so c can grow to ~log2 n -
 * fewer windows, i.e. fewer additions per point: 2^20 points, c = 20: 13 levels (1.5 GB), 3.93 ms against 4.32 ms for the per-call
 * pipeline on raw bases (profiles/r02_run17_static_tables.txt).  What bzk_groth16_prove uses for the (static) h query. */
int32_t bzk_msm_g1_table_build_c(bzk_ctx* ctx, const void* bases_dev, uint64_t n, uint32_t c, bzk_msm_table** out);
/* folded table: `levels` L < W levels tab[j][i] = 2^(c wpl j) * base_i, wpl = ceil(W / L): windows j * wpl + w' share bucket
 * set w' - the same number of point additions, 1 / L of the buckets to reduce, L x the base memory (levels = 0 or >= W: the
 * full table).  For a folded table the `windows` range of *_table_windows_dev selects bucket sets [w_begin, w_end) of the
 * bzk_msm_table_window_count() = wpl sets; the partial results still add up to the full MSM (bzk_g1_sum / bzk_g2_sum). */
int32_t bzk_msm_g1_table_build_levels(bzk_ctx* ctx, const void* bases_dev, uint64_t n, uint32_t levels, bzk_msm_table** out);
int32_t bzk_msm_g2_table_build_levels(bzk_ctx* ctx, const void* bases_dev, uint64_t n, uint32_t levels, bzk_msm_table** out);
uint32_t bzk_msm_table_levels(const bzk_msm_table* table);
int32_t bzk_msm_g1_table_run_dev(bzk_ctx* ctx, const bzk_msm_table* table, const void* scalars_dev, uint64_t n, uint32_t flags, uint8_t out[97]);
int32_t bzk_msm_g2_table_run_dev(bzk_ctx* ctx, const bzk_msm_table* table, const void* scalars_dev, uint64_t n, uint32_t flags, uint8_t out[193]);
int32_t bzk_msm_g1_table_windows_dev(bzk_ctx* ctx, const bzk_msm_table* table, const void* scalars_dev, uint64_t n, uint32_t flags,
                                     uint32_t w_begin, uint32_t w_end, uint8_t out[97]);
int32_t bzk_msm_g2_table_windows_dev(bzk_ctx* ctx, const bzk_msm_table* table, const void* scalars_dev, uint64_t n, uint32_t flags,
                                     uint32_t w_begin, uint32_t w_end, uint8_t out[193]);
uint32_t bzk_msm_table_window_count(const bzk_msm_table* table);
void bzk_msm_table_free(bzk_ctx* ctx, bzk_msm_table* table);

/* ---- resident base sets --------------------------------------------------------------------------------------------
 * A Groth16 CRS query is a STATIC point set (bellman `Parameters`: h, l, a, b_g1, b_g2): load converts it once into the
 * library's internal limb form (G1 112 B, G2 224 B per point) and keeps it in HBM; every later MSM over it - and every
 * rank of a window-sharded MSM - gathers from the resident set and converts nothing per call.  Same result bytes as
 * bzk_msm_*_dev on the raw bases.  n scalars (n <= set size) use bases [0, n).  A set is read-only after load and may be
 * shared by any number of contexts of the same device. */
typedef struct bzk_msm_bases bzk_msm_bases;
int32_t bzk_msm_g1_bases_load_dev(bzk_ctx* ctx, const void* bases_dev, uint64_t n, bzk_msm_bases** out);
int32_t bzk_msm_g2_bases_load_dev(bzk_ctx* ctx, const void* bases_dev, uint64_t n, bzk_msm_bases** out);
void bzk_msm_bases_free(bzk_ctx* ctx, bzk_msm_bases* bases);
uint64_t bzk_msm_bases_size(const bzk_msm_bases* bases);
/* What a load decided: *forms = 1 (the set alone) or the number of arrays held - the set plus its endomorphism images X^m P (G1: 2,
 * G2: 4), which whole-MSM calls flagged BZK_F_THROUGHPUT use (fewer bucket sets, same additions, same result).  The images are built
 * at load unless no call could use them (BZK_MSM_ENDO_G1 / _G2 = 0, BZK_MSM_NO_ENDO = 1, device groups) or they do not fit beside an
 * 8 GiB reserve; *device_bytes = HBM held by the set. */
int32_t bzk_msm_bases_info(const bzk_msm_bases* bases, uint64_t* n, int32_t* forms, uint64_t* device_bytes);
int32_t bzk_msm_g1_bases_run_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* scalars_dev, uint64_t n, uint32_t flags, uint8_t out[97]);
int32_t bzk_msm_g2_bases_run_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* scalars_dev, uint64_t n, uint32_t flags, uint8_t out[193]);
int32_t bzk_msm_g1_bases_windows_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* scalars_dev, uint64_t n, uint32_t flags,
                                     uint32_t w_begin, uint32_t w_end, uint8_t out[97]);
int32_t bzk_msm_g2_bases_windows_dev(bzk_ctx* ctx, const bzk_msm_bases* bases, const void* scalars_dev, uint64_t n, uint32_t flags,
                                     uint32_t w_begin, uint32_t w_end, uint8_t out[193]);

/* ---- row (e): multi-GPU ------------------------------------------------------------------------------------------------
 * A device GROUP behind the boundary: a Rust prover drives 1..8 MI355X of a node through these entry points alone (no
 * torch.distributed, no Python).  SURVEY.md 8b proposed `bzk_ctx_create(device_ids, n)`; the group is its own handle so that
 * the single-device ctx keeps its one-stream, one-thread contract.
 *   MSM    : sharded by SCALAR-WINDOW RANGE over the group (north star): rank r computes windows [W r / world, W (r + 1) / world)
 *            over all points from a resident base set replicated at load time; ONE all-gather of the W window sums (3 - 6 KB
 *            for the whole group; RCCL cannot add curve points, so gather + combine), then the Horner combine over the W windows
 *            on the host, as in the single-GPU call.  Same 97 / 193 result bytes as bzk_msm_g*_dev.
 *   proofs : replicas (the reference's own split of a block's proofs over provers, src/mpn/mod.rs:79-107): a pool of prover slots
 *            on every device takes proofs from one queue (submit / wait), CRS shared per device.
 * Deployments: bzk_mg_create - ONE process drives n devices (one persistent host thread each; device ids may repeat, e.g. for a
 * rehearsal on a one-GPU box); bzk_mg_create_rank - one process PER GPU (torchrun style): rank 0 draws bzk_mg_unique_id() and
 * the host hands the 128 bytes to the other ranks by its own means, every rank then calls bzk_mg_create_rank and later the same
 * bzk_mg_msm_* call with its own device pointers; every rank receives the result.
 * Exchange: BZK_MG_X_RCCL = ncclAllGather(uint8) over xGMI (librccl is loaded at run time); BZK_MG_X_HOST = pinned host memory
 * (one process) or a POSIX shared-memory segment (process per GPU, single node) - the only transport for ranks that SHARE a
 * device; BZK_MG_X_PEER = hipMemcpyPeerAsync into device 0 (one process).  BZK_MG_X_AUTO = RCCL when the group spans distinct
 * devices and librccl loads, else HOST.  One collective call at a time per group; bzk_mg_ctx(i) exposes the i-th local context
 * (allocation, uploads, profiling) and must not be used concurrently with a group call. */
#define BZK_MG_UID_BYTES 128
#define BZK_MG_X_AUTO 0u
#define BZK_MG_X_HOST 1u
#define BZK_MG_X_PEER 2u
#define BZK_MG_X_RCCL 3u
typedef struct bzk_mg bzk_mg;
typedef struct bzk_mg_bases bzk_mg_bases;
int32_t bzk_mg_unique_id(uint8_t uid[BZK_MG_UID_BYTES]);
/* what THIS process could contribute to a group on `device_id`, without creating anything that a peer would wait for: a bit mask,
 * bit 0 = the device exists and is a gfx950, bit 1 = librccl is loadable with every entry point the RCCL exchange needs.  A
 * process-per-GPU host publishes the answer of every rank (by its own means) BEFORE calling bzk_mg_create_rank with BZK_MG_X_RCCL:
 * ncclCommInitRank blocks until all ranks arrive, so a rank that cannot take part must be known beforehand.  Negative: BZK_E_*. */
int32_t bzk_mg_probe(int32_t device_id);
int32_t bzk_mg_create(const int32_t* device_ids, int32_t n_devices, uint32_t exchange, bzk_mg** out);
int32_t bzk_mg_create_rank(int32_t device_id, int32_t rank, int32_t world, const uint8_t uid[BZK_MG_UID_BYTES], uint32_t exchange, bzk_mg** out);
void bzk_mg_destroy(bzk_mg* mg);
int32_t bzk_mg_world(const bzk_mg* mg);      /* ranks of the group */
int32_t bzk_mg_local(const bzk_mg* mg);      /* devices this process drives */
int32_t bzk_mg_rank(const bzk_mg* mg);       /* rank of local device 0 */
uint32_t bzk_mg_exchange(const bzk_mg* mg);  /* the transport in use (BZK_MG_X_*) */
bzk_ctx* bzk_mg_ctx(bzk_mg* mg, int32_t local_index);
/* Where the window-sharded calls of this rank spent their time (cumulative; reset != 0 clears the per-call sums afterwards):
 *   out[0] calls   out[1] local stage, ms (this rank's windows, device side, local device 0)   out[2] exchange, ms (device side: the
 *   all-gather / peer copies and the read-back queued behind the local stage)   out[3] waiting for the other ranks, ms (host side:
 *   the shared-memory transport's barrier)   out[4] host combine (Horner), ms   out[5] group creation, s   out[6] of which the
 *   communicator (ncclCommInitRank / ncclCommInitAll), s   out[7] reserved.  A slow rank shows up as exchange / waiting time of the others. */
int32_t bzk_mg_stats(bzk_mg* mg, int32_t reset, double out[8]);
const char* bzk_mg_last_error(bzk_mg* mg);
/* replicate a static base set on every local device (host pointer: uploaded + converted per device; _dev: one device pointer
 * per local device, raw affine) */
int32_t bzk_mg_bases_g1_load(bzk_mg* mg, const uint8_t* bases_host, uint64_t n, bzk_mg_bases** out);
int32_t bzk_mg_bases_g2_load(bzk_mg* mg, const uint8_t* bases_host, uint64_t n, bzk_mg_bases** out);
int32_t bzk_mg_bases_g1_load_dev(bzk_mg* mg, const void* const* bases_dev, uint64_t n, bzk_mg_bases** out);
int32_t bzk_mg_bases_g2_load_dev(bzk_mg* mg, const void* const* bases_dev, uint64_t n, bzk_mg_bases** out);
void bzk_mg_bases_free(bzk_mg* mg, bzk_mg_bases* bases);
/* scalars: one host vector (broadcast to the local devices), or one DEVICE pointer per local device holding the same n scalars */
int32_t bzk_mg_msm_g1(bzk_mg* mg, const bzk_mg_bases* bases, const uint8_t* scalars, uint64_t n, uint32_t flags, uint8_t out[97]);
int32_t bzk_mg_msm_g2(bzk_mg* mg, const bzk_mg_bases* bases, const uint8_t* scalars, uint64_t n, uint32_t flags, uint8_t out[193]);
int32_t bzk_mg_msm_g1_dev(bzk_mg* mg, const bzk_mg_bases* bases, const void* const* scalars_dev, uint64_t n, uint32_t flags, uint8_t out[97]);
int32_t bzk_mg_msm_g2_dev(bzk_mg* mg, const bzk_mg_bases* bases, const void* const* scalars_dev, uint64_t n, uint32_t flags, uint8_t out[193]);

/* proof pool over the group's local devices: `slots_per_device` prover slots (context + lanes + scratch each) per device over ONE
 * CRS upload per device; one host thread per slot takes proofs from a common queue (whichever slot is free next).  The assignment
 * arrays and proof_out must stay valid until the ticket has been waited for; wait returns that proof's status. */
typedef struct bzk_mg_params bzk_mg_params;
int32_t bzk_mg_params_load(bzk_mg* mg, const bzk_params_desc* desc, uint32_t slots_per_device, bzk_mg_params** out);
void bzk_mg_params_free(bzk_mg* mg, bzk_mg_params* params);
uint32_t bzk_mg_params_slots(const bzk_mg_params* params);
int32_t bzk_mg_params_stats(bzk_mg_params* params, uint64_t* proofs_per_slot, uint32_t cap);
int32_t bzk_mg_prove_submit(bzk_mg* mg, bzk_mg_params* params, const bzk_assignment* asg, const uint8_t r[32], const uint8_t s[32],
                            uint8_t proof_out[387], uint64_t* ticket);
int32_t bzk_mg_prove_wait(bzk_mg* mg, bzk_mg_params* params, uint64_t ticket);
int32_t bzk_mg_prove(bzk_mg* mg, bzk_mg_params* params, const bzk_assignment* asg, const uint8_t r[32], const uint8_t s[32],
                     uint8_t proof_out[387]);

/* synthetic-input helpers (device side, for benches and tests): base_i = k_i * G with
 * k_i = SplitMix64(seed + 0x632BE59BD9B4E019 * (start+i)).next() | 1 ; raw affine out */
int32_t bzk_g1_synth_bases_dev(bzk_ctx* ctx, uint64_t seed, uint64_t start, uint64_t n, void* out_dev);
int32_t bzk_g2_synth_bases_dev(bzk_ctx* ctx, uint64_t seed, uint64_t start, uint64_t n, void* out_dev);

#ifdef __cplusplus
}
#endif
#endif /* BZK_H */
