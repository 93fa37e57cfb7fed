// a4-a7: Groth16 prove after witness synthesis, on gfx950.
//
// Replaces bellman 0.14 `groth16::create_proof` (third-party crate; reference call sites
// /root/reference/src/mpn/circuits/test.rs:135,175,215) from the point where the assignment
// (z, A.z, B.z, C.z) exists.  Layout of the computation follows SURVEY.md Appendix D:
//   a,b,c <- iNTT ; <- coset NTT (shift 7) ; a <- (a*b - c) / (7^m - 1) ; a <- inverse coset NTT
//   H = MSM(h, a[0..m-1)) ; L = MSM(l, aux) ; A = MSM(a, z|a_density) ; B1, B2 = MSM(b, z|b_density)
//   g_a = r*delta1 + alpha1 + A ;  g_b = s*delta2 + beta2 + B2
//   g_c = rs*delta1 + s*alpha1 + r*beta1 + s*A + r*B1 + H + L
// The NTTs and MSMs are the HIP kernels of ntt.hip / msm_g1.hip / msm_g2.hip; the last three lines
// are ~10 point operations done on the host.  Proof bytes = Groth16Proof layout
// (/root/reference/src/zk/groth16/mod.rs:33-38).
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "bzk_curve.cuh"
#include "bzk_internal.h"
#include "host_fp64.h"
#include "host_r1cs.h"  // DeferData / witfill_run_dev: witness values the device fills in (bzk_groth16_prove_r1cs)
namespace bzk { void r1cs_assignment(const bzk_r1cs* r, bzk_assignment* a, const DeferData** dd, uint64_t* n_in); }  // mpn.hip

// The CRS of one circuit on one device, shared (reference-counted, read-only once prepared) by every prover SLOT of that device:
// a slot = bzk_params = the shared CRS + its own per-proof scratch.  Four slots per GPU (bench.py) used to hold four CRS copies and
// four 1.5 GB h tables; now one of each.
struct CrsShared {
    std::atomic<int> refs{1};
    std::mutex m;  // guards the lazy preparation below
    int device = 0;
    uint32_t n_in = 0, n_aux = 0, log_m = 0, n_a = 0, n_b = 0;
    uint8_t vk[870];
    void *h = nullptr, *l = nullptr, *a = nullptr, *b_g1 = nullptr, *b_g2 = nullptr;  // raw affine, as loaded (bzk_params_read)
    uint32_t *a_idx = nullptr, *b_idx = nullptr;                                       // variable index of each dense entry
    // resident internal forms of the queries (bzk_msm_bases: converted once, no per-proof conversion), built on first use
    bzk_msm_bases *rl = nullptr, *ra = nullptr, *rb1 = nullptr, *rb2 = nullptr, *rh = nullptr;
    // [l | b_g1] as ONE base set: l + r b_g1 is then one MSM over (z_aux | r z_b) - see groth16_prove_impl (round 3)
    bzk_msm_bases* rlb1 = nullptr;
    // static-base table of the h query: the h bases never change and their scalars are never de-duplicated, so all windows can share one
    // bucket set at a window size of ~log2 m (fewer windows = fewer additions)
    // published with release / read with acquire: prover slots of this CRS read it without the lock
    std::atomic<bzk_msm_table*> h_table{nullptr};
    bool prepared = false;
};
struct bzk_params {
    CrsShared* crs;
    void *d_z, *d_a, *d_b, *d_c, *d_sa, *d_sb;  // device scratch sized for this circuit, one set per slot
};

namespace bzk {

__global__ void __launch_bounds__(256) g16_pointwise_kernel(Fr* __restrict__ a, const Fr* __restrict__ b,
                                                            const Fr* __restrict__ c, uint64_t m, Fr zinv) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Fr v = fe_mul<FrParams>(a[i], b[i]);
    v = fe_sub<FrParams>(v, c[i]);
    a[i] = fe_mul<FrParams>(v, zinv);
}

__global__ void __launch_bounds__(256) g16_gather_kernel(const Fr* __restrict__ z, const uint32_t* __restrict__ idx, uint32_t n,
                                                         Fr* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = z[idx[i]];
}

// out[i] = k * z[idx[i]] (Montgomery forms): the scalars of r b_g1 inside the merged l + r b_g1 MSM
__global__ void __launch_bounds__(256) g16_gather_mul_kernel(const Fr* __restrict__ z, const uint32_t* __restrict__ idx, uint32_t n, Fr k,
                                                             Fr* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fe_mul<FrParams>(z[idx[i]], k);
}

static Fr host_fr_pow(Fr b, uint64_t e) {
    Fr r = Fr::one();
    while (e) {
        if (e & 1) r = fe_mul<FrParams>(r, b);
        b = fe_sqr<FrParams>(b);
        e >>= 1;
    }
    return r;
}

int32_t groth16_h(bzk_ctx* ctx, void* a, void* b, void* c, uint32_t log_m) {
    if (log_m > 28) return BZK_E_ARG;
    // env BZK_H_UNFUSED=1: the round-1/2 form - seven stand-alone transforms and a pointwise kernel (A/B runs, and the parity
    // reference of the fused chain inside the GPU tests)
    static const bool unfused = env_on("BZK_H_UNFUSED");
    if (!unfused && log_m >= 1) return ntt_h_chain(ctx, a, b, c, log_m);
    const uint64_t m = (uint64_t)1 << log_m;
    void* v[3] = {a, b, c};
    for (int k = 0; k < 3; ++k) {
        BZK_TRY(ntt_run(ctx, v[k], log_m, 1, 0));
        BZK_TRY(ntt_run(ctx, v[k], log_m, 0, 1));
    }
    Fr seven = Fr::zero();
    seven.l[0] = 7;
    seven = fe_to_mont<FrParams>(seven);
    Fr zinv = fe_inv<FrParams>(fe_sub<FrParams>(host_fr_pow(seven, m), Fr::one()));
    BZK_LAUNCH(ctx, "g16_pointwise", g16_pointwise_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (Fr*)a, (const Fr*)b,
               (const Fr*)c, m, zinv);
    BZK_TRY(ntt_run(ctx, a, log_m, 1, 1));
    return BZK_OK;
}

// ---- proof assembly on the host (bellman prover.rs: g_a = alpha + a + r delta, g_b = beta + b + s delta,
// g_c = h + l + s g_a + r (beta_g1 + b_g1)): a handful of 255-bit scalar multiplications that nothing overlaps with once the
// MSMs are in.  They run on the 64-bit-limb host field (host_fp64.h); the two that depend only on (r, s) and the verifying key are
// computed on a lane thread while the device still works, the other two share one doubling chain (Shamir), and the three final
// inversions share one.  Round 3: with full-size (r, s) this was 2.4 ms at the end of every proof (six G1 + one G2 double-and-add
// chains on the 32-bit-limb field code); now ~0.3 ms.
typedef XyzzT<HFpOps> HG1;
typedef XyzzT<HFp2Ops> HG2;

static inline int fr_bit(const Fr& k, int i) { return (int)((k.l[i >> 5] >> (i & 31)) & 1u); }
static inline int fr_top_bit(const Fr& k) {
    for (int i = 254; i >= 0; --i)
        if (fr_bit(k, i)) return i;
    return -1;
}
// k * p, k canonical (not Montgomery)
template <class H>
static XyzzT<H> host_mul_fr(const XyzzT<H>& p, const Fr& k) {
    XyzzT<H> r = xyzz_identity<H>();
    for (int i = fr_top_bit(k); i >= 0; --i) {
        r = xyzz_dbl<H>(r);
        if (fr_bit(k, i)) xyzz_add<H>(r, p);
    }
    return r;
}
// k * p + l * q on one doubling chain
template <class H>
static XyzzT<H> host_mul2_fr(const XyzzT<H>& p, const Fr& k, const XyzzT<H>& q, const Fr& l) {
    XyzzT<H> pq = p;
    xyzz_add<H>(pq, q);
    const XyzzT<H>* tab[4] = {nullptr, &p, &q, &pq};
    XyzzT<H> r = xyzz_identity<H>();
    const int top = fr_top_bit(k) > fr_top_bit(l) ? fr_top_bit(k) : fr_top_bit(l);
    for (int i = top; i >= 0; --i) {
        r = xyzz_dbl<H>(r);
        const int sel = fr_bit(k, i) | (fr_bit(l, i) << 1);
        if (sel) xyzz_add<H>(r, *tab[sel]);
    }
    return r;
}
// (g_a, g_b, g_c) -> the 387-byte packed proof; one field inversion for the three points (Montgomery's trick; the G2 denominator
// through its norm).  An identity point packs as (0, 1, flag) like PointIO::pack.
static void pack_proof(const HG1& ga, const HG2& gb, const HG1& gc, uint8_t* proof) {
    typedef HFpOps F;
    const bool fa = !xyzz_is_identity<HFpOps>(ga), fb = !xyzz_is_identity<HFp2Ops>(gb), fc = !xyzz_is_identity<HFpOps>(gc);
    const HFp nb = F::add(F::sqr(gb.ZZZ.c0), F::sqr(gb.ZZZ.c1));  // |ZZZ_b|^2
    const HFp d0 = fa ? ga.ZZZ : F::one(), d1 = fb ? nb : F::one(), d2 = fc ? gc.ZZZ : F::one();
    const HFp d01 = F::mul(d0, d1);
    const HFp iall = F::inv(F::mul(d01, d2));
    const HFp i2 = F::mul(iall, d01);      // 1 / d2
    const HFp i01 = F::mul(iall, d2);      // 1 / (d0 d1)
    const HFp i0 = F::mul(i01, d1), i1 = F::mul(i01, d0);
    auto g1_out = [](const HG1& p, const HFp& i3, bool fin, uint8_t* out) {
        HFp x = F::zero(), y = F::one();
        if (fin) {
            const HFp iz2 = F::mul(F::sqr(p.ZZ), F::sqr(i3));  // 1/ZZ = ZZ^2 / ZZZ^2
            x = F::mul(p.X, iz2);
            y = F::mul(p.Y, i3);
        }
        memcpy(out, x.l, 48);
        memcpy(out + 48, y.l, 48);
        out[96] = fin ? 0 : 1;
    };
    g1_out(ga, i0, fa, proof);
    g1_out(gc, i2, fc, proof + 290);
    HFp2 x = HFp2Ops::zero(), y = HFp2Ops::one();
    if (fb) {
        const HFp2 i3 = {F::mul(gb.ZZZ.c0, i1), F::mul(F::neg(gb.ZZZ.c1), i1)};  // conj(ZZZ) / |ZZZ|^2
        const HFp2 iz2 = HFp2Ops::mul(HFp2Ops::sqr(gb.ZZ), HFp2Ops::sqr(i3));
        x = HFp2Ops::mul(gb.X, iz2);
        y = HFp2Ops::mul(gb.Y, i3);
    }
    memcpy(proof + 97, x.c0.l, 48);
    memcpy(proof + 97 + 48, x.c1.l, 48);
    memcpy(proof + 97 + 96, y.c0.l, 48);
    memcpy(proof + 97 + 144, y.c1.l, 48);
    proof[97 + 192] = fb ? 0 : 1;
}

static G1Xyzz unpack_g1(const uint8_t* in) {
    if (in[96]) return xyzz_identity<FpOps>();
    G1Affine a;
    memcpy(a.x.l, in, 48);
    memcpy(a.y.l, in + 48, 48);
    return xyzz_from_affine<FpOps>(a);
}
static G2Xyzz unpack_g2(const uint8_t* in) {
    if (in[192]) return xyzz_identity<Fp2Ops>();
    G2Affine a;
    memcpy(a.x.c0.l, in, 48);
    memcpy(a.x.c1.l, in + 48, 48);
    memcpy(a.y.c0.l, in + 96, 48);
    memcpy(a.y.c1.l, in + 144, 48);
    return xyzz_from_affine<Fp2Ops>(a);
}

}  // namespace bzk

using namespace bzk;

extern "C" {

static void crs_release(bzk_ctx* ctx, CrsShared* c) {
    if (!c || c->refs.fetch_sub(1) != 1) return;
    if (bzk_msm_table* t = c->h_table.load(std::memory_order_acquire)) bzk_msm_table_free(ctx, t);
    for (bzk_msm_bases* b : {c->rl, c->ra, c->rb1, c->rb2, c->rh, c->rlb1})
        if (b) bzk_msm_bases_free(ctx, b);
    void* bufs[] = {c->h, c->l, c->a, c->b_g1, c->b_g2, c->a_idx, c->b_idx};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete c;
}

void bzk_params_free(bzk_ctx* ctx, bzk_params* p) {
    if (!p) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        for (bzk_ctx* c : ctx->lanes) (void)hipStreamSynchronize(c->stream);
    }
    void* bufs[] = {p->d_z, p->d_a, p->d_b, p->d_c, p->d_sa, p->d_sb};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    crs_release(ctx, p->crs);
    delete p;
}

static int32_t slot_alloc(bzk_ctx* ctx, CrsShared* c, bzk_params** out) {
    bzk_params* p = new (std::nothrow) bzk_params();
    if (!p) return BZK_E_ALLOC;
    memset(p, 0, sizeof(*p));
    p->crs = c;
    const uint64_t m = (uint64_t)1 << c->log_m, nv = (uint64_t)c->n_in + c->n_aux;
    struct Al { void** dst; size_t bytes; };
    // d_z = [inputs | aux | r z_b]: the third part is written per proof when l and b_g1 run as one MSM (scalars = aux | r z_b, contiguous)
    Al als[] = {{&p->d_z, (size_t)(nv + c->n_b) * 32}, {&p->d_a, (size_t)m * 32}, {&p->d_b, (size_t)m * 32}, {&p->d_c, (size_t)m * 32},
                {&p->d_sa, (size_t)c->n_a * 32}, {&p->d_sb, (size_t)c->n_b * 32}};
    for (auto& a : als) {
        if (hipMalloc(a.dst, a.bytes ? a.bytes : 32) != hipSuccess) {
            ctx->last_error = "params: scratch allocation";
            (void)hipGetLastError();
            p->crs = nullptr;  // the caller still owns its reference
            bzk_params_free(ctx, p);
            return BZK_E_ALLOC;
        }
    }
    *out = p;
    return BZK_OK;
}

static int32_t params_build(bzk_ctx* ctx, const bzk_params_desc* d, bool upload_crs, bzk_params** out) {
    if (!ctx || !d || !out || !d->vk || !d->a_density || !d->b_density) return BZK_E_ARG;
    if (d->log_m > 28 || d->n_in == 0) return BZK_E_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    const uint64_t m = (uint64_t)1 << d->log_m, nv = (uint64_t)d->n_in + d->n_aux;
    std::vector<uint32_t> ia, ib;
    for (uint64_t v = 0; v < nv; ++v) {
        if (d->a_density[v]) ia.push_back((uint32_t)v);
        if (d->b_density[v]) ib.push_back((uint32_t)v);
    }
    if (ia.size() != d->n_a || ib.size() != d->n_b) return BZK_E_ARG;
    if (upload_crs && ((m > 1 && !d->h) || (d->n_aux && !d->l) || (d->n_a && !d->a) || (d->n_b && (!d->b_g1 || !d->b_g2))))
        return BZK_E_ARG;
    CrsShared* c = new (std::nothrow) CrsShared();
    if (!c) return BZK_E_ALLOC;
    c->device = ctx->device;
    c->n_in = d->n_in; c->n_aux = d->n_aux; c->log_m = d->log_m; c->n_a = d->n_a; c->n_b = d->n_b;
    memcpy(c->vk, d->vk, 870);
    struct Up { void** dst; const void* src; size_t bytes; };
    Up ups[] = {
        {&c->h, upload_crs ? d->h : nullptr, (size_t)(m - 1) * 96}, {&c->l, upload_crs ? d->l : nullptr, (size_t)d->n_aux * 96},
        {&c->a, upload_crs ? d->a : nullptr, (size_t)d->n_a * 96}, {&c->b_g1, upload_crs ? d->b_g1 : nullptr, (size_t)d->n_b * 96},
        {&c->b_g2, upload_crs ? d->b_g2 : nullptr, (size_t)d->n_b * 192},
        {(void**)&c->a_idx, ia.data(), ia.size() * 4}, {(void**)&c->b_idx, ib.data(), ib.size() * 4},
    };
    for (auto& u : ups) {
        hipError_t e = hipMalloc(u.dst, u.bytes ? u.bytes : 32);
        if (e == hipSuccess && u.src && u.bytes) e = hipMemcpyAsync(*u.dst, u.src, u.bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            ctx->last_error = std::string("params_load: ") + hipGetErrorString(e);
            (void)hipGetLastError();
            (void)hipStreamSynchronize(ctx->stream);
            crs_release(ctx, c);
            return BZK_E_ALLOC;
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
        crs_release(ctx, c);
        return BZK_E_DEVICE;
    }
    const int32_t st = slot_alloc(ctx, c, out);
    if (st != BZK_OK) crs_release(ctx, c);
    return st;
}

// another prover slot over the SAME device-resident CRS (tables and resident base sets included): own scratch, shared queries
int32_t bzk_params_slot(bzk_ctx* ctx, const bzk_params* src, bzk_params** out) {
    if (!ctx || !src || !out) return BZK_E_ARG;
    *out = nullptr;
    if (src->crs->device != ctx->device) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    src->crs->refs.fetch_add(1);
    const int32_t st = slot_alloc(ctx, src->crs, out);
    if (st != BZK_OK) src->crs->refs.fetch_sub(1);
    return st;
}

// Resident forms of the CRS, built once per device under the CRS mutex by whichever slot proves first:
//   l, a, b_g1, b_g2 (and h where no table is built) -> bzk_msm_bases (internal limb form: no conversion inside a proof)
//   h                                                -> full static table with a 20-bit window for 2^16 .. 2^24 domains (13 levels x 112 B x m:
//                                                        1.5 GB at 2^20, 24 GB at the production 2^24; env BZK_PROVE_H_TABLE_MAX_LOG moves the
//                                                        limit; BZK_PROVE_H_TABLE=0 switches the table off)
// Memory policy: a resident form is only built when hipMemGetInfo shows its size plus a reserve for the provers' grow-only
// workspaces free; anything that does not fit is simply not built (the per-call pipeline on the raw bases works without it), and
// bzk_groth16_prove drops the table and retries once if a workspace allocation fails later while it is the CRS's only user.
// env BZK_PROVE_MERGE_LB1=0: l and b_g1 as two MSMs (the round-2 form; A/B runs).  BZK_PROVE_LANES=4 implies it (b_g1 has its own lane there).
static bool prove_merge_lb1() {
    static const bool on = [] {
        const char* e = getenv("BZK_PROVE_MERGE_LB1");
        const char* l = getenv("BZK_PROVE_LANES");
        return !(e && atoi(e) == 0) && !(l && atoi(l) == 4);
    }();
    return on;
}
static bool crs_fits(size_t need, size_t reserve) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return false; }
    return fr >= need + reserve;
}
static void crs_prepare(bzk_ctx* ctx, CrsShared* c) {
    std::lock_guard<std::mutex> lk(c->m);
    if (c->prepared) return;
    c->prepared = true;
    const uint64_t m = (uint64_t)1 << c->log_m;
    static const bool want_tab = [] { const char* e = getenv("BZK_PROVE_H_TABLE"); return !e || atoi(e) != 0; }();
    static const bool want_res = [] { const char* e = getenv("BZK_PROVE_RESIDENT_BASES"); return !e || atoi(e) != 0; }();
    static const uint32_t max_log = [] {
        const char* e = getenv("BZK_PROVE_H_TABLE_MAX_LOG");
        const int v = e ? atoi(e) : 24;  // round 3: one table per DEVICE (shared by the slots), so the production 2^24 domain's 24 GB is affordable by default
        return (uint32_t)(v < 16 ? 16 : (v > 26 ? 26 : v));
    }();
    // reserve: what the slots of this device may still allocate - MSM workspaces of the five lanes (~40 B x 16 windows per point) x 4 slots
    const size_t reserve = ((size_t)8 << 30) + (size_t)m * 16 * 40 * 4;
    if (want_tab && c->log_m >= 16 && c->log_m <= max_log && m > 1) {
        const uint32_t cw = c->log_m > 20 ? 20u : c->log_m;
        const size_t need = (size_t)((256 + cw - 1) / cw) * 112 * (m - 1);
        bzk_msm_table* t = nullptr;
        if (!crs_fits(need, reserve) || bzk_msm_g1_table_build_c(ctx, c->h, m - 1, cw, &t) != BZK_OK) {
            t = nullptr;
            (void)hipGetLastError();
        }
        c->h_table.store(t, std::memory_order_release);
    }
    if (!want_res) return;
    // l and b_g1 as one set (prove_merge_lb1): [l | b_g1] concatenated in a temporary raw buffer, converted once
    if (prove_merge_lb1() && c->l && c->b_g1 && c->n_aux && c->n_b) {
        const size_t n_cat = (size_t)c->n_aux + c->n_b;
        void* cat = nullptr;
        if (crs_fits(n_cat * (96 + 112), reserve) && hipMalloc(&cat, n_cat * 96) == hipSuccess) {
            bool ok = hipMemcpyAsync(cat, c->l, (size_t)c->n_aux * 96, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
                      hipMemcpyAsync((char*)cat + (size_t)c->n_aux * 96, c->b_g1, (size_t)c->n_b * 96, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess;
            if (ok) ok = bzk_msm_g1_bases_load_dev(ctx, cat, n_cat, &c->rlb1) == BZK_OK;
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(cat);
            if (!ok) c->rlb1 = nullptr;
        }
        (void)hipGetLastError();
    }
    struct Q { const void* raw; uint64_t n; bzk_msm_bases** dst; bool g2; };
    Q qs[] = {{c->b_g2, c->n_b, &c->rb2, true}, {c->rlb1 ? nullptr : c->l, c->n_aux, &c->rl, false}, {c->a, c->n_a, &c->ra, false},
              {c->rlb1 ? nullptr : c->b_g1, c->n_b, &c->rb1, false}, {c->h_table.load(std::memory_order_acquire) ? nullptr : c->h, m - 1, &c->rh, false}};
    for (auto& q : qs) {
        if (!q.raw || !q.n) continue;
        if (!crs_fits((size_t)q.n * (q.g2 ? 224 : 112), reserve)) continue;
        const int32_t st = q.g2 ? bzk_msm_g2_bases_load_dev(ctx, q.raw, q.n, q.dst) : bzk_msm_g1_bases_load_dev(ctx, q.raw, q.n, q.dst);
        if (st != BZK_OK) { *q.dst = nullptr; (void)hipGetLastError(); }
    }
}

int32_t bzk_params_load(bzk_ctx* ctx, const bzk_params_desc* d, bzk_params** out) { return params_build(ctx, d, true, out); }

// which: 0 vk (870 B), 1 h, 2 l, 3 a, 4 b_g1, 5 b_g2 ; copies min(cap, size) bytes, *size_out = full size
int32_t bzk_params_read(bzk_ctx* ctx, const bzk_params* slot, int32_t which, uint8_t* out, uint64_t cap, uint64_t* size_out) {
    if (!ctx || !slot || (cap && !out)) return BZK_E_ARG;
    const CrsShared* p = slot->crs;
    (void)hipSetDevice(ctx->device);
    const uint64_t m = (uint64_t)1 << p->log_m;
    const void* src = nullptr;
    uint64_t size = 0;
    switch (which) {
        case 0: size = 870; break;
        case 1: src = p->h; size = (m - 1) * 96; break;
        case 2: src = p->l; size = (uint64_t)p->n_aux * 96; break;
        case 3: src = p->a; size = (uint64_t)p->n_a * 96; break;
        case 4: src = p->b_g1; size = (uint64_t)p->n_b * 96; break;
        case 5: src = p->b_g2; size = (uint64_t)p->n_b * 192; break;
        default: return BZK_E_ARG;
    }
    if (size_out) *size_out = size;
    const uint64_t nbytes = cap < size ? cap : size;
    if (!nbytes) return BZK_OK;
    if (which == 0) {
        memcpy(out, p->vk, nbytes);
        return BZK_OK;
    }
    BZK_HIP(ctx, hipMemcpyAsync(out, src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    BZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BZK_OK;
}

int32_t bzk_groth16_h_dev(bzk_ctx* ctx, void* a, void* b, void* c, uint32_t log_m) {
    if (!ctx || !a || !b || !c) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    return groth16_h(ctx, a, b, c, log_m);
}

static int32_t groth16_prove_impl(bzk_ctx* ctx, bzk_params* p, const bzk_assignment* asg, const bzk::DeferData* dd, const bzk_staged* st, const uint8_t r32[32],
                                  const uint8_t s32[32], uint8_t proof[387]);
static int32_t groth16_prove_entry(bzk_ctx* ctx, bzk_params* p, const bzk_assignment* asg, const bzk::DeferData* dd, const bzk_staged* st, const uint8_t r32[32],
                                   const uint8_t s32[32], uint8_t proof[387]);

int32_t bzk_groth16_prove(bzk_ctx* ctx, bzk_params* p, const bzk_assignment* asg, const uint8_t r32[32], const uint8_t s32[32],
                          uint8_t proof[387]) {
    return groth16_prove_entry(ctx, p, asg, nullptr, nullptr, r32, s32, proof);
}
// The same over an R1CS instance of the host generator (bzk_mpn_*_synthesize), INCLUDING one whose hash-dependent values were deferred
// (bzk_mpn_set_defer): the arrays are uploaded as they are and the instance's DeferProgram fills the rest in on the device before any
// MSM or transform reads them.  BZK_E_UNSAT: a deferred constraint does not hold, or a transition's computed state differs from the one
// the witness builder predicted (the caller may synthesize again without deferral for the exact first unsatisfied row).
int32_t bzk_groth16_prove_r1cs(bzk_ctx* ctx, bzk_params* p, const bzk_r1cs* r, const uint8_t r32[32], const uint8_t s32[32], uint8_t proof[387]) {
    if (!r) return BZK_E_ARG;
    bzk_assignment asg;
    const bzk::DeferData* dd = nullptr;
    bzk::r1cs_assignment(r, &asg, &dd, nullptr);
    return groth16_prove_entry(ctx, p, &asg, dd, nullptr, r32, s32, proof);
}

// ---- staged assignments: the uploads and the deferred-value program in the PRODUCER's pipeline ------------------------------------------------
// bzk_r1cs_stage puts an instance's z | A.z | B.z | C.z into HBM on `ctx`'s stream - a context of the witness producer's, not a prover slot's - runs the
// instance's deferred-value program behind the uploads and returns at once; bzk_groth16_prove_staged (any context of the same device) waits for that work in
// stream order and copies device to device (128 MB at HBM rate) instead of uploading.  Why (profiles/r05_run8...): inside bzk_groth16_prove_r1cs the program is
// 15 ms during which its slot feeds the device no MSM work - 50 proofs/s with live producers on the deferred generator against 62 - 63 on the plain one.
int32_t bzk_r1cs_stage(bzk_ctx* ctx, const bzk_r1cs* r, bzk_staged** out) {
    if (!ctx || !r || !out) return BZK_E_ARG;
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    bzk_assignment asg;
    const bzk::DeferData* dd = nullptr;
    uint64_t n_in = 0;
    bzk::r1cs_assignment(r, &asg, &dd, &n_in);
    const size_t bytes = (size_t)(asg.n_vars + 3 * asg.n_rows) * 32;
    bzk_staged* st = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->staged_mu);
        for (size_t i = 0; i < ctx->staged_pool.size(); ++i)
            if (ctx->staged_pool[i]->cap >= bytes) {
                st = ctx->staged_pool[i];
                ctx->staged_pool.erase(ctx->staged_pool.begin() + (long)i);
                break;
            }
    }
    auto give_back = [&] {
        std::lock_guard<std::mutex> lk(ctx->staged_mu);
        ctx->staged_pool.push_back(st);
    };
    if (!st) {
        st = new bzk_staged();
        st->owner = ctx;
        if (hipMalloc(&st->buf, bytes) != hipSuccess || hipEventCreateWithFlags(&st->ready, hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc((void**)&st->flags_host, 64) != hipSuccess) {
            (void)hipGetLastError();
            if (st->buf) (void)hipFree(st->buf);
            if (st->ready) (void)hipEventDestroy(st->ready);
            if (st->flags_host) (void)hipHostFree(st->flags_host);
            delete st;
            return BZK_E_ALLOC;
        }
        st->cap = bytes;
    }
    st->n_vars = asg.n_vars;
    st->n_rows = asg.n_rows;
    *st->flags_host = 0;
    const uint8_t* src[4] = {asg.z, asg.az, asg.bz, asg.cz};
    void* dst[4] = {st->z(), st->ev(0), st->ev(1), st->ev(2)};
    const size_t len[4] = {(size_t)asg.n_vars * 32, (size_t)asg.n_rows * 32, (size_t)asg.n_rows * 32, (size_t)asg.n_rows * 32};
    int32_t rc = BZK_OK;
    for (int k = 0; k < 4 && rc == BZK_OK; ++k)
        if (hipMemcpyAsync(dst[k], src[k], len[k], hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = BZK_E_DEVICE;
    if (rc == BZK_OK && dd)
        rc = bzk::witfill_run_dev(ctx, *dd, bzk::wf::Arrays{(Fr*)st->z() + n_in, (Fr*)st->ev(0), (Fr*)st->ev(1), (Fr*)st->ev(2)}, st->flags_host);
    if (rc == BZK_OK && hipEventRecord(st->ready, ctx->stream) != hipSuccess) rc = BZK_E_DEVICE;
    if (rc != BZK_OK) {
        (void)hipStreamSynchronize(ctx->stream);  // nothing may still be reading the instance's arrays
        bzk::witfill_quiesce(ctx);
        (void)hipGetLastError();
        give_back();
        return rc;
    }
    *out = st;
    return BZK_OK;
}
// blocks until the staging work has finished (the instance's host arrays may be freed from here on).  BZK_E_UNSAT: see bzk_groth16_prove_r1cs
int32_t bzk_staged_wait(bzk_staged* st) {
    if (!st) return BZK_E_ARG;
    if (hipEventSynchronize(st->ready) != hipSuccess) return BZK_E_DEVICE;
    return *st->flags_host ? BZK_E_UNSAT : BZK_OK;
}
// read-back of a staged array (0 z, 1 A.z, 2 B.z, 3 C.z) after the staging work - uploads and, if the instance had one, the deferred-value program on
// the DEVICE - has finished: what the device fill produced, for consumers that pin it on fixtures of the complete arrays (tests/test_gpu_defer.py) or
// want the complete witness back on the host without running the program there.  *size_out = the array's size in bytes; at most `cap` bytes are copied.
// The flags of the program are not judged here (bzk_staged_wait does that): a violated instance can still be inspected.
int32_t bzk_staged_read(const bzk_staged* st, int32_t which, uint8_t* out, uint64_t cap, uint64_t* size_out) {
    if (!st || which < 0 || which > 3 || (cap && !out)) return BZK_E_ARG;
    (void)hipSetDevice(st->owner->device);
    const uint64_t size = (which == 0 ? st->n_vars : st->n_rows) * 32;
    if (size_out) *size_out = size;
    if (hipEventSynchronize(st->ready) != hipSuccess) return BZK_E_DEVICE;
    const uint64_t nbytes = cap < size ? cap : size;
    if (!nbytes) return BZK_OK;
    const void* src = which == 0 ? st->z() : st->ev(which - 1);
    if (hipMemcpy(out, src, nbytes, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return BZK_E_DEVICE;
    }
    return BZK_OK;
}
// hands the buffers back to the staging context (callable from any thread, once no prove call is using them)
void bzk_staged_free(bzk_staged* st) {
    if (!st) return;
    std::lock_guard<std::mutex> lk(st->owner->staged_mu);
    st->owner->staged_pool.push_back(st);
}
int32_t bzk_groth16_prove_staged(bzk_ctx* ctx, bzk_params* p, const bzk_staged* st, const uint8_t r32[32], const uint8_t s32[32], uint8_t proof[387]) {
    if (!st || !ctx || st->owner->device != ctx->device) return BZK_E_ARG;
    bzk_assignment asg;
    asg.z = (const uint8_t*)st->z();
    asg.az = (const uint8_t*)st->ev(0);
    asg.bz = (const uint8_t*)st->ev(1);
    asg.cz = (const uint8_t*)st->ev(2);
    asg.n_rows = st->n_rows;
    asg.n_vars = st->n_vars;
    return groth16_prove_entry(ctx, p, &asg, nullptr, st, r32, s32, proof);
}

static int32_t groth16_prove_entry(bzk_ctx* ctx, bzk_params* p, const bzk_assignment* asg, const bzk::DeferData* dd, const bzk_staged* st_in,
                                   const uint8_t r32[32], const uint8_t s32[32], uint8_t proof[387]) {
    if (!ctx || !p || !asg || !r32 || !s32 || !proof || !asg->z || !asg->az || !asg->bz || !asg->cz) return BZK_E_ARG;
    if (p->crs->device != ctx->device) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    crs_prepare(ctx, p->crs);
    int32_t st = groth16_prove_impl(ctx, p, asg, dd, st_in, r32, s32, proof);
    auto quiesce = [&] {
        // the caller frees (re-uses) the assignment arrays as soon as this returns: no copy out of them may still be in
        // flight, on the main stream or on a lane
        (void)hipStreamSynchronize(ctx->stream);
        bzk::witfill_quiesce(ctx);
        if (ctx->hprio) (void)hipStreamSynchronize(ctx->hprio);
        for (bzk_ctx* c : ctx->lanes) (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
    };
    if (st == BZK_E_ALLOC) {
        // a grow-only workspace did not fit beside the resident h table: if this slot is the CRS's only user, give the table's memory
        // back and prove once more through the per-call pipeline; a shared table cannot be pulled from under other slots
        quiesce();
        CrsShared* c = p->crs;
        bool dropped = false;
        {
            std::lock_guard<std::mutex> lk(c->m);
            bzk_msm_table* t = c->h_table.load(std::memory_order_acquire);
            if (t && c->refs.load() == 1) {
                c->h_table.store(nullptr, std::memory_order_release);
                bzk_msm_table_free(ctx, t);
                dropped = true;
            }
        }
        if (dropped) st = groth16_prove_impl(ctx, p, asg, dd, st_in, r32, s32, proof);
    }
    if (st != BZK_OK) quiesce();
    return st;
}

// explicit control of the h table (1: build it now if it fits, 0: drop it - only while this slot is the CRS's sole user)
int32_t bzk_params_h_table(bzk_ctx* ctx, bzk_params* p, int32_t on) {
    if (!ctx || !p) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    CrsShared* c = p->crs;
    if (on) {
        crs_prepare(ctx, c);
        std::lock_guard<std::mutex> lk(c->m);
        if (c->h_table.load(std::memory_order_acquire)) return BZK_OK;  // (the build below is serialised by c->m: one table per CRS)
        const uint64_t m = (uint64_t)1 << c->log_m;
        if (c->log_m < 16 || m <= 1) return BZK_E_ARG;
        const uint32_t cw = c->log_m > 20 ? 20u : c->log_m;
        // built into a local and published once complete and synchronised: other slots of this CRS read c->h_table without the lock
        //  - they see either no table or a finished one
        bzk_msm_table* t = nullptr;
        const int32_t st = bzk_msm_g1_table_build_c(ctx, c->h, m - 1, cw, &t);
        if (st == BZK_OK) c->h_table.store(t, std::memory_order_release);
        return st;
    }
    std::lock_guard<std::mutex> lk(c->m);
    c->prepared = true;  // an explicit "off" also keeps the first proof from building it
    bzk_msm_table* t = c->h_table.load(std::memory_order_acquire);
    if (!t) return BZK_OK;
    if (c->refs.load() != 1) return BZK_E_ARG;
    (void)hipStreamSynchronize(ctx->stream);
    for (bzk_ctx* l : ctx->lanes) (void)hipStreamSynchronize(l->stream);
    c->h_table.store(nullptr, std::memory_order_release);
    bzk_msm_table_free(ctx, t);
    return BZK_OK;
}

static int32_t groth16_prove_impl(bzk_ctx* ctx, bzk_params* slot, const bzk_assignment* asg, const bzk::DeferData* dd, const bzk_staged* staged,
                                  const uint8_t r32[32], const uint8_t s32[32], uint8_t proof[387]) {
    // a staged assignment is device memory another stream has been filling: this stream waits for that work, and every "upload" below is a device-to-device copy
    const hipMemcpyKind up = staged ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (staged) BZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, staged->ready, 0));
    const CrsShared* p = slot->crs;
    const uint64_t m = (uint64_t)1 << p->log_m, nv = (uint64_t)p->n_in + p->n_aux;
    if (asg->n_rows > m) return BZK_E_ARG;
    if (asg->n_vars != nv) {  // an R1CS of another circuit shape than the CRS: refuse instead of reading past `z`
        ctx->last_error = "groth16_prove: assignment has " + std::to_string(asg->n_vars) + " variables, the parameters " + std::to_string(nv);
        return BZK_E_ARG;
    }
    // Schedule: the five MSMs are independent, and each ends in a latency-bound tail (bucket reduction, window
    // sums, read-back) that leaves most CUs idle.  They run on separate lanes (stream + workspace each, one PERSISTENT host
    // thread per lane - bzk::lane_post - since round 3; three std::thread creations per proof before) so that one MSM's tail
    // overlaps another's accumulation; l, a, b only need z, so they start while az/bz/cz are still being copied and the h
    // polynomial is computed on the main stream.
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    // env BZK_PROVE_LANES=4: b_g1 on a lane of its own instead of behind l (A/B: the end of a proof is then four tails side by side)
    static const int n_lanes = [] { const char* e = getenv("BZK_PROVE_LANES"); return e && atoi(e) == 4 ? 4 : 3; }();
    bzk_ctx* lane[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < n_lanes; ++i)
        if (!(lane[i] = bzk::ctx_lane(ctx, (size_t)i))) return BZK_E_DEVICE;
    BZK_HIP(ctx, hipMemcpyAsync(slot->d_z, asg->z, nv * 32, up, ctx->stream));
    if (dd) {
        // deferred witness values: z is not complete until the instance's program has run, and the program also writes into the three
        // evaluation arrays - so those are staged now (main_part skips them) and every consumer below is behind the fill in stream order
        void* ev[3] = {slot->d_a, slot->d_b, slot->d_c};
        const uint8_t* hv[3] = {asg->az, asg->bz, asg->cz};
        for (int k = 0; k < 3; ++k) {
            BZK_HIP(ctx, hipMemcpyAsync(ev[k], hv[k], asg->n_rows * 32, hipMemcpyHostToDevice, ctx->stream));
            if (m > asg->n_rows) BZK_HIP(ctx, hipMemsetAsync((char*)ev[k] + asg->n_rows * 32, 0, (m - asg->n_rows) * 32, ctx->stream));
        }
        BZK_TRY(bzk::witfill_run_dev(ctx, *dd, bzk::wf::Arrays{(Fr*)slot->d_z + p->n_in, (Fr*)slot->d_a, (Fr*)slot->d_b, (Fr*)slot->d_c}));
    }
    // density-filtered scalar vectors
    if (p->n_a)
        BZK_LAUNCH(ctx, "g16_gather", g16_gather_kernel, dim3((p->n_a + 255) / 256), dim3(256), 0, (const Fr*)slot->d_z, p->a_idx, p->n_a, (Fr*)slot->d_sa);
    if (p->n_b)
        BZK_LAUNCH(ctx, "g16_gather", g16_gather_kernel, dim3((p->n_b + 255) / 256), dim3(256), 0, (const Fr*)slot->d_z, p->b_idx, p->n_b, (Fr*)slot->d_sb);
    // l + r b_g1 as ONE MSM over the base set [l | b_g1] with the scalars (z_aux | r z_b), laid out contiguously behind z: one bucket
    // reduction, one set of window sums, one de-duplication pass and one read-back instead of two (a witness MSM of this size is
    // ~1.2 ms of accumulation inside ~3.8 ms of mostly latency-bound phases), and r b_g1 needs no host scalar multiplication.
    const bool merged_lb1 = p->rlb1 != nullptr && n_lanes == 3;
    if (merged_lb1) {
        Fr r_mont;
        memcpy(r_mont.l, r32, 32);
        BZK_LAUNCH(ctx, "g16_gather_mul", g16_gather_mul_kernel, dim3((p->n_b + 255) / 256), dim3(256), 0, (const Fr*)slot->d_z, p->b_idx, p->n_b,
                   r_mont, (Fr*)slot->d_z + nv);
    }
    if (!ctx->ev_z) BZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_z, hipEventDisableTiming));
    const hipEvent_t z_ready = ctx->ev_z;
    BZK_HIP(ctx, hipEventRecord(z_ready, ctx->stream));
    for (int i = 0; i < n_lanes; ++i) BZK_HIP(ctx, hipStreamWaitEvent(lane[i]->stream, z_ready, 0));
    const auto t1 = clk::now();
    uint8_t pH[97], pL[97], pA[97], pB1[97], pB2[193];
    int32_t st[4] = {BZK_OK, BZK_OK, BZK_OK, BZK_OK};
    const int dev = ctx->device;
    // env BZK_PROVE_SERIAL=1: the same five MSMs one after the other on the lanes' streams (clean per-kernel event
    // timings for profiling; the lanes otherwise overlap and stretch each other's intervals)
    // the witness MSMs (l, a, b_g1, b_g2) run on de-duplicated scalars: half of a Groth16 assignment is repeats
    // (msm_impl.cuh section 8); env BZK_PROVE_NODEDUP=1 switches that off for A/B measurements
    // BZK_F_THROUGHPUT: the five MSMs overlap each other and the neighbouring proofs of a pipelined prover, so the forms with
    // less arithmetic beat those with the shortest chain (env BZK_PROVE_LATENCY=1 switches the hint off for A/B measurements)
    static const uint32_t tflag = (env_on("BZK_PROVE_LATENCY")) ? 0u : BZK_F_THROUGHPUT;
    static const uint32_t wflags = ((env_on("BZK_PROVE_NODEDUP")) ? 0u : BZK_F_DEDUP) | tflag;
    static const bool serial = env_on("BZK_PROVE_SERIAL");
    const void* z_aux = (const char*)slot->d_z + (size_t)p->n_in * 32;
    // every query through its resident set where one was built (crs_prepare), through the raw bases otherwise
    auto g1 = [&](bzk_ctx* c, const bzk_msm_bases* res, const void* raw, const void* sc, uint64_t n, uint32_t fl, uint8_t* out) {
        return res ? bzk_msm_g1_bases_run_dev(c, res, sc, n, fl, out) : bzk_msm_g1_dev(c, raw, sc, n, fl, out);
    };
    double lane_done_ms[4] = {0, 0, 0, 0};  // BZK_TIMING: when each lane's host thread returned, relative to t0
    auto stamp = [&](int i) { lane_done_ms[i] = std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    // the (r, s)-only part of the assembly: host work on a thread of its own (one past the MSM lanes) while the device computes
    Fr r_canon, s_canon;
    {
        Fr r, s;
        memcpy(r.l, r32, 32);
        memcpy(s.l, s32, 32);
        r_canon = fe_from_mont<FrParams>(r);
        s_canon = fe_from_mont<FrParams>(s);
    }
    // g_c = s g_a + r (beta_g1 + b_g1) + h + l = [s (alpha + r delta) + r beta_g1] + s a + r b_g1 + h + l: the bracket needs no MSM
    // result (job_pre), s a and r b_g1 are computed by the lane threads that own a and b_g1 as soon as their MSM has returned -
    // what is left after the join is seven point additions and the packing.
    HG1 ga_pre, gc_pre, s_a, r_b1;
    HG2 gb_pre;
    auto job_pre = [&] {
        const uint8_t* vk = p->vk;
        const HG1 alpha = to_host_fast<FpOps>(unpack_g1(vk)), beta1 = to_host_fast<FpOps>(unpack_g1(vk + 97));
        ga_pre = host_mul_fr<HFpOps>(to_host_fast<FpOps>(unpack_g1(vk + 580)), r_canon);  // r delta_g1 + alpha
        xyzz_add<HFpOps>(ga_pre, alpha);
        gc_pre = host_mul2_fr<HFpOps>(ga_pre, s_canon, beta1, r_canon);
        gb_pre = host_mul_fr<HFp2Ops>(to_host_fast<Fp2Ops>(unpack_g2(vk + 677)), s_canon);  // s delta_g2 + beta_g2
        xyzz_add<HFp2Ops>(gb_pre, to_host_fast<Fp2Ops>(unpack_g2(vk + 194)));
    };
    auto after_a = [&] { if (st[2] == BZK_OK) s_a = host_mul_fr<HFpOps>(to_host_fast<FpOps>(unpack_g1(pA)), s_canon); };
    auto after_b1 = [&](int lane_i) { if (st[lane_i] == BZK_OK) r_b1 = host_mul_fr<HFpOps>(to_host_fast<FpOps>(unpack_g1(pB1)), r_canon); };
    auto job0 = [&] {
        (void)hipSetDevice(dev);
        st[0] = p->rb2 ? bzk_msm_g2_bases_run_dev(lane[0], p->rb2, slot->d_sb, p->n_b, wflags, pB2)
                       : bzk_msm_g2_dev(lane[0], p->b_g2, slot->d_sb, p->n_b, wflags, pB2);
        stamp(0);
    };
    auto job1a = [&] {
        if (merged_lb1) {  // pL = l + r b_g1
            st[1] = bzk_msm_g1_bases_run_dev(lane[1], p->rlb1, z_aux, (uint64_t)p->n_aux + p->n_b, wflags, pL);
            r_b1 = xyzz_identity<HFpOps>();
        } else {
            st[1] = g1(lane[1], p->rl, p->l, z_aux, p->n_aux, wflags, pL);
        }
    };
    auto job1b = [&] { if (!merged_lb1 && st[1] == BZK_OK) st[1] = g1(lane[1], p->rb1, p->b_g1, slot->d_sb, p->n_b, wflags, pB1); };
    auto job1 = [&] {
        (void)hipSetDevice(dev);
        job1a();
        if (n_lanes == 3) job1b();
        stamp(1);
        if (n_lanes == 3 && !merged_lb1) after_b1(1);
    };
    auto job3 = [&] {  // four-lane form: b_g1 beside l
        (void)hipSetDevice(dev);
        st[3] = g1(lane[3], p->rb1, p->b_g1, slot->d_sb, p->n_b, wflags, pB1);
        stamp(3);
        after_b1(3);
    };
    auto job2 = [&] {
        (void)hipSetDevice(dev);
        st[2] = g1(lane[2], p->ra, p->a, slot->d_sa, p->n_a, wflags, pA);
        stamp(2);
        after_a();
    };
    if (serial) {
        auto dump = [&](int i, const char* what) {
            if (!ctx->timing || !ctx->prof) return;
            char buf[2048];
            if (bzk_prof_dump(lane[i], buf, sizeof buf) == BZK_OK) {
                for (char* c = buf; *c; ++c)
                    if (*c == '\n') *c = ';';
                fprintf(stderr, "[bzk] serial %s: %s\n", what, buf);
            }
            (void)bzk_prof_reset(lane[i]);
        };
        job0(); dump(0, "b_g2");
        (void)hipSetDevice(dev);
        job1a(); dump(1, "l");
        job1b(); dump(1, "b_g1");
        if (!merged_lb1) after_b1(1);
        job2(); dump(2, "a");
        job_pre();
    } else {
        bzk::lane_post(ctx, 0, job0);
        bzk::lane_post(ctx, 1, job1);
        bzk::lane_post(ctx, 2, job2);
        if (n_lanes == 4) bzk::lane_post(ctx, 3, job3);
        bzk::lane_post(ctx, (size_t)n_lanes, job_pre);
    }
    // main stream: stage the evaluations, h polynomial, h MSM
    int32_t st_main = BZK_OK;
    // BZK_TIMING: device-side marks on the main stream (start of the proof, evaluations staged, h polynomial done)
    struct Marks {
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        ~Marks() {
            for (auto& x : e)
                if (x) (void)hipEventDestroy(x);
        }
    } marks;
    hipEvent_t* const tm = marks.e;
    if (ctx->timing)
        for (auto& e : marks.e)
            if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    // env BZK_PROVE_H_PRIO=1: the h polynomial on a highest-priority side stream.  Measured (profiles/r03_run21...): the transforms then
    // finish 2.2 ms into the proof instead of 11 ms, the h MSM behind them takes the time back (a proof is ~17 ms of throughput-bound
    // work on four streams whichever order it runs in): single proof and 4-slot rate unchanged, so the default stays the plain stream
    static const bool h_prio = [] { const char* e = getenv("BZK_PROVE_H_PRIO"); return e && atoi(e) != 0; }();
    if (h_prio && !ctx->hprio) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&ctx->hprio, hipStreamNonBlocking, hi) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_h, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (ctx->hprio) (void)hipStreamDestroy(ctx->hprio);
            ctx->hprio = nullptr;  // no side stream: everything on the main stream
        }
    }
    auto main_part = [&]() -> int32_t {
        void* ev[3] = {slot->d_a, slot->d_b, slot->d_c};
        const uint8_t* hv[3] = {asg->az, asg->bz, asg->cz};
        // the evaluations and the transforms on the high-priority side stream (BZK_LAUNCH and the transform code work on ctx->stream: it
        // is pointed at the side stream for this stretch); d_a/d_b/d_c are this slot's own and idle since the last proof's final sync
        struct StreamSwap {
            bzk_ctx* c;
            hipStream_t saved;
            StreamSwap(bzk_ctx* c_, hipStream_t s) : c(c_), saved(c_->stream) { if (s) c->stream = s; }
            ~StreamSwap() { c->stream = saved; }
        };
        const bool side = h_prio && ctx->hprio && !dd && !staged;
        {
        StreamSwap swap(ctx, side ? ctx->hprio : nullptr);
        if (tm[0]) (void)hipEventRecord(tm[0], ctx->stream);
        for (int k = 0; k < 3 && !dd; ++k) {
            BZK_HIP(ctx, hipMemcpyAsync(ev[k], hv[k], asg->n_rows * 32, up, ctx->stream));
            if (m > asg->n_rows)
                BZK_HIP(ctx, hipMemsetAsync((char*)ev[k] + asg->n_rows * 32, 0, (m - asg->n_rows) * 32, ctx->stream));
        }
        if (tm[1]) (void)hipEventRecord(tm[1], ctx->stream);
        BZK_TRY(groth16_h(ctx, slot->d_a, slot->d_b, slot->d_c, p->log_m));
        if (tm[2]) (void)hipEventRecord(tm[2], ctx->stream);
        if (side) BZK_HIP(ctx, hipEventRecord(ctx->ev_h, ctx->stream));
        }
        if (side) BZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_h, 0));
        if (bzk_msm_table* ht = p->h_table.load(std::memory_order_acquire)) return bzk_msm_g1_table_run_dev(ctx, ht, slot->d_a, m - 1, tflag, pH);
        return g1(ctx, p->rh, p->h, slot->d_a, m - 1, tflag, pH);
    };
    const auto t2 = clk::now();
    st_main = main_part();
    const auto t3 = clk::now();
    if (!serial)
        for (size_t i = 0; i <= (size_t)n_lanes; ++i) bzk::lane_wait(ctx, i);  // always: the jobs reference this frame
    const auto t4 = clk::now();
    if (st_main != BZK_OK) return st_main;
    for (int i = 0; i < n_lanes; ++i)
        if (st[i] != BZK_OK) {
            ctx->last_error = "lane " + std::to_string(i) + ": " + lane[i]->last_error;
            return st[i];
        }
    if (dd || staged) {  // (the h MSM's read-back has synchronised the main stream: the program's flags word has landed)
        const uint32_t f = staged ? *staged->flags_host : bzk::witfill_flags(ctx);
        if (f) {
            ctx->last_error = std::string("groth16_prove: deferred witness values -") + ((f & bzk::wf::FLAG_UNSATISFIED) ? " a deferred constraint does not hold" : "") +
                              ((f & bzk::wf::FLAG_CHAIN) ? " a transition's computed state differs from the witness builder's prediction" : "");
            return BZK_E_UNSAT;
        }
    }
    // assembly (host): g_a = (alpha + r delta) + a;  g_b = (beta + s delta) + b;  g_c = gc_pre + s a + r b_g1 + h + l
    // (= bellman's h + l + s g_a + r g_b1 - r s delta with g_b1 = beta_g1 + b_g1 + s delta)
    HG1 ga = ga_pre;
    xyzz_add<HFpOps>(ga, to_host_fast<FpOps>(unpack_g1(pA)));
    HG2 gb = gb_pre;
    xyzz_add<HFp2Ops>(gb, to_host_fast<Fp2Ops>(unpack_g2(pB2)));
    HG1 gc = gc_pre;
    xyzz_add<HFpOps>(gc, s_a);
    xyzz_add<HFpOps>(gc, r_b1);
    xyzz_add<HFpOps>(gc, to_host_fast<FpOps>(unpack_g1(pH)));
    xyzz_add<HFpOps>(gc, to_host_fast<FpOps>(unpack_g1(pL)));
    pack_proof(ga, gb, gc, proof);
    if (ctx->timing) {
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[bzk] groth16_prove: z staged %.2f ms, lanes started %.2f, h chain %.2f, lanes joined +%.2f, assembly %.2f; lanes done at "
                        "b_g2 %.2f, l%s %.2f, a %.2f, b_g1 %.2f, main %.2f ms\n",
                ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, clk::now()), lane_done_ms[0], n_lanes == 3 ? "+b_g1" : "", lane_done_ms[1],
                lane_done_ms[2], lane_done_ms[3], ms(t0, t3));
        if (tm[0] && tm[1] && tm[2]) {
            float e01 = 0, e12 = 0;
            (void)hipEventElapsedTime(&e01, tm[0], tm[1]);
            (void)hipEventElapsedTime(&e12, tm[1], tm[2]);
            fprintf(stderr, "[bzk] groth16_prove main stream: az/bz/cz staged in %.2f ms after z and the gathers, h polynomial %.2f ms, h MSM the rest\n", e01, e12);
        }
    }

    return BZK_OK;
}

}  // extern "C"

// internal hooks for setup.hip (CRS generated in place on the device)
int32_t bzk_params_alloc_internal(bzk_ctx* ctx, const bzk_params_desc* d, bzk_params** out) { return params_build(ctx, d, false, out); }
void bzk_params_buffers_internal(bzk_params* p, void** h, void** l, void** a, void** b_g1, void** b_g2) {
    *h = p->crs->h; *l = p->crs->l; *a = p->crs->a; *b_g1 = p->crs->b_g1; *b_g2 = p->crs->b_g2;
}
void bzk_params_set_vk_internal(bzk_params* p, const uint8_t vk[870]) { memcpy(p->crs->vk, vk, 870); }
