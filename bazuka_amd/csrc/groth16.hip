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

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "bzk_curve.cuh"
#include "bzk_internal.h"

struct bzk_params {
    uint32_t n_in, n_aux, log_m, n_a, n_b;
    uint8_t vk[870];
    void *h, *l, *a, *b_g1, *b_g2;      // device CRS
    uint32_t *a_idx, *b_idx;            // device: variable index of each dense entry
    void *d_z, *d_a, *d_b, *d_c, *d_sa, *d_sb;  // device scratch sized for this circuit
    // static-base table of the h query (built on the first proof): the h bases never change and their scalars are never
    // de-duplicated, so all windows can share one bucket set at a window size of ~log2 m (fewer windows = fewer additions)
    bzk_msm_table* h_table;
    bool h_table_tried;
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

template <class F>
static XyzzT<F> host_mul_fr(const XyzzT<F>& p, const Fr& k_mont) {
    Fr k = fe_from_mont<FrParams>(k_mont);
    XyzzT<F> r = xyzz_identity<F>();
    for (int i = 254; i >= 0; --i) {
        r = xyzz_dbl<F>(r);
        if ((k.l[i >> 5] >> (i & 31)) & 1) xyzz_add<F>(r, p);
    }
    return r;
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
static void pack_g1(const G1Xyzz& p, uint8_t* out) {
    G1Affine a;
    bool fin = xyzz_to_affine<FpOps>(p, a);
    memcpy(out, a.x.l, 48);
    memcpy(out + 48, a.y.l, 48);
    out[96] = fin ? 0 : 1;
}
static void pack_g2(const G2Xyzz& p, uint8_t* out) {
    G2Affine a;
    bool fin = xyzz_to_affine<Fp2Ops>(p, a);
    memcpy(out, a.x.c0.l, 48);
    memcpy(out + 48, a.x.c1.l, 48);
    memcpy(out + 96, a.y.c0.l, 48);
    memcpy(out + 144, a.y.c1.l, 48);
    out[192] = fin ? 0 : 1;
}

}  // namespace bzk

using namespace bzk;

extern "C" {

void bzk_params_free(bzk_ctx* ctx, bzk_params* p) {
    if (!p) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (p->h_table) bzk_msm_table_free(ctx, p->h_table);
    void* bufs[] = {p->h, p->l, p->a, p->b_g1, p->b_g2, p->a_idx, p->b_idx, p->d_z, p->d_a, p->d_b, p->d_c, p->d_sa, p->d_sb};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete p;
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
    bzk_params* p = new (std::nothrow) bzk_params();
    if (!p) return BZK_E_ALLOC;
    memset(p, 0, sizeof(*p));
    p->n_in = d->n_in; p->n_aux = d->n_aux; p->log_m = d->log_m; p->n_a = d->n_a; p->n_b = d->n_b;
    memcpy(p->vk, d->vk, 870);
    struct Up { void** dst; const void* src; size_t bytes; };
    Up ups[] = {
        {&p->h, upload_crs ? d->h : nullptr, (size_t)(m - 1) * 96}, {&p->l, upload_crs ? d->l : nullptr, (size_t)d->n_aux * 96},
        {&p->a, upload_crs ? d->a : nullptr, (size_t)d->n_a * 96}, {&p->b_g1, upload_crs ? d->b_g1 : nullptr, (size_t)d->n_b * 96},
        {&p->b_g2, upload_crs ? d->b_g2 : nullptr, (size_t)d->n_b * 192},
        {(void**)&p->a_idx, ia.data(), ia.size() * 4}, {(void**)&p->b_idx, ib.data(), ib.size() * 4},
        {&p->d_z, nullptr, (size_t)nv * 32}, {&p->d_a, nullptr, (size_t)m * 32}, {&p->d_b, nullptr, (size_t)m * 32},
        {&p->d_c, nullptr, (size_t)m * 32}, {&p->d_sa, nullptr, (size_t)d->n_a * 32}, {&p->d_sb, nullptr, (size_t)d->n_b * 32},
    };
    for (auto& u : ups) {
        hipError_t e = hipMalloc(u.dst, u.bytes ? u.bytes : 32);
        if (e == hipSuccess && u.src && u.bytes) e = hipMemcpyAsync(*u.dst, u.src, u.bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) {
            ctx->last_error = std::string("params_load: ") + hipGetErrorString(e);
            (void)hipGetLastError();
            bzk_params_free(ctx, p);
            return BZK_E_ALLOC;
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
        bzk_params_free(ctx, p);
        return BZK_E_DEVICE;
    }
    *out = p;
    return BZK_OK;
}

int32_t bzk_params_load(bzk_ctx* ctx, const bzk_params_desc* d, bzk_params** out) { return params_build(ctx, d, true, out); }

// which: 0 vk (870 B), 1 h, 2 l, 3 a, 4 b_g1, 5 b_g2 ; copies min(cap, size) bytes, *size_out = full size
int32_t bzk_params_read(bzk_ctx* ctx, const bzk_params* p, int32_t which, uint8_t* out, uint64_t cap, uint64_t* size_out) {
    if (!ctx || !p || (cap && !out)) return BZK_E_ARG;
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

static int32_t groth16_prove_impl(bzk_ctx* ctx, bzk_params* p, const bzk_assignment* asg, const uint8_t r32[32], const uint8_t s32[32],
                                  uint8_t proof[387]);

int32_t bzk_groth16_prove(bzk_ctx* ctx, bzk_params* p, const bzk_assignment* asg, const uint8_t r32[32], const uint8_t s32[32],
                          uint8_t proof[387]) {
    if (!ctx || !p || !asg || !r32 || !s32 || !proof || !asg->z || !asg->az || !asg->bz || !asg->cz) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    const int32_t st = groth16_prove_impl(ctx, p, asg, r32, s32, proof);
    if (st != BZK_OK) {
        // the caller frees (re-uses) the assignment arrays as soon as this returns: no copy out of them may still be in
        // flight, on the main stream or on a lane
        (void)hipStreamSynchronize(ctx->stream);
        for (bzk_ctx* c : ctx->lanes) (void)hipStreamSynchronize(c->stream);
        (void)hipGetLastError();
    }
    return st;
}

static int32_t groth16_prove_impl(bzk_ctx* ctx, bzk_params* p, const bzk_assignment* asg, const uint8_t r32[32], const uint8_t s32[32],
                                  uint8_t proof[387]) {
    const uint64_t m = (uint64_t)1 << p->log_m, nv = (uint64_t)p->n_in + p->n_aux;
    if (asg->n_rows > m) return BZK_E_ARG;
    if (asg->n_vars != nv) {  // an R1CS of another circuit shape than the CRS: refuse instead of reading past `z`
        ctx->last_error = "groth16_prove: assignment has " + std::to_string(asg->n_vars) + " variables, the parameters " + std::to_string(nv);
        return BZK_E_ARG;
    }
    // Schedule: the five MSMs are independent, and each ends in a latency-bound tail (bucket reduction, window
    // sums, 97-byte read-back) that leaves most CUs idle.  They run on separate lanes (stream + workspace each,
    // one host thread per lane) so that one MSM's tail overlaps another's accumulation; l, a, b only need z, so
    // they start while az/bz/cz are still being copied and the h polynomial is computed on the main stream.
    // h-query table: 2^16 .. 2^21 domains (13 levels x 112 B x m: 1.5 GB at 2^20, 3 GB at 2^21 - the production 2^24 domain would
    // need 24 GB per prover slot and keeps the per-call pipeline unless env BZK_PROVE_H_TABLE_MAX_LOG raises the limit - a deployment
    // with one or two slots per GPU can afford it); env BZK_PROVE_H_TABLE=0 switches the table off (A/B runs)
    if (!p->h_table_tried) {
        p->h_table_tried = true;
        static const bool want = [] { const char* e = getenv("BZK_PROVE_H_TABLE"); return !e || atoi(e) != 0; }();
        static const uint32_t max_log = [] {
            const char* e = getenv("BZK_PROVE_H_TABLE_MAX_LOG");
            const int v = e ? atoi(e) : 21;
            return (uint32_t)(v < 16 ? 16 : (v > 26 ? 26 : v));
        }();
        if (want && p->log_m >= 16 && p->log_m <= max_log) {
            const uint32_t c = p->log_m > 20 ? 20u : p->log_m;
            if (bzk_msm_g1_table_build_c(ctx, p->h, m - 1, c, &p->h_table) != BZK_OK) {
                p->h_table = nullptr;  // not enough memory: the per-call pipeline works without it
                (void)hipGetLastError();
            }
        }
    }
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    bzk_ctx* lane[3];
    for (int i = 0; i < 3; ++i)
        if (!(lane[i] = bzk::ctx_lane(ctx, (size_t)i))) return BZK_E_DEVICE;
    BZK_HIP(ctx, hipMemcpyAsync(p->d_z, asg->z, nv * 32, hipMemcpyHostToDevice, ctx->stream));
    // density-filtered scalar vectors
    if (p->n_a)
        BZK_LAUNCH(ctx, "g16_gather", g16_gather_kernel, dim3((p->n_a + 255) / 256), dim3(256), 0, (const Fr*)p->d_z, p->a_idx, p->n_a, (Fr*)p->d_sa);
    if (p->n_b)
        BZK_LAUNCH(ctx, "g16_gather", g16_gather_kernel, dim3((p->n_b + 255) / 256), dim3(256), 0, (const Fr*)p->d_z, p->b_idx, p->n_b, (Fr*)p->d_sb);
    if (!ctx->ev_z) BZK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_z, hipEventDisableTiming));
    const hipEvent_t z_ready = ctx->ev_z;
    BZK_HIP(ctx, hipEventRecord(z_ready, ctx->stream));
    for (int i = 0; i < 3; ++i) BZK_HIP(ctx, hipStreamWaitEvent(lane[i]->stream, z_ready, 0));
    const auto t1 = clk::now();
    uint8_t pH[97], pL[97], pA[97], pB1[97], pB2[193];
    int32_t st[3] = {BZK_OK, BZK_OK, BZK_OK};
    const int dev = ctx->device;
    // env BZK_PROVE_SERIAL=1: the same five MSMs one after the other on the lanes' streams (clean per-kernel event
    // timings for profiling; the lanes otherwise overlap and stretch each other's intervals)
    // the witness MSMs (l, a, b_g1, b_g2) run on de-duplicated scalars: half of a Groth16 assignment is repeats
    // (msm_impl.cuh section 8); env BZK_PROVE_NODEDUP=1 switches that off for A/B measurements
    // BZK_F_THROUGHPUT: the five MSMs overlap each other and the neighbouring proofs of a pipelined prover, so the forms with
    // less arithmetic beat those with the shortest chain (env BZK_PROVE_LATENCY=1 switches the hint off for A/B measurements)
    static const uint32_t tflag = (getenv("BZK_PROVE_LATENCY") && atoi(getenv("BZK_PROVE_LATENCY")) != 0) ? 0u : BZK_F_THROUGHPUT;
    static const uint32_t wflags = ((getenv("BZK_PROVE_NODEDUP") && atoi(getenv("BZK_PROVE_NODEDUP")) != 0) ? 0u : BZK_F_DEDUP) | tflag;
    static const bool serial = getenv("BZK_PROVE_SERIAL") && atoi(getenv("BZK_PROVE_SERIAL")) != 0;
    std::thread th[3];
    auto job0 = [&] {
        (void)hipSetDevice(dev);
        st[0] = bzk_msm_g2_dev(lane[0], p->b_g2, p->d_sb, p->n_b, wflags, pB2);
    };
    auto job1 = [&] {
        (void)hipSetDevice(dev);
        st[1] = bzk_msm_g1_dev(lane[1], p->l, (const char*)p->d_z + (size_t)p->n_in * 32, p->n_aux, wflags, pL);
        if (st[1] == BZK_OK) st[1] = bzk_msm_g1_dev(lane[1], p->b_g1, p->d_sb, p->n_b, wflags, pB1);
    };
    auto job2 = [&] {
        (void)hipSetDevice(dev);
        st[2] = bzk_msm_g1_dev(lane[2], p->a, p->d_sa, p->n_a, wflags, pA);
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
        st[1] = bzk_msm_g1_dev(lane[1], p->l, (const char*)p->d_z + (size_t)p->n_in * 32, p->n_aux, wflags, pL);
        dump(1, "l");
        if (st[1] == BZK_OK) st[1] = bzk_msm_g1_dev(lane[1], p->b_g1, p->d_sb, p->n_b, wflags, pB1);
        dump(1, "b_g1");
        job2(); dump(2, "a");
    } else {
        th[0] = std::thread(job0);
        th[1] = std::thread(job1);
        th[2] = std::thread(job2);
    }
    // main stream: stage the evaluations, h polynomial, h MSM
    int32_t st_main = BZK_OK;
    auto main_part = [&]() -> int32_t {
        void* ev[3] = {p->d_a, p->d_b, p->d_c};
        const uint8_t* hv[3] = {asg->az, asg->bz, asg->cz};
        for (int k = 0; k < 3; ++k) {
            BZK_HIP(ctx, hipMemcpyAsync(ev[k], hv[k], asg->n_rows * 32, hipMemcpyHostToDevice, ctx->stream));
            if (m > asg->n_rows)
                BZK_HIP(ctx, hipMemsetAsync((char*)ev[k] + asg->n_rows * 32, 0, (m - asg->n_rows) * 32, ctx->stream));
        }
        BZK_TRY(groth16_h(ctx, p->d_a, p->d_b, p->d_c, p->log_m));
        if (p->h_table) return bzk_msm_g1_table_run_dev(ctx, p->h_table, p->d_a, m - 1, tflag, pH);
        return bzk_msm_g1_dev(ctx, p->h, p->d_a, m - 1, tflag, pH);
    };
    const auto t2 = clk::now();
    st_main = main_part();
    const auto t3 = clk::now();
    if (!serial)
        for (auto& t : th) t.join();
    const auto t4 = clk::now();
    if (st_main != BZK_OK) return st_main;
    for (int i = 0; i < 3; ++i)
        if (st[i] != BZK_OK) {
            ctx->last_error = "lane " + std::to_string(i) + ": " + lane[i]->last_error;
            return st[i];
        }
    // assembly (host)
    Fr r, s;
    memcpy(r.l, r32, 32);
    memcpy(s.l, s32, 32);
    Fr rs = fe_mul<FrParams>(r, s);
    const uint8_t* vk = p->vk;
    G1Xyzz alpha = unpack_g1(vk), beta1 = unpack_g1(vk + 97), delta1 = unpack_g1(vk + 580);
    G2Xyzz beta2 = unpack_g2(vk + 194), delta2 = unpack_g2(vk + 677);
    G1Xyzz H = unpack_g1(pH), L = unpack_g1(pL), A = unpack_g1(pA), B1 = unpack_g1(pB1);
    G2Xyzz B2 = unpack_g2(pB2);
    G1Xyzz ga = host_mul_fr<FpOps>(delta1, r);
    xyzz_add<FpOps>(ga, alpha);
    xyzz_add<FpOps>(ga, A);
    G2Xyzz gb = host_mul_fr<Fp2Ops>(delta2, s);
    xyzz_add<Fp2Ops>(gb, beta2);
    xyzz_add<Fp2Ops>(gb, B2);
    G1Xyzz gc = host_mul_fr<FpOps>(delta1, rs);
    G1Xyzz t;
    t = host_mul_fr<FpOps>(alpha, s);  xyzz_add<FpOps>(gc, t);
    t = host_mul_fr<FpOps>(beta1, r);  xyzz_add<FpOps>(gc, t);
    t = host_mul_fr<FpOps>(A, s);      xyzz_add<FpOps>(gc, t);
    t = host_mul_fr<FpOps>(B1, r);     xyzz_add<FpOps>(gc, t);
    xyzz_add<FpOps>(gc, H);
    xyzz_add<FpOps>(gc, L);
    pack_g1(ga, proof);
    pack_g2(gb, proof + 97);
    pack_g1(gc, proof + 290);
    if (ctx->timing) {
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[bzk] groth16_prove: z staged %.2f ms, lanes started %.2f, h chain %.2f, lanes joined +%.2f, assembly %.2f\n",
                ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, clk::now()));
    }
    return BZK_OK;
}

}  // extern "C"

// internal hooks for setup.hip (CRS generated in place on the device)
int32_t bzk_params_alloc_internal(bzk_ctx* ctx, const bzk_params_desc* d, bzk_params** out) { return params_build(ctx, d, false, out); }
void bzk_params_buffers_internal(bzk_params* p, void** h, void** l, void** a, void** b_g1, void** b_g2) {
    *h = p->h; *l = p->l; *a = p->a; *b_g1 = p->b_g1; *b_g2 = p->b_g2;
}
void bzk_params_set_vk_internal(bzk_params* p, const uint8_t vk[870]) { memcpy(p->vk, vk, 870); }
