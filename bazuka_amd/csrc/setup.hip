// Groth16 CRS generation on the GPU (the reference's dev-mode counterpart is
// `bellman::groth16::generate_random_parameters` driven from src/config/blockchain.rs:355-417 with a
// seeded ChaChaRng, and MpnCircuit::empty(L, T, B) as the circuit - src/mpn/circuits/mod.rs:10-12).
//
// Layout of the result = bellman 0.14 `Parameters`:
//   L_k(tau)            : inverse NTT of (1, tau, tau^2, ...)                         [ntt.hip kernels]
//   A_v, B_v, C_v (tau) : sparse transposed mat-vec over the R1CS rows                 [host, O(nnz)]
//   h[i]   = tau^i (tau^m - 1)/delta * G1            i < m-1
//   l[v]   = (beta A_v + alpha B_v + C_v)/delta * G1 aux v ;  ic[v] = (...)/gamma * G1 input v
//   a[v]   = A_v * G1 ; b_g1[v] = B_v * G1 ; b_g2[v] = B_v * G2      (dense variables only)
// The ~5 n scalar multiplications are fixed-base: one lane per output point walks a 32-window table of
// 8-bit multiples of the generator (L2-resident, 8160 points) with XYZZ mixed adds on the reduced-radix
// field and converts to affine itself (Fermat inversion).  Integer-ALU bound like the MSM.
#include <string.h>

#include <vector>

#include "bzk_internal.h"
#include "msm_policy.cuh"

namespace bzk {

// out[i] = scalar_i * G  (raw affine 12 x 32-bit Montgomery limbs); flag |= 1 if any result is the identity
template <class C, class StdAff>
__global__ void __launch_bounds__(64) fixed_base_kernel(const typename C::DevAff* __restrict__ table, const Fr* __restrict__ scalars,
                                                        uint64_t n, StdAff* __restrict__ out, uint32_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr k = fe_from_mont<FrParams>(scalars[i]);
    typename C::Pt acc = C::identity();
#pragma unroll 1
    for (int w = 0; w < 32; ++w) {
        const uint32_t d = (k.l[w >> 2] >> ((w & 3) * 8)) & 0xff;
        if (d) {
            typename C::DevAff p = table[w * 255 + d - 1];
            C::add_mixed(acc, p, false);
        }
    }
    XyzzT<typename C::HostF> s = C::to_std(acc);
    StdAff a;
    if (!xyzz_to_affine<typename C::HostF>(s, a)) atomicOr(flag, 1u);
    out[i] = a;
}

template <class C>
static std::vector<typename C::DevAff> host_table(const AffineT<typename C::HostF>& gen);

template <>
std::vector<G1A28> host_table<G1Fast>(const G1Affine& gen) {
    std::vector<G1A28> t(32 * 255);
    G1Xyzz base = xyzz_from_affine<FpOps>(gen);
    for (int w = 0; w < 32; ++w) {
        G1Xyzz acc = base;
        for (int k = 1; k <= 255; ++k) {
            G1Affine a;
            xyzz_to_affine<FpOps>(acc, a);
            t[w * 255 + k - 1] = g1x28::affine_to28(a);
            xyzz_add<FpOps>(acc, base);
        }
        base = acc;  // 256 * base
    }
    return t;
}
template <>
std::vector<G2A28> host_table<G2Fast>(const G2Affine& gen) {
    std::vector<G2A28> t(32 * 255);
    G2Xyzz base = xyzz_from_affine<Fp2Ops>(gen);
    for (int w = 0; w < 32; ++w) {
        G2Xyzz acc = base;
        for (int k = 1; k <= 255; ++k) {
            G2Affine a;
            xyzz_to_affine<Fp2Ops>(acc, a);
            t[w * 255 + k - 1] = g2x28::affine_to28(a);
            xyzz_add<Fp2Ops>(acc, base);
        }
        base = acc;
    }
    return t;
}

static Fp fp_hex(const char* hex) {
    Fp c = Fp::zero();
    for (int i = 0; i < 96; ++i) {
        char ch = hex[i];
        uint32_t v = ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10;
        int nib = 95 - i;
        c.l[nib / 8] |= v << ((nib % 8) * 4);
    }
    return fe_to_mont<FpParams>(c);
}
static G1Affine g1_gen() {
    return {fp_hex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"),
            fp_hex("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")};
}
static G2Affine g2_gen() {
    return {{fp_hex("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"),
             fp_hex("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e")},
            {fp_hex("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"),
             fp_hex("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be")}};
}

// scalars (host) -> points (device buffer `out_dev`, raw affine)
template <class C, class StdAff>
static int32_t fixed_base_batch(bzk_ctx* ctx, const typename C::DevAff* table_dev, const std::vector<Fr>& scalars, void* out_dev,
                                uint32_t* flag_dev) {
    const uint64_t n = scalars.size();
    if (!n) return BZK_OK;
    void* ds = nullptr;
    BZK_HIP(ctx, hipMalloc(&ds, n * sizeof(Fr)));
    hipError_t e = hipMemcpyAsync(ds, scalars.data(), n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        auto k = fixed_base_kernel<C, StdAff>;
        hipLaunchKernelGGL(k, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, table_dev, (const Fr*)ds, n, (StdAff*)out_dev, flag_dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(ds);
    if (e != hipSuccess) {
        ctx->last_error = std::string("fixed_base_batch: ") + hipGetErrorString(e);
        return BZK_E_DEVICE;
    }
    return BZK_OK;
}

struct CsrView {
    uint64_t n_rows;
    const uint32_t* row_ptr;
    const uint32_t* col;
    const Fr* val;
};

}  // namespace bzk

using namespace bzk;

extern "C" {

// declared in bzk.h
int32_t bzk_groth16_setup(bzk_ctx* ctx, const bzk_csr* A, const bzk_csr* B, const bzk_csr* C, uint32_t n_in, uint32_t n_aux,
                          const uint8_t toxic[160], bzk_params** out_params, uint8_t* vk_out, uint64_t vk_cap) {
    if (!ctx || !A || !B || !C || !toxic || !out_params || n_in == 0) return BZK_E_ARG;
    if (A->n_rows != B->n_rows || A->n_rows != C->n_rows || A->n_rows == 0) return BZK_E_ARG;
    const uint64_t vk_need = 870 + 8 + 97ull * n_in;
    if (vk_out && vk_cap < vk_need) return BZK_E_ARG;
    (void)hipSetDevice(ctx->device);
    *out_params = nullptr;
    const uint64_t n_rows = A->n_rows, nv = (uint64_t)n_in + n_aux;
    uint32_t log_m = 0;
    while (((uint64_t)1 << log_m) < n_rows) ++log_m;
    if (log_m > 28) return BZK_E_ARG;
    const uint64_t m = (uint64_t)1 << log_m;
    Fr tau, alpha, beta, gamma, delta;
    memcpy(tau.l, toxic, 32); memcpy(alpha.l, toxic + 32, 32); memcpy(beta.l, toxic + 64, 32);
    memcpy(gamma.l, toxic + 96, 32); memcpy(delta.l, toxic + 128, 32);

    // 1. Lagrange basis at tau on the device: powers of tau, then inverse NTT
    std::vector<Fr> lag(m);
    {
        lag[0] = Fr::one();
        for (uint64_t i = 1; i < m; ++i) lag[i] = fe_mul<FrParams>(lag[i - 1], tau);
        void* d = nullptr;
        BZK_HIP(ctx, hipMalloc(&d, m * sizeof(Fr)));
        int32_t st = BZK_OK;
        if (hipMemcpyAsync(d, lag.data(), m * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = BZK_E_DEVICE;
        if (st == BZK_OK) st = ntt_run(ctx, d, log_m, 1, 0);
        if (st == BZK_OK && hipMemcpyAsync(lag.data(), d, m * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = BZK_E_DEVICE;
        if (st == BZK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = BZK_E_DEVICE;
        (void)hipFree(d);
        if (st != BZK_OK) return st;
    }
    Fr tau_m = Fr::one();
    {
        Fr b = tau;
        for (uint64_t e = m; e; e >>= 1) {
            if (e & 1) tau_m = fe_mul<FrParams>(tau_m, b);
            b = fe_sqr<FrParams>(b);
        }
    }
    // 2. A_v(tau), B_v(tau), C_v(tau) and densities
    std::vector<Fr> at(nv, Fr::zero()), bt(nv, Fr::zero()), ct(nv, Fr::zero());
    std::vector<uint8_t> a_d(nv, 0), b_d(nv, 0);
    auto accum = [&](const bzk_csr* M, std::vector<Fr>& dst, std::vector<uint8_t>* dens) -> bool {
        const Fr* val = (const Fr*)M->val;
        for (uint64_t r = 0; r < M->n_rows; ++r)
            for (uint32_t k = M->row_ptr[r]; k < M->row_ptr[r + 1]; ++k) {
                const uint32_t v = M->col[k];
                if (v >= nv) return false;
                dst[v] = fe_add<FrParams>(dst[v], fe_mul<FrParams>(val[k], lag[r]));
                if (dens) (*dens)[v] = 1;
            }
        return true;
    };
    if (!accum(A, at, &a_d) || !accum(B, bt, &b_d) || !accum(C, ct, nullptr)) return BZK_E_ARG;
    std::vector<uint32_t> ia, ib;
    for (uint64_t v = 0; v < nv; ++v) {
        if (a_d[v]) ia.push_back((uint32_t)v);
        if (b_d[v]) ib.push_back((uint32_t)v);
    }
    // 3. scalar vectors
    const Fr dinv = fe_inv<FrParams>(delta), ginv = fe_inv<FrParams>(gamma);
    const Fr zt = fe_sub<FrParams>(tau_m, Fr::one());
    std::vector<Fr> s_h(m - 1), s_l(n_aux), s_ic(n_in), s_a(ia.size()), s_b(ib.size()), s_vk1 = {alpha, beta, delta}, s_vk2 = {beta, gamma, delta};
    {
        Fr x = fe_mul<FrParams>(zt, dinv);
        for (uint64_t i = 0; i + 1 < m; ++i) {
            s_h[i] = x;
            x = fe_mul<FrParams>(x, tau);
        }
    }
    auto comb = [&](uint64_t v) {
        return fe_add<FrParams>(fe_add<FrParams>(fe_mul<FrParams>(beta, at[v]), fe_mul<FrParams>(alpha, bt[v])), ct[v]);
    };
    for (uint32_t v = 0; v < n_in; ++v) s_ic[v] = fe_mul<FrParams>(comb(v), ginv);
    for (uint32_t v = 0; v < n_aux; ++v) s_l[v] = fe_mul<FrParams>(comb(n_in + v), dinv);
    for (size_t i = 0; i < ia.size(); ++i) s_a[i] = at[ia[i]];
    for (size_t i = 0; i < ib.size(); ++i) s_b[i] = bt[ib[i]];

    // 4. fixed-base multiplications on the device
    std::vector<G1A28> t1 = host_table<G1Fast>(g1_gen());
    std::vector<G2A28> t2 = host_table<G2Fast>(g2_gen());
    void *dt1 = nullptr, *dt2 = nullptr;
    uint32_t* dflag = nullptr;
    bzk_params* p = nullptr;
    void *d_ic = nullptr, *d_vk1 = nullptr, *d_vk2 = nullptr;
    int32_t st = BZK_OK;
    auto fail = [&](int32_t code) {
        if (dt1) (void)hipFree(dt1);
        if (dt2) (void)hipFree(dt2);
        if (dflag) (void)hipFree(dflag);
        if (d_ic) (void)hipFree(d_ic);
        if (d_vk1) (void)hipFree(d_vk1);
        if (d_vk2) (void)hipFree(d_vk2);
        if (p) bzk_params_free(ctx, p);
        return code;
    };
    if (hipMalloc(&dt1, t1.size() * sizeof(G1A28)) != hipSuccess || hipMalloc(&dt2, t2.size() * sizeof(G2A28)) != hipSuccess ||
        hipMalloc((void**)&dflag, 4) != hipSuccess || hipMalloc(&d_ic, 96ull * n_in) != hipSuccess || hipMalloc(&d_vk1, 96 * 3) != hipSuccess ||
        hipMalloc(&d_vk2, 192 * 3) != hipSuccess)
        return fail(BZK_E_ALLOC);
    if (hipMemcpyAsync(dt1, t1.data(), t1.size() * sizeof(G1A28), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(dt2, t2.data(), t2.size() * sizeof(G2A28), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemsetAsync(dflag, 0, 4, ctx->stream) != hipSuccess)
        return fail(BZK_E_DEVICE);

    // a params object with empty CRS buffers of the right size, filled in place
    {
        bzk_params_desc d;
        memset(&d, 0, sizeof d);
        std::vector<uint8_t> zero_vk(870, 0);
        d.n_in = n_in; d.n_aux = n_aux; d.log_m = log_m; d.n_a = (uint32_t)ia.size(); d.n_b = (uint32_t)ib.size();
        d.vk = zero_vk.data(); d.a_density = a_d.data(); d.b_density = b_d.data();
        st = bzk_params_alloc_internal(ctx, &d, &p);
        if (st != BZK_OK) return fail(st);
    }
    void *ph, *pl, *pa, *pb1, *pb2;
    bzk_params_buffers_internal(p, &ph, &pl, &pa, &pb1, &pb2);
    if ((st = fixed_base_batch<G1Fast, G1Affine>(ctx, (const G1A28*)dt1, s_h, ph, dflag)) != BZK_OK) return fail(st);
    if ((st = fixed_base_batch<G1Fast, G1Affine>(ctx, (const G1A28*)dt1, s_l, pl, dflag)) != BZK_OK) return fail(st);
    if ((st = fixed_base_batch<G1Fast, G1Affine>(ctx, (const G1A28*)dt1, s_a, pa, dflag)) != BZK_OK) return fail(st);
    if ((st = fixed_base_batch<G1Fast, G1Affine>(ctx, (const G1A28*)dt1, s_b, pb1, dflag)) != BZK_OK) return fail(st);
    if ((st = fixed_base_batch<G2Fast, G2Affine>(ctx, (const G2A28*)dt2, s_b, pb2, dflag)) != BZK_OK) return fail(st);
    if ((st = fixed_base_batch<G1Fast, G1Affine>(ctx, (const G1A28*)dt1, s_ic, d_ic, dflag)) != BZK_OK) return fail(st);
    if ((st = fixed_base_batch<G1Fast, G1Affine>(ctx, (const G1A28*)dt1, s_vk1, d_vk1, dflag)) != BZK_OK) return fail(st);
    if ((st = fixed_base_batch<G2Fast, G2Affine>(ctx, (const G2A28*)dt2, s_vk2, d_vk2, dflag)) != BZK_OK) return fail(st);
    uint32_t flag = 0;
    std::vector<uint8_t> h_ic(96ull * n_in), h_vk1(96 * 3), h_vk2(192 * 3);
    if (hipMemcpy(&flag, dflag, 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(h_ic.data(), d_ic, h_ic.size(), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(h_vk1.data(), d_vk1, h_vk1.size(), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(h_vk2.data(), d_vk2, h_vk2.size(), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(BZK_E_DEVICE);
    if (flag) {  // an identity point in the CRS: a query polynomial vanished at tau (tau on the evaluation domain, a zero in the toxic waste)
        ctx->last_error = "bzk_groth16_setup: a CRS element is the identity for this toxic waste (tau on the evaluation domain, or a zero among alpha / beta / gamma / delta)";
        return fail(BZK_E_ARG);
    }
    // vk: alpha_g1 | beta_g1 | beta_g2 | gamma_g2 | delta_g1 | delta_g2 (packed, inf flag 0)
    uint8_t vk[870];
    memset(vk, 0, sizeof vk);
    memcpy(vk, &h_vk1[0], 96);            // alpha_g1
    memcpy(vk + 97, &h_vk1[96], 96);      // beta_g1
    memcpy(vk + 194, &h_vk2[0], 192);     // beta_g2
    memcpy(vk + 387, &h_vk2[192], 192);   // gamma_g2
    memcpy(vk + 580, &h_vk1[192], 96);    // delta_g1
    memcpy(vk + 677, &h_vk2[384], 192);   // delta_g2
    bzk_params_set_vk_internal(p, vk);
    if (vk_out) {
        memcpy(vk_out, vk, 870);
        const uint64_t len = n_in;
        memcpy(vk_out + 870, &len, 8);
        for (uint32_t v = 0; v < n_in; ++v) {
            memcpy(vk_out + 878 + 97ull * v, &h_ic[96ull * v], 96);
            vk_out[878 + 97ull * v + 96] = 0;
        }
    }
    (void)hipFree(dt1); (void)hipFree(dt2); (void)hipFree(dflag); (void)hipFree(d_ic); (void)hipFree(d_vk1); (void)hipFree(d_vk2);
    *out_params = p;
    return BZK_OK;
}

}  // extern "C"
