// BLS12-381 G1 / G2 group law in extended-Jacobian ("XYZZ") coordinates, generic over FpOps / Fp2Ops.
//
// x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2; identity <=> ZZ == 0.  XYZZ is chosen for the MSM bucket
// accumulators: a mixed add (bucket += affine CRS point) is 8M+2S against 7M+4S for Jacobian, with
// no inversion, and the accumulator is 4 field elements (48 VGPRs for G1).
// Affine results are canonical, so outputs match bellman/bls12_381 (Jacobian) byte for byte.
// Formulas: EFD madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1 (a = 0).
#pragma once
#include "bzk_field.cuh"

namespace bzk {

template <class F>
struct AffineT {
    typename F::T x, y;
};

template <class F>
struct XyzzT {
    typename F::T X, Y, ZZ, ZZZ;
};

template <class F>
BZK_HD XyzzT<F> xyzz_identity() {
    return {F::zero(), F::one(), F::zero(), F::zero()};
}

template <class F>
BZK_HD bool xyzz_is_identity(const XyzzT<F>& p) {
    return F::is_zero(p.ZZ);
}

template <class F>
BZK_HD XyzzT<F> xyzz_from_affine(const AffineT<F>& a) {
    return {a.x, a.y, F::one(), F::one()};
}

// 2 * (affine point)
template <class F>
BZK_HD XyzzT<F> xyzz_dbl_affine(const AffineT<F>& a) {
    typedef typename F::T T;
    T U = F::dbl(a.y), V = F::sqr(U), W = F::mul(U, V), S = F::mul(a.x, V);
    T xx = F::sqr(a.x);
    T M = F::add(F::dbl(xx), xx);
    XyzzT<F> r;
    r.X = F::sub(F::sqr(M), F::dbl(S));
    r.Y = F::sub(F::mul(M, F::sub(S, r.X)), F::mul(W, a.y));
    r.ZZ = V;
    r.ZZZ = W;
    return r;
}

template <class F>
BZK_HD XyzzT<F> xyzz_dbl(const XyzzT<F>& p) {
    typedef typename F::T T;
    if (xyzz_is_identity<F>(p)) return p;
    T U = F::dbl(p.Y), V = F::sqr(U), W = F::mul(U, V), S = F::mul(p.X, V);
    T xx = F::sqr(p.X);
    T M = F::add(F::dbl(xx), xx);
    XyzzT<F> r;
    r.X = F::sub(F::sqr(M), F::dbl(S));
    r.Y = F::sub(F::mul(M, F::sub(S, r.X)), F::mul(W, p.Y));
    r.ZZ = F::mul(V, p.ZZ);
    r.ZZZ = F::mul(W, p.ZZZ);
    return r;
}

// acc += q (affine, never the identity).  Handles acc == identity, q == acc (doubling), q == -acc.
template <class F>
BZK_HD void xyzz_add_mixed(XyzzT<F>& acc, const AffineT<F>& q) {
    typedef typename F::T T;
    if (xyzz_is_identity<F>(acc)) {
        acc = xyzz_from_affine<F>(q);
        return;
    }
    T U2 = F::mul(q.x, acc.ZZ), S2 = F::mul(q.y, acc.ZZZ);
    T Pp = F::sub(U2, acc.X), R = F::sub(S2, acc.Y);
    if (F::is_zero(Pp)) {
        if (F::is_zero(R)) acc = xyzz_dbl_affine<F>(q);
        else acc = xyzz_identity<F>();
        return;
    }
    T PP = F::sqr(Pp), PPP = F::mul(Pp, PP), Q = F::mul(acc.X, PP);
    T X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    T Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(acc.Y, PPP));
    acc.X = X3;
    acc.Y = Y3;
    acc.ZZ = F::mul(acc.ZZ, PP);
    acc.ZZZ = F::mul(acc.ZZZ, PPP);
}

template <class F>
BZK_HD void xyzz_add(XyzzT<F>& acc, const XyzzT<F>& q) {
    typedef typename F::T T;
    if (xyzz_is_identity<F>(q)) return;
    if (xyzz_is_identity<F>(acc)) {
        acc = q;
        return;
    }
    T U1 = F::mul(acc.X, q.ZZ), U2 = F::mul(q.X, acc.ZZ);
    T S1 = F::mul(acc.Y, q.ZZZ), S2 = F::mul(q.Y, acc.ZZZ);
    T Pp = F::sub(U2, U1), R = F::sub(S2, S1);
    if (F::is_zero(Pp)) {
        if (F::is_zero(R)) acc = xyzz_dbl<F>(acc);
        else acc = xyzz_identity<F>();
        return;
    }
    T PP = F::sqr(Pp), PPP = F::mul(Pp, PP), Q = F::mul(U1, PP);
    T X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    T Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(S1, PPP));
    acc.X = X3;
    acc.Y = Y3;
    acc.ZZ = F::mul(F::mul(acc.ZZ, q.ZZ), PP);
    acc.ZZZ = F::mul(F::mul(acc.ZZZ, q.ZZZ), PPP);
}

// acc += *q with q left in memory: each coordinate of q is read where it is used, so only acc and the formula's
// temporaries occupy registers (peak ~10 field elements instead of ~14 with q resident).  For the G2 tail kernels, where
// two resident 112-register points plus an addition's temporaries exceed what survives a call to the field product
// (everything beyond went through scratch memory: 1744 B per lane in msm_reduce).  Same formula and case analysis as
// xyzz_add; `fence` is a compiler barrier that keeps the loads from being hoisted back to the top.
template <class F>
BZK_HD void xyzz_add_mem(XyzzT<F>& acc, const XyzzT<F>* q) {
    typedef typename F::T T;
#if defined(__HIP_DEVICE_COMPILE__)
#define BZK_MEM_FENCE() __asm__ volatile("" ::: "memory")
#else
#define BZK_MEM_FENCE() ((void)0)
#endif
    {
        const T qzz = q->ZZ;
        if (F::is_zero(qzz)) return;  // q is the identity
        if (xyzz_is_identity<F>(acc)) {
            acc = *q;
            return;
        }
    }
    BZK_MEM_FENCE();
    T U1 = F::mul(acc.X, q->ZZ);
    BZK_MEM_FENCE();
    T U2 = F::mul(q->X, acc.ZZ);
    T Pp = F::sub(U2, U1);
    BZK_MEM_FENCE();
    T S1 = F::mul(acc.Y, q->ZZZ);
    BZK_MEM_FENCE();
    T S2 = F::mul(q->Y, acc.ZZZ);
    T R = F::sub(S2, S1);
    if (F::is_zero(Pp)) {
        if (F::is_zero(R)) acc = xyzz_dbl<F>(acc);
        else acc = xyzz_identity<F>();
        return;
    }
    T PP = F::sqr(Pp), PPP = F::mul(Pp, PP), Q = F::mul(U1, PP);
    T X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    T Y3 = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(S1, PPP));
    acc.X = X3;
    acc.Y = Y3;
    BZK_MEM_FENCE();
    acc.ZZ = F::mul(F::mul(acc.ZZ, q->ZZ), PP);
    BZK_MEM_FENCE();
    acc.ZZZ = F::mul(F::mul(acc.ZZZ, q->ZZZ), PPP);
#undef BZK_MEM_FENCE
}

template <class F>
BZK_HD XyzzT<F> xyzz_neg(const XyzzT<F>& p) {
    return {p.X, F::neg(p.Y), p.ZZ, p.ZZZ};
}

// affine (x, y); returns false for the identity
template <class F>
BZK_HD bool xyzz_to_affine(const XyzzT<F>& p, AffineT<F>& out) {
    typedef typename F::T T;
    if (xyzz_is_identity<F>(p)) {
        out.x = F::zero();
        out.y = F::one();
        return false;
    }
    // 1/ZZZ = i3 ; 1/ZZ = ZZ^2 * i3^2 (since ZZ^3 = ZZZ^2)
    T i3 = F::inv(p.ZZZ);
    T i2 = F::mul(F::sqr(p.ZZ), F::sqr(i3));
    out.x = F::mul(p.X, i2);
    out.y = F::mul(p.Y, i3);
    return true;
}

// k * p for a small scalar k (< 2^32), double-and-add, used in bucket-chunk recombination
template <class F>
BZK_HD XyzzT<F> xyzz_mul_u32(const XyzzT<F>& p, uint32_t k) {
    XyzzT<F> r = xyzz_identity<F>();
    for (int i = 31; i >= 0; --i) {
        r = xyzz_dbl<F>(r);
        if ((k >> i) & 1) xyzz_add<F>(r, p);
    }
    return r;
}

typedef AffineT<FpOps> G1Affine;
typedef AffineT<Fp2Ops> G2Affine;
typedef XyzzT<FpOps> G1Xyzz;
typedef XyzzT<Fp2Ops> G2Xyzz;

}  // namespace bzk
