// CPU ORACLE - TEST INFRASTRUCTURE ONLY (see field.hpp header).
//
// Short-Weierstrass y^2 = x^3 + b (a = 0) group law in Jacobian coordinates, generic over the base
// field (Fp for G1, Fp2 for G2).  Restates what the reference gets from bls12_381 0.8
// `G1Affine/G1Projective/G2Affine/G2Projective` (third-party; used at
// /root/reference/src/zk/groth16/mod.rs:4-17 and inside bellman's multiexp).  Formulas are the
// public EFD ones (dbl-2009-l, madd-2007-bl, add-2007-bl); affine results are canonical so the
// choice of projective system does not influence output bytes.
#pragma once
#include "field.hpp"

namespace orc {

template <class F>
struct Affine {
    F x, y;
    bool inf;
};

template <class F>
struct Jac {
    F X, Y, Z;
    static Jac identity() { return {F::one(), F::one(), F::zero()}; }
    bool is_identity() const { return Z.is_zero(); }
    static Jac from_affine(const Affine<F>& a) {
        if (a.inf) return identity();
        return {a.x, a.y, F::one()};
    }

    Jac dbl() const {
        if (is_identity()) return *this;
        F A = X.sqr(), B = Y.sqr(), C = B.sqr();
        F D = X.add(B).sqr().sub(A).sub(C).dbl();
        F E = A.dbl().add(A), Fq = E.sqr();
        Jac r;
        r.X = Fq.sub(D.dbl());
        r.Z = Y.mul(Z).dbl();
        r.Y = E.mul(D.sub(r.X)).sub(C.dbl().dbl().dbl());
        return r;
    }

    Jac add_mixed(const Affine<F>& q) const {
        if (q.inf) return *this;
        if (is_identity()) return from_affine(q);
        F Z1Z1 = Z.sqr();
        F U2 = q.x.mul(Z1Z1), S2 = q.y.mul(Z).mul(Z1Z1);
        F H = U2.sub(X), rr = S2.sub(Y);
        if (H.is_zero()) {
            if (rr.is_zero()) return dbl();
            return identity();
        }
        rr = rr.dbl();
        F HH = H.sqr(), I = HH.dbl().dbl(), J = H.mul(I), V = X.mul(I);
        Jac r;
        r.X = rr.sqr().sub(J).sub(V.dbl());
        r.Y = rr.mul(V.sub(r.X)).sub(Y.mul(J).dbl());
        r.Z = Z.add(H).sqr().sub(Z1Z1).sub(HH);
        return r;
    }

    Jac add(const Jac& q) const {
        if (q.is_identity()) return *this;
        if (is_identity()) return q;
        F Z1Z1 = Z.sqr(), Z2Z2 = q.Z.sqr();
        F U1 = X.mul(Z2Z2), U2 = q.X.mul(Z1Z1);
        F S1 = Y.mul(q.Z).mul(Z2Z2), S2 = q.Y.mul(Z).mul(Z1Z1);
        F H = U2.sub(U1), rr = S2.sub(S1);
        if (H.is_zero()) {
            if (rr.is_zero()) return dbl();
            return identity();
        }
        rr = rr.dbl();
        F I = H.dbl().sqr(), J = H.mul(I), V = U1.mul(I);
        Jac r;
        r.X = rr.sqr().sub(J).sub(V.dbl());
        r.Y = rr.mul(V.sub(r.X)).sub(S1.mul(J).dbl());
        r.Z = Z.add(q.Z).sqr().sub(Z1Z1).sub(Z2Z2).mul(H);
        return r;
    }

    Jac neg() const { return {X, Y.neg(), Z}; }

    Affine<F> to_affine() const {
        if (is_identity()) return {F::zero(), F::one(), true};
        F zi = Z.inv(), zi2 = zi.sqr();
        return {X.mul(zi2), Y.mul(zi2).mul(zi), false};
    }

    // scalar = canonical little-endian limbs
    Jac mul(const uint64_t* k, int nlimbs) const {
        Jac r = identity();
        for (int i = nlimbs * 64 - 1; i >= 0; --i) {
            r = r.dbl();
            if ((k[i / 64] >> (i % 64)) & 1) r = r.add(*this);
        }
        return r;
    }
};

typedef Affine<Fp> G1Affine;
typedef Affine<Fp2> G2Affine;
typedef Jac<Fp> G1;
typedef Jac<Fp2> G2;

}  // namespace orc
