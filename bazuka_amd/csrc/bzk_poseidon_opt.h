// Host-side derivation of the "sparse partial rounds" form of the Poseidon permutation (same function, fewer
// products): the reference (/root/reference/src/zk/poseidon/mod.rs:24-84) multiplies the state by the dense
// T x T MDS matrix in every one of its R_P = 56/57 partial rounds although only state[0] went through the S-box.
//
// 1. Round constants.  In a partial round the S-box layer P is the identity on coordinates 1..T-1, so a constant
//    vector c_i added before P can be written as M * (M^-1 c_i) and moved to the far side of the previous round's
//    matrix; its coordinates 1.. then slide through that round's P as well.  Working from the last partial round
//    to the first leaves ONE full vector `pre` added before the first partial round and a scalar s_i added to
//    state[0] after the S-box of round i.
// 2. Matrices.  Write M_i = [[m00, v],[w, Mh]] = diag(1, Mh) * [[m00, v],[Mh^-1 w, I]].  The left factor fixes
//    coordinate 0, so it commutes with the next round's S-box and scalar add and is absorbed into the next matrix:
//    M_{i+1} = M * diag(1, Mh_i).  Each partial round then costs 2T - 1 products (row 0 plus one per remaining
//    coordinate); the last left factor D = diag(1, Mh_{R-1}) is applied once after the block.
// The derived constants are checked against the plain form by tests/host/hostcheck.hip (the 16 reference KATs and
// random inputs, through the very device function) and at context start-up (poseidon.hip).
#pragma once
#include <vector>

#include "bzk_field.cuh"

namespace bzk {

namespace popt {
typedef std::vector<Fr> Mat;  // row-major n x n
inline Mat mat_mul(const Mat& a, const Mat& b, int n) {
    Mat r((size_t)n * n, Fr::zero());
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            Fr acc = Fr::zero();
            for (int k = 0; k < n; ++k) acc = fe_add<FrParams>(acc, fe_mul<FrParams>(a[i * n + k], b[k * n + j]));
            r[i * n + j] = acc;
        }
    return r;
}
inline bool mat_inv(const Mat& m, int n, Mat& out) {  // Gauss-Jordan over Fr
    Mat a(m);
    out.assign((size_t)n * n, Fr::zero());
    for (int i = 0; i < n; ++i) out[i * n + i] = Fr::one();
    for (int col = 0; col < n; ++col) {
        int piv = -1;
        for (int r = col; r < n; ++r)
            if (!a[r * n + col].is_zero()) { piv = r; break; }
        if (piv < 0) return false;
        if (piv != col)
            for (int k = 0; k < n; ++k) {
                std::swap(a[piv * n + k], a[col * n + k]);
                std::swap(out[piv * n + k], out[col * n + k]);
            }
        const Fr inv = fe_inv<FrParams>(a[col * n + col]);
        for (int k = 0; k < n; ++k) {
            a[col * n + k] = fe_mul<FrParams>(a[col * n + k], inv);
            out[col * n + k] = fe_mul<FrParams>(out[col * n + k], inv);
        }
        for (int r = 0; r < n; ++r) {
            if (r == col || a[r * n + col].is_zero()) continue;
            const Fr f = a[r * n + col];
            for (int k = 0; k < n; ++k) {
                a[r * n + k] = fe_sub<FrParams>(a[r * n + k], fe_mul<FrParams>(f, a[col * n + k]));
                out[r * n + k] = fe_sub<FrParams>(out[r * n + k], fe_mul<FrParams>(f, out[col * n + k]));
            }
        }
    }
    return true;
}
}  // namespace popt

// number of field elements of the optimized constant block for width T
inline size_t poseidon_opt_count(int T, int rf, int rp) {
    return (size_t)rf * T + T + (size_t)rp * 2 * T + (size_t)(T - 1) * (T - 1) + (size_t)T * T;
}

// rc: (rf + rp) * T round constants, mds: T * T (new[j] = sum_k mds[j*T+k] * st[k]), both Montgomery.
// flat = rc_first (rf/2 * T) | pre (T) | rp x { s_i, row0[T], what[T-1] } | D ((T-1)^2) | rc_second (rf/2 * T) | mds (T*T)
// returns false when a sub-matrix is singular (no optimized form; never the case for the reference's parameters)
inline bool poseidon_optimize(int T, int rf, int rp, const std::vector<Fr>& rc, const std::vector<Fr>& mds, std::vector<Fr>& flat) {
    using namespace popt;
    const int half = rf / 2, n1 = T - 1;
    if ((int)rc.size() != (rf + rp) * T || (int)mds.size() != T * T || rp < 1) return false;
    Mat Minv;
    if (!mat_inv(mds, T, Minv)) return false;
    // 1. constants
    std::vector<std::vector<Fr>> c((size_t)rp, std::vector<Fr>((size_t)T));
    for (int i = 0; i < rp; ++i)
        for (int k = 0; k < T; ++k) c[i][k] = rc[(size_t)(half + i) * T + k];
    std::vector<Fr> s((size_t)rp, Fr::zero());
    for (int i = rp - 1; i >= 1; --i) {
        for (int j = 0; j < T; ++j) {
            Fr e = Fr::zero();
            for (int k = 0; k < T; ++k) e = fe_add<FrParams>(e, fe_mul<FrParams>(Minv[j * T + k], c[i][k]));
            if (j == 0) s[i - 1] = e;
            else c[i - 1][j] = fe_add<FrParams>(c[i - 1][j], e);
        }
    }
    // 2. matrices
    flat.clear();
    flat.insert(flat.end(), rc.begin(), rc.begin() + (size_t)half * T);
    flat.insert(flat.end(), c[0].begin(), c[0].end());
    Mat Mi(mds), Mh((size_t)n1 * n1), Mh_inv;
    for (int i = 0; i < rp; ++i) {
        for (int a = 0; a < n1; ++a)
            for (int b = 0; b < n1; ++b) Mh[a * n1 + b] = Mi[(a + 1) * T + (b + 1)];
        if (!mat_inv(Mh, n1, Mh_inv)) return false;
        flat.push_back(s[i]);
        for (int k = 0; k < T; ++k) flat.push_back(Mi[k]);  // row 0: m00, v
        for (int a = 0; a < n1; ++a) {                      // what = Mh^-1 w
            Fr e = Fr::zero();
            for (int b = 0; b < n1; ++b) e = fe_add<FrParams>(e, fe_mul<FrParams>(Mh_inv[a * n1 + b], Mi[(b + 1) * T]));
            flat.push_back(e);
        }
        Mat Mp((size_t)T * T, Fr::zero());  // diag(1, Mh)
        Mp[0] = Fr::one();
        for (int a = 0; a < n1; ++a)
            for (int b = 0; b < n1; ++b) Mp[(a + 1) * T + (b + 1)] = Mh[a * n1 + b];
        Mi = mat_mul(mds, Mp, T);
    }
    flat.insert(flat.end(), Mh.begin(), Mh.end());  // D (the last left factor)
    flat.insert(flat.end(), rc.begin() + (size_t)(half + rp) * T, rc.end());
    flat.insert(flat.end(), mds.begin(), mds.end());
    return flat.size() == poseidon_opt_count(T, rf, rp);
}

}  // namespace bzk
