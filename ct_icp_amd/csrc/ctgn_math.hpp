// ctgn_math.hpp — fixed-size double-precision SE(3) / small linear algebra used by the gfx950 kernels and by
// the host side of libctgn (compiled by hipcc for both). Quaternions are (x, y, z, w), Eigen's coeffs() order.
//
// Semantics follow Eigen 3 as used by the reference at
//   src/ct_icp/ct_icp.cpp:716-717,813-816,914-962 and include/SlamCore/types.h:192-219,353-366,453-470
// (slerp with the |dot| >= 1-eps linear fallback, rotate = v + w*(2 q_v x v) + q_v x (2 q_v x v),
//  matrix->quaternion by the trace / largest-diagonal branches, diagonally pivoted LDL^T).
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

#define CTGN_HD __host__ __device__ __forceinline__

// Wave votes on a bool. HIP's __ballot / __any take an int: the predicate — already a lane mask in a scalar register pair — is first
// turned into 0 / 1 per lane (v_cndmask) and compared against zero again (v_cmp_ne); two vector instructions per vote that the builtin
// on a bool does not need. The search kernel votes several times per 16-candidate step.
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool any64(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

namespace ctgn {

struct Vec3 {
    double x, y, z;
};

CTGN_HD Vec3 v3(double x, double y, double z) { return Vec3{x, y, z}; }
CTGN_HD Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
CTGN_HD Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
CTGN_HD Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
CTGN_HD double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// |d|^2 exactly as the reference's stock build evaluates (p - q).squaredNorm() / .norm() of a Vector3d (map.h:279-286,313,491):
// Eigen's completely unrolled redux associates c0 + (c1 + c2) (Core/Redux.h), and a build without -march cannot fuse multiply-adds.
// This value feeds discrete decisions (radius test, k-th best, minimum-distance insert, eviction), so it is kept bit-identical.
CTGN_HD double sq_norm3(double dx, double dy, double dz) {
#pragma clang fp contract(off)
    const double xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return xx + (yy + zz);
}
CTGN_HD Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Quat {
    double x, y, z, w;
};

CTGN_HD Quat quat_normalized(Quat q) {
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return {q.x / n, q.y / n, q.z / n, q.w / n};
}

// Eigen QuaternionBase::_transformVector
CTGN_HD Vec3 quat_rotate(Quat q, Vec3 v) {
    Vec3 qv{q.x, q.y, q.z};
    Vec3 uv = cross(qv, v);
    uv = uv + uv;
    Vec3 c = cross(qv, uv);
    return {v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z};
}

// Slerp coefficients of Eigen QuaternionBase::slerp for a FIXED pair (a, b): they depend on the pair only
// through d = a.b, theta = acos(|d|), sin(theta); those are hoisted out of the per-keypoint work.
struct SlerpPair {
    double theta;      // acos(|d|)
    double sin_theta;  // sin(theta)
    int linear;        // |d| >= 1 - eps  -> scale0 = 1-t, scale1 = t
    int negate;        // d < 0 -> scale1 = -scale1
};

CTGN_HD SlerpPair slerp_prepare(Quat a, Quat b) {
    SlerpPair s;
    const double one = 1.0 - DBL_EPSILON;
    double d = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    double ad = fabs(d);
    s.linear = (ad >= one) ? 1 : 0;
    s.negate = (d < 0) ? 1 : 0;
    if (s.linear) {
        s.theta = 0.0;
        s.sin_theta = 1.0;
    } else {
        s.theta = acos(ad);
        s.sin_theta = sin(s.theta);
    }
    return s;
}

CTGN_HD Quat slerp_eval(Quat a, Quat b, SlerpPair s, double t) {
    double s0, s1;
    if (s.linear) {
        s0 = 1.0 - t;
        s1 = t;
    } else {
        s0 = sin((1.0 - t) * s.theta) / s.sin_theta;
        s1 = sin(t * s.theta) / s.sin_theta;
    }
    if (s.negate) s1 = -s1;
    return {s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z, s0 * a.w + s1 * b.w};
}

// TPose::GetAlphaTimestamp (reference include/SlamCore/types.h:192-219): 0 below the min AND above the max.
CTGN_HD double alpha_timestamp(double t, double t_begin, double t_end) {
    double lo = fmin(t_begin, t_end), hi = fmax(t_begin, t_end);
    if (lo > t) return 0.0;
    if (hi < t) return 0.0;
    if (lo == hi) return 1.0;
    return (t - lo) / (hi - lo);
}

// Eigen toRotationMatrix, row-major.
CTGN_HD void quat_to_matrix(Quat q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// Eigen Quaterniond(Matrix3d).
CTGN_HD Quat matrix_to_quat(const double R[9]) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        const double w = 0.5 * t;
        t = 0.5 / t;
        return {(R[7] - R[5]) * t, (R[2] - R[6]) * t, (R[3] - R[1]) * t, w};
    }
    // largest diagonal entry, first maximum wins (i = 0; if R11 > R00: i = 1; if R22 > R_ii: i = 2), then j, k cyclic. The three cases
    // are spelled out with constant indices: a run-time index into R[] or q[] puts both arrays into scratch memory on the device.
    const bool i1 = R[4] > R[0];
    const bool i2 = R[8] > (i1 ? R[4] : R[0]);
    if (i2) {                       // i = 2, j = 0, k = 1
        t = sqrt(R[8] - R[0] - R[4] + 1.0);
        const double qi = 0.5 * t;
        t = 0.5 / t;
        return {(R[2] + R[6]) * t, (R[5] + R[7]) * t, qi, (R[3] - R[1]) * t};
    }
    if (i1) {                       // i = 1, j = 2, k = 0
        t = sqrt(R[4] - R[8] - R[0] + 1.0);
        const double qi = 0.5 * t;
        t = 0.5 / t;
        return {(R[1] + R[3]) * t, qi, (R[7] + R[5]) * t, (R[2] - R[6]) * t};
    }
    t = sqrt(R[0] - R[4] - R[8] + 1.0);   // i = 0, j = 1, k = 2
    const double qi = 0.5 * t;
    t = 0.5 / t;
    return {qi, (R[3] + R[1]) * t, (R[6] + R[2]) * t, (R[7] - R[5]) * t};
}

// Rz(gamma) Ry(beta) Rx(alpha) exactly as spelled at reference src/ct_icp/ct_icp.cpp:919-932.
CTGN_HD void euler_rotation_sc(double sa, double ca, double sb, double cb, double sg, double cg, double R[9]) {
    R[0] = cg * cb; R[1] = -sg * ca + cg * sb * sa; R[2] = sg * sa + cg * sb * ca;
    R[3] = sg * cb; R[4] = cg * ca + sg * sb * sa;  R[5] = -cg * sa + sg * sb * ca;
    R[6] = -sb;     R[7] = cb * sa;                 R[8] = cb * ca;
}
CTGN_HD void euler_rotation(double al, double be, double ga, double R[9]) {
    double sa = sin(al), ca = cos(al), sb = sin(be), cb = cos(be), sg = sin(ga), cg = cos(ga);
    R[0] = cg * cb; R[1] = -sg * ca + cg * sb * sa; R[2] = sg * sa + cg * sb * ca;
    R[3] = sg * cb; R[4] = cg * ca + sg * sb * sa;  R[5] = -cg * sa + sg * sb * ca;
    R[6] = -sb;     R[7] = cb * sa;                 R[8] = cb * ca;
}

CTGN_HD void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// Smallest-|eigenvalue| eigenvector (the reference's `normal`, V.col(2) of JacobiSVD, neighborhood.h:293-303)
// and a2D = (sqrt|s1| - sqrt|s2|) / sqrt|s0| (:304-311) of a symmetric 3x3 given by its 6 unique entries.
// Cyclic Jacobi in registers: every index is a compile-time constant so nothing spills to scratch.
struct Sym3 {
    double xx, xy, xz, yy, yz, zz;
};

// 1/x and 1/sqrt(x) for well-scaled positive doubles: hardware seed + two Newton steps (~1e-16 relative). The IEEE
// sequences the compiler emits for `/` and sqrt() (v_div_scale / v_div_fmas / v_div_fixup ...) are 3-4x longer, and the
// Jacobi rotations below are nothing but divisions and square roots: they were 70 % of k_residual_reduce.
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x;
    y = fma(fma(-h * y, y, 0.5), y, y);
    y = fma(fma(-h * y, y, 0.5), y, y);
    return y;
}

// One Jacobi rotation annihilating a_pq: t = b / (d + sgn(d) sqrt(d^2 + b^2)) with d = a_qq - a_pp, b = 2 a_pq (the
// smaller root of t^2 + 2 tau t - 1 = 0), c = 1 / sqrt(1 + t^2), s = t c — one square root, one reciprocal, one
// reciprocal square root.
#define CTGN_JACOBI_ROT(app, aqq, apq, arp, arq, vp0, vq0, vp1, vq1, vp2, vq2)                       \
    const double d_##apq = aqq - app, b_##apq = 2.0 * apq;                                           \
    const double r2_##apq = fma(d_##apq, d_##apq, b_##apq * b_##apq);                                \
    if (r2_##apq < 1e-290) apq = 0.0;              /* both underflowed: nothing to annihilate */      \
    if (apq != 0.0) {                                                                                \
        const double d_ = d_##apq, b_ = b_##apq, r2_ = r2_##apq;                                     \
        const double r_ = r2_ * fast_rsqrt(r2_);                                                     \
        const double t_ = b_ * fast_rcp(d_ + copysign(r_, d_));                                      \
        const double c_ = fast_rsqrt(fma(t_, t_, 1.0)), s_ = t_ * c_;                                \
        app -= t_ * apq;                                                                             \
        aqq += t_ * apq;                                                                             \
        apq = 0.0;                                                                                   \
        double n_rp = c_ * arp - s_ * arq, n_rq = s_ * arp + c_ * arq;                               \
        arp = n_rp; arq = n_rq;                                                                      \
        double n0 = c_ * vp0 - s_ * vq0, m0 = s_ * vp0 + c_ * vq0; vp0 = n0; vq0 = m0;              \
        double n1 = c_ * vp1 - s_ * vq1, m1 = s_ * vp1 + c_ * vq1; vp1 = n1; vq1 = m1;              \
        double n2 = c_ * vp2 - s_ * vq2, m2 = s_ * vp2 + c_ * vq2; vp2 = n2; vq2 = m2;              \
    }

__device__ __forceinline__ void sym3_normal_a2d(Sym3 c, Vec3 &normal, double &a2d) {
    double a00 = c.xx, a01 = c.xy, a02 = c.xz, a11 = c.yy, a12 = c.yz, a22 = c.zz;
    // V columns: (v00,v10,v20) = eigenvector 0, etc.
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
    for (int sweep = 0; sweep < 24; ++sweep) {
        double off = fabs(a01) + fabs(a02) + fabs(a12);
        double scale = fabs(a00) + fabs(a11) + fabs(a22);
        // off-diagonal mass shrinks quadratically per sweep; below 1e-14 of the diagonal the eigenvalues are exact to
        // rounding (second-order error) and the eigenvectors to ~1e-14 / relative gap
        if (off <= 1e-300 + 1e-14 * scale) break;
        // (p,q) = (0,1), r = 2 : arp = a02, arq = a12
        CTGN_JACOBI_ROT(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21)
        // (p,q) = (0,2), r = 1 : arp = a01, arq = a12
        CTGN_JACOBI_ROT(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22)
        // (p,q) = (1,2), r = 0 : arp = a01, arq = a02
        CTGN_JACOBI_ROT(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22)
    }
    double m0 = fabs(a00), m1 = fabs(a11), m2 = fabs(a22);
    // singular values = |eigenvalues| sorted descending; the normal belongs to the smallest
    double smin = m0, smid, smax;
    Vec3 n{v00, v10, v20};
    if (m1 < smin) { smin = m1; n = Vec3{v01, v11, v21}; }
    if (m2 < smin) { smin = m2; n = Vec3{v02, v12, v22}; }
    smax = fmax(m0, fmax(m1, m2));
    smid = fmax(fmin(m0, m1), fmin(fmax(m0, m1), m2));   // median of three, exact
    normal = n;
    a2d = (sqrt(smid) - sqrt(smin)) / sqrt(smax);
}

// Eigen::JacobiSVD<Matrix3d>(C, ComputeFullV) of the reference's ComputeNeighborhoodInfo (include/SlamCore/experimental/
// neighborhood.h:293) restated operation for operation (Eigen/src/SVD/JacobiSVD.h two-sided Jacobi over (p, q) = (1,0), (2,0), (2,1)
// with real_2x2_jacobi_svd and JacobiRotation::makeJacobi from Eigen/src/Jacobi/Jacobi.h; threshold 2 eps max|diag|; singular
// values = |diag| * scale, sorted decreasing together with the columns of V), with correctly rounded division and square root and
// no fused multiply-add — i.e. the arithmetic of a stock x86-64 build. The fast cyclic Jacobi above agrees with it to rounding
// wherever the normal is defined; on a rank-deficient covariance (collinear neighbours) the third singular vector is decided by
// the roundings themselves, and the robust route gives such neighbourhoods a non-zero weight (ct_icp.cpp:574-579), so that route
// uses this one. C row-major full symmetric matrix; sv descending; V row-major.
CTGN_HD void jacobi_svd3_exact(const double Cin[9], double sv[3], double V[9]) {
#pragma clang fp contract(off)
    double W[3][3], Vm[3][3];
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) if (fabs(Cin[i]) > scale) scale = fabs(Cin[i]);
    if (scale == 0.0) scale = 1.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { W[i][j] = Cin[3 * i + j] / scale; Vm[i][j] = (i == j) ? 1.0 : 0.0; }
    const double precision = 2.0 * DBL_EPSILON, tiny = DBL_MIN;
    double max_diag = 0.0;
    for (int i = 0; i < 3; ++i) if (fabs(W[i][i]) > max_diag) max_diag = fabs(W[i][i]);
    bool finished = false;
    while (!finished) {
        finished = true;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 0 ? 1 : 2, q = pq == 2 ? 1 : 0;
            double thr = precision * max_diag;
            if (thr < tiny) thr = tiny;
            if (!(fabs(W[p][q]) > thr || fabs(W[q][p]) > thr)) continue;
            finished = false;
            const double m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
            double r1c, r1s;
            const double t = m00 + m11, d = m10 - m01;
            if (fabs(d) < tiny) { r1s = 0.0; r1c = 1.0; }
            else { const double u = t / d; const double tmp = sqrt(1.0 + u * u); r1s = 1.0 / tmp; r1c = u / tmp; }
            const double n00 = r1c * m00 + r1s * m10, n01 = r1c * m01 + r1s * m11, n11 = -r1s * m01 + r1c * m11;
            double jc, js;
            const double deno = 2.0 * fabs(n01);
            if (deno < tiny) { jc = 1.0; js = 0.0; }
            else {
                const double tau = (n00 - n11) / deno, w = sqrt(tau * tau + 1.0);
                const double t2 = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
                const double sign_t = t2 > 0.0 ? 1.0 : -1.0;
                const double n = 1.0 / sqrt(t2 * t2 + 1.0);
                js = -sign_t * (n01 / fabs(n01)) * fabs(t2) * n;
                jc = n;
            }
            const double lc = r1c * jc - r1s * (-js), ls = r1c * (-js) + r1s * jc;
            if (!(lc == 1.0 && ls == 0.0)) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double xi = W[p][k], yi = W[q][k];
                    W[p][k] = lc * xi + ls * yi;
                    W[q][k] = -ls * xi + lc * yi;
                }
            }
            const double tc = jc, ts = -js;
            if (!(tc == 1.0 && ts == 0.0)) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double xi = W[k][p], yi = W[k][q];
                    W[k][p] = tc * xi + ts * yi;
                    W[k][q] = -ts * xi + tc * yi;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double xi = Vm[k][p], yi = Vm[k][q];
                    Vm[k][p] = tc * xi + ts * yi;
                    Vm[k][q] = -ts * xi + tc * yi;
                }
            }
            const double md = fabs(W[p][p]) > fabs(W[q][q]) ? fabs(W[p][p]) : fabs(W[q][q]);
            if (md > max_diag) max_diag = md;
        }
    }
    for (int i = 0; i < 3; ++i) sv[i] = fabs(W[i][i]) * scale;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int pos = 0;
        double mx = sv[i];
        for (int k = i + 1; k < 3; ++k) if (sv[k] > mx) { mx = sv[k]; pos = k - i; }
        if (mx == 0.0) break;
        if (pos) {
            pos += i;
            const double ts = sv[i]; sv[i] = sv[pos]; sv[pos] = ts;
            for (int k = 0; k < 3; ++k) { const double tv = Vm[k][pos]; Vm[k][pos] = Vm[k][i]; Vm[k][i] = tv; }
        }
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[3 * i + j] = Vm[i][j];
}

// mean / covariance / JacobiSVD of ComputeNeighborhood (neighborhood.h:236-244,293-311) from the running sums, in the reference
// build's arithmetic (no fused multiply-add): normal = V[:, 2], a2D = (sqrt s1 - sqrt s2) / sqrt s0
CTGN_HD void normal_a2d_exact(int n, Vec3 S, const double SS9[9], Vec3 &normal, double &a2d) {
#pragma clang fp contract(off)
    const double dn = (double) n;
    const double b[3] = {S.x / dn, S.y / dn, S.z / dn};
    double C[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { const double m = SS9[3 * r + c] / dn; C[3 * r + c] = m - b[r] * b[c]; }
    double sv[3], V[9];
    jacobi_svd3_exact(C, sv, V);
    normal = Vec3{V[2], V[5], V[8]};
    a2d = (sqrt(fabs(sv[1])) - sqrt(fabs(sv[2]))) / sqrt(fabs(sv[0]));
}

// Diagonally pivoted LDL^T solve of a symmetric 12x12 system (Eigen LDLT::compute + solve, ct_icp.cpp:914).
// m is the full matrix, row-major, destroyed.
CTGN_HD void ldlt_solve12(double *m, const double *b, double *x) {
    const int N = 12;
    int transp[12];
    double temp[12];
    for (int k = 0; k < N; ++k) {
        int big = k;
        double best = fabs(m[k * N + k]);
        for (int i = k + 1; i < N; ++i)
            if (fabs(m[i * N + i]) > best) { best = fabs(m[i * N + i]); big = i; }
        transp[k] = big;
        if (big != k) {
            for (int j = 0; j < k; ++j) { double t = m[k * N + j]; m[k * N + j] = m[big * N + j]; m[big * N + j] = t; }
            for (int i = big + 1; i < N; ++i) { double t = m[i * N + k]; m[i * N + k] = m[i * N + big]; m[i * N + big] = t; }
            for (int i = k + 1; i < big; ++i) { double t = m[i * N + k]; m[i * N + k] = m[big * N + i]; m[big * N + i] = t; }
            double t = m[k * N + k]; m[k * N + k] = m[big * N + big]; m[big * N + big] = t;
        }
        if (k > 0) {
            for (int j = 0; j < k; ++j) temp[j] = m[j * N + j] * m[k * N + j];
            double acc = 0;
            for (int j = 0; j < k; ++j) acc += m[k * N + j] * temp[j];
            m[k * N + k] -= acc;
            for (int i = k + 1; i < N; ++i) {
                double a2 = 0;
                for (int j = 0; j < k; ++j) a2 += m[i * N + j] * temp[j];
                m[i * N + k] -= a2;
            }
        }
        double akk = m[k * N + k];
        if (fabs(akk) > 0)
            for (int i = k + 1; i < N; ++i) m[i * N + k] /= akk;
    }
    double y[12];
    for (int i = 0; i < N; ++i) y[i] = b[i];
    for (int k = 0; k < N; ++k)
        if (transp[k] != k) { double t = y[k]; y[k] = y[transp[k]]; y[transp[k]] = t; }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < i; ++j) y[i] -= m[i * N + j] * y[j];
    for (int i = 0; i < N; ++i) y[i] = (fabs(m[i * N + i]) > DBL_MIN) ? y[i] / m[i * N + i] : 0.0;
    for (int i = N - 1; i >= 0; --i)
        for (int j = i + 1; j < N; ++j) y[i] -= m[j * N + i] * y[j];
    for (int k = N - 1; k >= 0; --k)
        if (transp[k] != k) { double t = y[k]; y[k] = y[transp[k]]; y[transp[k]] = t; }
    for (int i = 0; i < N; ++i) x[i] = y[i];
}

}  // namespace ctgn
