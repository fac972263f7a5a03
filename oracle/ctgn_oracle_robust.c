/*
 * ctgn_oracle_robust.c — CPU restatement of the robust-loss (CERES-profile) registration of jedeschaud/ct_icp:
 * CT_ICP_Registration::DoRegisterCeres (reference src/ct_icp/ct_icp.cpp:457-707) for
 * parametrization = CONTINUOUS_TIME, distance = POINT_TO_PLANE — the configuration every shipped config selects.
 *
 * TEST INFRASTRUCTURE ONLY (see ctgn_oracle.h). Pinned against oracle/_ref (the reference's DoRegisterCeres and cost
 * functors on a restated Ceres: tests/test_oracle_vs_ref.py, 5 losses x 4 configurations, poses to ~1e-16) — which pins
 * this file to the reference's code, NOT to a real Ceres build: the inner solver of
 * this route is the third-party Ceres Solver (pulled in through an external superbuild at `master`,
 * superbuild/CMakeLists.txt:20-33 — version unpinned; absent from the reference tree and from this image).
 * Restated here from its published algorithm:
 *   - ceres::AutoDiffCostFunction      -> forward-mode dual numbers ("jets") over the reference's functor
 *                                         (include/ct_icp/cost_functions.h:46-58,200-225), exact derivatives;
 *   - ceres::EigenQuaternionParameterization (Plus: q' = [sin|d|/|d| d, cos|d|] (x) q, and its 4x3 Jacobian);
 *   - ceres::CauchyLoss / HuberLoss / TolerantLoss (loss_function.cc) and ct_icp::TruncatedLoss
 *     (src/ct_icp/cost_function.cpp:5-15); the Triggs corrector (corrector.cc);
 *   - the trust-region minimiser with the Levenberg-Marquardt strategy at the defaults the reference leaves
 *     untouched (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc of Ceres 2.0: initial radius 1e4,
 *     min_relative_decrease 1e-3, min/max LM diagonal 1e-6/1e32, Jacobi scaling from the first Jacobian,
 *     function / gradient / parameter tolerances 1e-6 / 1e-10 / 1e-8, radius update r/max(1/3, 1-(2q-1)^3),
 *     halving with a doubling factor on rejection). The linear solve is a dense Cholesky of the damped normal
 *     equations (Ceres would use QR or sparse Cholesky: same minimiser of the same linear least squares).
 * Equivalence with a real Ceres build is therefore "same algorithm, same stationary points", not bit-exact.
 */
#include "ctgn_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * jets: value + 14 partials (pose layout: begin qx qy qz qw tx ty tz | end ...)
 * ---------------------------------------------------------------------------------------------- */
#define NJ 14
typedef struct {
    double v, d[NJ];
} jet;

static jet j_const(double v) { jet r; r.v = v; memset(r.d, 0, sizeof(r.d)); return r; }
static jet j_var(double v, int k) { jet r = j_const(v); r.d[k] = 1.0; return r; }
static jet j_add(jet a, jet b) { jet r; r.v = a.v + b.v; for (int i = 0; i < NJ; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static jet j_sub(jet a, jet b) { jet r; r.v = a.v - b.v; for (int i = 0; i < NJ; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
static jet j_mul(jet a, jet b) { jet r; r.v = a.v * b.v; for (int i = 0; i < NJ; ++i) r.d[i] = a.v * b.d[i] + a.d[i] * b.v; return r; }
static jet j_div(jet a, jet b) {
    jet r; double inv = 1.0 / b.v; r.v = a.v * inv;
    for (int i = 0; i < NJ; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
static jet j_neg(jet a) { jet r; r.v = -a.v; for (int i = 0; i < NJ; ++i) r.d[i] = -a.d[i]; return r; }
static jet j_scale(double s, jet a) { jet r; r.v = s * a.v; for (int i = 0; i < NJ; ++i) r.d[i] = s * a.d[i]; return r; }
static jet j_sqrt(jet a) { jet r; r.v = sqrt(a.v); double k = 0.5 / r.v; for (int i = 0; i < NJ; ++i) r.d[i] = k * a.d[i]; return r; }
static jet j_sin(jet a) { jet r; r.v = sin(a.v); double k = cos(a.v); for (int i = 0; i < NJ; ++i) r.d[i] = k * a.d[i]; return r; }
static jet j_acos(jet a) { jet r; r.v = acos(a.v); double k = -1.0 / sqrt(1.0 - a.v * a.v); for (int i = 0; i < NJ; ++i) r.d[i] = k * a.d[i]; return r; }

typedef struct { jet x, y, z, w; } jquat;

static jquat jq_normalized(jquat q) {
    jet n2 = j_add(j_add(j_mul(q.x, q.x), j_mul(q.y, q.y)), j_add(j_mul(q.z, q.z), j_mul(q.w, q.w)));
    jet n = j_sqrt(n2);
    jquat r = {j_div(q.x, n), j_div(q.y, n), j_div(q.z, n), j_div(q.w, n)};
    return r;
}

/* Eigen QuaternionBase::slerp on jets (branches decided on the scalar parts, as Ceres' Jet comparisons do) */
static jquat jq_slerp(jquat a, double t, jquat b) {
    const double one = 1.0 - DBL_EPSILON;
    jet d = j_add(j_add(j_mul(a.x, b.x), j_mul(a.y, b.y)), j_add(j_mul(a.z, b.z), j_mul(a.w, b.w)));
    jet absd = d.v < 0 ? j_neg(d) : d;
    jet s0, s1;
    if (absd.v >= one) {
        s0 = j_const(1.0 - t);
        s1 = j_const(t);
    } else {
        jet theta = j_acos(absd), sin_theta = j_sin(theta);
        s0 = j_div(j_sin(j_scale(1.0 - t, theta)), sin_theta);
        s1 = j_div(j_sin(j_scale(t, theta)), sin_theta);
    }
    if (d.v < 0) s1 = j_neg(s1);
    jquat r = {j_add(j_mul(s0, a.x), j_mul(s1, b.x)), j_add(j_mul(s0, a.y), j_mul(s1, b.y)),
               j_add(j_mul(s0, a.z), j_mul(s1, b.z)), j_add(j_mul(s0, a.w), j_mul(s1, b.w))};
    return r;
}

/* Eigen q * v on jets (v constant) */
static void jq_rotate(jquat q, const double v[3], jet out[3]) {
    jet uvx = j_sub(j_scale(v[2], q.y), j_scale(v[1], q.z));
    jet uvy = j_sub(j_scale(v[0], q.z), j_scale(v[2], q.x));
    jet uvz = j_sub(j_scale(v[1], q.x), j_scale(v[0], q.y));
    uvx = j_add(uvx, uvx); uvy = j_add(uvy, uvy); uvz = j_add(uvz, uvz);
    out[0] = j_add(j_add(j_const(v[0]), j_mul(q.w, uvx)), j_sub(j_mul(q.y, uvz), j_mul(q.z, uvy)));
    out[1] = j_add(j_add(j_const(v[1]), j_mul(q.w, uvy)), j_sub(j_mul(q.z, uvx), j_mul(q.x, uvz)));
    out[2] = j_add(j_add(j_const(v[2]), j_mul(q.w, uvz)), j_sub(j_mul(q.x, uvy), j_mul(q.y, uvx)));
}

/* EigenQuaternionParameterization::ComputeJacobian (4x3, rows x y z w) */
static void quat_plus_jacobian(const double q[4], double J[12]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    J[0] = w;  J[1] = z;  J[2] = -y;
    J[3] = -z; J[4] = w;  J[5] = x;
    J[6] = y;  J[7] = -x; J[8] = w;
    J[9] = -x; J[10] = -y; J[11] = -z;
}

/* EigenQuaternionParameterization::Plus */
static void quat_plus(const double q[4], const double delta[3], double out[4]) {
    const double n = sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    if (n == 0.0) { memcpy(out, q, 4 * sizeof(double)); return; }
    const double k = sin(n) / n;
    const double dx = k * delta[0], dy = k * delta[1], dz = k * delta[2], dw = cos(n);
    /* dq (x) q, Eigen quaternion product */
    out[3] = dw * q[3] - dx * q[0] - dy * q[1] - dz * q[2];
    out[0] = dw * q[0] + dx * q[3] + dy * q[2] - dz * q[1];
    out[1] = dw * q[1] + dy * q[3] + dz * q[0] - dx * q[2];
    out[2] = dw * q[2] + dz * q[3] + dx * q[1] - dy * q[0];
}

/* tangent layout of the whole problem: [begin_quat(3) | end_quat(3) | begin_t(3) | end_t(3)] —
 * the order AddParameterBlocks registers the blocks (ct_icp.cpp:227-230) */
static void pose_plus(const double pose[14], const double delta[12], double out[14]) {
    quat_plus(pose, delta, out);
    quat_plus(pose + 7, delta + 3, out + 7);
    for (int i = 0; i < 3; ++i) { out[4 + i] = pose[4 + i] + delta[6 + i]; out[11 + i] = pose[11 + i] + delta[9 + i]; }
}

static void ambient_to_tangent(const double pose[14], const double g14[14], double g12[12]) {
    double Jb[12], Je[12];
    quat_plus_jacobian(pose, Jb);
    quat_plus_jacobian(pose + 7, Je);
    for (int c = 0; c < 3; ++c) {
        g12[c] = g14[0] * Jb[c] + g14[1] * Jb[3 + c] + g14[2] * Jb[6 + c] + g14[3] * Jb[9 + c];
        g12[3 + c] = g14[7] * Je[c] + g14[8] * Je[3 + c] + g14[9] * Je[6 + c] + g14[10] * Je[9 + c];
        g12[6 + c] = g14[4 + c];
        g12[9 + c] = g14[11 + c];
    }
}

/* CTFunctor<FunctorPointToPlane>::operator() (cost_functions.h:209-225 -> :46-58). residual and its 12 tangent
 * partials (jac may be NULL). */
void orc_ct_point_to_plane(const double pose[14], double alpha, const double raw[3], const double ref[3],
                           const double normal[3], double weight, double *residual, double *jac) {
    jquat qb = {j_var(pose[0], 0), j_var(pose[1], 1), j_var(pose[2], 2), j_var(pose[3], 3)};
    jquat qe = {j_var(pose[7], 7), j_var(pose[8], 8), j_var(pose[9], 9), j_var(pose[10], 10)};
    jquat qi = jq_normalized(jq_slerp(jq_normalized(qb), alpha, jq_normalized(qe)));     /* :214-217 */
    jet tr[3];
    for (int i = 0; i < 3; ++i)                                                            /* :219-222 */
        tr[i] = j_add(j_scale(1.0 - alpha, j_var(pose[4 + i], 4 + i)), j_scale(alpha, j_var(pose[11 + i], 11 + i)));
    jet p[3];
    jq_rotate(jq_normalized(qi), raw, p);                                                  /* :48-49 */
    jet prod = j_const(0.0);
    for (int i = 0; i < 3; ++i) {
        p[i] = j_add(p[i], tr[i]);                                                         /* :50-52 */
        prod = j_add(prod, j_scale(normal[i], j_sub(j_const(ref[i]), p[i])));              /* :54-55 */
    }
    jet r = j_scale(weight, prod);                                                         /* :56 */
    *residual = r.v;
    if (jac) ambient_to_tangent(pose, r.d, jac);
}

/* ceres::LossFunction::Evaluate for the five choices of ct_icp.cpp:170-187. rho = {rho, rho', rho''}. */
void orc_loss_evaluate(int kind, double sigma, double tolerant_min, double s, double rho[3]) {
    switch (kind) {
        case ORC_LOSS_CAUCHY: {
            const double b = sigma * sigma, c = 1.0 / b;
            const double sum = 1.0 + s * c, inv = 1.0 / sum;
            rho[0] = b * log(sum);
            rho[1] = inv > DBL_MIN ? inv : DBL_MIN;
            rho[2] = -c * (inv * inv);
            return;
        }
        case ORC_LOSS_HUBER: {
            const double a = sigma, b = sigma * sigma;
            if (s > b) {
                const double r = sqrt(s);
                rho[0] = 2.0 * a * r - b;
                rho[1] = (a / r) > DBL_MIN ? (a / r) : DBL_MIN;
                rho[2] = -rho[1] / (2.0 * s);
            } else {
                rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
            }
            return;
        }
        case ORC_LOSS_TOLERANT: {                         /* TolerantLoss(a = ls_tolerant_min_threshold, b = ls_sigma) */
            const double a = tolerant_min, b = sigma;
            const double c = b * log(1.0 + exp(-a / b));
            const double x = (s - a) / b;
            const double kLog2Pow53 = 36.7;
            if (x > kLog2Pow53) {
                rho[0] = s - a - c; rho[1] = 1.0; rho[2] = 0.0;
            } else {
                const double e_x = exp(x);
                rho[0] = b * log(1.0 + e_x) - c;
                rho[1] = (e_x / (1.0 + e_x)) > DBL_MIN ? (e_x / (1.0 + e_x)) : DBL_MIN;
                rho[2] = 0.5 / (b * (1.0 + cosh(x)));
            }
            return;
        }
        case ORC_LOSS_TRUNCATED: {                        /* cost_function.cpp:5-15 */
            const double s2 = sigma * sigma;
            if (s < s2) { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
            else { rho[0] = s2; rho[1] = 0.0; rho[2] = 0.0; }
            return;
        }
        default:
            rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

/* ------------------------------------------------------------------------------------------------
 * the least-squares problem of one ICP iteration
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    size_t n;                     /* residual blocks */
    const double *raw;            /* [n][3] */
    const double *ref;            /* [n][3] */
    const double *normal;         /* [n][3] */
    const double *weight, *alpha; /* [n] */
    const orc_robust_options *opt;
    const orc_robust_prior *prior;
} problem_t;

static void add_outer(double H[144], double g[12], const double J[12], double r) {
    for (int i = 0; i < 12; ++i) {
        g[i] += J[i] * r;
        for (int j = 0; j < 12; ++j) H[12 * i + j] += J[i] * J[j];
    }
}

/* Evaluate cost = 1/2 sum rho(r^2) (+ regularisers) and, if H != NULL, H = J~^T J~, g = J~^T r~ with the
 * loss-corrected residuals / Jacobians (corrector.cc). */
static double problem_evaluate(const problem_t *P, const double pose[14], double H[144], double g[12]) {
    const orc_robust_options *o = P->opt;
    double cost = 0.0;
    if (H) { memset(H, 0, 144 * sizeof(double)); memset(g, 0, 12 * sizeof(double)); }
    for (size_t k = 0; k < P->n; ++k) {
        double r, J[12];
        orc_ct_point_to_plane(pose, P->alpha[k], P->raw + 3 * k, P->ref + 3 * k, P->normal + 3 * k, P->weight[k], &r,
                              H ? J : NULL);
        const double s = r * r;
        if (o->loss_function == ORC_LOSS_STANDARD) {      /* loss_function == nullptr */
            cost += 0.5 * s;
            if (H) add_outer(H, g, J, r);
            continue;
        }
        double rho[3];
        orc_loss_evaluate(o->loss_function, o->ls_sigma, o->ls_tolerant_min_threshold, s, rho);
        cost += 0.5 * rho[0];
        if (!H) continue;
        const double sqrt_rho1 = sqrt(rho[1]);
        double residual_scaling, alpha_sq_norm;
        if (s == 0.0 || rho[2] <= 0.0) {
            residual_scaling = sqrt_rho1;
            alpha_sq_norm = 0.0;
        } else {
            const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
            const double al = 1.0 - sqrt(D);
            residual_scaling = sqrt_rho1 / (1.0 - al);
            alpha_sq_norm = al / s;
        }
        const double jscale = sqrt_rho1 * (1.0 - alpha_sq_norm * s);     /* scalar residual: J - a r r^T J */
        for (int i = 0; i < 12; ++i) J[i] *= jscale;
        add_outer(H, g, J, r * residual_scaling);
    }
    /* PreviousFrameMotionModel::AddConstraintsToCeresProblem (src/ct_icp/motion_model.cpp:12-61), no loss */
    if (P->prior) {
        const orc_robust_prior *p = P->prior;
        const double nres = (double) P->n;
        const double *tb = pose + 4, *te = pose + 11;
        if (p->beta_location_consistency > 0.) {                         /* :18-27, cost_functions.h:262-282 */
            const double beta = sqrt(nres * p->beta_location_consistency);
            for (int c = 0; c < 3; ++c) {
                const double r = beta * (tb[c] - p->previous_end_tr[c]);
                cost += 0.5 * r * r;
                if (H) { double J[12] = {0}; J[6 + c] = beta; add_outer(H, g, J, r); }
            }
        }
        if (p->beta_orientation_consistency > 0.) {                      /* :31-39, cost_functions.h:285-306 */
            const double beta = sqrt(nres * p->beta_orientation_consistency);
            const double *q = pose, *qp = p->previous_end_quat;
            const double sc = q[0] * qp[0] + q[1] * qp[1] + q[2] * qp[2] + q[3] * qp[3];
            const double r = beta * (1.0 - sc * sc);
            cost += 0.5 * r * r;
            if (H) {
                double g14[14] = {0}, J[12];
                for (int c = 0; c < 4; ++c) g14[c] = -2.0 * beta * sc * qp[c];
                ambient_to_tangent(pose, g14, J);
                for (int c = 3; c < 12; ++c) J[c] = 0.0;
                add_outer(H, g, J, r);
            }
        }
        if (p->beta_constant_velocity > 0.) {                            /* :42-50, cost_functions.h:309-330 */
            const double beta = sqrt(nres * p->beta_constant_velocity);
            for (int c = 0; c < 3; ++c) {
                const double vel = p->previous_end_tr[c] - p->previous_begin_tr[c];
                const double r = beta * (te[c] - tb[c] - vel);
                cost += 0.5 * r * r;
                if (H) { double J[12] = {0}; J[6 + c] = -beta; J[9 + c] = beta; add_outer(H, g, J, r); }
            }
        }
        if (p->beta_small_velocity > 0.) {                               /* :53-60, cost_functions.h:339-354 */
            const double beta = sqrt(nres * p->beta_small_velocity);
            for (int c = 0; c < 3; ++c) {
                const double r = beta * (tb[c] - te[c]);
                cost += 0.5 * r * r;
                if (H) { double J[12] = {0}; J[6 + c] = beta; J[9 + c] = -beta; add_outer(H, g, J, r); }
            }
        }
    }
    return cost;
}

/* dense Cholesky solve of a 12x12 SPD system; returns 0 when a pivot is not positive / finite */
static int cholesky_solve12(const double A[144], const double b[12], double x[12]) {
    double L[144];
    memset(L, 0, sizeof(L));
    for (int j = 0; j < 12; ++j) {
        double d = A[12 * j + j];
        for (int k = 0; k < j; ++k) d -= L[12 * j + k] * L[12 * j + k];
        if (!(d > 0.0) || !isfinite(d)) return 0;
        L[12 * j + j] = sqrt(d);
        for (int i = j + 1; i < 12; ++i) {
            double v = A[12 * i + j];
            for (int k = 0; k < j; ++k) v -= L[12 * i + k] * L[12 * j + k];
            L[12 * i + j] = v / L[12 * j + j];
        }
    }
    double y[12];
    for (int i = 0; i < 12; ++i) {
        double v = b[i];
        for (int k = 0; k < i; ++k) v -= L[12 * i + k] * y[k];
        y[i] = v / L[12 * i + i];
    }
    for (int i = 11; i >= 0; --i) {
        double v = y[i];
        for (int k = i + 1; k < 12; ++k) v -= L[12 * k + i] * x[k];
        x[i] = v / L[12 * i + i];
    }
    for (int i = 0; i < 12; ++i) if (!isfinite(x[i])) return 0;
    return 1;
}

static double gradient_max_norm(const double pose[14], const double g[12]) {
    double neg[12], moved[14], m = 0.0;
    for (int i = 0; i < 12; ++i) neg[i] = -g[i];
    pose_plus(pose, neg, moved);
    for (int i = 0; i < 14; ++i) { double d = fabs(pose[i] - moved[i]); if (d > m) m = d; }
    return m;
}

/* ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT, max_num_iterations = ls_max_num_iters (ct_icp.cpp:484-487).
 * Returns the termination: 0 NO_CONVERGENCE (iteration budget), 1 CONVERGENCE, -1 FAILURE (solution unusable). */
static int lm_minimize(const problem_t *P, double pose[14], int max_num_iterations, orc_lm_report *rep) {
    const double min_relative_decrease = 1e-3, min_diag = 1e-6, max_diag = 1e32;
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double max_radius = 1e16, min_radius = 1e-32;
    double radius = 1e4, decrease_factor = 2.0;
    double H[144], g[12], scale[12];
    double x_cost = problem_evaluate(P, pose, H, g);
    if (rep) { memset(rep, 0, sizeof(*rep)); rep->initial_cost = x_cost; rep->final_cost = x_cost; }
    if (!isfinite(x_cost)) return -1;
    for (int i = 0; i < 12; ++i) scale[i] = 1.0 / (1.0 + sqrt(H[13 * i]));      /* jacobi_scaling, first Jacobian only */
    int invalid = 0, iteration = 0, term = 0;
    for (;;) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        if (iteration >= max_num_iterations) { term = 0; break; }
        if (gradient_max_norm(pose, g) <= gradient_tolerance) { term = 1; break; }
        if (radius < min_radius) { term = 1; break; }
        ++iteration;
        /* LevenbergMarquardtStrategy::ComputeStep on the column-scaled Jacobian */
        double Hs[144], gs[12], A[144], rhs[12], y[12];
        for (int i = 0; i < 12; ++i) {
            gs[i] = scale[i] * g[i];
            for (int j = 0; j < 12; ++j) Hs[12 * i + j] = scale[i] * H[12 * i + j] * scale[j];
        }
        memcpy(A, Hs, sizeof(A));
        for (int i = 0; i < 12; ++i) {
            double d = Hs[13 * i];
            d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d);
            A[13 * i] += d / radius;
            rhs[i] = -gs[i];
        }
        int ok = cholesky_solve12(A, rhs, y);
        double model_cost_change = 0.0;
        if (ok) {                                       /* -(J y).(r + J y / 2) */
            double yg = 0.0, yHy = 0.0;
            for (int i = 0; i < 12; ++i) {
                yg += y[i] * gs[i];
                double row = 0.0;
                for (int j = 0; j < 12; ++j) row += Hs[12 * i + j] * y[j];
                yHy += y[i] * row;
            }
            model_cost_change = -(yg + 0.5 * yHy);
            ok = model_cost_change > 0.0;
        }
        if (!ok) {                                      /* HandleInvalidStep */
            if (++invalid >= 5) { term = -1; break; }
            radius *= 0.5;
            continue;
        }
        invalid = 0;
        double delta[12], cand[14];
        for (int i = 0; i < 12; ++i) delta[i] = y[i] * scale[i];
        pose_plus(pose, delta, cand);
        const double cand_cost = problem_evaluate(P, cand, NULL, NULL);
        double step2 = 0.0, x2 = 0.0;
        for (int i = 0; i < 14; ++i) { step2 += (pose[i] - cand[i]) * (pose[i] - cand[i]); x2 += pose[i] * pose[i]; }
        if (sqrt(step2) <= parameter_tolerance * (sqrt(x2) + parameter_tolerance)) { term = 1; break; }
        const double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= function_tolerance * x_cost) { term = 1; break; }
        const double relative_decrease = cost_change / model_cost_change;
        if (isfinite(cand_cost) && relative_decrease > min_relative_decrease) {      /* HandleSuccessfulStep */
            memcpy(pose, cand, 14 * sizeof(double));
            x_cost = problem_evaluate(P, pose, H, g);
            double f = 1.0 - pow(2.0 * relative_decrease - 1.0, 3.0);
            radius = radius / (f > 1.0 / 3.0 ? f : 1.0 / 3.0);
            if (radius > max_radius) radius = max_radius;
            decrease_factor = 2.0;
            if (rep) rep->num_successful_steps++;
        } else {                                        /* StepRejected */
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            if (rep) rep->num_unsuccessful_steps++;
        }
    }
    if (rep) { rep->final_cost = x_cost; rep->iterations = iteration; rep->termination = term; rep->final_radius = radius; }
    return term;
}

/* entry point for tests: solve one fixed-correspondence problem */
int orc_robust_solve_fixed(const double *raw, const double *ref, const double *normal, const double *weight,
                           const double *alpha, size_t n, const orc_robust_options *opts, const orc_robust_prior *prior,
                           double pose[14], int max_num_iterations, orc_lm_report *rep) {
    problem_t P = {n, raw, ref, normal, weight, alpha, opts, prior};
    return lm_minimize(&P, pose, max_num_iterations, rep);
}

/* cost / normal equations of a fixed-correspondence problem at `pose` (tests) */
double orc_robust_evaluate_fixed(const double *raw, const double *ref, const double *normal, const double *weight,
                                 const double *alpha, size_t n, const orc_robust_options *opts,
                                 const orc_robust_prior *prior, const double pose[14], double *H, double *g) {
    problem_t P = {n, raw, ref, normal, weight, alpha, opts, prior};
    return problem_evaluate(&P, pose, H, g);
}

/* slam::AngularDistance (include/SlamCore/types.h:142-150), degrees */
static double angular_distance_deg(const double qa[4], const double qb[4]) {
    double Ra[9], Rb[9];
    double a[4] = {qa[0], qa[1], qa[2], qa[3]}, b[4] = {qb[0], qb[1], qb[2], qb[3]};
    orc_quat_normalize(a); orc_quat_normalize(b);          /* TSE3::Rotation(): quat.normalized() */
    orc_quat_to_matrix(a, Ra);
    orc_quat_to_matrix(b, Rb);
    double tr = 0.0;                                       /* trace(Ra Rb^T) = sum_ij Ra_ij Rb_ij */
    for (int i = 0; i < 9; ++i) tr += Ra[i] * Rb[i];
    double c = (tr - 1.0) / 2.0;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    return acos(c) * (180.0 / M_PI);
}

/* Build the residual blocks of one ICP iteration (ct_icp.cpp:540-600): neighbourhood, normal, weight, one block per
 * (keypoint, i < num_closest_neighbors), then the max_num_residuals cap of GetProblem (:415-426). Arrays hold
 * n * num_closest_neighbors entries; returns the number of blocks kept (packed to the front, index order). */
size_t orc_robust_build(const orc_map *m, const double *raw_xyz, const double *world_xyz, const double *t, size_t n,
                        const double tbe[2], const orc_robust_options *o, int heap_mode, double *raw_out,
                        double *ref_out, double *normal_out, double *weight_out, double *alpha_out,
                        int32_t *keypoint_out) {
    double lw = fabs(o->weight_alpha), ln = fabs(o->weight_neighborhood);       /* :525-532 */
    const double sum = lw + ln;
    lw /= sum; ln /= sum;
    size_t kept = 0;
    for (size_t k = 0; k < n; ++k) {
        double nb[3 * ORC_MAX_NEIGHBORS];
        const double *world = world_xyz + 3 * k;
        /* DefaultNearestNeighborStrategy -> map.ComputeNeighborhoodInPlace(world, max_num_neighbors)
         * (neighborhood_strategy.h:77-83); strategy max_num_neighbors taken from max_number_neighbors */
        int cnt = orc_map_radius_search(m, world, -1.0, o->max_number_neighbors, heap_mode, nb);
        if (cnt < o->min_number_neighbors) continue;                            /* :566-567 */
        double normal[3], a2d;
        if (!orc_neighborhood(nb, cnt, normal, &a2d)) continue;                 /* :569 (undefined below 5 points) */
        /* :570-573: normal . (BeginTr - BeginTr) < 0 never holds: the normal keeps the eigen-solver's sign */
        double weight = pow(a2d, o->power_planarity);                           /* :574 */
        const double dx = nb[0] - world[0], dy = nb[1] - world[1], dz = nb[2] - world[2];
        weight = lw * weight + ln * exp(-sqrt(dx * dx + (dy * dy + dz * dz)) /
                                        (o->max_dist_to_plane_ct_icp * o->min_number_neighbors));   /* :576-579 */
        const double alpha = orc_alpha_timestamp(t[k], tbe[0], tbe[1]);        /* :592 */
        for (int i = 0; i < o->num_closest_neighbors; ++i) {                    /* :585-595 */
            if (o->max_num_residuals > 0 && kept >= (size_t) o->max_num_residuals) break;   /* :418-423 */
            memcpy(raw_out + 3 * kept, raw_xyz + 3 * k, 3 * sizeof(double));
            memcpy(ref_out + 3 * kept, nb + 3 * i, 3 * sizeof(double));
            memcpy(normal_out + 3 * kept, normal, 3 * sizeof(double));
            weight_out[kept] = weight;
            alpha_out[kept] = alpha;
            if (keypoint_out) keypoint_out[kept] = (int32_t) k;
            ++kept;
        }
    }
    return kept;
}

/* DoRegisterCeres (ct_icp.cpp:457-707), CONTINUOUS_TIME + POINT_TO_PLANE. world_xyz is overwritten (the reference
 * re-derives it from the poses at the top of every iteration, :499-515,537). Returns 0, -5 on a timestamp outside
 * [t_begin, t_end] (glog CHECK in the reference), -3 when the solver reports an unusable solution (:628-631 throws). */
int orc_register_robust(const orc_map *m, const double *raw_xyz, double *world_xyz, const double *t, size_t n,
                        double pose[14], const double tbe[2], const orc_robust_options *o, const orc_robust_prior *prior,
                        int heap_mode, orc_summary *summary) {
    memset(summary, 0, sizeof(*summary));
    for (size_t i = 0; i < n; ++i)
        if (!(tbe[0] <= t[i] && t[i] <= tbe[1])) return -5;
    if (!(fabs(o->weight_alpha) + fabs(o->weight_neighborhood) > 0.0)) return -1;      /* CHECK at :529 */
    if (o->num_closest_neighbors < 1 || o->num_closest_neighbors > o->min_number_neighbors) return -1;
    orc_quat_normalize(pose);                                                   /* :476-477 */
    orc_quat_normalize(pose + 7);
    double prev[14];
    memcpy(prev, pose, sizeof(prev));                                           /* :500-501 */
    const size_t cap = n * (size_t) o->num_closest_neighbors;
    double *buf = (double *) malloc((cap ? cap : 1) * 11 * sizeof(double));
    double *raw_r = buf, *ref_r = buf + 3 * cap, *nrm_r = buf + 6 * cap, *w_r = buf + 9 * cap, *a_r = buf + 10 * cap;
    int iter = 0, rc = 0, nres = 0;
    for (; iter < o->num_iters_icp; ++iter) {                                   /* :535 */
        for (size_t i = 0; i < n; ++i)                                          /* transform_keypoints :537 */
            orc_transform_point(pose, tbe, t[i], raw_xyz + 3 * i, world_xyz + 3 * i);
        nres = (int) orc_robust_build(m, raw_xyz, world_xyz, t, n, tbe, o, heap_mode, raw_r, ref_r, nrm_r, w_r, a_r, NULL);
        if (nres < o->min_number_neighbors) {                                   /* :612-624 */
            snprintf(summary->error_log, sizeof(summary->error_log),
                     "[CT_ICP] Error : not enough keypoints selected in ct-icp !\n[CT_ICP] number_of_residuals : %d\n", nres);
            if (o->debug_print) fputs(summary->error_log, stdout);
            summary->success = 0;
            summary->num_residuals_used = nres;
            free(buf);
            return 0;
        }
        problem_t P = {(size_t) nres, raw_r, ref_r, nrm_r, w_r, a_r, o, prior};
        if (lm_minimize(&P, pose, o->ls_max_num_iters, NULL) < 0) { rc = -3; break; }   /* :627-637 */
        orc_quat_normalize(pose);                                               /* :633-634, :640-641 */
        orc_quat_normalize(pose + 7);
        double diff_trans = 0.0, db = 0.0, de = 0.0;                            /* :643-646 */
        for (int c = 0; c < 3; ++c) {
            db += (prev[4 + c] - pose[4 + c]) * (prev[4 + c] - pose[4 + c]);
            de += (prev[11 + c] - pose[11 + c]) * (prev[11 + c] - pose[11 + c]);
        }
        diff_trans = sqrt(db) + sqrt(de);
        const double diff_rot = angular_distance_deg(pose, prev) + angular_distance_deg(pose + 7, prev + 7);
        memcpy(prev, pose, sizeof(prev));                                       /* :648-649 */
        summary->last_step_norm = diff_trans;
        if (diff_rot < o->threshold_orientation_norm && diff_trans < o->threshold_translation_norm) break;   /* :662-667 */
    }
    free(buf);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i)                                              /* :685 */
        orc_transform_point(pose, tbe, t[i], raw_xyz + 3 * i, world_xyz + 3 * i);
    summary->success = 1;                                                       /* :691-693 */
    summary->num_residuals_used = nres;
    summary->num_iters = iter;
    orc_quat_normalize(pose);                                                   /* :698-699 */
    orc_quat_normalize(pose + 7);
    return 0;
}
