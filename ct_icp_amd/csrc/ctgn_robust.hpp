// ctgn_robust.hpp — robust-loss (CERES-profile) registration on the GPU: SURVEY.md section 8f row 4.
//
// Replaces CT_ICP_Registration::DoRegisterCeres (reference src/ct_icp/ct_icp.cpp:457-707) for
// parametrization = CONTINUOUS_TIME, distance = POINT_TO_PLANE. Per ICP iteration:
//   k_accumulate_rows   the same neighbour search as the GN route (ctgn_kernels.hpp)
//   k_robust_prepare    lane per keypoint: normal + a2D + weight (:569-579), the num_closest_neighbors reference points
//   k_robust_cap        max_num_residuals cap in keypoint order (:415-426), soft failure (:612-624), solver reset
//   (ls_max_num_iters + 1) x [ k_robust_eval   residual + closed-form Jacobian + loss -> packed J^T J | J^T r | cost, at the
//                                              start pose first, then at each candidate
//                              k_robust_step ] regularisers (motion_model.cpp:12-61); accept / reject the candidate just evaluated
//                                              and update the radius; checks; next Levenberg-Marquardt step -> next candidate
//   One full evaluation per LM iteration instead of Ceres' two (cost at the candidate, then Jacobian at the accepted point): an
//   accepted candidate's normal equations ARE the next iteration's, a rejected one leaves the previous ones in place.
//   k_robust_outer      normalise, ICP stop test (:640-667)
// Everything stays on the device; the host enqueues the fixed worst-case sequence and the kernels turn into no-ops once
// the device-side flags say the inner solve or the ICP loop is finished.
//
// The inner solver is Ceres' trust-region Levenberg-Marquardt minimiser restated (DESIGN.md section 9 lists what is
// taken from Ceres' published algorithm and at which defaults); the derivative Ceres obtains by automatic differentiation is evaluated in closed form:
// with R_b^T R_e = Exp(theta u) and R(a) = R_b Exp(a theta u),
//     d p / d w_b = -[R raw]x + R [raw]x W_b R_b^T,    d p / d w_e = -R [raw]x W_e R_e^T,
//     W = a u u^T + s (cos(psi) (I - u u^T) + sin(psi) [u]x),   s = sin(a theta/2) / sin(theta/2),
//     psi_e = (1 - a) theta/2,  psi_b = -(1 + a) theta/2,
// in the tangent of Ceres' EigenQuaternionParameterization (a left rotation by twice the tangent vector).
#pragma once

#include "ctgn_kernels.hpp"

namespace ctgn {

enum { LOSS_STANDARD = 0, LOSS_CAUCHY = 1, LOSS_HUBER = 2, LOSS_TOLERANT = 3, LOSS_TRUNCATED = 4 };

struct RobustParams {
    int min_nb, max_nb, num_closest, max_res, loss, ls_max_iters;
    double lambda_w, lambda_n, power, nbr_scale;       // nbr_scale = max_dist_to_plane * min_number_neighbors
    double sigma, tol_min;
    double thr_rot_deg, thr_trans;
    int has_prior;
    double beta_loc, beta_vel, beta_small, beta_orient;
    double prev_b[3], prev_e[3], prev_q[4];
};

struct PoseCtx {             // a begin|end pose with everything that depends on the pair only
    double pose[14];
    double theta, sin_theta;  // slerp constants (quaternion half-angle of the relative rotation)
    int linear, negate;
    double ux, uy, uz;        // axis of R_b^T R_e in the begin/end body frame
};

struct RobustState {
    PoseCtx x, cand;
    double prev[14];          // pose at the end of the previous ICP iteration (:500-501, :648-649)
    double H[144], g[12], scale[12];
    double x_cost, cand_cost, model_cost_change, radius, decrease_factor;
    double diff_trans, diff_rot;
    int have_scale, ls_iter, ls_done, ls_term, invalid, step_valid;
    int eval_at_x, _pad0;     // 1: the pending evaluation is of x (start of a solve), 0: of cand
    int ls_iters_total, ls_accepted_total;
    int n_res;                // residual blocks of the current ICP iteration
    int icp_iter;             // the reference's loop counter `iter` (:535) as ICPSummary::num_iters reports it
    int error;                // the inner solver gave up (reference throws, :628-631)
    int converged;
    unsigned long long step_cycles[8];   // shader clocks of the last k_robust_step: stage-in+reduce | control | scale | solve | candidate | write-back
};

struct RobustBuf {            // per-keypoint output of k_robust_prepare, SoA with stride cap
    double *nx, *ny, *nz, *w, *alpha;
    double *ref;              // [3][num_closest][cap]
    int *rank;                // k_robust_prepare: 1 valid / 0 not;  k_robust_cap: rank of the keypoint's first block, -1
    size_t cap;
};

// ------------------------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ Quat quat_mul(Quat a, Quat b) {       // Eigen a * b
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}

__device__ inline void pose_ctx_prepare(PoseCtx &c) {
    const Quat qb{c.pose[0], c.pose[1], c.pose[2], c.pose[3]}, qe{c.pose[7], c.pose[8], c.pose[9], c.pose[10]};
    const SlerpPair sp = slerp_prepare(qb, qe);
    c.theta = sp.theta; c.sin_theta = sp.sin_theta; c.linear = sp.linear; c.negate = sp.negate;
    Quat qr = quat_mul(Quat{-qb.x, -qb.y, -qb.z, qb.w}, qe);
    if (qr.w < 0) qr = Quat{-qr.x, -qr.y, -qr.z, -qr.w};
    const double n = sqrt(qr.x * qr.x + qr.y * qr.y + qr.z * qr.z);
    if (sp.linear || !(n > 0)) { c.ux = 1.0; c.uy = 0.0; c.uz = 0.0; }
    else { c.ux = qr.x / n; c.uy = qr.y / n; c.uz = qr.z / n; }
}

// EigenQuaternionParameterization::Plus on both quaternions + plain addition on the translations; tangent layout
// [begin_quat | end_quat | begin_t | end_t] (the order of AddParameterBlocks, ct_icp.cpp:227-230)
__device__ inline void quat_plus(const double *q, const double *d, double *out) {
    const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n == 0.0) { out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3]; return; }
    const double k = sin(n) / n;
    const Quat r = quat_mul(Quat{k * d[0], k * d[1], k * d[2], cos(n)}, Quat{q[0], q[1], q[2], q[3]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
__device__ inline void pose_plus(const double *pose, const double *delta, double *out) {
    quat_plus(pose, delta, out);
    quat_plus(pose + 7, delta + 3, out + 7);
    for (int i = 0; i < 3; ++i) { out[4 + i] = pose[4 + i] + delta[6 + i]; out[11 + i] = pose[11 + i] + delta[9 + i]; }
}

// ceres::LossFunction::Evaluate for the five choices of ct_icp.cpp:170-187
__device__ __forceinline__ void loss_evaluate(int kind, double sigma, double tol_min, double s, double rho[3]) {
    switch (kind) {
        case LOSS_CAUCHY: {
            const double b = sigma * sigma, c = 1.0 / b, sum = 1.0 + s * c, inv = 1.0 / sum;
            rho[0] = b * log(sum); rho[1] = fmax(DBL_MIN, inv); rho[2] = -c * (inv * inv);
            return;
        }
        case LOSS_HUBER: {
            const double b = sigma * sigma;
            if (s > b) {
                const double r = sqrt(s);
                rho[0] = 2.0 * sigma * r - b; rho[1] = fmax(DBL_MIN, sigma / r); rho[2] = -rho[1] / (2.0 * s);
            } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
            return;
        }
        case LOSS_TOLERANT: {
            const double a = tol_min, b = sigma, c = b * log(1.0 + exp(-a / b)), x = (s - a) / b;
            if (x > 36.7) { rho[0] = s - a - c; rho[1] = 1.0; rho[2] = 0.0; }
            else {
                const double ex = exp(x);
                rho[0] = b * log(1.0 + ex) - c; rho[1] = fmax(DBL_MIN, ex / (1.0 + ex)); rho[2] = 0.5 / (b * (1.0 + cosh(x)));
            }
            return;
        }
        case LOSS_TRUNCATED: {
            const double s2 = sigma * sigma;
            if (s < s2) { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
            else { rho[0] = s2; rho[1] = 0.0; rho[2] = 0.0; }
            return;
        }
        default: rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

// CTFunctor<FunctorPointToPlane> (include/ct_icp/cost_functions.h:46-58,200-225): r = m . (ref - p(alpha)), m = weight * normal,
// and (JAC) its 12 tangent partials in closed form.
template <bool JAC>
__device__ __forceinline__ double ct_residual(const PoseCtx &c, double alpha, Vec3 raw, Vec3 ref, Vec3 m, double *J) {
    const Quat qb{c.pose[0], c.pose[1], c.pose[2], c.pose[3]}, qe{c.pose[7], c.pose[8], c.pose[9], c.pose[10]};
    const SlerpPair sp{c.theta, c.sin_theta, c.linear, c.negate};
    const Quat qi = quat_normalized(slerp_eval(qb, qe, sp, alpha));
    const Vec3 a = quat_rotate(qi, raw);
    const double oma = 1.0 - alpha;
    const Vec3 p{a.x + (oma * c.pose[4] + alpha * c.pose[11]), a.y + (oma * c.pose[5] + alpha * c.pose[12]),
                 a.z + (oma * c.pose[6] + alpha * c.pose[13])};
    const double r = dot(m, ref - p);
    if (JAC) {
        const Vec3 mb = quat_rotate(Quat{-qi.x, -qi.y, -qi.z, qi.w}, m);      // normal in the interpolated body frame
        const Vec3 cc = cross(mb, raw);
        Vec3 wb, we;
        if (c.linear) {
            wb = alpha * cc; we = wb;
        } else {
            const Vec3 u{c.ux, c.uy, c.uz};
            const double s = sin(alpha * c.theta) / c.sin_theta;
            const double uc = dot(u, cc);
            const Vec3 perp = cc - uc * u, uxc = cross(u, cc), ax = (alpha * uc) * u;
            double sb, cb, se, ce;
            sincos(-(1.0 + alpha) * c.theta, &sb, &cb);
            sincos(oma * c.theta, &se, &ce);
            wb = ax + s * (cb * perp - sb * uxc);        // W_b^T c
            we = ax + s * (ce * perp - se * uxc);        // W_e^T c
        }
        // alpha == 1: slerp(q_b, q_e, 1) is q_e whatever q_b, and the reference's Jets say so exactly (Eigen's slerp gives scale0 = sin(0 * theta) /
        // sin(theta) and scale1 = sin(theta) / sin(theta), whose derivatives are 0 and (da - 1 * db) / b with da == db). The closed form
        // cancels a x m against R_b W_b^T c only to rounding, and a column of 1e-16s is not harmless: Levenberg-Marquardt damps a (near)
        // zero column with min_diagonal / radius = 1e-10, so gradient noise of 1e-14 becomes a step of 1e-4. Frames 0 and 1 of a sequence
        // carry the end timestamp on every point (odometry.cpp:354-359): there the begin orientation must stay where it was, bit for bit.
        const Vec3 jb = oma == 0.0 ? Vec3{0.0, 0.0, 0.0} : cross(a, m) + quat_rotate(qb, wb), je = quat_rotate(qe, we);
        J[0] = -2.0 * jb.x; J[1] = -2.0 * jb.y; J[2] = -2.0 * jb.z;
        J[3] = 2.0 * je.x; J[4] = 2.0 * je.y; J[5] = 2.0 * je.z;
        J[6] = -oma * m.x; J[7] = -oma * m.y; J[8] = -oma * m.z;
        J[9] = -alpha * m.x; J[10] = -alpha * m.y; J[11] = -alpha * m.z;
    }
    return r;
}

// ================================================================================================
// k_robust_prepare — lane per keypoint (ct_icp.cpp:548-596 without the solver bookkeeping)
// ================================================================================================
__global__ __launch_bounds__(256) void k_robust_prepare(MapView map, KpView kp, const GnState *st, RobustParams prm, RobustBuf rb) {
    __shared__ TieScratch s_tie[4];
    if (st->done) return;
    const char *pbase = reinterpret_cast<const char *>(map.blocks);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < kp.n; k += gridDim.x * blockDim.x) {
        const uint4 *in4 = reinterpret_cast<const uint4 *>(kp.sel + (size_t) k * SEL_STRIDE);
        uint32_t rec32[SEL_STRIDE];
#pragma unroll
        for (int q = 0; q < SEL_STRIDE / 4; ++q) {
            const uint4 v4 = in4[q];
            rec32[4 * q] = v4.x; rec32[4 * q + 1] = v4.y; rec32[4 * q + 2] = v4.z; rec32[4 * q + 3] = v4.w;
        }
        // (nearly) tied candidates: the reference's own queue is replayed for this keypoint (resolve_ties, ctgn_kernels.hpp)
        resolve_ties(map, kp, k, true, rec32, s_tie[threadIdx.x >> 6], (int) (threadIdx.x & 63), prm.max_nb);
        const int n_all = min((int) rec32[0], KMAX);
        const int n = (n_all >= prm.min_nb && n_all >= 5) ? n_all : 0;          // invalid below (:566-567): nothing to gather, and the search kernel hands over no offsets
        Vec3 S{0, 0, 0}, q0{0, 0, 0};
        double SS9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        // the reference's neighbour vector is farthest-first and summed front to back (neighborhood.h:236-240): the record is nearest-first,
        // walked from entry n - 1 down to entry 0. WITHOUT fused multiply-adds: this route weighs rank-deficient neighbourhoods
        // too (:574-579), whose "normal" is decided by the roundings (jacobi_svd3_exact)
#pragma unroll
        for (int g = KMAX / 8 - 1; g >= 0; --g) {
            if (8 * g < n) {
                double gx[8], gy[8], gz[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t off = (8 * g + q < n) ? rec32[1 + 8 * g + q] : 0u;
                    load_point(pbase, off, gx[q], gy[q], gz[q]);
                }
#pragma unroll
                for (int q = 7; q >= 0; --q) {
                    if (8 * g + q < n) {
#pragma clang fp contract(off)
                        const double x = gx[q], y = gy[q], z = gz[q];
                        const int i_ref = n - 1 - (8 * g + q);                    // its index in the reference's vector
                        if (i_ref == 0) q0 = Vec3{x, y, z};
                        S.x += x; S.y += y; S.z += z;
                        const double xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z, zz = z * z;
                        SS9[0] += xx; SS9[1] += xy; SS9[2] += xz; SS9[3] += xy; SS9[4] += yy; SS9[5] += yz; SS9[6] += xz; SS9[7] += yz; SS9[8] += zz;
                        if (i_ref < prm.num_closest) {                            // neighborhood.points[i], :585-595
                            const size_t at = (size_t) i_ref * rb.cap + k;
                            rb.ref[at] = x;
                            rb.ref[(size_t) prm.num_closest * rb.cap + at] = y;
                            rb.ref[(size_t) 2 * prm.num_closest * rb.cap + at] = z;
                        }
                    }
                }
            }
        }
        const bool valid = n >= prm.min_nb && n >= 5;                          // :566-567 ; neighborhood.h:227-230
        Vec3 nrm{0, 0, 0};
        double weight = 0.0;
        if (valid) {
            double a2d;
            normal_a2d_exact(n, S, SS9, nrm, a2d);      // :570-573 never flips: normal . (BeginTr - BeginTr) = 0
            const Vec3 p{kp.wx[k], kp.wy[k], kp.wz[k]};
            const Vec3 d = q0 - p;
            weight = prm.lambda_w * pow(a2d, prm.power) + prm.lambda_n * exp(-sqrt(dot(d, d)) / prm.nbr_scale);   // :574-579
        }
        rb.nx[k] = nrm.x; rb.ny[k] = nrm.y; rb.nz[k] = nrm.z;
        rb.w[k] = weight;
        rb.alpha[k] = alpha_timestamp(kp.t[k], st->tbe[0], st->tbe[1]);         // :592
        rb.rank[k] = valid ? 1 : 0;
    }
}

// ================================================================================================
// k_robust_cap — one block: rank of every valid keypoint's first residual block in index order, the
// max_num_residuals cap (GetProblem, :415-426), the soft failure (:612-624) and the reset of the inner solver.
// ================================================================================================
constexpr int CAP_BLOCK = 1024;
__global__ __launch_bounds__(CAP_BLOCK) void k_robust_cap(GnState *st, RobustState *rs, RobustParams prm, RobustBuf rb, int n) {
    if (st->done) return;
    const int tid = threadIdx.x;
    const int chunk = (n + CAP_BLOCK - 1) / CAP_BLOCK;
    const int lo = min(n, tid * chunk), hi = min(n, lo + chunk);
    int cnt = 0;
    for (int k = lo; k < hi; ++k) cnt += rb.rank[k];
    // exclusive scan of the 1024 chunk counts: shuffle scan inside each wave, then the 16 wave totals
    const int lane = tid & 63, wave = tid >> 6;
    int incl = cnt;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __shared__ int s_wave[CAP_BLOCK / 64];
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, all = 0;
    for (int w = 0; w < CAP_BLOCK / 64; ++w) { const int v = s_wave[w]; if (w < wave) base += v; all += v; }
    int run = base + incl - cnt;
    if (tid == 0) {
        long long total = (long long) all * prm.num_closest;
        if (prm.max_res > 0 && total > prm.max_res) total = prm.max_res;
        rs->n_res = (int) total;
        st->n_used = (int) total;
        if (total < prm.min_nb) {                              // :612-624
            st->failed = 1;
            st->done = 1;
        } else {                                               // a fresh ceres::Solve: new strategy, new scaling
            PoseCtx c;                                         // in registers: no global read-after-write chains
            for (int i = 0; i < 14; ++i) c.pose[i] = st->pose[i];
            pose_ctx_prepare(c);
            rs->x = c;
            rs->radius = 1e4; rs->decrease_factor = 2.0;
            rs->have_scale = 0; rs->ls_iter = 0; rs->ls_done = 0; rs->ls_term = 0; rs->invalid = 0; rs->step_valid = 0;
            rs->eval_at_x = 1;
        }
    }
    for (int k = lo; k < hi; ++k) {
        const int v = rb.rank[k];
        rb.rank[k] = v ? run * prm.num_closest : -1;
        run += v;
    }
}

// ================================================================================================
// k_robust_eval — lane per keypoint, its num_closest residual blocks in turn: residual + Jacobian at rs->x (first
// evaluation of a solve) or rs->cand, loss + Triggs corrector (Ceres corrector.cc), packed J^T J | -J^T r | cost.
// ================================================================================================
constexpr int EVAL_BLOCK = 256;
constexpr int EVAL_REC = 15;                 // 12 J | r | cost | pad (odd stride: conflict-free LDS writes)

typedef double rb_d4_t __attribute__((ext_vector_type(4)));

// One wave's share of an evaluation: its lanes take keypoints tile * BLK + tid of the tiles tile_first, tile_first + tile_step, ...;
// the wave's records go through `rec` (64 x EVAL_REC doubles of LDS) into the 16 x 16 FP64 MFMA accumulator accm (U^T U).
template <int BLK>
__device__ __forceinline__ void robust_eval_tiles(const KpView &kp, const PoseCtx &ctx, const RobustParams &prm, const RobustBuf &rb, int tile_first,
                                                  int tile_step, int tid, double *rec, rb_d4_t &accm) {
    const int lane = tid & 63;
    const int ntiles = (kp.n + BLK - 1) / BLK;
    const size_t ncap = (size_t) prm.num_closest * rb.cap;
    for (int tile = tile_first; tile < ntiles; tile += tile_step) {
        const int k = tile * BLK + tid;
        int rank = -1;
        Vec3 raw{0, 0, 0}, m{0, 0, 0};
        double alpha = 0.0;
        if (k < kp.n) {
            rank = rb.rank[k];
            if (rank >= 0) {
                raw = Vec3{kp.rx[k], kp.ry[k], kp.rz[k]};
                const double w = rb.w[k];
                m = Vec3{w * rb.nx[k], w * rb.ny[k], w * rb.nz[k]};
                alpha = rb.alpha[k];
            }
        }
        for (int i = 0; i < prm.num_closest; ++i) {
            const bool used = rank >= 0 && (prm.max_res <= 0 || rank + i < prm.max_res);
            double J[12], r = 0.0, cost = 0.0;
            if (used) {
                const size_t at = (size_t) i * rb.cap + k;
                const Vec3 ref{rb.ref[at], rb.ref[ncap + at], rb.ref[2 * ncap + at]};
                r = ct_residual<true>(ctx, alpha, raw, ref, m, J);
                const double s = r * r;
                if (prm.loss == LOSS_STANDARD) {
                    cost = 0.5 * s;
                } else {
                    double rho[3];
                    loss_evaluate(prm.loss, prm.sigma, prm.tol_min, s, rho);
                    cost = 0.5 * rho[0];
                    const double sqrt_rho1 = sqrt(rho[1]);
                    double residual_scaling, alpha_sq_norm;
                    if (s == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
                    else {
                        const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
                        const double al = 1.0 - sqrt(D);
                        residual_scaling = sqrt_rho1 / (1.0 - al);
                        alpha_sq_norm = al / s;
                    }
                    const double js = sqrt_rho1 * (1.0 - alpha_sq_norm * s);
#pragma unroll
                    for (int c = 0; c < 12; ++c) J[c] *= js;
                    r *= residual_scaling;
                }
            }
            // U = the wave's 64 x 15 records (J | r | cost | 1): the packed sums are entries of U^T U, sixteen FP64 MFMAs
            // (see k_residual_reduce); cost rides along as (U^T U)[13][14]
            double *my = rec + lane * EVAL_REC;
#pragma unroll
            for (int c = 0; c < 12; ++c) my[c] = used ? J[c] : 0.0;
            my[12] = used ? r : 0.0;
            my[13] = cost;
            my[14] = 1.0;
            const int comp = lane & 15;
#pragma unroll 4
            for (int k0 = 0; k0 < 64; k0 += 4) {
                const double v = (comp < EVAL_REC) ? rec[(k0 + (lane >> 4)) * EVAL_REC + comp] : 0.0;
                accm = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, accm, 0, 0, 0);
            }
        }
    }
}
// the wave's packed J^T J | -J^T r | cost out of the accumulator (the f64 MFMA's C / D layout) into comb[SYS_N]
__device__ __forceinline__ void robust_unpack(const rb_d4_t &accm, int lane, double *comb) {
    if (lane < SYS_N - SYS_USED) comb[SYS_USED + lane] = 0.0;
    const int col = lane & 15;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = (lane >> 4) + 4 * m;
        if (row < 12 && col >= row && col < 12) comb[row * 12 - (row * (row - 1)) / 2 + (col - row)] = accm[m];
        if (row < 12 && col == 12) comb[78 + row] = -accm[m];
        if (row == 13 && col == 14) comb[90] = accm[m];
    }
}

__global__ __launch_bounds__(EVAL_BLOCK) void k_robust_eval(KpView kp, const GnState *st, const RobustState *rs, RobustParams prm,
                                                             RobustBuf rb, double *partials) {
    __shared__ double s_rec[EVAL_BLOCK / 64][64 * EVAL_REC];
    __shared__ double s_comb[EVAL_BLOCK / 64][SYS_N];
    __shared__ PoseCtx s_ctx;
    if (st->done || rs->ls_done) return;
    const int at_x = rs->eval_at_x;
    if (!at_x && !rs->step_valid) return;        // the last step was invalid: nothing new to evaluate
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_ctx = at_x ? rs->x : rs->cand;
    __syncthreads();
    rb_d4_t accm = {0.0, 0.0, 0.0, 0.0};
    robust_eval_tiles<EVAL_BLOCK>(kp, s_ctx, prm, rb, (int) blockIdx.x, (int) gridDim.x, tid, s_rec[wave], accm);
    robust_unpack(accm, lane, s_comb[wave]);
    __syncthreads();
    if (tid < SYS_N) {
        double t = 0.0;
        for (int w = 0; w < EVAL_BLOCK / 64; ++w) t += s_comb[w][tid];
        partials[(size_t) blockIdx.x * SYS_N + tid] = t;            // block-major (reduce_partials, ctgn_kernels.hpp)
    }
}

// regularisers of PreviousFrameMotionModel::AddConstraintsToCeresProblem (src/ct_icp/motion_model.cpp:12-61), no loss.
// H (full 12x12) and g may be null (cost only).
__device__ inline double robust_regularisers(const RobustParams &prm, int n_res, const double *pose, double *H, double *g) {
    if (!prm.has_prior) return 0.0;
    double cost = 0.0;
    const double nres = (double) n_res;
    const double *tb = pose + 4, *te = pose + 11;
    if (prm.beta_loc > 0.) {                                          // :18-27
        const double beta = sqrt(nres * prm.beta_loc);
        for (int c = 0; c < 3; ++c) {
            const double r = beta * (tb[c] - prm.prev_e[c]);
            cost += 0.5 * r * r;
            if (H) { H[13 * (6 + c)] += beta * beta; g[6 + c] += beta * r; }
        }
    }
    if (prm.beta_orient > 0.) {                                       // :31-39
        const double beta = sqrt(nres * prm.beta_orient);
        const double sc = pose[0] * prm.prev_q[0] + pose[1] * prm.prev_q[1] + pose[2] * prm.prev_q[2] + pose[3] * prm.prev_q[3];
        const double r = beta * (1.0 - sc * sc);
        cost += 0.5 * r * r;
        if (H) {
            const double k = -2.0 * beta * sc;
            const double ax = k * prm.prev_q[0], ay = k * prm.prev_q[1], az = k * prm.prev_q[2], aw = k * prm.prev_q[3];
            const double x = pose[0], y = pose[1], z = pose[2], w = pose[3];
            // ambient gradient times the 4x3 Jacobian of Plus at zero
            const double J[3] = {ax * w - ay * z + az * y - aw * x, ax * z + ay * w - az * x - aw * y,
                                 -ax * y + ay * x + az * w - aw * z};
            for (int i = 0; i < 3; ++i) {
                g[i] += J[i] * r;
                for (int j = 0; j < 3; ++j) H[12 * i + j] += J[i] * J[j];
            }
        }
    }
    if (prm.beta_vel > 0.) {                                          // :42-50
        const double beta = sqrt(nres * prm.beta_vel);
        for (int c = 0; c < 3; ++c) {
            const double r = beta * (te[c] - tb[c] - (prm.prev_e[c] - prm.prev_b[c]));
            cost += 0.5 * r * r;
            if (H) {
                const double b2 = beta * beta;
                H[13 * (6 + c)] += b2; H[13 * (9 + c)] += b2;
                H[12 * (6 + c) + 9 + c] -= b2; H[12 * (9 + c) + 6 + c] -= b2;
                g[6 + c] -= beta * r; g[9 + c] += beta * r;
            }
        }
    }
    if (prm.beta_small > 0.) {                                        // :53-60
        const double beta = sqrt(nres * prm.beta_small);
        for (int c = 0; c < 3; ++c) {
            const double r = beta * (tb[c] - te[c]);
            cost += 0.5 * r * r;
            if (H) {
                const double b2 = beta * beta;
                H[13 * (6 + c)] += b2; H[13 * (9 + c)] += b2;
                H[12 * (6 + c) + 9 + c] -= b2; H[12 * (9 + c) + 6 + c] -= b2;
                g[6 + c] += beta * r; g[9 + c] -= beta * r;
            }
        }
    }
    return cost;
}

// Lane i < 12 holds row i of a symmetric positive definite 12x12 matrix (rowr) and b_i; returns x_i of A x = b.
// Right-looking LDL^T with the pivot row broadcast by v_readlane (no LDS), then the two substitutions the same way.
// All 64 lanes must call it; ok = every pivot positive.
#define CTGN_BCAST(v, src) __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src))
__device__ __forceinline__ double wave_spd_solve12(double (&rowr)[12], double bi, int i, bool &ok) {
    ok = true;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const double dk = CTGN_BCAST(rowr[k], k);
        ok = ok && (dk > 0.0) && isfinite(dk);
        const double lik = (i > k) ? rowr[k] / dk : 0.0;
#pragma unroll
        for (int j = k + 1; j < 12; ++j) {
            const double vkj = CTGN_BCAST(rowr[j], k);
            if (i > k) rowr[j] -= lik * vkj;
        }
        if (i > k) rowr[k] = lik;
    }
    double y = bi, dii = 1.0;
#pragma unroll
    for (int j = 0; j < 12; ++j) {                // forward: after step j, y_j is final
        const double yj = CTGN_BCAST(y, j);
        if (i > j) y -= rowr[j] * yj;
        if (i == j) dii = rowr[j];
    }
    y = y / dii;
#pragma unroll
    for (int j = 11; j >= 1; --j) {               // backward with L^T: L[j][c] lives in lane j, register c
        const double xj = CTGN_BCAST(y, j);
#pragma unroll
        for (int c = 0; c < j; ++c) {
            const double ljc = CTGN_BCAST(rowr[c], j);
            if (i == c) y -= ljc * xj;
        }
    }
    return y;
}

// ================================================================================================
// k_robust_step — one block, after every k_robust_eval:
//   reduce the partials of the pose just evaluated (x at the start of a solve, otherwise the candidate) and add the
//   regularisers; for a candidate: function tolerance, accept / reject, trust-region radius; then the checks of
//   FinalizeIterationAndCheckIfMinimizerCanContinue, the next Levenberg-Marquardt step and the next candidate.
// The 12x12 work runs on wave 0 with one matrix row per lane; only the quaternion bookkeeping is single-lane. The solver
// state is staged in LDS for the whole kernel and written back once at the end: a dependent chain of global-memory
// read-after-writes on one lane costs ~1 us per link on this part.
// ================================================================================================
constexpr int STEP_BLOCK = 1024;
#define RWSYNC() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

struct StepScratch {                   // LDS of one Levenberg-Marquardt step (wave 0)
    double sys[SYS_N];
    __attribute__((aligned(8))) RobustState R;
    double Hc[144], gc[12];            // normal equations of the pose just evaluated
    double delta[12];
    int flag;
};
static_assert(sizeof(RobustState) % 8 == 0, "RobustState is copied as doubles");

// Everything of a step behind the reduction, by ONE wave: S.sys holds the packed sums of the pose just evaluated (if have_eval), S.R the
// solver state, which is written back to rs at the end.
__device__ __forceinline__ void robust_step_wave0(StepScratch &S, bool have_eval, GnState *st, RobustState *rs, const RobustParams &prm, int lane,
                                                  unsigned long long tc0) {
    RobustState &R = S.R;
    constexpr int NW = (int) (sizeof(RobustState) / 8);
    unsigned long long tc1 = tc0, tc2 = tc0, tc3 = tc0, tc4 = tc0, tc5 = tc0;
    const double min_relative_decrease = 1e-3, min_diag = 1e-6, max_diag = 1e32;
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double max_radius = 1e16, min_radius = 1e-32;
    bool stop_all = false;                                 // lane 0: raise GnState::done
    tc1 = __builtin_readcyclecounter();
    const int have_scale = R.have_scale;
    const int i = lane < 12 ? lane : 0;
    if (have_eval) {
        for (int e = lane; e < 78; e += 64) {
            const int r = c_tri_i[e], c = c_tri_j[e];
            S.Hc[12 * r + c] = S.sys[e];
            S.Hc[12 * c + r] = S.sys[e];
        }
        if (lane < 12) S.gc[lane] = -S.sys[78 + lane];
    }
    RWSYNC();
    if (lane == 0) {
        int flag = 0, take = 0;                            // take: S.Hc / S.gc become the current normal equations
        if (have_eval) {
            const double *pose_e = R.eval_at_x ? R.x.pose : R.cand.pose;
            const double cost = S.sys[90] + robust_regularisers(prm, R.n_res, pose_e, S.Hc, S.gc);
            if (R.eval_at_x) {                             // Ceres' iteration 0
                R.x_cost = cost;
                R.eval_at_x = 0;
                take = 1;
                if (!isfinite(cost)) { R.error = 1; R.ls_done = 1; stop_all = true; flag = 1; }
            } else {                                       // the candidate of the step computed last time
                R.cand_cost = cost;
                const double cost_change = R.x_cost - cost;
                if (fabs(cost_change) <= function_tolerance * R.x_cost) { R.ls_done = 1; R.ls_term = 1; flag = 1; }
                else {
                    const double rd = cost_change / R.model_cost_change;
                    if (isfinite(cost) && rd > min_relative_decrease) {          // HandleSuccessfulStep
                        R.x = R.cand;
                        for (int c = 0; c < 14; ++c) st->pose[c] = R.cand.pose[c];
                        st->slerp_theta = R.cand.theta; st->slerp_sin = R.cand.sin_theta;
                        st->slerp_linear = R.cand.linear; st->slerp_negate = R.cand.negate;
                        const double t = 2.0 * rd - 1.0, f = 1.0 - t * t * t;
                        R.radius = fmin(max_radius, R.radius / fmax(1.0 / 3.0, f));
                        R.decrease_factor = 2.0;
                        R.ls_accepted_total += 1;
                        R.x_cost = cost;
                        take = 1;
                    } else {                                                     // StepRejected
                        R.radius = R.radius / R.decrease_factor;
                        R.decrease_factor *= 2.0;
                    }
                }
                R.step_valid = 0;
            }
        }
        S.flag = flag | (take << 1);
    }
    RWSYNC();
    if (S.flag & 2) {                                      // adopt the evaluated normal equations
        for (int e = lane; e < 144; e += 64) R.H[e] = S.Hc[e];
        if (lane < 12) R.g[lane] = S.gc[lane];
    }
    RWSYNC();
    const double radius = R.radius;
    if (lane == 0 && !(S.flag & 1)) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        int flag = 0;
        if (R.ls_iter >= prm.ls_max_iters) { R.ls_done = 1; R.ls_term = 0; flag = 1; }
        else {
            double neg[12], moved[14], mx = 0.0;              // gradient tolerance: max norm of x - Plus(x, -g)
#pragma unroll
            for (int c = 0; c < 12; ++c) neg[c] = -R.g[c];
            pose_plus(R.x.pose, neg, moved);
#pragma unroll
            for (int c = 0; c < 14; ++c) mx = fmax(mx, fabs(R.x.pose[c] - moved[c]));
            if (mx <= gradient_tolerance || radius < min_radius) { R.ls_done = 1; R.ls_term = 1; flag = 1; }
        }
        if (!flag) { R.ls_iter += 1; R.ls_iters_total += 1; }
        S.flag = flag;
    }
    RWSYNC();
    tc2 = __builtin_readcyclecounter();
    // jacobi_scaling: from the first Jacobian of the solve only
    const double sc_i = have_scale ? R.scale[i] : 1.0 / (1.0 + sqrt(R.H[13 * i]));
    RWSYNC();
    if (lane < 12) R.scale[lane] = sc_i;
    if (lane == 0) R.have_scale = 1;
    RWSYNC();
    tc3 = __builtin_readcyclecounter();
    tc4 = tc3;
    if (!(S.flag & 1)) {
        // LevenbergMarquardtStrategy::ComputeStep on the column-scaled Jacobian: (Hs + clamp(diag Hs) / radius) y = -gs
        double rowr[12], hs[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) { hs[j] = sc_i * R.H[12 * i + j] * R.scale[j]; rowr[j] = hs[j]; }
#pragma unroll
        for (int j = 0; j < 12; ++j)
            if (j == i) rowr[j] += fmin(fmax(hs[j], min_diag), max_diag) / radius;
        const double gs_i = sc_i * R.g[i];
        bool ok;
        const double y = wave_spd_solve12(rowr, -gs_i, lane < 12 ? lane : 63, ok);
        double row = 0.0;
#pragma unroll
        for (int j = 0; j < 12; ++j) row += hs[j] * CTGN_BCAST(y, j);
        const double model_cost_change = -wave_sum_fixed(lane < 12 ? y * (gs_i + 0.5 * row) : 0.0);   // -(J y).(r + J y / 2)
        const bool finite_all = ballot64(lane < 12 && !isfinite(y)) == 0ull;
        ok = ok && finite_all && model_cost_change > 0.0;
        if (lane < 12) S.delta[lane] = y * sc_i;
        RWSYNC();
        tc4 = __builtin_readcyclecounter();
        if (lane == 0) {
            if (!ok) {                                      // HandleInvalidStep
                R.step_valid = 0;
                if (++R.invalid >= 5) { R.error = 1; R.ls_done = 1; stop_all = true; }
                else R.radius = radius * 0.5;
            } else {
                R.invalid = 0;
                R.model_cost_change = model_cost_change;
                double delta[12];
#pragma unroll
                for (int c = 0; c < 12; ++c) delta[c] = S.delta[c];
                pose_plus(R.x.pose, delta, R.cand.pose);
                pose_ctx_prepare(R.cand);
                R.step_valid = 1;
                double step2 = 0.0, x2 = 0.0;
#pragma unroll
                for (int c = 0; c < 14; ++c) {
                    const double d = R.x.pose[c] - R.cand.pose[c];
                    step2 += d * d;
                    x2 += R.x.pose[c] * R.x.pose[c];
                }
                if (sqrt(step2) <= parameter_tolerance * (sqrt(x2) + parameter_tolerance)) { R.ls_done = 1; R.ls_term = 1; }
            }
        }
    }
    RWSYNC();
    tc5 = __builtin_readcyclecounter();
    if (lane == 0) {
        R.step_cycles[0] = tc1 - tc0; R.step_cycles[1] = tc2 - tc1; R.step_cycles[2] = tc3 - tc2; R.step_cycles[3] = tc4 - tc3;
        R.step_cycles[4] = tc5 - tc4;
    }
    RWSYNC();
    for (int w = lane; w < NW; w += 64) reinterpret_cast<double *>(rs)[w] = reinterpret_cast<const double *>(&R)[w];
    if (lane == 0 && stop_all) st->done = 1;
}

__global__ __launch_bounds__(STEP_BLOCK) void k_robust_step(const double *partials, int nblocks, GnState *st, RobustState *rs,
                                                             RobustParams prm) {
    __shared__ StepScratch S;
    constexpr int NW = (int) (sizeof(RobustState) / 8);
    if (st->done || rs->ls_done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long tc0 = __builtin_readcyclecounter();
    for (int w = tid; w < NW; w += STEP_BLOCK) reinterpret_cast<double *>(&S.R)[w] = reinterpret_cast<const double *>(rs)[w];
    __syncthreads();
    const bool have_eval = S.R.eval_at_x || S.R.step_valid;        // did the preceding k_robust_eval run?
    // wave w sums entries w, w + 16, ... over the blocks: lane-strided, then a fixed shuffle tree (deterministic)
    if (have_eval) {
        for (int e = wave; e < SYS_USED; e += STEP_BLOCK / 64) {
            double acc = 0.0;
            for (int b = lane; b < nblocks; b += 64) acc += partials[(size_t) b * SYS_N + e];
            acc = wave_sum_fixed(acc);
            if (lane == 0) S.sys[e] = acc;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    robust_step_wave0(S, have_eval, st, rs, prm, lane, tc0);
}

// slam::AngularDistance (include/SlamCore/types.h:142-150), degrees
__device__ inline double angular_distance_deg(const double *qa, const double *qb) {
    double Ra[9], Rb[9];
    quat_to_matrix(quat_normalized(Quat{qa[0], qa[1], qa[2], qa[3]}), Ra);
    quat_to_matrix(quat_normalized(Quat{qb[0], qb[1], qb[2], qb[3]}), Rb);
    double tr = 0.0;
    for (int i = 0; i < 9; ++i) tr += Ra[i] * Rb[i];
    double c = (tr - 1.0) / 2.0;
    c = fmin(1.0, fmax(-1.0, c));
    return acos(c) * (180.0 / M_PI);
}

// End of one ICP iteration (ct_icp.cpp:633-667): normalise, pose change since the previous iteration, stop test.
__device__ __forceinline__ void robust_outer_body(GnState *st, RobustState *rs, const RobustParams &prm) {       // ONE lane
    if (st->done) return;
    double p[14], prev[14];
    for (int i = 0; i < 14; ++i) { p[i] = st->pose[i]; prev[i] = rs->prev[i]; }
    const int icp_iter = rs->icp_iter, iter = st->iter;
    const Quat qb = quat_normalized(Quat{p[0], p[1], p[2], p[3]});
    const Quat qe = quat_normalized(Quat{p[7], p[8], p[9], p[10]});
    p[0] = qb.x; p[1] = qb.y; p[2] = qb.z; p[3] = qb.w;
    p[7] = qe.x; p[8] = qe.y; p[9] = qe.z; p[10] = qe.w;
    const SlerpPair sp = slerp_prepare(qb, qe);
    double db = 0.0, de = 0.0;
    for (int c = 0; c < 3; ++c) {
        db += (prev[4 + c] - p[4 + c]) * (prev[4 + c] - p[4 + c]);
        de += (prev[11 + c] - p[11 + c]) * (prev[11 + c] - p[11 + c]);
    }
    const double diff_trans = sqrt(db) + sqrt(de);
    const double diff_rot = angular_distance_deg(p, prev) + angular_distance_deg(p + 7, prev + 7);
    for (int i = 0; i < 14; ++i) { st->pose[i] = p[i]; rs->prev[i] = p[i]; }
    st->slerp_theta = sp.theta; st->slerp_sin = sp.sin_theta; st->slerp_linear = sp.linear; st->slerp_negate = sp.negate;
    rs->diff_trans = diff_trans; rs->diff_rot = diff_rot;
    st->step_norm = diff_trans;
    st->iter = iter + 1;
    if (diff_rot < prm.thr_rot_deg && diff_trans < prm.thr_trans) { rs->converged = 1; st->done = 1; }   // :662-667: break
    else rs->icp_iter = icp_iter + 1;
}

__global__ void k_robust_outer(GnState *st, RobustState *rs, RobustParams prm) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    robust_outer_body(st, rs, prm);
}

// Small frames (round 5): evaluation AND step in one launch by one block — the reference's own keypoint counts (<= max_num_residuals
// blocks of 1-3 k keypoints) keep one compute unit busy for a few microseconds, while the two launches they replace cost a launch
// latency each, thirty times per registration (5 ICP iterations x 6 evaluations). Same arithmetic per residual block; the packed sums
// are taken per wave over its tiles, then over the block's waves in index order (a fixed order of its own, like every path here).
constexpr int FUSE_BLOCK = 512;
struct FuseScratch {
    double rec[FUSE_BLOCK / 64][64 * EVAL_REC];
    double comb[FUSE_BLOCK / 64][SYS_N];
    StepScratch step;
};
// last: this is the inner solver's final launch of the ICP iteration — the end-of-iteration bookkeeping (k_robust_outer: normalise, pose
// change, stop test) follows in the same launch, whether or not the inner solver had already stopped.
__global__ __launch_bounds__(FUSE_BLOCK) void k_robust_eval_step(KpView kp, GnState *st, RobustState *rs, RobustParams prm, RobustBuf rb, int last) {
    extern __shared__ __attribute__((aligned(16))) char fuse_smem[];
    FuseScratch &F = *reinterpret_cast<FuseScratch *>(fuse_smem);
    StepScratch &S = F.step;
    constexpr int NW = (int) (sizeof(RobustState) / 8);
    if (st->done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (rs->ls_done) {
        if (last && tid == 0) robust_outer_body(st, rs, prm);
        return;
    }
    const unsigned long long tc0 = __builtin_readcyclecounter();
    for (int w = tid; w < NW; w += FUSE_BLOCK) reinterpret_cast<double *>(&S.R)[w] = reinterpret_cast<const double *>(rs)[w];
    __syncthreads();
    const bool have_eval = S.R.eval_at_x || S.R.step_valid;        // (an invalid last step: nothing new to evaluate)
    if (have_eval) {
        rb_d4_t accm = {0.0, 0.0, 0.0, 0.0};
        robust_eval_tiles<FUSE_BLOCK>(kp, S.R.eval_at_x ? S.R.x : S.R.cand, prm, rb, 0, 1, tid, F.rec[wave], accm);
        robust_unpack(accm, lane, F.comb[wave]);
    }
    __syncthreads();
    if (have_eval && tid < SYS_N) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < FUSE_BLOCK / 64; ++w) t += F.comb[w][tid];
        S.sys[tid] = t;
    }
    __syncthreads();
    if (wave != 0) return;
    robust_step_wave0(S, have_eval, st, rs, prm, lane, tc0);
    if (last) {
        __threadfence();                                   // the state this wave just wrote back, before lane 0 reads it again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) robust_outer_body(st, rs, prm);
    }
}

__global__ void k_robust_init(const GnState *st, RobustState *rs) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    PoseCtx c;
    for (int i = 0; i < 14; ++i) c.pose[i] = st->pose[i];
    pose_ctx_prepare(c);
    for (int i = 0; i < 14; ++i) rs->prev[i] = c.pose[i];
    rs->x = c;
    rs->cand = c;
    rs->x_cost = 0.0; rs->cand_cost = 0.0; rs->model_cost_change = 0.0; rs->radius = 1e4; rs->decrease_factor = 2.0;
    rs->diff_trans = 0.0; rs->diff_rot = 0.0;
    rs->have_scale = 0; rs->ls_iter = 0; rs->ls_done = 0; rs->ls_term = 0; rs->invalid = 0; rs->step_valid = 0;
    rs->eval_at_x = 1; rs->_pad0 = 0;
    rs->ls_iters_total = 0; rs->ls_accepted_total = 0;
    rs->n_res = 0; rs->icp_iter = 0; rs->error = 0; rs->converged = 0;
    for (int i = 0; i < 144; ++i) rs->H[i] = 0.0;
    for (int i = 0; i < 12; ++i) { rs->g[i] = 0.0; rs->scale[i] = 1.0; }
    for (int i = 0; i < 8; ++i) rs->step_cycles[i] = 0;
}

}  // namespace ctgn
