// oracle/_ref shim for <ceres/ceres.h> (TEST INFRASTRUCTURE ONLY, see ../mini_eigen.h).
//
// Ceres Solver is a third-party dependency of the reference (pulled by its superbuild at an unpinned revision, absent
// from /root/reference and from this image).  This header gives the reference's own DoRegisterCeres
// (src/ct_icp/ct_icp.cpp:457-707), its cost functors (include/ct_icp/cost_functions.h) and its motion-model
// regularisers (src/ct_icp/motion_model.cpp:12-61) something to compile and RUN against:
//   * Jet<T, N> forward-mode dual numbers and AutoDiffCostFunction (what Ceres does with the functors);
//   * the loss functions the reference instantiates, from their published closed forms (loss_function.h);
//   * the corrector of Triggs et al. as published (corrector.cc);
//   * EigenQuaternionParameterization (local_parameterization.cc: q <- [sin|d|/|d| d, cos|d|] (x) q);
//   * Solve(): TRUST_REGION / LEVENBERG_MARQUARDT with Ceres 2.0's documented defaults (trust_region_minimizer.cc,
//     levenberg_marquardt_strategy.cc): Jacobi column scaling from the first Jacobian, step from
//     (J^T J + clamp(diag, 1e-6, 1e32) / radius) y = -J^T r, initial radius 1e4, min_relative_decrease 1e-3,
//     radius update radius / max(1/3, 1 - (2 rho - 1)^3), rejection radius / f with f doubling, function /
//     gradient / parameter tolerances 1e-6 / 1e-10 / 1e-8, 5 consecutive invalid steps = failure.
// It is a restatement of published algorithms, written here; it is NOT Ceres, so the robust route stays "parity
// unpinned against a real Ceres build"  -- what this buys is that everything AROUND the
// minimiser (residual definition, weights, block selection, regularisers, outer loop, stop tests) is the reference's
// literal code.
#ifndef CTGN_ORACLE_CERES_SHIM_H
#define CTGN_ORACLE_CERES_SHIM_H

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace ceres {

    // ---------------------------------------------------------------------------------------------------------------
    template<typename T, int N>
    struct Jet {
        T a;
        T v[N];
        Jet() : a() { for (int i = 0; i < N; ++i) v[i] = T(0); }
        Jet(const T &value) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(0); }   // NOLINT implicit on purpose (as in Ceres)
        template<typename U, typename = typename std::enable_if<std::is_arithmetic<U>::value && !std::is_same<U, T>::value>::type>
        Jet(const U &value) : a(T(value)) { for (int i = 0; i < N; ++i) v[i] = T(0); } // NOLINT
        Jet(const T &value, int k) : a(value) { for (int i = 0; i < N; ++i) v[i] = T(0); v[k] = T(1); }
        Jet &operator+=(const Jet &y) { a += y.a; for (int i = 0; i < N; ++i) v[i] += y.v[i]; return *this; }
        Jet &operator-=(const Jet &y) { a -= y.a; for (int i = 0; i < N; ++i) v[i] -= y.v[i]; return *this; }
        Jet &operator*=(const Jet &y) { *this = *this * y; return *this; }
        Jet &operator/=(const Jet &y) { *this = *this / y; return *this; }
    };
    template<typename T, int N> inline Jet<T, N> operator+(const Jet<T, N> &f) { return f; }
    template<typename T, int N> inline Jet<T, N> operator-(const Jet<T, N> &f) { Jet<T, N> r; r.a = -f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }
    template<typename T, int N> inline Jet<T, N> operator+(const Jet<T, N> &f, const Jet<T, N> &g) { Jet<T, N> r; r.a = f.a + g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i]; return r; }
    template<typename T, int N> inline Jet<T, N> operator-(const Jet<T, N> &f, const Jet<T, N> &g) { Jet<T, N> r; r.a = f.a - g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i]; return r; }
    template<typename T, int N> inline Jet<T, N> operator*(const Jet<T, N> &f, const Jet<T, N> &g) { Jet<T, N> r; r.a = f.a * g.a; for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
    template<typename T, int N> inline Jet<T, N> operator/(const Jet<T, N> &f, const Jet<T, N> &g) {
        const T g_a_inverse = T(1.0) / g.a;
        const T f_a_by_g_a = f.a * g_a_inverse;
        Jet<T, N> r; r.a = f_a_by_g_a;
        for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
        return r;
    }
#define CTGN_JET_SCALAR_OPS(op)                                                                                              \
    template<typename T, int N> inline Jet<T, N> operator op(const Jet<T, N> &f, T s) { return f op Jet<T, N>(s); }          \
    template<typename T, int N> inline Jet<T, N> operator op(T s, const Jet<T, N> &f) { return Jet<T, N>(s) op f; }
    CTGN_JET_SCALAR_OPS(+) CTGN_JET_SCALAR_OPS(-) CTGN_JET_SCALAR_OPS(*) CTGN_JET_SCALAR_OPS(/)
#undef CTGN_JET_SCALAR_OPS
#define CTGN_JET_CMP(op)                                                                                                     \
    template<typename T, int N> inline bool operator op(const Jet<T, N> &f, const Jet<T, N> &g) { return f.a op g.a; }       \
    template<typename T, int N> inline bool operator op(const Jet<T, N> &f, const T &g) { return f.a op g; }                 \
    template<typename T, int N> inline bool operator op(const T &f, const Jet<T, N> &g) { return f op g.a; }
    CTGN_JET_CMP(<) CTGN_JET_CMP(<=) CTGN_JET_CMP(>) CTGN_JET_CMP(>=) CTGN_JET_CMP(==) CTGN_JET_CMP(!=)
#undef CTGN_JET_CMP
    template<typename T, int N> inline Jet<T, N> jet_chain(const Jet<T, N> &f, T value, T deriv) { Jet<T, N> r; r.a = value; for (int i = 0; i < N; ++i) r.v[i] = deriv * f.v[i]; return r; }
    template<typename T, int N> inline Jet<T, N> abs(const Jet<T, N> &f) { return f.a < T(0) ? -f : f; }
    template<typename T, int N> inline Jet<T, N> fabs(const Jet<T, N> &f) { return abs(f); }
    template<typename T, int N> inline Jet<T, N> sqrt(const Jet<T, N> &f) { T s = std::sqrt(f.a); return jet_chain(f, s, T(1.0) / (T(2.0) * s)); }
    template<typename T, int N> inline Jet<T, N> sin(const Jet<T, N> &f) { return jet_chain(f, std::sin(f.a), std::cos(f.a)); }
    template<typename T, int N> inline Jet<T, N> cos(const Jet<T, N> &f) { return jet_chain(f, std::cos(f.a), -std::sin(f.a)); }
    template<typename T, int N> inline Jet<T, N> tan(const Jet<T, N> &f) { T t = std::tan(f.a); return jet_chain(f, t, T(1.0) + t * t); }
    template<typename T, int N> inline Jet<T, N> acos(const Jet<T, N> &f) { return jet_chain(f, std::acos(f.a), -T(1.0) / std::sqrt(T(1.0) - f.a * f.a)); }
    template<typename T, int N> inline Jet<T, N> asin(const Jet<T, N> &f) { return jet_chain(f, std::asin(f.a), T(1.0) / std::sqrt(T(1.0) - f.a * f.a)); }
    template<typename T, int N> inline Jet<T, N> atan(const Jet<T, N> &f) { return jet_chain(f, std::atan(f.a), T(1.0) / (T(1.0) + f.a * f.a)); }
    template<typename T, int N> inline Jet<T, N> exp(const Jet<T, N> &f) { T e = std::exp(f.a); return jet_chain(f, e, e); }
    template<typename T, int N> inline Jet<T, N> log(const Jet<T, N> &f) { return jet_chain(f, std::log(f.a), T(1.0) / f.a); }
    template<typename T, int N> inline Jet<T, N> atan2(const Jet<T, N> &g, const Jet<T, N> &f) {
        T tmp = T(1.0) / (f.a * f.a + g.a * g.a);
        Jet<T, N> r; r.a = std::atan2(g.a, f.a);
        for (int i = 0; i < N; ++i) r.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
        return r;
    }
    template<typename T, int N> inline Jet<T, N> pow(const Jet<T, N> &f, double g) { T p = std::pow(f.a, g - 1.0); return jet_chain(f, p * f.a, g * p); }
    template<typename T, int N> inline Jet<T, N> fmax(const Jet<T, N> &f, const Jet<T, N> &g) { return f < g ? g : f; }
    template<typename T, int N> inline Jet<T, N> fmin(const Jet<T, N> &f, const Jet<T, N> &g) { return f < g ? f : g; }
    template<typename T, int N> inline bool isfinite(const Jet<T, N> &f) {
        if (!std::isfinite(f.a)) return false;
        for (int i = 0; i < N; ++i) if (!std::isfinite(f.v[i])) return false;
        return true;
    }
    template<typename T, int N> inline bool isnan(const Jet<T, N> &f) {
        if (std::isnan(f.a)) return true;
        for (int i = 0; i < N; ++i) if (std::isnan(f.v[i])) return true;
        return false;
    }
    template<typename T, int N> inline std::ostream &operator<<(std::ostream &s, const Jet<T, N> &z) { return s << "[" << z.a << " ; ...]"; }
    using std::abs; using std::sqrt; using std::sin; using std::cos; using std::acos; using std::asin; using std::atan2;
    using std::exp; using std::log; using std::pow; using std::fmax; using std::fmin; using std::isfinite; using std::isnan;

    // ---------------------------------------------------------------------------------------------------------------
    enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
    enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
    enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
    enum MinimizerType { LINE_SEARCH, TRUST_REGION };
    enum LoggingType { SILENT, PER_MINIMIZER_ITERATION };
    enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE, USER_SUCCESS, USER_FAILURE };

    class CostFunction {
    public:
        virtual ~CostFunction() {}
        virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
        const std::vector<int> &parameter_block_sizes() const { return parameter_block_sizes_; }
        int num_residuals() const { return num_residuals_; }
    protected:
        std::vector<int> *mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
        void set_num_residuals(int n) { num_residuals_ = n; }
    private:
        std::vector<int> parameter_block_sizes_;
        int num_residuals_ = 0;
    };

    namespace internal {
        template<int... Ns> struct Sum;
        template<> struct Sum<> { static constexpr int value = 0; };
        template<int N0, int... Ns> struct Sum<N0, Ns...> { static constexpr int value = N0 + Sum<Ns...>::value; };
        template<typename F, typename T, std::size_t... I>
        inline bool call_functor(const F &f, T **params, T *residuals, std::index_sequence<I...>) { return f(params[I]..., residuals); }
    }

    template<typename CostFunctor, int kNumResiduals, int... Ns>
    class AutoDiffCostFunction : public CostFunction {
    public:
        explicit AutoDiffCostFunction(CostFunctor *functor, Ownership ownership = TAKE_OWNERSHIP) : functor_(functor), ownership_(ownership) {
            set_num_residuals(kNumResiduals);
            *mutable_parameter_block_sizes() = std::vector<int>{Ns...};
        }
        ~AutoDiffCostFunction() override { if (ownership_ == TAKE_OWNERSHIP) delete functor_; }
        bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override {
            constexpr int kBlocks = sizeof...(Ns);
            constexpr int kTotal = internal::Sum<Ns...>::value;
            const int sizes[kBlocks] = {Ns...};
            if (!jacobians) {
                double *p[kBlocks];
                for (int b = 0; b < kBlocks; ++b) p[b] = const_cast<double *>(parameters[b]);
                return internal::call_functor(*functor_, p, residuals, std::make_index_sequence<kBlocks>());
            }
            typedef Jet<double, kTotal> JetT;
            JetT x[kTotal];
            JetT *p[kBlocks];
            int off = 0;
            for (int b = 0; b < kBlocks; ++b) {
                p[b] = x + off;
                for (int k = 0; k < sizes[b]; ++k) x[off + k] = JetT(parameters[b][k], off + k);
                off += sizes[b];
            }
            JetT out[kNumResiduals];
            if (!internal::call_functor(*functor_, p, out, std::make_index_sequence<kBlocks>())) return false;
            for (int r = 0; r < kNumResiduals; ++r) residuals[r] = out[r].a;
            off = 0;
            for (int b = 0; b < kBlocks; ++b) {
                if (jacobians[b])
                    for (int r = 0; r < kNumResiduals; ++r)
                        for (int k = 0; k < sizes[b]; ++k) jacobians[b][r * sizes[b] + k] = out[r].v[off + k];   // row-major
                off += sizes[b];
            }
            return true;
        }
    private:
        CostFunctor *functor_;
        Ownership ownership_;
    };

    // ---------------------------------------------------------------------------------------------------------------
    class LossFunction {
    public:
        virtual ~LossFunction() {}
        virtual void Evaluate(double sq_norm, double out[3]) const = 0;
    };
    class TrivialLoss : public LossFunction {
    public:
        void Evaluate(double s, double rho[3]) const override { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
    };
    class HuberLoss : public LossFunction {
    public:
        explicit HuberLoss(double a) : a_(a), b_(a * a) {}
        void Evaluate(double s, double rho[3]) const override {
            if (s > b_) {
                const double r = std::sqrt(s);
                rho[0] = 2.0 * a_ * r - b_;
                rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
                rho[2] = -rho[1] / (2.0 * s);
            } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
        }
    private:
        const double a_, b_;
    };
    class SoftLOneLoss : public LossFunction {
    public:
        explicit SoftLOneLoss(double a) : b_(a * a), c_(1 / b_) {}
        void Evaluate(double s, double rho[3]) const override {
            const double sum = 1.0 + s * c_, tmp = std::sqrt(sum);
            rho[0] = 2.0 * b_ * (tmp - 1.0);
            rho[1] = std::max(std::numeric_limits<double>::min(), 1.0 / tmp);
            rho[2] = -(c_ * rho[1]) / (2.0 * sum);
        }
    private:
        const double b_, c_;
    };
    class CauchyLoss : public LossFunction {
    public:
        explicit CauchyLoss(double a) : b_(a * a), c_(1 / b_) {}
        void Evaluate(double s, double rho[3]) const override {
            const double sum = 1.0 + s * c_, inv = 1.0 / sum;
            rho[0] = b_ * std::log(sum);
            rho[1] = std::max(std::numeric_limits<double>::min(), inv);
            rho[2] = -c_ * (inv * inv);
        }
    private:
        const double b_, c_;
    };
    class ArctanLoss : public LossFunction {
    public:
        explicit ArctanLoss(double a) : a_(a), b_(1 / (a * a)) {}
        void Evaluate(double s, double rho[3]) const override {
            const double sum = 1 + s * s * b_, inv = 1 / sum;
            rho[0] = a_ * std::atan2(s, a_);
            rho[1] = std::max(std::numeric_limits<double>::min(), inv);
            rho[2] = -2.0 * s * b_ * (inv * inv);
        }
    private:
        const double a_, b_;
    };
    class TolerantLoss : public LossFunction {
    public:
        TolerantLoss(double a, double b) : a_(a), b_(b), c_(b * std::log(1.0 + std::exp(-a / b))) {}
        void Evaluate(double s, double rho[3]) const override {
            const double x = (s - a_) / b_;
            const double kLog2Pow53 = 36.7;
            if (x > kLog2Pow53) {
                rho[0] = s - a_ - c_; rho[1] = 1.0; rho[2] = 0.0;
            } else {
                const double e_x = std::exp(x);
                rho[0] = b_ * std::log(1.0 + e_x) - c_;
                rho[1] = std::max(std::numeric_limits<double>::min(), e_x / (1.0 + e_x));
                rho[2] = 0.5 / (b_ * (1.0 + std::cosh(x)));
            }
        }
    private:
        const double a_, b_, c_;
    };
    class TukeyLoss : public LossFunction {
    public:
        explicit TukeyLoss(double a) : a_squared_(a * a) {}
        void Evaluate(double s, double rho[3]) const override {
            if (s <= a_squared_) {
                const double value = 1.0 - s / a_squared_, value_sq = value * value;
                rho[0] = a_squared_ / 3.0 * (1.0 - value_sq * value);
                rho[1] = value_sq;
                rho[2] = -2.0 / a_squared_ * value;
            } else { rho[0] = a_squared_ / 3.0; rho[1] = 0.0; rho[2] = 0.0; }
        }
    private:
        const double a_squared_;
    };

    // ---------------------------------------------------------------------------------------------------------------
    class LocalParameterization {
    public:
        virtual ~LocalParameterization() {}
        virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
        virtual bool ComputeJacobian(const double *x, double *jacobian) const = 0;   // GlobalSize x LocalSize, row-major
        virtual int GlobalSize() const = 0;
        virtual int LocalSize() const = 0;
    };
    // Eigen coefficient order (x, y, z, w)
    class EigenQuaternionParameterization : public LocalParameterization {
    public:
        bool Plus(const double *x, const double *delta, double *out) const override {
            const double norm_delta = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
            if (norm_delta > 0.0) {
                const double s = std::sin(norm_delta) / norm_delta;
                const double dq[4] = {s * delta[0], s * delta[1], s * delta[2], std::cos(norm_delta)};   // x y z w
                // out = dq * x (Hamilton product)
                const double ax = dq[0], ay = dq[1], az = dq[2], aw = dq[3], bx = x[0], by = x[1], bz = x[2], bw = x[3];
                out[3] = aw * bw - ax * bx - ay * by - az * bz;
                out[0] = aw * bx + ax * bw + ay * bz - az * by;
                out[1] = aw * by + ay * bw + az * bx - ax * bz;
                out[2] = aw * bz + az * bw + ax * by - ay * bx;
            } else { for (int i = 0; i < 4; ++i) out[i] = x[i]; }
            return true;
        }
        bool ComputeJacobian(const double *x, double *j) const override {
            j[0] = x[3];  j[1] = x[2];   j[2] = -x[1];
            j[3] = -x[2]; j[4] = x[3];   j[5] = x[0];
            j[6] = x[1];  j[7] = -x[0];  j[8] = x[3];
            j[9] = -x[0]; j[10] = -x[1]; j[11] = -x[2];
            return true;
        }
        int GlobalSize() const override { return 4; }
        int LocalSize() const override { return 3; }
    };

    // ---------------------------------------------------------------------------------------------------------------
    struct ResidualBlock {
        CostFunction *cost = nullptr;
        LossFunction *loss = nullptr;
        std::vector<double *> params;
    };
    typedef ResidualBlock *ResidualBlockId;

    class Problem {
    public:
        struct Options {
            Ownership cost_function_ownership = TAKE_OWNERSHIP;
            Ownership loss_function_ownership = TAKE_OWNERSHIP;
            Ownership local_parameterization_ownership = TAKE_OWNERSHIP;
        };
        struct ParameterBlock { double *ptr; int size; LocalParameterization *local; bool constant; };
        Problem() {}
        explicit Problem(const Options &o) : options_(o) {}
        Problem(const Problem &) = delete;
        ~Problem() {
            // Ceres owns cost functions, losses and parameterisations by default; the same pointer may be shared
            std::vector<CostFunction *> costs; std::vector<LossFunction *> losses; std::vector<LocalParameterization *> locals;
            for (auto &b: blocks_) { if (b->cost) costs.push_back(b->cost); if (b->loss) losses.push_back(b->loss); }
            for (auto &p: params_) if (p.local) locals.push_back(p.local);
            auto uniq_delete = [](auto &v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); for (auto *p: v) delete p; };
            if (options_.cost_function_ownership == TAKE_OWNERSHIP) uniq_delete(costs);
            if (options_.loss_function_ownership == TAKE_OWNERSHIP) uniq_delete(losses);
            if (options_.local_parameterization_ownership == TAKE_OWNERSHIP) uniq_delete(locals);
        }
        void AddParameterBlock(double *values, int size, LocalParameterization *local = nullptr) {
            for (auto &p: params_) if (p.ptr == values) { if (local) p.local = local; return; }
            params_.push_back(ParameterBlock{values, size, local, false});
        }
        template<typename... Ptrs>
        ResidualBlockId AddResidualBlock(CostFunction *cost, LossFunction *loss, double *x0, Ptrs *... xs) {
            return AddResidualBlock(cost, loss, std::vector<double *>{x0, xs...});
        }
        ResidualBlockId AddResidualBlock(CostFunction *cost, LossFunction *loss, const std::vector<double *> &ps) {
            auto b = std::make_unique<ResidualBlock>();
            b->cost = cost; b->loss = loss; b->params = ps;
            const auto &sizes = cost->parameter_block_sizes();
            for (std::size_t i = 0; i < ps.size(); ++i) AddParameterBlock(ps[i], sizes[i]);
            blocks_.push_back(std::move(b));
            return blocks_.back().get();
        }
        void SetParameterBlockConstant(double *values) { for (auto &p: params_) if (p.ptr == values) p.constant = true; }
        void SetParameterBlockVariable(double *values) { for (auto &p: params_) if (p.ptr == values) p.constant = false; }
        void SetParameterization(double *values, LocalParameterization *local) { for (auto &p: params_) if (p.ptr == values) p.local = local; }
        // Ceres 2.0: cost = 1/2 rho(|r|^2) (or 1/2 |r|^2 without the loss), residuals loss-corrected when apply_loss_function
        bool EvaluateResidualBlock(ResidualBlockId id, bool apply_loss_function, double *cost, double *residuals, double **jacobians) const {
            (void) jacobians;
            const int nr = id->cost->num_residuals();
            std::vector<double> res(static_cast<std::size_t>(nr), 0.0);
            std::vector<const double *> pp(id->params.begin(), id->params.end());
            if (!id->cost->Evaluate(pp.data(), res.data(), nullptr)) return false;
            double s = 0.0;
            for (double r: res) s += r * r;
            double rho[3] = {s, 1.0, 0.0};
            if (apply_loss_function && id->loss) id->loss->Evaluate(s, rho);
            if (cost) *cost = 0.5 * rho[0];
            if (residuals) {
                double scaling = 1.0;
                if (apply_loss_function && id->loss) {
                    const double sqrt_rho1 = std::sqrt(rho[1]);
                    if (s == 0.0 || rho[2] <= 0.0) scaling = sqrt_rho1;
                    else scaling = sqrt_rho1 / std::sqrt(1.0 + 2.0 * s * rho[2] / rho[1]);
                }
                for (int r = 0; r < nr; ++r) residuals[r] = scaling * res[static_cast<std::size_t>(r)];
            }
            return true;
        }
        int NumResidualBlocks() const { return int(blocks_.size()); }
        int NumParameterBlocks() const { return int(params_.size()); }
        int NumResiduals() const { int n = 0; for (auto &b: blocks_) n += b->cost->num_residuals(); return n; }
        const std::vector<std::unique_ptr<ResidualBlock>> &residual_blocks() const { return blocks_; }
        const std::vector<ParameterBlock> &parameter_blocks() const { return params_; }
    private:
        Options options_;
        std::vector<std::unique_ptr<ResidualBlock>> blocks_;
        std::vector<ParameterBlock> params_;
    };

    class Solver {
    public:
        struct Options {
            MinimizerType minimizer_type = TRUST_REGION;
            TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
            LinearSolverType linear_solver_type = DENSE_QR;
            LoggingType logging_type = PER_MINIMIZER_ITERATION;
            int max_num_iterations = 50;
            int num_threads = 1;
            double max_solver_time_in_seconds = 1e9;
            double initial_trust_region_radius = 1e4;
            double max_trust_region_radius = 1e16;
            double min_trust_region_radius = 1e-32;
            double min_relative_decrease = 1e-3;
            double min_lm_diagonal = 1e-6;
            double max_lm_diagonal = 1e32;
            int max_num_consecutive_invalid_steps = 5;
            double function_tolerance = 1e-6;
            double gradient_tolerance = 1e-10;
            double parameter_tolerance = 1e-8;
            bool jacobi_scaling = true;
            bool minimizer_progress_to_stdout = false;
            bool use_nonmonotonic_steps = false;
        };
        struct Summary {
            TerminationType termination_type = FAILURE;
            std::string message;
            double initial_cost = -1, final_cost = -1;
            int num_successful_steps = 0, num_unsuccessful_steps = 0;
            int num_iterations = 0;               // LM iterations started
            double final_trust_region_radius = 0;
            double total_time_in_seconds = 0;
            bool IsSolutionUsable() const { return termination_type == CONVERGENCE || termination_type == NO_CONVERGENCE || termination_type == USER_SUCCESS; }
            std::string BriefReport() const {
                std::ostringstream ss;
                ss << "mini-ceres: iterations " << num_iterations << ", initial cost " << initial_cost << ", final cost " << final_cost
                   << ", termination " << int(termination_type);
                return ss.str();
            }
            std::string FullReport() const { return BriefReport() + " " + message; }
        };
    };

    namespace internal {
        struct DenseProblem {
            Problem *problem;
            std::vector<int> global_off, local_off;     // per parameter block (local_off < 0: constant)
            int n_global = 0, n_local = 0;
            std::map<double *, int> index;
            explicit DenseProblem(Problem *p) : problem(p) {
                for (auto &pb: p->parameter_blocks()) {
                    index[pb.ptr] = int(global_off.size());
                    global_off.push_back(n_global); n_global += pb.size;
                    if (pb.constant) local_off.push_back(-1);
                    else { local_off.push_back(n_local); n_local += pb.local ? pb.local->LocalSize() : pb.size; }
                }
            }
            void gather(std::vector<double> &x) const {
                x.resize(std::size_t(n_global));
                const auto &pbs = problem->parameter_blocks();
                for (std::size_t b = 0; b < pbs.size(); ++b) std::memcpy(&x[std::size_t(global_off[b])], pbs[b].ptr, sizeof(double) * std::size_t(pbs[b].size));
            }
            void scatter(const std::vector<double> &x) const {
                const auto &pbs = problem->parameter_blocks();
                for (std::size_t b = 0; b < pbs.size(); ++b) std::memcpy(pbs[b].ptr, &x[std::size_t(global_off[b])], sizeof(double) * std::size_t(pbs[b].size));
            }
            void plus(const std::vector<double> &x, const std::vector<double> &delta, std::vector<double> &out) const {
                out = x;
                const auto &pbs = problem->parameter_blocks();
                for (std::size_t b = 0; b < pbs.size(); ++b) {
                    if (local_off[b] < 0) continue;
                    const double *xb = &x[std::size_t(global_off[b])]; const double *db = &delta[std::size_t(local_off[b])];
                    double *ob = &out[std::size_t(global_off[b])];
                    if (pbs[b].local) pbs[b].local->Plus(xb, db, ob);
                    else for (int k = 0; k < pbs[b].size; ++k) ob[k] = xb[k] + db[k];
                }
            }
            // cost = 1/2 sum rho(|r|^2); if H: H = J~^T J~ (n_local x n_local, row-major), g = J~^T r~ in the tangent space
            bool evaluate(const std::vector<double> &x, double *cost, std::vector<double> *H, std::vector<double> *g) const {
                const auto &pbs = problem->parameter_blocks();
                *cost = 0.0;
                if (H) { H->assign(std::size_t(n_local) * std::size_t(n_local), 0.0); g->assign(std::size_t(n_local), 0.0); }
                std::vector<double> res, jl;
                std::vector<std::vector<double>> jac;
                for (auto &blk: problem->residual_blocks()) {
                    const int nr = blk->cost->num_residuals();
                    const std::size_t nb = blk->params.size();
                    std::vector<const double *> pp(nb);
                    std::vector<int> bi(nb);
                    for (std::size_t i = 0; i < nb; ++i) { bi[i] = index.at(blk->params[i]); pp[i] = &x[std::size_t(global_off[std::size_t(bi[i])])]; }
                    res.assign(std::size_t(nr), 0.0);
                    std::vector<double *> jp(nb, nullptr);
                    if (H) {
                        jac.resize(nb);
                        for (std::size_t i = 0; i < nb; ++i) { jac[i].assign(std::size_t(nr * pbs[std::size_t(bi[i])].size), 0.0); jp[i] = jac[i].data(); }
                    }
                    if (!blk->cost->Evaluate(pp.data(), res.data(), H ? jp.data() : nullptr)) return false;
                    double s = 0.0;
                    for (int r = 0; r < nr; ++r) s += res[std::size_t(r)] * res[std::size_t(r)];
                    double rho[3] = {s, 1.0, 0.0};
                    if (blk->loss) blk->loss->Evaluate(s, rho);
                    *cost += 0.5 * rho[0];
                    if (!H) continue;
                    // local Jacobian, nr x n_local (only this block's columns are non-zero): J_local = J_global * dPlus
                    jl.assign(std::size_t(nr) * std::size_t(n_local), 0.0);
                    for (std::size_t i = 0; i < nb; ++i) {
                        const auto &pb = pbs[std::size_t(bi[i])];
                        const int lo = local_off[std::size_t(bi[i])];
                        if (lo < 0) continue;
                        if (pb.local) {
                            const int gs = pb.local->GlobalSize(), ls = pb.local->LocalSize();
                            std::vector<double> pj(std::size_t(gs * ls));
                            pb.local->ComputeJacobian(pp[i], pj.data());
                            for (int r = 0; r < nr; ++r) for (int c = 0; c < ls; ++c) {
                                double v = 0.0;
                                for (int k = 0; k < gs; ++k) v += jac[i][std::size_t(r * gs + k)] * pj[std::size_t(k * ls + c)];
                                jl[std::size_t(r) * std::size_t(n_local) + std::size_t(lo + c)] += v;
                            }
                        } else {
                            for (int r = 0; r < nr; ++r) for (int c = 0; c < pb.size; ++c)
                                jl[std::size_t(r) * std::size_t(n_local) + std::size_t(lo + c)] += jac[i][std::size_t(r * pb.size + c)];
                        }
                    }
                    // corrector (Triggs): r~ = sqrt(rho') / (1 - alpha) r ; J~ = sqrt(rho') (J - alpha/|r|^2 r r^T J)
                    if (blk->loss) {
                        const double sqrt_rho1 = std::sqrt(rho[1]);
                        double residual_scaling, alpha_sq_norm;
                        if (s == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
                        else {
                            const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
                            const double alpha = 1.0 - std::sqrt(D);
                            residual_scaling = sqrt_rho1 / (1 - alpha);
                            alpha_sq_norm = alpha / s;
                        }
                        if (alpha_sq_norm == 0.0) {
                            for (auto &v: jl) v *= sqrt_rho1;
                        } else {
                            for (int c = 0; c < n_local; ++c) {
                                double r_dot_j = 0.0;
                                for (int r = 0; r < nr; ++r) r_dot_j += jl[std::size_t(r) * std::size_t(n_local) + std::size_t(c)] * res[std::size_t(r)];
                                for (int r = 0; r < nr; ++r) {
                                    double &v = jl[std::size_t(r) * std::size_t(n_local) + std::size_t(c)];
                                    v = sqrt_rho1 * (v - alpha_sq_norm * res[std::size_t(r)] * r_dot_j);
                                }
                            }
                        }
                        for (int r = 0; r < nr; ++r) res[std::size_t(r)] *= residual_scaling;
                    }
                    for (int r = 0; r < nr; ++r) {
                        const double *jr = &jl[std::size_t(r) * std::size_t(n_local)];
                        for (int i = 0; i < n_local; ++i) {
                            if (jr[i] == 0.0) continue;
                            (*g)[std::size_t(i)] += jr[i] * res[std::size_t(r)];
                            for (int j = 0; j < n_local; ++j) (*H)[std::size_t(i) * std::size_t(n_local) + std::size_t(j)] += jr[i] * jr[j];
                        }
                    }
                }
                return true;
            }
        };
        inline bool cholesky_solve(int n, const std::vector<double> &A, const std::vector<double> &b, std::vector<double> &x) {
            const std::size_t un = std::size_t(n);
            std::vector<double> L(un * un, 0.0), y(un, 0.0);
            auto at = [n](std::vector<double> &m, int i, int j) -> double & { return m[std::size_t(i) * std::size_t(n) + std::size_t(j)]; };
            for (int j = 0; j < n; ++j) {
                double d = A[std::size_t(j) * std::size_t(n) + std::size_t(j)];
                for (int k = 0; k < j; ++k) d -= at(L, j, k) * at(L, j, k);
                if (!(d > 0.0) || !std::isfinite(d)) return false;
                at(L, j, j) = std::sqrt(d);
                for (int i = j + 1; i < n; ++i) {
                    double v = A[std::size_t(i) * std::size_t(n) + std::size_t(j)];
                    for (int k = 0; k < j; ++k) v -= at(L, i, k) * at(L, j, k);
                    at(L, i, j) = v / at(L, j, j);
                }
            }
            for (int i = 0; i < n; ++i) { double v = b[std::size_t(i)]; for (int k = 0; k < i; ++k) v -= at(L, i, k) * y[std::size_t(k)]; y[std::size_t(i)] = v / at(L, i, i); }
            x.assign(std::size_t(n), 0.0);
            for (int i = n - 1; i >= 0; --i) { double v = y[std::size_t(i)]; for (int k = i + 1; k < n; ++k) v -= at(L, k, i) * x[std::size_t(k)]; x[std::size_t(i)] = v / at(L, i, i); }
            for (int i = 0; i < n; ++i) if (!std::isfinite(x[std::size_t(i)])) return false;
            return true;
        }
    }

    inline void Solve(const Solver::Options &o, Problem *problem, Solver::Summary *summary) {
        internal::DenseProblem P(problem);
        const int n = P.n_local;
        Solver::Summary &S = *summary;
        S = Solver::Summary();
        const std::size_t un = std::size_t(n);
        std::vector<double> x, cand, H, g, scale(un, 1.0), delta(un, 0.0), neg(un, 0.0), moved;
        P.gather(x);
        double x_cost = 0.0;
        if (!P.evaluate(x, &x_cost, &H, &g) || !std::isfinite(x_cost)) { S.termination_type = FAILURE; S.message = "initial evaluation failed"; return; }
        S.initial_cost = S.final_cost = x_cost;
        if (o.jacobi_scaling) for (int i = 0; i < n; ++i) scale[std::size_t(i)] = 1.0 / (1.0 + std::sqrt(H[std::size_t(i) * std::size_t(n) + std::size_t(i)]));
        double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
        int invalid = 0, iteration = 0;
        TerminationType term = NO_CONVERGENCE;
        auto gradient_max_norm = [&]() {
            for (int i = 0; i < n; ++i) neg[std::size_t(i)] = -g[std::size_t(i)];
            P.plus(x, neg, moved);
            double m = 0.0;
            for (std::size_t i = 0; i < x.size(); ++i) m = std::max(m, std::fabs(x[i] - moved[i]));
            return m;
        };
        for (;;) {
            if (iteration >= o.max_num_iterations) { term = NO_CONVERGENCE; break; }
            if (gradient_max_norm() <= o.gradient_tolerance) { term = CONVERGENCE; break; }
            if (radius < o.min_trust_region_radius) { term = CONVERGENCE; break; }
            ++iteration;
            std::vector<double> Hs(H.size(), 0.0), gs(un, 0.0), A, rhs(un, 0.0), y;
            for (int i = 0; i < n; ++i) {
                gs[std::size_t(i)] = scale[std::size_t(i)] * g[std::size_t(i)];
                for (int j = 0; j < n; ++j) Hs[std::size_t(i) * std::size_t(n) + std::size_t(j)] = scale[std::size_t(i)] * H[std::size_t(i) * std::size_t(n) + std::size_t(j)] * scale[std::size_t(j)];
            }
            A = Hs;
            for (int i = 0; i < n; ++i) {
                double d = Hs[std::size_t(i) * std::size_t(n) + std::size_t(i)];
                d = std::min(std::max(d, o.min_lm_diagonal), o.max_lm_diagonal);
                A[std::size_t(i) * std::size_t(n) + std::size_t(i)] += d / radius;
                rhs[std::size_t(i)] = -gs[std::size_t(i)];
            }
            bool ok = internal::cholesky_solve(n, A, rhs, y);
            double model_cost_change = 0.0;
            if (ok) {
                double yg = 0.0, yHy = 0.0;
                for (int i = 0; i < n; ++i) {
                    yg += y[std::size_t(i)] * gs[std::size_t(i)];
                    double row = 0.0;
                    for (int j = 0; j < n; ++j) row += Hs[std::size_t(i) * std::size_t(n) + std::size_t(j)] * y[std::size_t(j)];
                    yHy += y[std::size_t(i)] * row;
                }
                model_cost_change = -(yg + 0.5 * yHy);
                ok = model_cost_change > 0.0;
            }
            if (!ok) {
                if (++invalid >= o.max_num_consecutive_invalid_steps) { term = FAILURE; S.message = "too many invalid steps"; break; }
                radius *= 0.5;
                continue;
            }
            invalid = 0;
            for (int i = 0; i < n; ++i) delta[std::size_t(i)] = y[std::size_t(i)] * scale[std::size_t(i)];
            P.plus(x, delta, cand);
            double cand_cost = 0.0;
            if (!P.evaluate(cand, &cand_cost, nullptr, nullptr)) cand_cost = std::numeric_limits<double>::infinity();
            double step2 = 0.0, x2 = 0.0;
            for (std::size_t i = 0; i < x.size(); ++i) { step2 += (x[i] - cand[i]) * (x[i] - cand[i]); x2 += x[i] * x[i]; }
            if (std::sqrt(step2) <= o.parameter_tolerance * (std::sqrt(x2) + o.parameter_tolerance)) { term = CONVERGENCE; break; }
            const double cost_change = x_cost - cand_cost;
            if (std::fabs(cost_change) <= o.function_tolerance * x_cost) { term = CONVERGENCE; break; }
            const double relative_decrease = cost_change / model_cost_change;
            if (std::isfinite(cand_cost) && relative_decrease > o.min_relative_decrease) {
                x = cand;
                P.evaluate(x, &x_cost, &H, &g);
                const double f = 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3.0);
                radius = std::min(o.max_trust_region_radius, radius / std::max(1.0 / 3.0, f));
                decrease_factor = 2.0;
                S.num_successful_steps++;
            } else {
                radius = radius / decrease_factor;
                decrease_factor *= 2.0;
                S.num_unsuccessful_steps++;
            }
        }
        P.scatter(x);
        S.final_cost = x_cost; S.num_iterations = iteration; S.termination_type = term; S.final_trust_region_radius = radius;
    }
}

#endif
