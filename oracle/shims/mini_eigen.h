// Minimal stand-in for the subset of Eigen 3 that the reference's GN registration path uses.
//
// TEST INFRASTRUCTURE ONLY (oracle/_ref): this header lets the reference's *own* source files under /root/reference
// (ct_icp.cpp, map.h, neighborhood.h, types.h, pointcloud.h ...) compile in an image that has no Eigen, so that their
// literal control flow, gates, visit orders and call sequences can be executed as a second checker next to
// oracle/ctgn_oracle.c.  Nothing under ct_icp_amd/ or include/ may include it (tests/test_layout.py).
//
// What is restated here (arithmetic of the third-party library, Eigen 3.3/3.4 -- unpinned by the reference, see
// SURVEY.md section 8c) and what is not:
//   * dense fixed/dynamic matrices are evaluated eagerly (no expression templates); element-wise operations are the plain
//     loops; REDUCTIONS (sum, squaredNorm, norm, dot, trace-free redux) of fixed-size objects follow Eigen's association
//     order, because it decides discrete results (`(p - q).norm() > radius`, map.h:491-493): the completely unrolled
//     scalar redux splits a range [s, s+n) into halves of n/2 and n - n/2 (Core/Redux.h, redux_novec_unroller) -- for a
//     3-vector that is c0 + (c1 + c2) -- and even-sized double vectors go through 2-wide packets (stock x86-64 build,
//     SSE2: redux_vec_unroller over packets, then predux = p[0] + p[1]) -- for a 4-vector (c0 + c2) + (c1 + c3);
//     dynamic-size reductions are summed left to right;
//   * Quaternion: product, conjugate, inverse, normalize(d), _transformVector (v + w*uv + qv x uv with uv = 2 qv x v),
//     toRotationMatrix, the matrix -> quaternion branches (trace / largest diagonal), slerp with the
//     |d| >= 1 - eps linear fallback: restated from Eigen/src/Geometry/Quaternion.h as documented;
//   * JacobiSVD: two-sided Jacobi with real_2x2_jacobi_svd / makeJacobi, threshold 2 eps * max diagonal, singular values
//     sorted in decreasing order: restated from Eigen/src/SVD/JacobiSVD.h + Jacobi/Jacobi.h as documented;
//   * LDLT: the diagonally pivoted unblocked in-place factorisation and its solve (pseudo-inverse of D with the
//     min-positive tolerance): restated from Eigen/src/Cholesky/LDLT.h as documented.
// The shim is written from the documented algorithms, not copied from Eigen (which is absent here).
#ifndef CTGN_ORACLE_MINI_EIGEN_H
#define CTGN_ORACLE_MINI_EIGEN_H

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_ALIGNED_ALLOCATOR Eigen::aligned_allocator
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 4
#define EIGEN_MINOR_VERSION 0
#define EIGEN_DEVICE_FUNC
#define EIGEN_STRONG_INLINE inline

namespace Eigen {

    typedef std::ptrdiff_t Index;
    const int Dynamic = -1;
    enum { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2, Unaligned = 0, Aligned = 16 };
    enum { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20 };
    enum TransformTraits { Isometry = 0x1, Affine = 0x2, AffineCompact = 0x10 | Affine, Projective = 0x20 };
    enum { Lower = 1, Upper = 2 };

    template<typename T> using aligned_allocator = std::allocator<T>;

    template<typename S> struct NumTraits {
        typedef S Real;
        static S epsilon() { return std::numeric_limits<S>::epsilon(); }
        static S lowest() { return std::numeric_limits<S>::lowest(); }
        static S highest() { return (std::numeric_limits<S>::max)(); }
        static S dummy_precision() { return S(1e-12); }
    };
    template<> struct NumTraits<float> {
        typedef float Real;
        static float epsilon() { return std::numeric_limits<float>::epsilon(); }
        static float lowest() { return std::numeric_limits<float>::lowest(); }
        static float highest() { return (std::numeric_limits<float>::max)(); }
        static float dummy_precision() { return 1e-5f; }
    };

    template<typename S, int R, int C, int Opt = 0, int MR = R, int MC = C> class Matrix;
    template<typename D> class MatrixBase;
    template<typename X, int R, int C> class Block;
    template<typename P, int MapOpt = 0, typename Stride = void> class Map;
    template<typename S, int Opt = 0> class Quaternion;
    template<typename M> class LDLT;
    template<typename M, int QR = 0> class JacobiSVD;

    namespace internal {
        template<typename D> struct traits;
        template<typename S, int R, int C, int O, int MR, int MC> struct traits<Matrix<S, R, C, O, MR, MC>> {
            typedef S Scalar;
            enum { Rows = R, Cols = C, Writable = 1 };
        };
        template<typename X, int R, int C> struct traits<Block<X, R, C>> {
            typedef typename traits<typename std::remove_const<X>::type>::Scalar Scalar;
            enum { Rows = R, Cols = C, Writable = !std::is_const<X>::value };
        };
        template<typename P, int O, typename St> struct traits<Map<P, O, St>> {
            typedef typename traits<typename std::remove_const<P>::type>::Scalar Scalar;
            enum { Rows = traits<typename std::remove_const<P>::type>::Rows,
                   Cols = traits<typename std::remove_const<P>::type>::Cols, Writable = !std::is_const<P>::value };
        };
        constexpr int pick_dim(int a, int b) { return a != Dynamic ? a : b; }
        // Eigen aligns fixed-size objects whose byte size is a multiple of 16 (EIGEN_MAX_STATIC_ALIGN_BYTES >= 16)
        template<typename S, int N> struct storage_align {
            static constexpr std::size_t value = (N > 0 && (sizeof(S) * std::size_t(N > 0 ? N : 1)) % 16 == 0) ? 16 : alignof(S);
        };

        template<typename S, int R, int C> struct DenseStorage {
            alignas(storage_align<S, R * C>::value) S d[R * C];
            DenseStorage() {}
            DenseStorage(Index, Index) {}
            static constexpr Index rows() { return R; }
            static constexpr Index cols() { return C; }
            void resize(Index r, Index c) { assert(r == R && c == C); (void) r; (void) c; }
            S *data() { return d; }
            const S *data() const { return d; }
        };
        template<typename S> struct DenseStorage<S, Dynamic, Dynamic> {
            std::vector<S> d; Index r_ = 0, c_ = 0;
            DenseStorage() {}
            DenseStorage(Index r, Index c) : d(std::size_t(r * c)), r_(r), c_(c) {}
            Index rows() const { return r_; }
            Index cols() const { return c_; }
            void resize(Index r, Index c) { d.resize(std::size_t(r * c)); r_ = r; c_ = c; }
            S *data() { return d.data(); }
            const S *data() const { return d.data(); }
        };
        template<typename S, int C> struct DenseStorage<S, Dynamic, C> {
            std::vector<S> d; Index r_ = 0;
            DenseStorage() {}
            DenseStorage(Index r, Index) : d(std::size_t(r * C)), r_(r) {}
            Index rows() const { return r_; }
            static constexpr Index cols() { return C; }
            void resize(Index r, Index c) { assert(c == C); (void) c; d.resize(std::size_t(r * C)); r_ = r; }
            S *data() { return d.data(); }
            const S *data() const { return d.data(); }
        };
        template<typename S, int R> struct DenseStorage<S, R, Dynamic> {
            std::vector<S> d; Index c_ = 0;
            DenseStorage() {}
            DenseStorage(Index, Index c) : d(std::size_t(R * c)), c_(c) {}
            static constexpr Index rows() { return R; }
            Index cols() const { return c_; }
            void resize(Index r, Index c) { assert(r == R); (void) r; d.resize(std::size_t(R * c)); c_ = c; }
            S *data() { return d.data(); }
            const S *data() const { return d.data(); }
        };

        using std::abs; using std::sqrt; using std::sin; using std::cos; using std::acos;
        template<typename S> inline S abs2(const S &x) { return x * x; }
    }

    template<typename XprType> class CommaInitializer {
    public:
        typedef typename internal::traits<XprType>::Scalar Scalar;
        CommaInitializer(XprType &x, const Scalar &s) : x_(x), r_(0), c_(1), blk_rows_(1) { x_.coeffRef(0, 0) = s; }
        template<typename OD> CommaInitializer(XprType &x, const MatrixBase<OD> &o) : x_(x), r_(0), c_(0), blk_rows_(o.rows()) {
            put(o);
        }
        CommaInitializer &operator,(const Scalar &s) {
            if (c_ == x_.cols()) { r_ += blk_rows_; c_ = 0; blk_rows_ = 1; }
            x_.coeffRef(r_, c_) = s; ++c_;
            return *this;
        }
        template<typename OD> CommaInitializer &operator,(const MatrixBase<OD> &o) {
            if (c_ == x_.cols()) { r_ += blk_rows_; c_ = 0; blk_rows_ = o.rows(); }
            put(o);
            return *this;
        }
        XprType &finished() { return x_; }
    private:
        template<typename OD> void put(const MatrixBase<OD> &o) {
            for (Index i = 0; i < o.rows(); ++i) for (Index j = 0; j < o.cols(); ++j) x_.coeffRef(r_ + i, c_ + j) = o.coeff(i, j);
            c_ += o.cols();
        }
        XprType &x_; Index r_, c_, blk_rows_;
    };

    // ------------------------------------------------------------------------------------------------------------
    template<typename D> class MatrixBase {
    public:
        typedef typename internal::traits<D>::Scalar Scalar;
        typedef Scalar RealScalar;
        typedef Scalar value_type;
        enum { RowsAtCompileTime = internal::traits<D>::Rows, ColsAtCompileTime = internal::traits<D>::Cols,
               SizeAtCompileTime = (RowsAtCompileTime == Dynamic || ColsAtCompileTime == Dynamic) ? Dynamic
                                                                                                 : RowsAtCompileTime * ColsAtCompileTime,
               IsVectorAtCompileTime = (RowsAtCompileTime == 1 || ColsAtCompileTime == 1) };
        typedef Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime> PlainObject;
        typedef Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> TransposeReturn;
        typedef Eigen::Index Index;

        D &derived() { return *static_cast<D *>(this); }
        const D &derived() const { return *static_cast<const D *>(this); }
        Index rows() const { return derived().rows_(); }
        Index cols() const { return derived().cols_(); }
        Index size() const { return rows() * cols(); }
        const Scalar &coeff(Index i, Index j) const { return derived().coeff_(i, j); }
        Scalar &coeffRef(Index i, Index j) { return derived().coeffRef_(i, j); }
        const Scalar &coeff(Index i) const { return coeff(i % rows(), i / rows()); }
        Scalar &coeffRef(Index i) { return coeffRef(i % rows(), i / rows()); }

        const Scalar &operator()(Index i, Index j) const { return coeff(i, j); }
        Scalar &operator()(Index i, Index j) { return coeffRef(i, j); }
        const Scalar &operator()(Index i) const { return coeff(i); }
        Scalar &operator()(Index i) { return coeffRef(i); }
        const Scalar &operator[](Index i) const { return coeff(i); }
        Scalar &operator[](Index i) { return coeffRef(i); }
        const Scalar &x() const { return coeff(0); }
        const Scalar &y() const { return coeff(1); }
        const Scalar &z() const { return coeff(2); }
        const Scalar &w() const { return coeff(3); }
        Scalar &x() { return coeffRef(0); }
        Scalar &y() { return coeffRef(1); }
        Scalar &z() { return coeffRef(2); }
        Scalar &w() { return coeffRef(3); }

        PlainObject eval() const {
            PlainObject r(rows(), cols());
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) = coeff(i, j);
            return r;
        }
        D &noalias() { return derived(); }

        // ---- assignment helpers
        template<typename OD> D &assign(const MatrixBase<OD> &o) {
            typename MatrixBase<OD>::PlainObject tmp = o.eval();   // alias-safe
            derived().resize_(tmp.rows(), tmp.cols());
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = tmp.coeff(i, j);
            return derived();
        }
        template<typename OD> D &operator+=(const MatrixBase<OD> &o) {
            assert(rows() == o.rows() && cols() == o.cols());
            typename MatrixBase<OD>::PlainObject tmp = o.eval();
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) += tmp.coeff(i, j);
            return derived();
        }
        template<typename OD> D &operator-=(const MatrixBase<OD> &o) {
            assert(rows() == o.rows() && cols() == o.cols());
            typename MatrixBase<OD>::PlainObject tmp = o.eval();
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) -= tmp.coeff(i, j);
            return derived();
        }
        D &operator*=(const Scalar &s) {
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) *= s;
            return derived();
        }
        D &operator/=(const Scalar &s) {
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) /= s;
            return derived();
        }
        template<typename OD> D &operator*=(const MatrixBase<OD> &o) { return assign((*this) * o); }

        D &setZero() { return setConstant(Scalar(0)); }
        D &setOnes() { return setConstant(Scalar(1)); }
        D &setConstant(const Scalar &s) {
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = s;
            return derived();
        }
        D &fill(const Scalar &s) { return setConstant(s); }
        D &setIdentity() {
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = (i == j) ? Scalar(1) : Scalar(0);
            return derived();
        }
        D &setRandom() {
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i)
                coeffRef(i, j) = Scalar(2.0 * (double(std::rand()) / double(RAND_MAX)) - 1.0);
            return derived();
        }

        CommaInitializer<D> operator<<(const Scalar &s) { return CommaInitializer<D>(derived(), s); }
        template<typename OD> CommaInitializer<D> operator<<(const MatrixBase<OD> &o) { return CommaInitializer<D>(derived(), o); }

        // ---- reductions
        // Eigen's redux association order (see the header): completely unrolled halving tree for fixed sizes, through
        // 2-wide packets when the scalar is double and the size is even; left to right for dynamic sizes
        template<typename F> Scalar redux_(F coeff_at) const {
            const Index n = size();
            if (n == 0) return Scalar(0);
            if (SizeAtCompileTime == Dynamic) {
                Scalar s = coeff_at(0);
                for (Index i = 1; i < n; ++i) s = s + coeff_at(i);
                return s;
            }
            if (std::is_same<Scalar, double>::value && n % 2 == 0 && n >= 2) {
                struct P { Scalar a, b; };
                struct T {
                    static P run(F &f, Index start, Index len) {          // start, len in packets
                        if (len == 1) return P{f(2 * start), f(2 * start + 1)};
                        const Index half = len / 2;
                        P l = run(f, start, half), r = run(f, start + half, len - half);
                        return P{l.a + r.a, l.b + r.b};
                    }
                };
                P p = T::run(coeff_at, 0, n / 2);
                return p.a + p.b;
            }
            struct T {
                static Scalar run(F &f, Index start, Index len) {
                    if (len == 1) return f(start);
                    const Index half = len / 2;
                    Scalar l = run(f, start, half);
                    Scalar r = run(f, start + half, len - half);
                    return l + r;
                }
            };
            return T::run(coeff_at, 0, n);
        }
        Scalar squaredNorm() const {
            auto f = [this](Index i) { return internal::abs2(this->coeff(i)); };
            return redux_(f);
        }
        Scalar norm() const { using std::sqrt; return sqrt(squaredNorm()); }
        Scalar sum() const {
            auto f = [this](Index i) { return this->coeff(i); };
            return redux_(f);
        }
        Scalar mean() const { return sum() / Scalar(size()); }
        Scalar prod() const { Scalar s(1); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) s = s * coeff(i, j); return s; }
        Scalar trace() const { Scalar s = coeff(0, 0); for (Index i = 1; i < std::min(rows(), cols()); ++i) s = s + coeff(i, i); return s; }
        Scalar maxCoeff() const { Scalar m = coeff(0, 0); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) if (coeff(i, j) > m) m = coeff(i, j); return m; }
        Scalar minCoeff() const { Scalar m = coeff(0, 0); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) if (coeff(i, j) < m) m = coeff(i, j); return m; }
        template<typename I> Scalar maxCoeff(I *idx) const {
            Scalar m = coeff(0); *idx = 0;
            for (Index i = 1; i < size(); ++i) if (coeff(i) > m) { m = coeff(i); *idx = I(i); }
            return m;
        }
        template<typename I> Scalar minCoeff(I *idx) const {
            Scalar m = coeff(0); *idx = 0;
            for (Index i = 1; i < size(); ++i) if (coeff(i) < m) { m = coeff(i); *idx = I(i); }
            return m;
        }
        bool hasNaN() const { for (Index i = 0; i < size(); ++i) if (!(coeff(i) == coeff(i))) return true; return false; }
        bool allFinite() const { using std::isfinite; for (Index i = 0; i < size(); ++i) if (!isfinite(coeff(i))) return false; return true; }
        template<typename OD> Scalar dot(const MatrixBase<OD> &o) const {
            assert(size() == o.size());
            auto f = [this, &o](Index i) { return this->coeff(i) * o.coeff(i); };
            return redux_(f);
        }
        template<typename OD> Matrix<Scalar, 3, 1> cross(const MatrixBase<OD> &o) const {
            Matrix<Scalar, 3, 1> r;
            r.coeffRef(0, 0) = coeff(1) * o.coeff(2) - coeff(2) * o.coeff(1);
            r.coeffRef(1, 0) = coeff(2) * o.coeff(0) - coeff(0) * o.coeff(2);
            r.coeffRef(2, 0) = coeff(0) * o.coeff(1) - coeff(1) * o.coeff(0);
            return r;
        }
        PlainObject normalized() const {
            // Eigen 3.3+: z = squaredNorm(); z > 0 ? this / sqrt(z) : this
            using std::sqrt;
            Scalar z = squaredNorm();
            PlainObject r = eval();
            if (z > Scalar(0)) r /= sqrt(z);
            return r;
        }
        void normalize() {
            using std::sqrt;
            Scalar z = squaredNorm();
            if (z > Scalar(0)) derived() /= sqrt(z);
        }
        TransposeReturn transpose() const {
            TransposeReturn r(cols(), rows());
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r.coeffRef(j, i) = coeff(i, j);
            return r;
        }
        TransposeReturn adjoint() const { return transpose(); }
        void transposeInPlace() { assign(transpose()); }
        PlainObject cwiseAbs() const { using std::abs; PlainObject r = eval(); for (Index i = 0; i < r.size(); ++i) r.coeffRef(i) = abs(r.coeff(i)); return r; }
        PlainObject cwiseAbs2() const { PlainObject r = eval(); for (Index i = 0; i < r.size(); ++i) r.coeffRef(i) = r.coeff(i) * r.coeff(i); return r; }
        PlainObject cwiseSqrt() const { using std::sqrt; PlainObject r = eval(); for (Index i = 0; i < r.size(); ++i) r.coeffRef(i) = sqrt(r.coeff(i)); return r; }
        PlainObject cwiseInverse() const { PlainObject r = eval(); for (Index i = 0; i < r.size(); ++i) r.coeffRef(i) = Scalar(1) / r.coeff(i); return r; }
        template<typename OD> PlainObject cwiseProduct(const MatrixBase<OD> &o) const { PlainObject r = eval(); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) *= o.coeff(i, j); return r; }
        template<typename OD> PlainObject cwiseQuotient(const MatrixBase<OD> &o) const { PlainObject r = eval(); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) /= o.coeff(i, j); return r; }
        template<typename OD> PlainObject cwiseMin(const MatrixBase<OD> &o) const { PlainObject r = eval(); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) = std::min(coeff(i, j), o.coeff(i, j)); return r; }
        template<typename OD> PlainObject cwiseMax(const MatrixBase<OD> &o) const { PlainObject r = eval(); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) = std::max(coeff(i, j), o.coeff(i, j)); return r; }
        const D &array() const { return derived(); }
        const D &matrix() const { return derived(); }
        template<typename OD> bool isApprox(const MatrixBase<OD> &o, Scalar prec = NumTraits<Scalar>::dummy_precision()) const {
            Scalar d = ((*this) - o).squaredNorm();
            return d <= prec * prec * std::min(squaredNorm(), o.squaredNorm());
        }
        template<typename OD> bool operator==(const MatrixBase<OD> &o) const {
            if (rows() != o.rows() || cols() != o.cols()) return false;
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) if (!(coeff(i, j) == o.coeff(i, j))) return false;
            return true;
        }
        template<typename OD> bool operator!=(const MatrixBase<OD> &o) const { return !(*this == o); }

        template<typename T> Matrix<T, RowsAtCompileTime, ColsAtCompileTime> cast() const {
            Matrix<T, RowsAtCompileTime, ColsAtCompileTime> r(rows(), cols());
            for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r.coeffRef(i, j) = static_cast<T>(coeff(i, j));
            return r;
        }

        // ---- blocks
        template<int BR, int BC> Block<D, BR, BC> block(Index i, Index j) { return Block<D, BR, BC>(derived(), i, j, BR, BC); }
        template<int BR, int BC> Block<const D, BR, BC> block(Index i, Index j) const { return Block<const D, BR, BC>(derived(), i, j, BR, BC); }
        Block<D, Dynamic, Dynamic> block(Index i, Index j, Index r, Index c) { return Block<D, Dynamic, Dynamic>(derived(), i, j, r, c); }
        Block<const D, Dynamic, Dynamic> block(Index i, Index j, Index r, Index c) const { return Block<const D, Dynamic, Dynamic>(derived(), i, j, r, c); }
        template<int BR, int BC> Block<D, BR, BC> topLeftCorner() { return block<BR, BC>(0, 0); }
        template<int BR, int BC> Block<const D, BR, BC> topLeftCorner() const { return block<BR, BC>(0, 0); }
        template<int BR, int BC> Block<D, BR, BC> topRightCorner() { return block<BR, BC>(0, cols() - BC); }
        template<int BR, int BC> Block<const D, BR, BC> topRightCorner() const { return block<BR, BC>(0, cols() - BC); }
        Block<D, RowsAtCompileTime, 1> col(Index j) { return Block<D, RowsAtCompileTime, 1>(derived(), 0, j, rows(), 1); }
        Block<const D, RowsAtCompileTime, 1> col(Index j) const { return Block<const D, RowsAtCompileTime, 1>(derived(), 0, j, rows(), 1); }
        Block<D, 1, ColsAtCompileTime> row(Index i) { return Block<D, 1, ColsAtCompileTime>(derived(), i, 0, 1, cols()); }
        Block<const D, 1, ColsAtCompileTime> row(Index i) const { return Block<const D, 1, ColsAtCompileTime>(derived(), i, 0, 1, cols()); }
        // vector segments (column or row vectors)
        template<int N> Block<D, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> segment(Index s) {
            return vec_block<N>(s, N);
        }
        template<int N> Block<const D, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> segment(Index s) const {
            return vec_block<N>(s, N);
        }
        template<int N> auto head() { return segment<N>(0); }
        template<int N> auto head() const { return segment<N>(0); }
        template<int N> auto tail() { return segment<N>(size() - N); }
        template<int N> auto tail() const { return segment<N>(size() - N); }
        auto segment(Index s, Index n) { return vec_block<Dynamic>(s, n); }
        auto segment(Index s, Index n) const { return vec_block<Dynamic>(s, n); }
        auto head(Index n) { return segment(0, n); }
        auto head(Index n) const { return segment(0, n); }
        auto tail(Index n) { return segment(size() - n, n); }
        auto tail(Index n) const { return segment(size() - n, n); }
        Matrix<Scalar, Dynamic, 1> diagonal() const {
            Index n = std::min(rows(), cols());
            Matrix<Scalar, Dynamic, 1> r(n, 1);
            for (Index i = 0; i < n; ++i) r.coeffRef(i, 0) = coeff(i, i);
            return r;
        }
        Matrix<Scalar, Dynamic, Dynamic> asDiagonal() const {
            Index n = size();
            Matrix<Scalar, Dynamic, Dynamic> r(n, n);
            r.setZero();
            for (Index i = 0; i < n; ++i) r.coeffRef(i, i) = coeff(i);
            return r;
        }

        // ---- small dense algebra
        Scalar determinant() const {
            assert(rows() == cols());
            const Index n = rows();
            if (n == 1) return coeff(0, 0);
            if (n == 2) return coeff(0, 0) * coeff(1, 1) - coeff(1, 0) * coeff(0, 1);
            if (n == 3) {
                // Eigen's 3x3 helper: det = a00 * (a11 a22 - a21 a12) - a10 * (a01 a22 - a21 a02) + a20 * (a01 a12 - a11 a02)
                auto d3 = [&](int a, int b, int c) {
                    return coeff(0, a) * (coeff(1, b) * coeff(2, c) - coeff(1, c) * coeff(2, b));
                };
                return d3(0, 1, 2) - d3(1, 0, 2) + d3(2, 0, 1);
            }
            // generic: partial-pivot LU
            PlainObject m = eval();
            Scalar det(1);
            for (Index k = 0; k < n; ++k) {
                using std::abs;
                Index p = k;
                for (Index i = k + 1; i < n; ++i) if (abs(m.coeff(i, k)) > abs(m.coeff(p, k))) p = i;
                if (m.coeff(p, k) == Scalar(0)) return Scalar(0);
                if (p != k) { for (Index j = 0; j < n; ++j) std::swap(m.coeffRef(k, j), m.coeffRef(p, j)); det = -det; }
                det = det * m.coeff(k, k);
                for (Index i = k + 1; i < n; ++i) {
                    Scalar f = m.coeff(i, k) / m.coeff(k, k);
                    for (Index j = k; j < n; ++j) m.coeffRef(i, j) -= f * m.coeff(k, j);
                }
            }
            return det;
        }
        PlainObject inverse() const {
            assert(rows() == cols());
            const Index n = rows();
            PlainObject a = eval();
            PlainObject inv(n, n);
            inv.setIdentity();
            using std::abs;
            for (Index k = 0; k < n; ++k) {
                Index p = k;
                for (Index i = k + 1; i < n; ++i) if (abs(a.coeff(i, k)) > abs(a.coeff(p, k))) p = i;
                if (p != k) for (Index j = 0; j < n; ++j) { std::swap(a.coeffRef(k, j), a.coeffRef(p, j)); std::swap(inv.coeffRef(k, j), inv.coeffRef(p, j)); }
                Scalar d = a.coeff(k, k);
                for (Index j = 0; j < n; ++j) { a.coeffRef(k, j) /= d; inv.coeffRef(k, j) /= d; }
                for (Index i = 0; i < n; ++i) if (i != k) {
                    Scalar f = a.coeff(i, k);
                    if (f == Scalar(0)) continue;
                    for (Index j = 0; j < n; ++j) { a.coeffRef(i, j) -= f * a.coeff(k, j); inv.coeffRef(i, j) -= f * inv.coeffRef(k, j); }
                }
            }
            return inv;
        }
        LDLT<PlainObject> ldlt() const { return LDLT<PlainObject>(eval()); }
        JacobiSVD<PlainObject> jacobiSvd(unsigned int opts = 0) const { return JacobiSVD<PlainObject>(eval(), opts); }

    private:
        template<int N> Block<D, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> vec_block(Index s, Index n) {
            typedef Block<D, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> B;
            return ColsAtCompileTime == 1 ? B(derived(), s, 0, n, 1) : B(derived(), 0, s, 1, n);
        }
        template<int N> Block<const D, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> vec_block(Index s, Index n) const {
            typedef Block<const D, (ColsAtCompileTime == 1 ? N : 1), (ColsAtCompileTime == 1 ? 1 : N)> B;
            return ColsAtCompileTime == 1 ? B(derived(), s, 0, n, 1) : B(derived(), 0, s, 1, n);
        }
    };

    // ------------------------------------------------------------------------------------------------------------
    template<typename S, int R, int C, int Opt, int MR, int MC>
    class Matrix : public MatrixBase<Matrix<S, R, C, Opt, MR, MC>> {
        typedef MatrixBase<Matrix<S, R, C, Opt, MR, MC>> Base;
    public:
        typedef S Scalar;
        using Base::operator+=; using Base::operator-=; using Base::operator*=; using Base::operator/=;
        Matrix() : st_() { }
        explicit Matrix(Index n) : st_(R == Dynamic ? n : R, C == Dynamic ? (R == Dynamic ? 1 : n) : C) {}   // Vector(n) for dynamic vectors
        Matrix(Index r, Index c) : st_(r, c) {}
        template<typename T0, typename T1, typename = typename std::enable_if<(R * C == 2) && std::is_convertible<T0, S>::value && !std::is_integral<T0>::value>::type>
        Matrix(const T0 &x, const T1 &y) : st_() { st_.data()[0] = S(x); st_.data()[1] = S(y); }
        Matrix(const S &x, const S &y, const S &z) : st_() { static_assert(R * C == 3, "3-vector ctor"); st_.data()[0] = x; st_.data()[1] = y; st_.data()[2] = z; }
        Matrix(const S &x, const S &y, const S &z, const S &w) : st_() { static_assert(R * C == 4, "4-vector ctor"); st_.data()[0] = x; st_.data()[1] = y; st_.data()[2] = z; st_.data()[3] = w; }
        explicit Matrix(const S *data) : st_() { for (Index i = 0; i < R * C; ++i) st_.data()[i] = data[i]; }
        Matrix(const Matrix &) = default;
        Matrix(Matrix &&) = default;
        Matrix &operator=(const Matrix &) = default;
        Matrix &operator=(Matrix &&) = default;
        template<typename OD> Matrix(const MatrixBase<OD> &o) : st_(o.rows(), o.cols()) {
            static_assert(std::is_same<typename MatrixBase<OD>::Scalar, S>::value, "mixing scalar types needs cast<>()");
            assert((R == Dynamic || R == o.rows()) && (C == Dynamic || C == o.cols()));
            for (Index j = 0; j < o.cols(); ++j) for (Index i = 0; i < o.rows(); ++i) st_.data()[i + j * o.rows()] = o.coeff(i, j);
        }
        template<typename OD> Matrix &operator=(const MatrixBase<OD> &o) {
            static_assert(std::is_same<typename MatrixBase<OD>::Scalar, S>::value, "mixing scalar types needs cast<>()");
            return Base::assign(o);
        }

        Index rows_() const { return st_.rows(); }
        Index cols_() const { return st_.cols(); }
        const S &coeff_(Index i, Index j) const { return st_.data()[i + j * st_.rows()]; }
        S &coeffRef_(Index i, Index j) { return st_.data()[i + j * st_.rows()]; }
        void resize_(Index r, Index c) { if (r != st_.rows() || c != st_.cols()) st_.resize(r, c); }
        void resize(Index r, Index c) { st_.resize(r, c); }
        void resize(Index n) { if (C == 1 || (R == Dynamic && C == Dynamic)) st_.resize(n, 1); else st_.resize(1, n); }
        void conservativeResize(Index r, Index c) {
            Matrix old = *this; st_.resize(r, c);
            for (Index j = 0; j < std::min(c, old.cols()); ++j) for (Index i = 0; i < std::min(r, old.rows()); ++i) coeffRef_(i, j) = old.coeff_(i, j);
        }
        S *data() { return st_.data(); }
        const S *data() const { return st_.data(); }
        // a 1 x 1 product converts to its scalar (Eigen allows `T s = a.transpose() * b;`)
        template<int R_ = R, int C_ = C, typename = typename std::enable_if<R_ == 1 && C_ == 1>::type>
        operator S() const { return st_.data()[0]; }

        static Matrix Zero() { Matrix m; m.setZero(); return m; }
        static Matrix Zero(Index r, Index c) { Matrix m(r, c); m.setZero(); return m; }
        static Matrix Zero(Index n) { Matrix m(n); m.setZero(); return m; }
        static Matrix Ones() { Matrix m; m.setOnes(); return m; }
        static Matrix Ones(Index r, Index c) { Matrix m(r, c); m.setOnes(); return m; }
        static Matrix Constant(const S &s) { Matrix m; m.setConstant(s); return m; }
        static Matrix Constant(Index r, Index c, const S &s) { Matrix m(r, c); m.setConstant(s); return m; }
        static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
        static Matrix Identity(Index r, Index c) { Matrix m(r, c); m.setIdentity(); return m; }
        static Matrix Random() { Matrix m; m.setRandom(); return m; }
        static Matrix Random(Index r, Index c) { Matrix m(r, c); m.setRandom(); return m; }
        static Matrix UnitX() { Matrix m; m.setZero(); m.coeffRef(0) = S(1); return m; }
        static Matrix UnitY() { Matrix m; m.setZero(); m.coeffRef(1) = S(1); return m; }
        static Matrix UnitZ() { Matrix m; m.setZero(); m.coeffRef(2) = S(1); return m; }
    private:
        internal::DenseStorage<S, R, C> st_;
    };

    // ------------------------------------------------------------------------------------------------------------
    template<typename X, int R, int C>
    class Block : public MatrixBase<Block<X, R, C>> {
        typedef MatrixBase<Block<X, R, C>> Base;
    public:
        typedef typename Base::Scalar Scalar;
        using Base::operator+=; using Base::operator-=; using Base::operator*=; using Base::operator/=;
        Block(X &x, Index i0, Index j0, Index r, Index c) : x_(x), i0_(i0), j0_(j0), r_(r), c_(c) {
            assert(i0 >= 0 && j0 >= 0 && i0 + r <= x.rows() && j0 + c <= x.cols());
        }
        Block(const Block &) = default;
        Index rows_() const { return r_; }
        Index cols_() const { return c_; }
        const Scalar &coeff_(Index i, Index j) const { return x_.coeff(i0_ + i, j0_ + j); }
        Scalar &coeffRef_(Index i, Index j) { return x_.coeffRef(i0_ + i, j0_ + j); }
        void resize_(Index r, Index c) { assert(r == r_ && c == c_); (void) r; (void) c; }
        template<typename OD> Block &operator=(const MatrixBase<OD> &o) { return Base::assign(o); }
        Block &operator=(const Block &o) { return Base::assign(o); }
    private:
        X &x_; Index i0_, j0_, r_, c_;
    };

    // ------------------------------------------------------------------------------------------------------------
    template<typename P, int MapOpt, typename Stride>
    class Map : public MatrixBase<Map<P, MapOpt, Stride>> {
        typedef MatrixBase<Map<P, MapOpt, Stride>> Base;
        typedef typename std::remove_const<P>::type Plain;
    public:
        typedef typename Base::Scalar Scalar;
        typedef typename std::conditional<std::is_const<P>::value, const Scalar *, Scalar *>::type Ptr;
        using Base::operator+=; using Base::operator-=; using Base::operator*=; using Base::operator/=;
        enum { R = internal::traits<Plain>::Rows, C = internal::traits<Plain>::Cols };
        explicit Map(Ptr p) : p_(p), r_(R), c_(C) {}
        Map(Ptr p, Index n) : p_(p), r_(C == 1 ? n : (R == Dynamic ? n : R)), c_(C == 1 ? 1 : (R == 1 ? n : C)) {}
        Map(Ptr p, Index r, Index c) : p_(p), r_(r), c_(c) {}
        Map(const Map &) = default;
        Index rows_() const { return r_; }
        Index cols_() const { return c_; }
        const Scalar &coeff_(Index i, Index j) const { return p_[i + j * r_]; }
        Scalar &coeffRef_(Index i, Index j) { return const_cast<Scalar *>(p_)[i + j * r_]; }
        void resize_(Index r, Index c) { assert(r == r_ && c == c_); (void) r; (void) c; }
        template<typename OD> Map &operator=(const MatrixBase<OD> &o) { return Base::assign(o); }
        Map &operator=(const Map &o) { return Base::assign(o); }
        Ptr data() const { return p_; }
    private:
        Ptr p_; Index r_, c_;
    };

    // ------------------------------------------------------------------------------------------------------------
    // free operators (all eager)
    template<typename A, typename B>
    Matrix<typename MatrixBase<A>::Scalar, internal::pick_dim(MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::RowsAtCompileTime),
           internal::pick_dim(MatrixBase<A>::ColsAtCompileTime, MatrixBase<B>::ColsAtCompileTime)>
    operator+(const MatrixBase<A> &a, const MatrixBase<B> &b) {
        assert(a.rows() == b.rows() && a.cols() == b.cols());
        Matrix<typename MatrixBase<A>::Scalar, internal::pick_dim(MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::RowsAtCompileTime),
               internal::pick_dim(MatrixBase<A>::ColsAtCompileTime, MatrixBase<B>::ColsAtCompileTime)> r(a.rows(), a.cols());
        for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) + b.coeff(i, j);
        return r;
    }
    template<typename A, typename B>
    Matrix<typename MatrixBase<A>::Scalar, internal::pick_dim(MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::RowsAtCompileTime),
           internal::pick_dim(MatrixBase<A>::ColsAtCompileTime, MatrixBase<B>::ColsAtCompileTime)>
    operator-(const MatrixBase<A> &a, const MatrixBase<B> &b) {
        assert(a.rows() == b.rows() && a.cols() == b.cols());
        Matrix<typename MatrixBase<A>::Scalar, internal::pick_dim(MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::RowsAtCompileTime),
               internal::pick_dim(MatrixBase<A>::ColsAtCompileTime, MatrixBase<B>::ColsAtCompileTime)> r(a.rows(), a.cols());
        for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) - b.coeff(i, j);
        return r;
    }
    template<typename A> typename MatrixBase<A>::PlainObject operator-(const MatrixBase<A> &a) {
        typename MatrixBase<A>::PlainObject r(a.rows(), a.cols());
        for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = -a.coeff(i, j);
        return r;
    }
    template<typename A> typename MatrixBase<A>::PlainObject operator*(const MatrixBase<A> &a, const typename MatrixBase<A>::Scalar &s) {
        typename MatrixBase<A>::PlainObject r(a.rows(), a.cols());
        for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) * s;
        return r;
    }
    template<typename A> typename MatrixBase<A>::PlainObject operator*(const typename MatrixBase<A>::Scalar &s, const MatrixBase<A> &a) {
        typename MatrixBase<A>::PlainObject r(a.rows(), a.cols());
        for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = s * a.coeff(i, j);
        return r;
    }
    template<typename A> typename MatrixBase<A>::PlainObject operator/(const MatrixBase<A> &a, const typename MatrixBase<A>::Scalar &s) {
        typename MatrixBase<A>::PlainObject r(a.rows(), a.cols());
        for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r.coeffRef(i, j) = a.coeff(i, j) / s;
        return r;
    }
    template<typename A, typename B>
    Matrix<typename MatrixBase<A>::Scalar, MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::ColsAtCompileTime>
    operator*(const MatrixBase<A> &a, const MatrixBase<B> &b) {
        assert(a.cols() == b.rows());
        typedef typename MatrixBase<A>::Scalar S;
        Matrix<S, MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::ColsAtCompileTime> r(a.rows(), b.cols());
        for (Index j = 0; j < b.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) {
            S s = a.coeff(i, 0) * b.coeff(0, j);
            for (Index k = 1; k < a.cols(); ++k) s = s + a.coeff(i, k) * b.coeff(k, j);
            r.coeffRef(i, j) = s;
        }
        return r;
    }
    template<typename A> std::ostream &operator<<(std::ostream &os, const MatrixBase<A> &a) {
        for (Index i = 0; i < a.rows(); ++i) {
            for (Index j = 0; j < a.cols(); ++j) { if (j) os << " "; os << a.coeff(i, j); }
            if (i + 1 < a.rows()) os << "\n";
        }
        return os;
    }

    // ------------------------------------------------------------------------------------------------------------
    // Quaternion (coefficients stored x, y, z, w like Eigen)
    template<typename D> class QuaternionBase;
    namespace internal {
        template<typename S, int O> struct traits<Quaternion<S, O>> { typedef S Scalar; typedef Matrix<S, 4, 1> Coeffs; };
        template<typename S, int O, int MO, typename St> struct traits<Map<Quaternion<S, O>, MO, St>> { typedef S Scalar; typedef Map<Matrix<S, 4, 1>> Coeffs; };
        template<typename S, int O, int MO, typename St> struct traits<Map<const Quaternion<S, O>, MO, St>> { typedef S Scalar; typedef Map<const Matrix<S, 4, 1>> Coeffs; };
    }

    template<typename D> class QuaternionBase {
    public:
        typedef typename internal::traits<D>::Scalar Scalar;
        typedef Matrix<Scalar, 3, 1> Vector3;
        typedef Matrix<Scalar, 3, 3> Matrix3;
        D &derived() { return *static_cast<D *>(this); }
        const D &derived() const { return *static_cast<const D *>(this); }
        const Scalar &x() const { return derived().coeffs().coeff(0); }
        const Scalar &y() const { return derived().coeffs().coeff(1); }
        const Scalar &z() const { return derived().coeffs().coeff(2); }
        const Scalar &w() const { return derived().coeffs().coeff(3); }
        Scalar &x() { return derived().coeffs().coeffRef(0); }
        Scalar &y() { return derived().coeffs().coeffRef(1); }
        Scalar &z() { return derived().coeffs().coeffRef(2); }
        Scalar &w() { return derived().coeffs().coeffRef(3); }
        Vector3 vec() const { return Vector3(x(), y(), z()); }
        Scalar squaredNorm() const { return derived().coeffs().squaredNorm(); }
        Scalar norm() const { return derived().coeffs().norm(); }
        void normalize() { derived().coeffs().normalize(); }
        Quaternion<Scalar> normalized() const { return Quaternion<Scalar>(Matrix<Scalar, 4, 1>(derived().coeffs().normalized())); }
        template<typename OD> Scalar dot(const QuaternionBase<OD> &o) const { return derived().coeffs().dot(o.derived().coeffs()); }
        Quaternion<Scalar> conjugate() const { return Quaternion<Scalar>(w(), -x(), -y(), -z()); }
        Quaternion<Scalar> inverse() const {
            Scalar n2 = squaredNorm();
            if (n2 > Scalar(0)) { Quaternion<Scalar> c = conjugate(); return Quaternion<Scalar>(Matrix<Scalar, 4, 1>(c.coeffs() / n2)); }
            return Quaternion<Scalar>(Matrix<Scalar, 4, 1>::Zero());
        }
        D &setIdentity() { x() = Scalar(0); y() = Scalar(0); z() = Scalar(0); w() = Scalar(1); return derived(); }
        template<typename OD> Quaternion<Scalar> operator*(const QuaternionBase<OD> &b) const {
            const QuaternionBase &a = *this;
            return Quaternion<Scalar>(
                    a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                    a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                    a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                    a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
        }
        template<typename OD> D &operator*=(const QuaternionBase<OD> &b) { Quaternion<Scalar> r = (*this) * b; derived().coeffs() = r.coeffs(); return derived(); }
        // rotation of a vector: v + w * uv + qv x uv, uv = 2 (qv x v)
        template<typename VD> Vector3 _transformVector(const MatrixBase<VD> &v) const {
            Vector3 uv = this->vec().cross(v);
            uv += uv;
            Vector3 vv = v.eval();
            return vv + this->w() * uv + this->vec().cross(uv);
        }
        template<typename VD> Vector3 operator*(const MatrixBase<VD> &v) const { return _transformVector(v); }
        Matrix3 toRotationMatrix() const {
            Matrix3 res;
            const Scalar tx = Scalar(2) * this->x();
            const Scalar ty = Scalar(2) * this->y();
            const Scalar tz = Scalar(2) * this->z();
            const Scalar twx = tx * this->w();
            const Scalar twy = ty * this->w();
            const Scalar twz = tz * this->w();
            const Scalar txx = tx * this->x();
            const Scalar txy = ty * this->x();
            const Scalar txz = tz * this->x();
            const Scalar tyy = ty * this->y();
            const Scalar tyz = tz * this->y();
            const Scalar tzz = tz * this->z();
            res.coeffRef(0, 0) = Scalar(1) - (tyy + tzz);
            res.coeffRef(0, 1) = txy - twz;
            res.coeffRef(0, 2) = txz + twy;
            res.coeffRef(1, 0) = txy + twz;
            res.coeffRef(1, 1) = Scalar(1) - (txx + tzz);
            res.coeffRef(1, 2) = tyz - twx;
            res.coeffRef(2, 0) = txz - twy;
            res.coeffRef(2, 1) = tyz + twx;
            res.coeffRef(2, 2) = Scalar(1) - (txx + tyy);
            return res;
        }
        Matrix3 matrix() const { return toRotationMatrix(); }
        template<typename OD> Quaternion<Scalar> slerp(const Scalar &t, const QuaternionBase<OD> &other) const {
            using std::acos; using std::sin; using std::abs;
            const Scalar one = Scalar(1) - NumTraits<Scalar>::epsilon();
            Scalar d = this->dot(other);
            Scalar absD = abs(d);
            Scalar scale0, scale1;
            if (absD >= one) {
                scale0 = Scalar(1) - t;
                scale1 = t;
            } else {
                Scalar theta = acos(absD);
                Scalar sinTheta = sin(theta);
                scale0 = sin((Scalar(1) - t) * theta) / sinTheta;
                scale1 = sin((t * theta)) / sinTheta;
            }
            if (d < Scalar(0)) scale1 = -scale1;
            return Quaternion<Scalar>(Matrix<Scalar, 4, 1>(scale0 * derived().coeffs() + scale1 * other.derived().coeffs()));
        }
        template<typename OD> Scalar angularDistance(const QuaternionBase<OD> &other) const {
            using std::atan2; using std::abs;
            Quaternion<Scalar> d = (*this) * other.conjugate();
            return Scalar(2) * atan2(d.vec().norm(), abs(d.w()));
        }
        template<typename OD> bool isApprox(const QuaternionBase<OD> &o, Scalar prec = NumTraits<Scalar>::dummy_precision()) const {
            return derived().coeffs().isApprox(o.derived().coeffs(), prec);
        }
        template<typename T> Quaternion<T> cast() const { return Quaternion<T>(static_cast<T>(w()), static_cast<T>(x()), static_cast<T>(y()), static_cast<T>(z())); }
    protected:
        template<typename MD> void set_from_matrix(const MatrixBase<MD> &mat) {
            using std::sqrt;
            Scalar t = mat.trace();
            if (t > Scalar(0)) {
                t = sqrt(t + Scalar(1.0));
                w() = Scalar(0.5) * t;
                t = Scalar(0.5) / t;
                x() = (mat.coeff(2, 1) - mat.coeff(1, 2)) * t;
                y() = (mat.coeff(0, 2) - mat.coeff(2, 0)) * t;
                z() = (mat.coeff(1, 0) - mat.coeff(0, 1)) * t;
            } else {
                Index i = 0;
                if (mat.coeff(1, 1) > mat.coeff(0, 0)) i = 1;
                if (mat.coeff(2, 2) > mat.coeff(i, i)) i = 2;
                Index j = (i + 1) % 3;
                Index k = (j + 1) % 3;
                t = sqrt(mat.coeff(i, i) - mat.coeff(j, j) - mat.coeff(k, k) + Scalar(1.0));
                derived().coeffs().coeffRef(i) = Scalar(0.5) * t;
                t = Scalar(0.5) / t;
                w() = (mat.coeff(k, j) - mat.coeff(j, k)) * t;
                derived().coeffs().coeffRef(j) = (mat.coeff(j, i) + mat.coeff(i, j)) * t;
                derived().coeffs().coeffRef(k) = (mat.coeff(k, i) + mat.coeff(i, k)) * t;
            }
        }
    };

    template<typename S, int Opt>
    class Quaternion : public QuaternionBase<Quaternion<S, Opt>> {
    public:
        typedef S Scalar;
        typedef Matrix<S, 4, 1> Coefficients;
        Quaternion() {}
        Quaternion(const S &w, const S &x, const S &y, const S &z) : c_(x, y, z, w) {}
        explicit Quaternion(const S *data) : c_(data) {}
        explicit Quaternion(const Matrix<S, 4, 1> &c) : c_(c) {}
        Quaternion(const Quaternion &) = default;
        Quaternion &operator=(const Quaternion &) = default;
        template<typename OD> Quaternion(const QuaternionBase<OD> &o) : c_(o.derived().coeffs()) {}
        template<typename OD> Quaternion &operator=(const QuaternionBase<OD> &o) { c_ = o.derived().coeffs(); return *this; }
        template<typename MD, typename = typename std::enable_if<MatrixBase<MD>::RowsAtCompileTime == 3 && MatrixBase<MD>::ColsAtCompileTime == 3>::type>
        explicit Quaternion(const MatrixBase<MD> &m) { this->set_from_matrix(m); }
        template<typename MD, typename = typename std::enable_if<MatrixBase<MD>::RowsAtCompileTime == 3 && MatrixBase<MD>::ColsAtCompileTime == 3>::type>
        Quaternion &operator=(const MatrixBase<MD> &m) { this->set_from_matrix(m); return *this; }
        Coefficients &coeffs() { return c_; }
        const Coefficients &coeffs() const { return c_; }
        static Quaternion Identity() { return Quaternion(S(1), S(0), S(0), S(0)); }
        static Quaternion UnitRandom() {
            Matrix<S, 4, 1> c = Matrix<S, 4, 1>::Random();
            Quaternion q(c); q.normalize(); return q;
        }
    private:
        Coefficients c_;
    };

    template<typename S, int O, int MO, typename St>
    class Map<Quaternion<S, O>, MO, St> : public QuaternionBase<Map<Quaternion<S, O>, MO, St>> {
    public:
        typedef S Scalar;
        explicit Map(S *p) : c_(p) {}
        Map(const Map &) = default;
        Map<Matrix<S, 4, 1>> &coeffs() { return c_; }
        const Map<Matrix<S, 4, 1>> &coeffs() const { return c_; }
        template<typename OD> Map &operator=(const QuaternionBase<OD> &o) { Matrix<S, 4, 1> t = o.derived().coeffs(); c_ = t; return *this; }
        Map &operator=(const Map &o) { Matrix<S, 4, 1> t = o.coeffs(); c_ = t; return *this; }
    private:
        Map<Matrix<S, 4, 1>> c_;
    };
    template<typename S, int O, int MO, typename St>
    class Map<const Quaternion<S, O>, MO, St> : public QuaternionBase<Map<const Quaternion<S, O>, MO, St>> {
    public:
        typedef S Scalar;
        explicit Map(const S *p) : c_(p) {}
        Map(const Map &) = default;
        const Map<const Matrix<S, 4, 1>> &coeffs() const { return c_; }
    private:
        Map<const Matrix<S, 4, 1>> c_;
    };
    template<typename D> std::ostream &operator<<(std::ostream &os, const QuaternionBase<D> &q) {
        return os << q.x() << "i + " << q.y() << "j + " << q.z() << "k + " << q.w();
    }

    template<typename S> class AngleAxis {
    public:
        AngleAxis(const S &angle, const Matrix<S, 3, 1> &axis) : angle_(angle), axis_(axis) {}
        Matrix<S, 3, 3> toRotationMatrix() const {
            using std::sin; using std::cos;
            Matrix<S, 3, 3> res;
            Matrix<S, 3, 1> sin_axis = sin(angle_) * axis_;
            S c = cos(angle_);
            Matrix<S, 3, 1> cos1_axis = (S(1) - c) * axis_;
            S tmp;
            tmp = cos1_axis.x() * axis_.y(); res.coeffRef(0, 1) = tmp - sin_axis.z(); res.coeffRef(1, 0) = tmp + sin_axis.z();
            tmp = cos1_axis.x() * axis_.z(); res.coeffRef(0, 2) = tmp + sin_axis.y(); res.coeffRef(2, 0) = tmp - sin_axis.y();
            tmp = cos1_axis.y() * axis_.z(); res.coeffRef(1, 2) = tmp - sin_axis.x(); res.coeffRef(2, 1) = tmp + sin_axis.x();
            res.coeffRef(0, 0) = cos1_axis.x() * axis_.x() + c;
            res.coeffRef(1, 1) = cos1_axis.y() * axis_.y() + c;
            res.coeffRef(2, 2) = cos1_axis.z() * axis_.z() + c;
            return res;
        }
        Matrix<S, 3, 3> matrix() const { return toRotationMatrix(); }
        operator Quaternion<S>() const {
            using std::sin; using std::cos;
            S h = S(0.5) * angle_;
            Matrix<S, 3, 1> v = sin(h) * axis_;
            return Quaternion<S>(cos(h), v.x(), v.y(), v.z());
        }
    private:
        S angle_; Matrix<S, 3, 1> axis_;
    };
    typedef AngleAxis<double> AngleAxisd;
    typedef AngleAxis<float> AngleAxisf;

    // ------------------------------------------------------------------------------------------------------------
    template<typename S, int Dim, int Mode>
    class Transform {
    public:
        typedef Matrix<S, Dim + 1, Dim + 1> MatrixType;
        Transform() { m_.setIdentity(); }
        template<typename MD> explicit Transform(const MatrixBase<MD> &m) {
            m_.setIdentity();
            for (Index j = 0; j < m.cols(); ++j) for (Index i = 0; i < m.rows(); ++i) m_.coeffRef(i, j) = m.coeff(i, j);
        }
        static Transform Identity() { return Transform(); }
        MatrixType &matrix() { return m_; }
        const MatrixType &matrix() const { return m_; }
        Block<MatrixType, Dim, Dim> linear() { return m_.template block<Dim, Dim>(0, 0); }
        Block<const MatrixType, Dim, Dim> linear() const { return m_.template block<Dim, Dim>(0, 0); }
        Block<MatrixType, Dim, Dim> rotation() { return linear(); }
        Block<const MatrixType, Dim, Dim> rotation() const { return linear(); }
        Block<MatrixType, Dim, 1> translation() { return m_.template block<Dim, 1>(0, Dim); }
        Block<const MatrixType, Dim, 1> translation() const { return m_.template block<Dim, 1>(0, Dim); }
        Transform operator*(const Transform &o) const { Transform r; r.m_ = m_ * o.m_; return r; }
        template<typename VD> Matrix<S, Dim, 1> operator*(const MatrixBase<VD> &v) const {
            Matrix<S, Dim, 1> lin = Matrix<S, Dim, Dim>(linear()) * v;
            return lin + Matrix<S, Dim, 1>(translation());
        }
        Transform inverse() const {
            Transform r;
            Matrix<S, Dim, Dim> rt = Matrix<S, Dim, Dim>(linear()).transpose();
            r.m_.template block<Dim, Dim>(0, 0) = rt;
            r.m_.template block<Dim, 1>(0, Dim) = -(rt * Matrix<S, Dim, 1>(translation()));
            return r;
        }
        template<typename T> Transform<T, Dim, Mode> cast() const { Transform<T, Dim, Mode> r; r.matrix() = m_.template cast<T>(); return r; }
    private:
        MatrixType m_;
    };
    typedef Transform<double, 3, Isometry> Isometry3d;
    typedef Transform<float, 3, Isometry> Isometry3f;
    typedef Transform<double, 3, Affine> Affine3d;

    // ------------------------------------------------------------------------------------------------------------
    // Jacobi rotation helpers + two-sided Jacobi SVD (square real matrices)
    template<typename S> struct JacobiRotation {
        S c_, s_;
        JacobiRotation() : c_(1), s_(0) {}
        JacobiRotation(const S &c, const S &s) : c_(c), s_(s) {}
        S &c() { return c_; }
        S &s() { return s_; }
        const S &c() const { return c_; }
        const S &s() const { return s_; }
        JacobiRotation operator*(const JacobiRotation &o) const { return JacobiRotation(c_ * o.c_ - s_ * o.s_, c_ * o.s_ + s_ * o.c_); }
        JacobiRotation transpose() const { return JacobiRotation(c_, -s_); }
        JacobiRotation adjoint() const { return JacobiRotation(c_, -s_); }
        bool makeJacobi(const S &x, const S &y, const S &z) {
            using std::sqrt; using std::abs;
            S deno = S(2) * abs(y);
            if (deno < (std::numeric_limits<S>::min)()) {
                c_ = S(1); s_ = S(0);
                return false;
            }
            S tau = (x - z) / deno;
            S w = sqrt(internal::abs2(tau) + S(1));
            S t;
            if (tau > S(0)) t = S(1) / (tau + w);
            else t = S(1) / (tau - w);
            S sign_t = t > S(0) ? S(1) : S(-1);
            S n = S(1) / sqrt(internal::abs2(t) + S(1));
            s_ = -sign_t * (y / abs(y)) * abs(t) * n;
            c_ = n;
            return true;
        }
    };
    namespace internal {
        // x_i' = c x_i + s y_i ; y_i' = -s x_i + c y_i
        template<typename M, typename S> void rot_rows(M &m, Index p, Index q, const JacobiRotation<S> &j) {
            if (j.c() == S(1) && j.s() == S(0)) return;
            for (Index k = 0; k < m.cols(); ++k) {
                S xi = m.coeff(p, k), yi = m.coeff(q, k);
                m.coeffRef(p, k) = j.c() * xi + j.s() * yi;
                m.coeffRef(q, k) = -j.s() * xi + j.c() * yi;
            }
        }
        // applyOnTheRight(p, q, j) == rotation of columns p, q by j.transpose()
        template<typename M, typename S> void rot_cols(M &m, Index p, Index q, const JacobiRotation<S> &j) {
            JacobiRotation<S> jt = j.transpose();
            if (jt.c() == S(1) && jt.s() == S(0)) return;
            for (Index k = 0; k < m.rows(); ++k) {
                S xi = m.coeff(k, p), yi = m.coeff(k, q);
                m.coeffRef(k, p) = jt.c() * xi + jt.s() * yi;
                m.coeffRef(k, q) = -jt.s() * xi + jt.c() * yi;
            }
        }
    }

    template<typename M, int QR>
    class JacobiSVD {
    public:
        typedef typename MatrixBase<M>::Scalar S;
        typedef Matrix<S, MatrixBase<M>::RowsAtCompileTime, 1> SingularValuesType;
        JacobiSVD() {}
        JacobiSVD(const M &matrix, unsigned int options = 0) { compute(matrix, options); }
        JacobiSVD &compute(const M &matrix, unsigned int options = 0) {
            using std::abs; using std::sqrt;
            assert(matrix.rows() == matrix.cols() && "mini JacobiSVD: square matrices only");
            const Index n = matrix.rows();
            const bool wantU = options & (ComputeFullU | ComputeThinU), wantV = options & (ComputeFullV | ComputeThinV);
            const S precision = S(2) * NumTraits<S>::epsilon();
            const S considerAsZero = (std::numeric_limits<S>::min)();
            S scale = matrix.cwiseAbs().maxCoeff();
            if (!(scale == scale) || std::isinf(double(scale))) { info_ok_ = false; return *this; }
            if (scale == S(0)) scale = S(1);
            M W = matrix / scale;
            U_ = M::Identity(n, n); V_ = M::Identity(n, n);
            S maxDiagEntry = S(0);
            for (Index i = 0; i < n; ++i) maxDiagEntry = std::max(maxDiagEntry, abs(W.coeff(i, i)));
            bool finished = false;
            while (!finished) {
                finished = true;
                for (Index p = 1; p < n; ++p) {
                    for (Index q = 0; q < p; ++q) {
                        S threshold = std::max(considerAsZero, precision * maxDiagEntry);
                        if (abs(W.coeff(p, q)) > threshold || abs(W.coeff(q, p)) > threshold) {
                            finished = false;
                            JacobiRotation<S> j_left, j_right;
                            real_2x2(W, p, q, &j_left, &j_right);
                            internal::rot_rows(W, p, q, j_left);
                            if (wantU) internal::rot_cols(U_, p, q, j_left.transpose());
                            internal::rot_cols(W, p, q, j_right);
                            if (wantV) internal::rot_cols(V_, p, q, j_right);
                            maxDiagEntry = std::max(maxDiagEntry, std::max(abs(W.coeff(p, p)), abs(W.coeff(q, q))));
                        }
                    }
                }
            }
            sv_ = SingularValuesType(n, 1);
            for (Index i = 0; i < n; ++i) {
                S a = abs(W.coeff(i, i));
                sv_.coeffRef(i) = a;
                if (wantU && a != S(0)) { S f = W.coeff(i, i) / a; for (Index k = 0; k < n; ++k) U_.coeffRef(k, i) *= f; }
            }
            sv_ *= scale;
            nonzero_ = n;
            for (Index i = 0; i < n; ++i) {
                Index pos = 0; S maxRemaining = sv_.coeff(i);
                for (Index k = i + 1; k < n; ++k) if (sv_.coeff(k) > maxRemaining) { maxRemaining = sv_.coeff(k); pos = k - i; }
                if (maxRemaining == S(0)) { nonzero_ = i; break; }
                if (pos) {
                    pos += i;
                    std::swap(sv_.coeffRef(i), sv_.coeffRef(pos));
                    if (wantU) for (Index k = 0; k < n; ++k) std::swap(U_.coeffRef(k, pos), U_.coeffRef(k, i));
                    if (wantV) for (Index k = 0; k < n; ++k) std::swap(V_.coeffRef(k, pos), V_.coeffRef(k, i));
                }
            }
            return *this;
        }
        const M &matrixU() const { return U_; }
        const M &matrixV() const { return V_; }
        const SingularValuesType &singularValues() const { return sv_; }
        Index nonzeroSingularValues() const { return nonzero_; }
    private:
        static void real_2x2(const M &W, Index p, Index q, JacobiRotation<S> *j_left, JacobiRotation<S> *j_right) {
            using std::sqrt; using std::abs;
            S m00 = W.coeff(p, p), m01 = W.coeff(p, q), m10 = W.coeff(q, p), m11 = W.coeff(q, q);
            JacobiRotation<S> rot1;
            S t = m00 + m11;
            S d = m10 - m01;
            if (abs(d) < (std::numeric_limits<S>::min)()) {
                rot1.s() = S(0); rot1.c() = S(1);
            } else {
                S u = t / d;
                S tmp = sqrt(S(1) + internal::abs2(u));
                rot1.s() = S(1) / tmp;
                rot1.c() = u / tmp;
            }
            // m.applyOnTheLeft(0, 1, rot1)
            S n00 = rot1.c() * m00 + rot1.s() * m10, n01 = rot1.c() * m01 + rot1.s() * m11;
            S n11 = -rot1.s() * m01 + rot1.c() * m11;
            j_right->makeJacobi(n00, n01, n11);
            *j_left = rot1 * j_right->transpose();
        }
        M U_, V_; SingularValuesType sv_; Index nonzero_ = 0; bool info_ok_ = true;
    };

    // ------------------------------------------------------------------------------------------------------------
    // LDLT with diagonal pivoting (lower, unblocked, in place) + solve
    template<typename M>
    class LDLT {
    public:
        typedef typename MatrixBase<M>::Scalar S;
        explicit LDLT(const M &a) : m_(a), n_(a.rows()), tr_(std::size_t(a.rows())) { factor(); }
        template<typename BD> typename MatrixBase<BD>::PlainObject solve(const MatrixBase<BD> &b) const {
            using std::abs;
            typename MatrixBase<BD>::PlainObject x = b.eval();
            const Index nc = x.cols();
            // x = P b
            for (Index k = 0; k < n_; ++k) if (tr_[std::size_t(k)] != k) for (Index c = 0; c < nc; ++c) std::swap(x.coeffRef(k, c), x.coeffRef(tr_[std::size_t(k)], c));
            // L^-1 (unit lower), column oriented forward substitution
            for (Index c = 0; c < nc; ++c)
                for (Index i = 0; i < n_; ++i) {
                    S xi = x.coeff(i, c);
                    if (xi != S(0)) for (Index r = i + 1; r < n_; ++r) x.coeffRef(r, c) -= xi * m_.coeff(r, i);
                }
            // D^+ : tolerance = smallest positive normal (Eigen: 1 / highest())
            const S tolerance = (std::numeric_limits<S>::min)();
            for (Index i = 0; i < n_; ++i) {
                S di = m_.coeff(i, i);
                for (Index c = 0; c < nc; ++c) {
                    if (abs(di) > tolerance) x.coeffRef(i, c) /= di; else x.coeffRef(i, c) = S(0);
                }
            }
            // L^-T, backward substitution (row oriented on the transposed = column dot products)
            for (Index c = 0; c < nc; ++c)
                for (Index i = n_ - 1; i >= 0; --i) {
                    S s = x.coeff(i, c);
                    for (Index r = i + 1; r < n_; ++r) s -= m_.coeff(r, i) * x.coeff(r, c);
                    x.coeffRef(i, c) = s;
                }
            // P^T
            for (Index k = n_ - 1; k >= 0; --k) if (tr_[std::size_t(k)] != k) for (Index c = 0; c < nc; ++c) std::swap(x.coeffRef(k, c), x.coeffRef(tr_[std::size_t(k)], c));
            return x;
        }
        bool isPositive() const { return sign_ >= 0; }
        Matrix<S, Dynamic, 1> vectorD() const { return m_.diagonal(); }
        const M &matrixLDLT() const { return m_; }
    private:
        void factor() {
            using std::abs;
            sign_ = 0;
            if (n_ <= 1) { if (n_ == 1) tr_[0] = 0; return; }
            std::vector<S> temp(std::size_t(n_), S(0));
            for (Index k = 0; k < n_; ++k) {
                // biggest |diagonal| of the trailing part
                Index piv = k; S big = abs(m_.coeff(k, k));
                for (Index i = k + 1; i < n_; ++i) if (abs(m_.coeff(i, i)) > big) { big = abs(m_.coeff(i, i)); piv = i; }
                tr_[std::size_t(k)] = piv;
                if (k != piv) {
                    // symmetric swap on the lower triangle
                    Index s = n_ - piv - 1;
                    for (Index j = 0; j < k; ++j) std::swap(m_.coeffRef(k, j), m_.coeffRef(piv, j));
                    for (Index i = 0; i < s; ++i) std::swap(m_.coeffRef(piv + 1 + i, k), m_.coeffRef(piv + 1 + i, piv));
                    std::swap(m_.coeffRef(k, k), m_.coeffRef(piv, piv));
                    for (Index i = k + 1; i < piv; ++i) std::swap(m_.coeffRef(i, k), m_.coeffRef(piv, i));
                }
                Index rs = n_ - k - 1;
                if (k > 0) {
                    for (Index j = 0; j < k; ++j) temp[std::size_t(j)] = m_.coeff(j, j) * m_.coeff(k, j);
                    S acc = m_.coeff(k, 0) * temp[0];
                    for (Index j = 1; j < k; ++j) acc = acc + m_.coeff(k, j) * temp[std::size_t(j)];
                    m_.coeffRef(k, k) -= acc;
                    for (Index i = 0; i < rs; ++i) {
                        S a = m_.coeff(k + 1 + i, 0) * temp[0];
                        for (Index j = 1; j < k; ++j) a = a + m_.coeff(k + 1 + i, j) * temp[std::size_t(j)];
                        m_.coeffRef(k + 1 + i, k) -= a;
                    }
                }
                S akk = m_.coeff(k, k);
                bool pivot_is_valid = abs(akk) > S(0);
                if (k == 0 && !pivot_is_valid) {
                    sign_ = 0;
                    for (Index j = 0; j < n_; ++j) tr_[std::size_t(j)] = j;
                    return;
                }
                if (rs > 0 && pivot_is_valid) for (Index i = 0; i < rs; ++i) m_.coeffRef(k + 1 + i, k) /= akk;
                if (akk > S(0)) { if (sign_ == 0 && k == 0) sign_ = 1; else if (sign_ < 0) sign_ = 2; }
                else if (akk < S(0)) { if (sign_ == 0 && k == 0) sign_ = -1; else if (sign_ == 1) sign_ = 2; }
            }
        }
        M m_; Index n_; std::vector<Index> tr_; int sign_ = 0;
    };

    // ------------------------------------------------------------------------------------------------------------
#define CTGN_MINI_EIGEN_TYPEDEFS(Type, Suf)                        \
    typedef Matrix<Type, 2, 1> Vector2##Suf;                        \
    typedef Matrix<Type, 3, 1> Vector3##Suf;                        \
    typedef Matrix<Type, 4, 1> Vector4##Suf;                        \
    typedef Matrix<Type, Dynamic, 1> VectorX##Suf;                  \
    typedef Matrix<Type, 1, 2> RowVector2##Suf;                     \
    typedef Matrix<Type, 1, 3> RowVector3##Suf;                     \
    typedef Matrix<Type, 1, 4> RowVector4##Suf;                     \
    typedef Matrix<Type, 1, Dynamic> RowVectorX##Suf;               \
    typedef Matrix<Type, 2, 2> Matrix2##Suf;                        \
    typedef Matrix<Type, 3, 3> Matrix3##Suf;                        \
    typedef Matrix<Type, 4, 4> Matrix4##Suf;                        \
    typedef Matrix<Type, Dynamic, Dynamic> MatrixX##Suf;
    CTGN_MINI_EIGEN_TYPEDEFS(double, d)
    CTGN_MINI_EIGEN_TYPEDEFS(float, f)
    CTGN_MINI_EIGEN_TYPEDEFS(int, i)
#undef CTGN_MINI_EIGEN_TYPEDEFS
    template<typename T> using Vector3 = Matrix<T, 3, 1>;
    template<typename T> using Vector4 = Matrix<T, 4, 1>;
    template<typename T> using Matrix3 = Matrix<T, 3, 3>;
    template<typename T> using Matrix4 = Matrix<T, 4, 4>;
    typedef Quaternion<double> Quaterniond;
    typedef Quaternion<float> Quaternionf;
}

#endif
