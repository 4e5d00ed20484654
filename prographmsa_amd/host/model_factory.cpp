// model_factory.cpp — substitution models (reference src/ModelFactory.{h,cpp},
// ModelFactoryWag.cpp:17-441, ModelFactoryEcm.cpp:20-3763).
//
// The reference sets up Q with Eigen's general (non-symmetric) EigenSolver and evaluates
// P(t) = V exp(sigma t) V^-1.  Here: a symmetrised Jacobi eigen-decomposition when the generator is reversible
// to rounding error (WAG), a scaling-and-squaring matrix exponential otherwise (ECM); both agree with the
// reference to ~1e-14 (double), far below the float cast applied before P reaches the DP.
#include "pgm_host.h"

#include <algorithm>
#include <cmath>
#include <fstream>

namespace pgm {

// Solve A x = b (n x n, column-major A) by Gaussian elimination with partial pivoting.
static std::vector<double> solve(std::vector<double> A, std::vector<double> b, int n) {
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(A[r + n * c]) > std::fabs(A[piv + n * c])) piv = r;
        if (piv != c) {
            for (int k = 0; k < n; ++k) std::swap(A[c + n * k], A[piv + n * k]);
            std::swap(b[c], b[piv]);
        }
        for (int r = c + 1; r < n; ++r) {
            double f = A[r + n * c] / A[c + n * c];
            if (f == 0) continue;
            for (int k = c; k < n; ++k) A[r + n * k] -= f * A[c + n * k];
            b[r] -= f * b[c];
        }
    }
    std::vector<double> x(n);
    for (int r = n - 1; r >= 0; --r) {
        double s = b[r];
        for (int k = r + 1; k < n; ++k) s -= A[r + n * k] * x[k];
        x[r] = s / A[r + n * r];
    }
    return x;
}

// Cyclic Jacobi for a symmetric matrix (column-major).  Returns eigenvalues; U holds eigenvectors
// as columns.
static std::vector<double> jacobi(std::vector<double> A, std::vector<double> &U, int n) {
    U.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) U[i + n * i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += A[p + n * q] * A[p + n * q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p) {
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p + n * q];
                if (apq == 0) continue;
                double theta = (A[q + n * q] - A[p + n * p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; ++k) {
                    double akp = A[k + n * p], akq = A[k + n * q];
                    A[k + n * p] = c * akp - s * akq;
                    A[k + n * q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = A[p + n * k], aqk = A[q + n * k];
                    A[p + n * k] = c * apk - s * aqk;
                    A[q + n * k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double ukp = U[k + n * p], ukq = U[k + n * q];
                    U[k + n * p] = c * ukp - s * ukq;
                    U[k + n * q] = s * ukp + c * ukq;
                }
            }
        }
    }
    std::vector<double> ev(n);
    for (int i = 0; i < n; ++i) ev[i] = A[i + n * i];
    return ev;
}

// C = A * B (n x n, column-major)
static void matmul(const std::vector<double> &A, const std::vector<double> &B, int n, std::vector<double> &C) {
    C.assign((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < n; ++k) {
            const double b = B[k + n * j];
            if (b == 0) continue;
            const double *a = &A[(size_t)n * k];
            double *c = &C[(size_t)n * j];
            for (int i = 0; i < n; ++i) c[i] += a[i] * b;
        }
}

// exp(A) by scaling and squaring with a Taylor series (||A/2^s||_1 <= 1/2, 20 terms: truncation < 1e-20).
// The reference evaluates P(d) = V exp(sigma d) V^-1 from Eigen's non-symmetric EigenSolver; both are exp(Q d) to
// ~1e-14, far below the float cast applied before P reaches the DP, and this form needs no reversibility assumption
// (the ECM data is reversible only to ~1e-6).
static std::vector<double> expm(std::vector<double> A, int n) {
    double norm = 0;
    for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int i = 0; i < n; ++i) s += std::fabs(A[i + n * j]);
        norm = std::max(norm, s);
    }
    int sq = 0;
    while (norm > 0.5) { norm *= 0.5; ++sq; }
    const double scale = std::ldexp(1.0, -sq);
    for (double &a : A) a *= scale;
    std::vector<double> E((size_t)n * n, 0.0), term((size_t)n * n, 0.0), tmp;
    for (int i = 0; i < n; ++i) { E[i + n * i] = 1.0; term[i + n * i] = 1.0; }
    for (int k = 1; k <= 20; ++k) {
        matmul(term, A, n, tmp);
        const double inv = 1.0 / k;
        for (size_t i = 0; i < tmp.size(); ++i) { term[i] = tmp[i] * inv; E[i] += term[i]; }
    }
    for (int k = 0; k < sq; ++k) { matmul(E, E, n, tmp); E.swap(tmp); }
    return E;
}

ModelFactory::ModelFactory(int dim, const std::string &qmat_file) : dim_(dim) {
    const int n = dim;
    std::ifstream in(qmat_file.c_str());
    int r = 0, c = 0;
    in >> r >> c;
    if (!in || r != n || c != n) error("cannot read rate matrix %s", qmat_file.c_str());
    Q_.resize((size_t)n * n);
    for (size_t i = 0; i < Q_.size(); ++i) in >> Q_[i];  // column-major, Q(i,j) = data[i + n*j]
    if (!in) error("truncated rate matrix %s", qmat_file.c_str());

    // pi = null vector of Q^T, normalised to sum 1 (ModelFactoryWag.cpp:422-430): solve
    // Q^T pi = 0 with the last equation replaced by sum(pi) = 1.
    std::vector<double> A((size_t)n * n), b(n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[i + n * j] = Q_[j + n * i];
    for (int j = 0; j < n; ++j) A[(n - 1) + n * j] = 1.0;
    b[n - 1] = 1.0;
    freqs_ = solve(A, b, n);

    // normalise rate (ModelFactoryWag.cpp:432-435)
    for (int i = 0; i < n; ++i) Q_[i + n * i] = 0;
    for (int i = 0; i < n; ++i) {
        double s = 0;
        for (int j = 0; j < n; ++j) s += Q_[i + n * j];
        Q_[i + n * i] = -s;
    }
    double rate = 0;
    for (int i = 0; i < n; ++i) rate += freqs_[i] * Q_[i + n * i];
    for (double &q : Q_) q /= -rate;

    // Eigen-decomposition of the normalised Q.  NOTE (reference quirk, kept): the literal arrays are
    // mapped column-major (ModelFactoryWag.cpp:420), which makes Q the TRANSPOSE of the usual WAG/ECM
    // generator; its columns sum to zero, so the "null vector of Q^T" above is the uniform vector and
    // freqs = 1/dim, while the diagonal reset makes rows sum to zero.  The resulting Q = diag(w) * L
    // (L symmetric) is reversible with respect to ITS OWN stationary distribution u (u^T Q = 0), not
    // with respect to freqs, so the symmetrisation B = U^1/2 Q U^-1/2 uses u.
    std::vector<double> At((size_t)n * n), bt(n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) At[i + n * j] = Q_[j + n * i];
    for (int j = 0; j < n; ++j) At[(n - 1) + n * j] = 1.0;
    bt[n - 1] = 1.0;
    std::vector<double> u = solve(At, bt, n);
    std::vector<double> B((size_t)n * n), U;
    double asym = 0, scale = 0;
    for (int i = 0; i < n; ++i) {
        if (!(u[i] > 0)) error("rate matrix has no positive stationary distribution");
        for (int j = 0; j < n; ++j) B[i + n * j] = std::sqrt(u[i]) * Q_[i + n * j] / std::sqrt(u[j]);
    }
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            asym = std::max(asym, std::fabs(B[i + n * j] - B[j + n * i]));
            scale = std::max(scale, std::fabs(B[i + n * j]));
            double m = 0.5 * (B[i + n * j] + B[j + n * i]);
            B[i + n * j] = B[j + n * i] = m;
        }
    // WAG is reversible to rounding error; the ECM data only to ~1e-6: then P(d) = exp(Q d) is evaluated with a
    // scaling-and-squaring matrix exponential instead (no reversibility assumed), see fillP.
    use_eigen_ = asym <= 1e-9 * std::max(scale, 1.0);
    if (!use_eigen_) return;
    sigma_ = jacobi(B, U, n);
    V_.resize((size_t)n * n);
    Vi_.resize((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            V_[i + n * j] = U[i + n * j] / std::sqrt(u[i]);
            Vi_[j + n * i] = U[i + n * j] * std::sqrt(u[i]);
        }
}

ModelFactory *ModelFactory::getDefault(const Alphabet &a) {
    if (a.kind == ALPHA_AA) return new ModelFactory(20, data_dir() + "/wag.qmat");
    return new ModelFactory(61, data_dir() + "/ecm.qmat");
}

double ModelFactory::getDelta(distance_t distance) const {
    return (1.0 - std::exp(-distance * cmdlineopts.indel_rate)) / 2.0;
}

void ModelFactory::parseDistance(distance_t distance, Model &model) {
    distance = std::max(0.0, distance);
    if (cmdlineopts.mldist_flag || cmdlineopts.mldist_gap_flag) {
        if (std::isnan(distance)) distance = 5.2;
        model.distance = distance;
        double ed = std::exp(model.distance);
        model.divergence = -0.5 * (5.0 * ed - std::sqrt(45.0 * ed * ed - 20.0 * ed)) / ed;
    } else {
        if (std::isnan(distance)) distance = 1.0;
        if (distance > 0.85) model.distance = 5.2;
        else model.distance = -std::log(1.0 - distance - 0.2 * distance * distance);
        model.divergence = distance;
    }
    model.distance = std::max(std::min(model.distance, cmdlineopts.max_dist), cmdlineopts.min_dist);
    model.divergence = std::max(std::min(model.divergence, cmdlineopts.max_pdist), cmdlineopts.min_pdist);
}

void ModelFactory::fillP(Model &model) const {
    const int n = dim_;
    model.dim = n;
    if (use_eigen_) {
        std::vector<double> e(n), tmp((size_t)n * n);
        for (int k = 0; k < n; ++k) e[k] = std::exp(sigma_[k] * model.distance);
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < n; ++k) tmp[i + n * k] = V_[i + n * k] * e[k];
        model.P.assign((size_t)n * n, 0.0);
        for (int j = 0; j < n; ++j)
            for (int k = 0; k < n; ++k) {
                const double v = Vi_[k + n * j];
                for (int i = 0; i < n; ++i) model.P[i + n * j] += tmp[i + n * k] * v;
            }
    } else {
        std::vector<double> A(Q_);
        for (double &a : A) a *= model.distance;
        model.P = expm(A, n);
    }
    model.M.resize((size_t)n * n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) model.M[i + n * j] = freqs_[i] * model.P[i + n * j];
    if (cmdlineopts.mldist_flag || cmdlineopts.mldist_gap_flag) {
        double tr = 0;
        for (int i = 0; i < n; ++i) tr += model.M[i + n * i];
        model.divergence = 1.0 - tr;
    }
    model.pi = freqs_;
    model.Q = Q_;
}

Model ModelFactory::getModel(distance_t distance) const {
    Model model;
    parseDistance(distance, model);
    fillP(model);
    model.epsilon = getEpsilon(model.distance);
    model.delta = getDelta(model.distance);
    return model;
}

Model ModelFactory::getModel(distance_t distance, distance_t gap_distance) const {
    Model model;
    parseDistance(gap_distance, model);
    model.epsilon = getEpsilon(model.distance);
    model.delta = getDelta(model.distance);
    parseDistance(distance, model);
    fillP(model);
    return model;
}

}  // namespace pgm
