// ceres/ceres.h -- SHIM, test infrastructure only (oracle/_ref).  It exists so that the reference's own
// lib/csrc/uncertainty_pnp/src/uncertainty_pnp.cpp compiles WHERE IT LIES (nothing of it is copied): that file needs
// <ceres/rotation.h> -- header-only, taken from the reference's vendored include/ -- and <ceres/ceres.h>, whose library
// (the vendored libceres.so.2.0.0) cannot be linked in this image (libglog / libspqr / libcholmod / liblapack missing).
//
// What the shim provides, and what that pins:
//   * AutoDiffCostFunction<F, 2, 6>: forward-mode derivatives of the REFERENCE'S functor through the reference's own
//     vendored ceres/jet.h -- residuals and Jacobians are exactly what real Ceres would evaluate;
//   * Problem / Solver / Solve: a plain dense Levenberg-Marquardt with Ceres' documented default schedule (initial
//     radius 1e4, rho > 1e-3 accepts, radius /= max(1/3, 1 - (2 rho - 1)^3) on success, halved with a doubling factor on
//     failure, function / gradient / parameter tolerances 1e-6 / 1e-10 / 1e-8, 50 iterations).  This is NOT Ceres'
//     code: the iterate path of the real library stays unpinned; what is pinned is the function being minimised and
//     the minimum it has (cross-checked against scipy.optimize.least_squares in tests/test_pnp.py).
#ifndef PVV_SHIM_CERES_H_
#define PVV_SHIM_CERES_H_

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "ceres/jet.h"

namespace ceres {

class CostFunction {
public:
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
    virtual int num_residuals() const = 0;
    virtual int num_parameters() const = 0;
};

template <typename Functor, int kNumResiduals, int N0>
class AutoDiffCostFunction : public CostFunction {
public:
    explicit AutoDiffCostFunction(Functor *f) : functor_(f) {}
    ~AutoDiffCostFunction() override { delete functor_; }
    int num_residuals() const override { return kNumResiduals; }
    int num_parameters() const override { return N0; }
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const override
    {
        if (!jacobians || !jacobians[0]) return (*functor_)(parameters[0], residuals);
        typedef Jet<double, N0> J;
        J x[N0], r[kNumResiduals];
        for (int i = 0; i < N0; ++i) x[i] = J(parameters[0][i], i);
        if (!(*functor_)(x, r)) return false;
        for (int k = 0; k < kNumResiduals; ++k) {
            residuals[k] = r[k].a;
            for (int i = 0; i < N0; ++i) jacobians[0][k * N0 + i] = r[k].v[i];     // row-major [residual, parameter]
        }
        return true;
    }

private:
    Functor *functor_;
};

class LossFunction;

class Problem {
public:
    ~Problem() { for (size_t i = 0; i < blocks.size(); ++i) delete blocks[i]; }
    void AddResidualBlock(CostFunction *f, LossFunction *, double *x) { blocks.push_back(f); params = x; }
    std::vector<CostFunction *> blocks;
    double *params = nullptr;
};

enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, DENSE_SCHUR, SPARSE_SCHUR, SPARSE_NORMAL_CHOLESKY };

class Solver {
public:
    struct Options {
        LinearSolverType linear_solver_type = DENSE_QR;
        bool minimizer_progress_to_stdout = false;
        int max_num_iterations = 50;
        double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
        double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
        double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
    };
    struct Summary {
        double initial_cost = 0, final_cost = 0;
        int num_iterations = 0, termination = 0;
        std::string FullReport() const { return "shim"; }
    };
};

void Solve(const Solver::Options &options, Problem *problem, Solver::Summary *summary);

}  // namespace ceres
#endif
