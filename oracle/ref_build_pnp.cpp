// ref_build_pnp.cpp -- oracle/_ref/libref_uncertainty_pnp.so: the reference's own uncertainty_pnp.cpp, #included where
// it lies under /root/reference (read-only, nothing copied), against oracle/ref_shim_pnp/ceres/ceres.h (see there) and
// the reference's vendored header-only ceres/rotation.h, ceres/jet.h and Eigen.  Test infrastructure only.
//   exported: uncertainty_pnp (the reference's own C symbol, uncertainty_pnp.cpp:61-92: ITS functor, the shim's LM),
//             refpnp_eval (residuals + Jacobian of the reference's functor at a pose), refpnp_set_max_iterations,
//             refpnp_last_summary.
#include "ceres/ceres.h"

#define main refpnp_unused_main      /* the file carries a demo main() */
#include "uncertainty_pnp.cpp"
#undef main

namespace {
int g_max_iter = -1;
ceres::Solver::Summary g_last;

// solve the 6x6 system (A + diag(d)) x = b by Cholesky; false when not positive definite
bool chol_solve(const double *A, const double *d, const double *b, double *x, int n)
{
    double L[36];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * n + j] + (i == j ? d[i] : 0.0);
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                L[i * n + i] = std::sqrt(s);
            } else {
                L[i * n + j] = s / L[j * n + j];
            }
        }
    double y[6];
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
        y[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
    return true;
}

double evaluate(const ceres::Problem &p, const double *x, double *JtJ, double *g)
{
    const int n = 6;
    if (JtJ) { std::memset(JtJ, 0, sizeof(double) * n * n); std::memset(g, 0, sizeof(double) * n); }
    double cost = 0;
    for (size_t bi = 0; bi < p.blocks.size(); ++bi) {
        double r[2], J[12];
        double *jac[1] = {J};
        const double *par[1] = {x};
        p.blocks[bi]->Evaluate(par, r, JtJ ? jac : nullptr);
        cost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
        if (JtJ)
            for (int k = 0; k < 2; ++k)
                for (int i = 0; i < n; ++i) {
                    g[i] += J[k * n + i] * r[k];
                    for (int j = 0; j < n; ++j) JtJ[i * n + j] += J[k * n + i] * J[k * n + j];
                }
    }
    return cost;
}
}  // namespace

namespace ceres {
void Solve(const Solver::Options &o, Problem *problem, Solver::Summary *summary)
{
    const int n = 6;
    double *x = problem->params;
    double JtJ[36], g[6], radius = o.initial_trust_region_radius, decrease = 2.0;
    double cost = evaluate(*problem, x, JtJ, g);
    summary->initial_cost = cost;
    double s2[6];                        // Jacobi scaling of the initial point, squared: 1 / (1 + |J_i|)^2
    for (int i = 0; i < n; ++i) { const double sc = 1.0 / (1.0 + std::sqrt(JtJ[i * n + i])); s2[i] = sc * sc; }
    const int max_iter = g_max_iter >= 0 ? g_max_iter : o.max_num_iterations;
    int it = 0;
    summary->termination = 0;
    for (; it < max_iter; ++it) {
        double gmax = 0;
        for (int i = 0; i < n; ++i) gmax = std::fmax(gmax, std::fabs(g[i]));
        if (gmax <= o.gradient_tolerance) { summary->termination = 1; break; }
        double d[6], step[6], neg_g[6];
        for (int i = 0; i < n; ++i) {
            const double di = std::fmin(std::fmax(JtJ[i * n + i] * s2[i], o.min_lm_diagonal), o.max_lm_diagonal);
            d[i] = di / (s2[i] * radius);
            neg_g[i] = -g[i];
        }
        bool ok = chol_solve(JtJ, d, neg_g, step, n);
        double xn = 0, sn = 0, model = 0;
        if (ok) {
            for (int i = 0; i < n; ++i) {
                xn += x[i] * x[i]; sn += step[i] * step[i];
                double Js = 0;
                for (int j = 0; j < n; ++j) Js += JtJ[i * n + j] * step[j];
                model -= step[i] * (g[i] + 0.5 * Js);
            }
            if (std::sqrt(sn) <= o.parameter_tolerance * (std::sqrt(xn) + o.parameter_tolerance)) { summary->termination = 2; break; }
        }
        double cand[6], new_cost = cost;
        double rho = -1;
        if (ok && model > 0) {
            for (int i = 0; i < n; ++i) cand[i] = x[i] + step[i];
            new_cost = evaluate(*problem, cand, nullptr, nullptr);
            rho = (cost - new_cost) / model;
        }
        if (rho > o.min_relative_decrease && std::isfinite(new_cost)) {
            const double change = cost - new_cost;
            std::memcpy(x, cand, sizeof(cand));
            const double t = 2.0 * rho - 1.0;
            radius = std::fmin(radius / std::fmax(1.0 / 3.0, 1.0 - t * t * t), o.max_trust_region_radius);
            decrease = 2.0;
            const double old = cost;
            cost = evaluate(*problem, x, JtJ, g);
            if (std::fabs(change) <= o.function_tolerance * old) { summary->termination = 3; ++it; break; }
        } else {
            radius /= decrease;
            decrease *= 2.0;
            if (radius < o.min_trust_region_radius) { summary->termination = 4; break; }
        }
    }
    summary->final_cost = cost;
    summary->num_iterations = it;
    g_last = *summary;
}
}  // namespace ceres

extern "C" {
// residuals [pn,2] and Jacobian [pn,2,6] of the reference's functor (ReprojectionErrorArray, uncertainty_pnp.cpp:7-55)
// at pose rt[6]; returns the cost 0.5 sum r^2
__attribute__((visibility("default"))) double refpnp_eval(const double *pts2d, const double *pts3d, const double *wgt2d,
                                                           const double *K, const double *rt, int pn, double *res, double *jac)
{
    double cost = 0;
    for (int i = 0; i < pn; ++i) {
        ceres::CostFunction *f = ReprojectionErrorArray::Create(pts2d[i * 2], pts2d[i * 2 + 1], pts3d[i * 3], pts3d[i * 3 + 1],
                                                                pts3d[i * 3 + 2], wgt2d[i * 3], wgt2d[i * 3 + 1], wgt2d[i * 3 + 2],
                                                                K[0], K[4], K[2], K[5]);
        double r[2], J[12];
        double *jp[1] = {J};
        const double *par[1] = {rt};
        f->Evaluate(par, r, jp);
        cost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
        if (res) { res[2 * i] = r[0]; res[2 * i + 1] = r[1]; }
        if (jac) std::memcpy(jac + 12 * i, J, sizeof(J));
        delete f;
    }
    return cost;
}
__attribute__((visibility("default"))) void refpnp_set_max_iterations(int n) { g_max_iter = n; }
__attribute__((visibility("default"))) void refpnp_last_summary(double *initial_cost, double *final_cost, int *iters, int *term)
{
    *initial_cost = g_last.initial_cost; *final_cost = g_last.final_cost; *iters = g_last.num_iterations; *term = g_last.termination;
}
}
