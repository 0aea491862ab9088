// Host-side Gaussian fit of the O2 histogram and the threshold that follows from it
// (reference muse_origin/lib_origin.py:1004-1024):
//
//     ind = argmax(hist); mod = edges[ind]; ind2 = argmin((hist[ind]/2 - hist[:ind])**2)
//     sigma = (mod - edges[ind2]) / sqrt(2 ln 2);  x = bin centres;  keep x < mod + fwhm/2
//     (amplitude, mean, stddev) = LevMarLSQFitter()(Gaussian1D(max(hist), mod, sigma), x, hist)
//     thresO2 = mean - stddev * norm.ppf(pfa)
//
// astropy's LevMarLSQFitter is scipy.optimize.leastsq = MINPACK lmder with the analytic Jacobian
// of Gaussian1D, ftol 1.49012e-8, xtol 1e-7, gtol 0, maxfev 100, factor 100, automatic
// scaling.  Round 1 kept SciPy for it: 36 fits with Python callbacks = 4.2 ms on the critical
// path of a 67 ms step.  This file is the same algorithm (Levenberg-Marquardt in Moré's
// formulation: QR with column pivoting, the trust-region parameter from the secular equation,
// the same acceptance ratios, scaling and stopping rules, in the same order) written for a
// three-parameter model, run for all areas on the histogram worker pool.  The model values
// follow the NumPy expressions of origin_amd/thresholds.py operation by operation; the
// exponential is libm's where NumPy uses its own vector exp, so fits agree with SciPy's to
// ~1e-13 relative, not bit for bit (tests/test_host_logic.py states the bound; the reference
// pins G3 are met at 1e-10 like before).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

#include "common.h"

namespace {

constexpr int NP = 3;                       // amplitude, mean, stddev
constexpr double EPSMCH = 2.220446049250313e-16;
constexpr double DWARF = 2.2250738585072014e-308;
constexpr double FLT_TINY = 1.1754943508222875e-38;  // np.finfo(np.float32).tiny: stddev bound

// Euclidean norm guarded against overflow / underflow (three accumulators by magnitude)
double enorm(int n, const double *x) {
  const double rdwarf = 3.834e-20, rgiant = 1.304e19;
  double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
  const double agiant = rgiant / (double)n;
  for (int i = 0; i < n; ++i) {
    const double xabs = std::fabs(x[i]);
    if (xabs > rdwarf && xabs < agiant) {
      s2 += xabs * xabs;
    } else if (xabs <= rdwarf) {
      if (xabs > x3max) {
        const double r = x3max / xabs;
        s3 = 1.0 + s3 * (r * r);
        x3max = xabs;
      } else if (xabs != 0.0) {
        const double r = xabs / x3max;
        s3 += r * r;
      }
    } else {
      if (xabs > x1max) {
        const double r = x1max / xabs;
        s1 = 1.0 + s1 * (r * r);
        x1max = xabs;
      } else {
        const double r = xabs / x1max;
        s1 += r * r;
      }
    }
  }
  if (s1 != 0.0) return x1max * std::sqrt(s1 + (s2 / x1max) / x1max);
  if (s2 != 0.0) {
    if (s2 >= x3max) return std::sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
    return std::sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
  }
  return x3max * std::sqrt(s3);
}

// column-major m x NP matrix
struct Mat {
  int m;
  double *a;
  double &operator()(int i, int j) { return a[(size_t)j * m + i]; }
  double *col(int j) { return a + (size_t)j * m; }
};

// Householder QR with column pivoting: A P = Q R.  On return the strict upper triangle of A
// holds R, the lower trapezoid the Householder vectors, rdiag the diagonal of R, acnorm the
// norms of the input columns.
void qrfac(Mat A, int *ipvt, double *rdiag, double *acnorm, double *wa) {
  const int m = A.m, n = NP;
  for (int j = 0; j < n; ++j) {
    acnorm[j] = enorm(m, A.col(j));
    rdiag[j] = acnorm[j];
    wa[j] = rdiag[j];
    ipvt[j] = j;
  }
  const int minmn = std::min(m, n);
  for (int j = 0; j < minmn; ++j) {
    int kmax = j;
    for (int k = j; k < n; ++k)
      if (rdiag[k] > rdiag[kmax]) kmax = k;
    if (kmax != j) {
      for (int i = 0; i < m; ++i) std::swap(A(i, j), A(i, kmax));
      rdiag[kmax] = rdiag[j];
      wa[kmax] = wa[j];
      std::swap(ipvt[j], ipvt[kmax]);
    }
    double ajnorm = enorm(m - j, A.col(j) + j);
    if (ajnorm != 0.0) {
      if (A(j, j) < 0.0) ajnorm = -ajnorm;
      for (int i = j; i < m; ++i) A(i, j) /= ajnorm;
      A(j, j) += 1.0;
      for (int k = j + 1; k < n; ++k) {
        double sum = 0.0;
        for (int i = j; i < m; ++i) sum += A(i, j) * A(i, k);
        const double temp = sum / A(j, j);
        for (int i = j; i < m; ++i) A(i, k) -= temp * A(i, j);
        if (rdiag[k] != 0.0) {
          double t = A(j, k) / rdiag[k];
          t = std::max(0.0, 1.0 - t * t);
          rdiag[k] *= std::sqrt(t);
          const double r = rdiag[k] / wa[k];
          if (0.05 * (r * r) <= EPSMCH) {
            rdiag[k] = enorm(m - j - 1, A.col(k) + j + 1);
            wa[k] = rdiag[k];
          }
        }
      }
    }
    rdiag[j] = -ajnorm;
  }
}

// Solve R z = Q^T b together with D z = 0 in the least-squares sense (Givens rotations that
// eliminate the diagonal matrix D); r is n x n with leading dimension ldr, upper triangle = R.
// On return sdiag holds the diagonal of the triangular S, the strict lower triangle of r its
// transpose.
void qrsolv(double *r, int ldr, const int *ipvt, const double *diag, const double *qtb, double *x,
            double *sdiag, double *wa) {
  const int n = NP;
  auto R = [&](int i, int j) -> double & { return r[(size_t)j * ldr + i]; };
  for (int j = 0; j < n; ++j) {
    for (int i = j; i < n; ++i) R(i, j) = R(j, i);
    x[j] = R(j, j);
    wa[j] = qtb[j];
  }
  for (int j = 0; j < n; ++j) {
    const int l = ipvt[j];
    if (diag[l] != 0.0) {
      for (int k = j; k < n; ++k) sdiag[k] = 0.0;
      sdiag[j] = diag[l];
      double qtbpj = 0.0;
      for (int k = j; k < n; ++k) {
        if (sdiag[k] == 0.0) continue;
        double c, s;
        if (std::fabs(R(k, k)) < std::fabs(sdiag[k])) {
          const double cotan = R(k, k) / sdiag[k];
          s = 0.5 / std::sqrt(0.25 + 0.25 * (cotan * cotan));
          c = s * cotan;
        } else {
          const double tanv = sdiag[k] / R(k, k);
          c = 0.5 / std::sqrt(0.25 + 0.25 * (tanv * tanv));
          s = c * tanv;
        }
        R(k, k) = c * R(k, k) + s * sdiag[k];
        const double temp = c * wa[k] + s * qtbpj;
        qtbpj = -s * wa[k] + c * qtbpj;
        wa[k] = temp;
        for (int i = k + 1; i < n; ++i) {
          const double t2 = c * R(i, k) + s * sdiag[i];
          sdiag[i] = -s * R(i, k) + c * sdiag[i];
          R(i, k) = t2;
        }
      }
    }
    sdiag[j] = R(j, j);
    R(j, j) = x[j];
  }
  int nsing = n;
  for (int j = 0; j < n; ++j) {
    if (sdiag[j] == 0.0 && nsing == n) nsing = j;
    if (nsing < n) wa[j] = 0.0;
  }
  for (int k = 0; k < nsing; ++k) {
    const int j = nsing - 1 - k;
    double sum = 0.0;
    for (int i = j + 1; i < nsing; ++i) sum += R(i, j) * wa[i];
    wa[j] = (wa[j] - sum) / sdiag[j];
  }
  for (int j = 0; j < n; ++j) x[ipvt[j]] = wa[j];
}

// Levenberg-Marquardt parameter: par >= 0 such that the solution x of
// (R^T R + par D^2) x = R^T Q^T b has ||D x|| within 10 % of delta (or par = 0 and ||D x|| <=
// 1.1 delta).
void lmpar(double *r, int ldr, const int *ipvt, const double *diag, const double *qtb,
           double delta, double *par, double *x, double *sdiag, double *wa1, double *wa2) {
  const int n = NP;
  auto R = [&](int i, int j) -> double & { return r[(size_t)j * ldr + i]; };
  // Gauss-Newton direction (least-squares solution for a rank-deficient R)
  int nsing = n;
  for (int j = 0; j < n; ++j) {
    wa1[j] = qtb[j];
    if (R(j, j) == 0.0 && nsing == n) nsing = j;
    if (nsing < n) wa1[j] = 0.0;
  }
  for (int k = 0; k < nsing; ++k) {
    const int j = nsing - 1 - k;
    wa1[j] /= R(j, j);
    const double temp = wa1[j];
    for (int i = 0; i < j; ++i) wa1[i] -= R(i, j) * temp;
  }
  for (int j = 0; j < n; ++j) x[ipvt[j]] = wa1[j];
  int iter = 0;
  for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
  double dxnorm = enorm(n, wa2);
  double fp = dxnorm - delta;
  if (fp <= 0.1 * delta) {
    *par = 0.0;
    return;
  }
  // lower bound parl from the Newton step of the secular function (full rank only)
  double parl = 0.0;
  if (nsing >= n) {
    for (int j = 0; j < n; ++j) {
      const int l = ipvt[j];
      wa1[j] = diag[l] * (wa2[l] / dxnorm);
    }
    for (int j = 0; j < n; ++j) {
      double sum = 0.0;
      for (int i = 0; i < j; ++i) sum += R(i, j) * wa1[i];
      wa1[j] = (wa1[j] - sum) / R(j, j);
    }
    const double temp = enorm(n, wa1);
    parl = ((fp / delta) / temp) / temp;
  }
  // upper bound paru
  for (int j = 0; j < n; ++j) {
    double sum = 0.0;
    for (int i = 0; i <= j; ++i) sum += R(i, j) * qtb[i];
    wa1[j] = sum / diag[ipvt[j]];
  }
  const double gnorm = enorm(n, wa1);
  double paru = gnorm / delta;
  if (paru == 0.0) paru = DWARF / std::min(delta, 0.1);
  *par = std::max(*par, parl);
  *par = std::min(*par, paru);
  if (*par == 0.0) *par = gnorm / dxnorm;
  for (;;) {
    ++iter;
    if (*par == 0.0) *par = std::max(DWARF, 0.001 * paru);
    const double sq = std::sqrt(*par);
    for (int j = 0; j < n; ++j) wa1[j] = sq * diag[j];
    qrsolv(r, ldr, ipvt, wa1, qtb, x, sdiag, wa2);
    for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    dxnorm = enorm(n, wa2);
    const double temp = fp;
    fp = dxnorm - delta;
    if (std::fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10)
      return;
    // Newton correction
    for (int j = 0; j < n; ++j) {
      const int l = ipvt[j];
      wa1[j] = diag[l] * (wa2[l] / dxnorm);
    }
    for (int j = 0; j < n; ++j) {
      wa1[j] /= sdiag[j];
      const double t = wa1[j];
      for (int i = j + 1; i < n; ++i) wa1[i] -= R(i, j) * t;
    }
    const double t2 = enorm(n, wa1);
    const double parc = ((fp / delta) / t2) / t2;
    if (fp > 0.0) parl = std::max(parl, *par);
    if (fp < 0.0) paru = std::min(paru, *par);
    *par = std::max(parl, *par + parc);
  }
}

// Gaussian1D residuals and Jacobian with the operation order of thresholds.fit_gauss1d
struct GaussModel {
  const double *x, *y;
  int m;
  void resid(const double *p, double *f) const {
    const double a = p[0], mu = p[1], s = std::max(p[2], FLT_TINY);
    const double s2 = s * s;
    for (int i = 0; i < m; ++i) {
      const double d = x[i] - mu;
      f[i] = a * std::exp(-0.5 * (d * d) / s2) - y[i];
    }
  }
  void jac(const double *p, Mat J) const {
    const double a = p[0], mu = p[1], s = std::max(p[2], FLT_TINY);
    const double s2 = s * s, s3 = std::pow(s, 3.0), c = -0.5 / s2;
    for (int i = 0; i < m; ++i) {
      const double d = x[i] - mu, d2 = d * d;
      const double g = std::exp(c * d2);
      J(i, 0) = g;
      J(i, 1) = a * g * d / s2;
      J(i, 2) = a * g * d2 / s3;
    }
  }
};

// returns MINPACK's info code; p holds the start on entry and the solution on return
int lm_fit(const GaussModel &model, double *p, double ftol, double xtol, double gtol, int maxfev,
           double factor, int *nfev_out) {
  const int m = model.m, n = NP;
  if (m < n) return 0;
  std::vector<double> store((size_t)m * (n + 2));
  Mat J{m, store.data()};
  double *fvec = store.data() + (size_t)m * n, *wa4 = fvec + m;
  double diag[NP], qtf[NP], wa1[NP], wa2[NP], wa3[NP], sdiag[NP], wb1[NP], wb2[NP];
  int ipvt[NP];
  int info = 0, nfev = 1, iter = 1;
  double par = 0.0, delta = 0.0, xnorm = 0.0;
  model.resid(p, fvec);
  double fnorm = enorm(m, fvec);
  for (;;) {
    model.jac(p, J);
    qrfac(J, ipvt, wa1, wa2, wa3);
    if (iter == 1) {
      for (int j = 0; j < n; ++j) {
        diag[j] = wa2[j];
        if (wa2[j] == 0.0) diag[j] = 1.0;
      }
      for (int j = 0; j < n; ++j) wa3[j] = diag[j] * p[j];
      xnorm = enorm(n, wa3);
      delta = factor * xnorm;
      if (delta == 0.0) delta = factor;
    }
    // first n components of Q^T fvec
    std::memcpy(wa4, fvec, sizeof(double) * m);
    for (int j = 0; j < n; ++j) {
      if (J(j, j) != 0.0) {
        double sum = 0.0;
        for (int i = j; i < m; ++i) sum += J(i, j) * wa4[i];
        const double temp = -sum / J(j, j);
        for (int i = j; i < m; ++i) wa4[i] += J(i, j) * temp;
      }
      J(j, j) = wa1[j];
      qtf[j] = wa4[j];
    }
    // norm of the scaled gradient
    double gnorm = 0.0;
    if (fnorm != 0.0) {
      for (int j = 0; j < n; ++j) {
        const int l = ipvt[j];
        if (wa2[l] == 0.0) continue;
        double sum = 0.0;
        for (int i = 0; i <= j; ++i) sum += J(i, j) * (qtf[i] / fnorm);
        gnorm = std::max(gnorm, std::fabs(sum / wa2[l]));
      }
    }
    if (gnorm <= gtol) {
      info = 4;
      break;
    }
    for (int j = 0; j < n; ++j) diag[j] = std::max(diag[j], wa2[j]);
    double ratio = 0.0;
    do {
      lmpar(J.a, m, ipvt, diag, qtf, delta, &par, wa1, sdiag, wb1, wb2);
      for (int j = 0; j < n; ++j) {
        wa1[j] = -wa1[j];
        wa2[j] = p[j] + wa1[j];
        wa3[j] = diag[j] * wa1[j];
      }
      const double pnorm = enorm(n, wa3);
      if (iter == 1) delta = std::min(delta, pnorm);
      model.resid(wa2, wa4);
      ++nfev;
      const double fnorm1 = enorm(m, wa4);
      double actred = -1.0;
      if (0.1 * fnorm1 < fnorm) {
        const double t = fnorm1 / fnorm;
        actred = 1.0 - t * t;
      }
      for (int j = 0; j < n; ++j) {
        wa3[j] = 0.0;
        const int l = ipvt[j];
        const double temp = wa1[l];
        for (int i = 0; i <= j; ++i) wa3[i] += J(i, j) * temp;
      }
      const double temp1 = enorm(n, wa3) / fnorm;
      const double temp2 = (std::sqrt(par) * pnorm) / fnorm;
      const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
      const double dirder = -(temp1 * temp1 + temp2 * temp2);
      ratio = prered != 0.0 ? actred / prered : 0.0;
      if (ratio <= 0.25) {
        double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        delta = temp * std::min(delta, pnorm / 0.1);
        par /= temp;
      } else if (par == 0.0 || ratio >= 0.75) {
        delta = pnorm / 0.5;
        par *= 0.5;
      }
      if (ratio >= 1e-4) {  // successful step
        for (int j = 0; j < n; ++j) {
          p[j] = wa2[j];
          wa2[j] = diag[j] * p[j];
        }
        std::memcpy(fvec, wa4, sizeof(double) * m);
        xnorm = enorm(n, wa2);
        fnorm = fnorm1;
        ++iter;
      }
      if (std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0) info = 1;
      if (delta <= xtol * xnorm) info = 2;
      if (std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0 && info == 2) info = 3;
      if (info != 0) break;
      if (nfev >= maxfev) info = 5;
      if (std::fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
      if (delta <= EPSMCH * xnorm) info = 7;
      if (gnorm <= EPSMCH) info = 8;
      if (info != 0) break;
    } while (ratio < 1e-4);
    if (info != 0) break;
  }
  if (nfev_out) *nfev_out = nfev;
  return info;
}

}  // namespace

void origin_host_pool_run(int n, const std::function<void(int)> &task);  // thresh.hip

extern "C" int origin_gauss_fit(const double *h_x, const double *h_y, long m, double *h_p,
                                int *info, int *nfev) {
  ORIGIN_CHECK_ARG(h_x && h_y && h_p && m >= 0, "bad arguments");
  GaussModel model{h_x, h_y, (int)m};
  int nf = 0;
  const int rc = lm_fit(model, h_p, 1.49012e-8, 1e-7, 0.0, 100, 100.0, &nf);
  if (info) *info = rc;
  if (nfev) *nfev = nf;
  if (m < NP) {
    origin_set_error("Gaussian fit needs at least 3 histogram bins left of the cut, got %ld", m);
    return ORIGIN_E_ARG;
  }
  return ORIGIN_OK;
}

// One area: histogram -> (thresO2, mean, stddev) in res[0..2]; returns the status code of
// origin_o2_threshold_batch.
static int o2_threshold_one(const double *hist, const double *edges, long nb, double coef,
                            double *res) {
    res[0] = res[1] = res[2] = NAN;
    long ind = 0;                                      // np.argmax: first maximum
    for (long i = 1; i < nb; ++i)
      if (hist[i] > hist[ind]) ind = i;
    if (ind == 0) {
      return 1;
    }
    const double mod = edges[ind];
    const double half = hist[ind] / 2;
    long ind2 = 0;                                     // np.argmin((half - hist[:ind])**2)
    double bestv = (half - hist[0]) * (half - hist[0]);
    for (long i = 1; i < ind; ++i) {
      const double v = (half - hist[i]) * (half - hist[i]);
      if (v < bestv) bestv = v, ind2 = i;
    }
    const double fwhm = mod - edges[ind2];
    const double sigma = fwhm / std::sqrt(2 * std::log(2.0));
    const double s2f = 2.0 * std::sqrt(2.0 * std::log(2.0));
    const double xcut = mod + s2f * sigma / 2;
    std::vector<double> x, y;
    x.reserve(nb);
    y.reserve(nb);
    double hmax = hist[ind];
    for (long i = 0; i < nb; ++i) {
      const double c = (edges[i + 1] + edges[i]) / 2;
      if (c < xcut) x.push_back(c), y.push_back(hist[i]);
    }
    if ((long)x.size() < NP) {
      return 2;
    }
    double p[3] = {hmax, mod, sigma};
    GaussModel model{x.data(), y.data(), (int)x.size()};
    lm_fit(model, p, 1.49012e-8, 1e-7, 0.0, 100, 100.0, nullptr);
    const double mea = p[1], sd = std::max(p[2], FLT_TINY);
    res[0] = mea - sd * coef;
    res[1] = mea;
    res[2] = sd;
    return 0;
}

// Per area: histogram (h_hist / h_edges with row pitch cap_bins + 1, h_nbins) ->
// (thresO2, mean, stddev) in h_res[a][0..2].  coef = norm.ppf(pfa) computed by the caller.
// status[a]: 0 ok, 1 = the histogram maximum is the first bin (the reference's argmin over an
// empty slice raises ValueError), 2 = fewer than 3 bins to fit.
extern "C" int origin_o2_threshold_batch(const double *h_hist, const double *h_edges,
                                         const long *h_nbins, int na, long cap_bins, double coef,
                                         double *h_res, int *h_status) {
  ORIGIN_CHECK_ARG(h_hist && h_edges && h_nbins && h_res && h_status && na >= 0, "bad arguments");
  origin_host_pool_run(na, [&](int a) {
    h_status[a] = o2_threshold_one(h_hist + (size_t)a * (cap_bins + 1),
                                   h_edges + (size_t)a * (cap_bins + 1), h_nbins[a], coef,
                                   h_res + (size_t)a * 3);
  });
  return ORIGIN_OK;
}

// Gather + clip + histogram + fit of every area in ONE pass over the worker pool: what
// ComputePCAThreshold.run does per area (steps.py:610-631, lib_origin.py:977-1024), from the O2
// map and the areas' spaxel lists.  h_map: float64 [S]; h_idx: concatenated flat spaxel indices,
// area a = [h_off[a], h_off[a+1]); h_data (out): the gathered O2 values in the same layout (the
// reference's testO2).  Histogram outputs as origin_o2_histogram_batch, results / status as
// origin_o2_threshold_batch (status 3: the histogram itself failed).
extern "C" int origin_o2_areas_fit(const double *h_map, const int *h_idx, const long *h_off, int na,
                                   double sigclip, int maxiters, double coef, double *h_data,
                                   double *h_hist, double *h_edges, long cap_bins, long *h_nbins,
                                   double *h_res, int *h_status) {
  ORIGIN_CHECK_ARG(h_map && h_idx && h_off && h_data && h_hist && h_edges && h_nbins && h_res &&
                       h_status && na >= 0,
                   "bad arguments");
  origin_host_pool_run(na, [&](int a) {
    const long n = h_off[a + 1] - h_off[a];
    double *d = h_data + h_off[a];
    const int *ix = h_idx + h_off[a];
    for (long i = 0; i < n; ++i) d[i] = h_map[ix[i]];
    double *hist = h_hist + (size_t)a * (cap_bins + 1);
    double *edges = h_edges + (size_t)a * (cap_bins + 1);
    long nk = 0;
    double *res = h_res + (size_t)a * 3;
    res[0] = res[1] = res[2] = NAN;
    if (origin_o2_histogram(d, n, sigclip, maxiters, hist, edges, cap_bins, h_nbins + a, &nk) !=
        ORIGIN_OK) {
      h_status[a] = 3;
      return;
    }
    h_status[a] = o2_threshold_one(hist, edges, h_nbins[a], coef, res);
  });
  return ORIGIN_OK;
}
