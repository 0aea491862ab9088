// Host-side preparation of the O2 threshold fit (reference
// muse_origin/lib_origin.py:999-1002): keep data > 0, sigma-clip (astropy defaults: median
// centre, population std, at most 5 iterations), Freedman-Diaconis histogram with
// density=True.  ~1e4 numbers per area, scalar work: it stays on the host by design, but as
// native code -- the NumPy version cost 0.55 ms per area (20 ms per 600x600 cube), more than
// the GPU spends on the DCT.  The Levenberg-Marquardt fit that follows is lmfit.hip.
//
// The arithmetic follows NumPy operation by operation (percentile 'linear' interpolation,
// linspace edges, the uniform-bin index formula of np.histogram with its edge corrections),
// tests/test_host_logic.py checks bit-equality with np.histogram(bins='fd', density=True).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace {

double median_of(std::vector<double> &w) {  // numpy.median: mean of the two middle values
  const size_t n = w.size();
  const size_t h = n / 2;
  std::nth_element(w.begin(), w.begin() + h, w.end());
  const double hi = w[h];
  if (n & 1) return hi;
  const double lo = *std::max_element(w.begin(), w.begin() + h);
  return (lo + hi) / 2.0;  // np.mean of two values
}

// numpy's pairwise summation (pairwise_sum_DOUBLE), so that mean / std match bit for bit
double pairwise_sum(const double *a, long n) {
  if (n < 8) {
    double r = 0.0;
    for (long i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    long i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  long n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

double std_of(const std::vector<double> &w, std::vector<double> &tmp) {  // np.std, ddof=0
  const long n = (long)w.size();
  const double mean = pairwise_sum(w.data(), n) / (double)n;
  tmp.resize(n);
  for (long i = 0; i < n; ++i) {
    const double d = w[i] - mean;
    tmp[i] = d * d;
  }
  return std::sqrt(pairwise_sum(tmp.data(), n) / (double)n);
}

double percentile_sorted_select(std::vector<double> &w, double q) {  // method='linear'
  const long n = (long)w.size();
  const double virt = (double)(n - 1) * (q / 100.0);
  long lo = (long)std::floor(virt);
  lo = std::max(0L, std::min(lo, n - 1));
  const long hi = std::min(lo + 1, n - 1);
  const double g = virt - (double)lo;
  std::nth_element(w.begin(), w.begin() + lo, w.end());
  const double a = w[lo];
  double b = a;
  if (hi != lo) b = *std::min_element(w.begin() + lo + 1, w.end());
  const double diff = b - a;
  double r = a + diff * g;  // numpy _lerp
  if (g >= 0.5) r = b - diff * (1.0 - g);
  if (diff == 0.0) r = a;
  return r;
}

}  // namespace

extern "C" int origin_o2_histogram(const double *h_data, long n, double sigclip, int maxiters,
                                   double *h_hist, double *h_edges, long cap_bins, long *nbins,
                                   long *nkept) {
  ORIGIN_CHECK_ARG(h_data && h_hist && h_edges && nbins && n >= 0 && cap_bins >= 1,
                   "bad arguments");
  // data = data[data > 0]                                            (lib_origin.py:999)
  std::vector<double> d;
  d.reserve(n);
  for (long i = 0; i < n; ++i)
    if (h_data[i] > 0) d.push_back(h_data[i]);
  // sigma_clip(data, sigclip).compressed()                           (lib_origin.py:1000-1001)
  std::vector<double> filt;
  filt.reserve(d.size());
  for (double v : d)
    if (std::isfinite(v)) filt.push_back(v);
  double lo = -INFINITY, hi = INFINITY;
  std::vector<double> work, tmp;
  long changed = 1;
  for (int it = 0; changed != 0 && it < maxiters && !filt.empty(); ++it) {
    const long size = (long)filt.size();
    work = filt;
    const double cen = median_of(work);
    const double sd = std_of(filt, tmp);
    lo = cen - sd * sigclip;
    hi = cen + sd * sigclip;
    size_t k = 0;
    for (size_t i = 0; i < filt.size(); ++i)
      if (filt[i] >= lo && filt[i] <= hi) filt[k++] = filt[i];
    filt.resize(k);
    changed = size - (long)k;
  }
  std::vector<double> x;
  x.reserve(d.size());
  for (double v : d)
    if (v >= lo && v <= hi) x.push_back(v);
  const long m = (long)x.size();
  if (nkept) *nkept = m;
  ORIGIN_CHECK_ARG(m > 0, "no positive O2 value left after sigma clipping");
  // np.histogram(data, bins='fd', density=True)                      (lib_origin.py:1002)
  double first = x[0], last = x[0];
  for (double v : x) first = std::min(first, v), last = std::max(last, v);
  if (first == last) {  // numpy widens a degenerate range by +-0.5
    first -= 0.5;
    last += 0.5;
  }
  work = x;
  const double p75 = percentile_sorted_select(work, 75.0);
  work = x;
  const double p25 = percentile_sorted_select(work, 25.0);
  const double width = 2.0 * (p75 - p25) * std::pow((double)m, -1.0 / 3.0);
  long nb = 1;
  if (width > 0) nb = (long)std::ceil((last - first) / width);
  if (nb < 1) nb = 1;
  *nbins = nb;
  if (nb + 1 > cap_bins + 1) {
    origin_set_error("histogram needs %ld bins, buffer holds %ld", nb, cap_bins);
    return ORIGIN_E_ARG;
  }
  // np.linspace(first, last, nb + 1)
  const double step = (last - first) / (double)nb;
  for (long i = 0; i <= nb; ++i) h_edges[i] = (double)i * step + first;
  h_edges[nb] = last;
  std::vector<long> cnt(nb, 0);
  const double denom = last - first;
  for (double v : x) {
    long idx = (long)(((v - first) / denom) * (double)nb);
    if (idx == nb) idx -= 1;
    if (v < h_edges[idx]) idx -= 1;
    else if (v >= h_edges[idx + 1] && idx != nb - 1) idx += 1;
    cnt[idx] += 1;
  }
  for (long i = 0; i < nb; ++i)
    h_hist[i] = (double)cnt[i] / (h_edges[i + 1] - h_edges[i]) / (double)m;
  return ORIGIN_OK;
}

// The same for `na` areas at once on a few host threads (no GIL involved): area a reads
// h_data[off[a] .. off[a+1]) and writes hist / edges at a * (cap_bins + 1).
namespace {

// Worker threads that outlive the calls: creating 16-36 threads costs more (~20 us each) than
// the histograms of a 600 x 600 field take.  One batch at a time (the callers are serial); the
// calling thread works too.  Threads are detached and sleep on the condition variable between
// batches.
class HistPool {
 public:
  static HistPool &get() {
    static HistPool *pool = new HistPool();  // never destroyed: workers may be asleep at exit
    return *pool;
  }
  // runs task(i) for i in [0, n) on the pool and the caller; returns when all are done
  void run(int n, const std::function<void(int)> &task) {
    std::unique_lock<std::mutex> batch(batch_mutex_);  // one batch at a time
    {
      std::lock_guard<std::mutex> lk(m_);
      task_ = &task;
      n_ = n;
      next_.store(0);
      pending_ = n;
      ++generation_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return pending_ == 0; });
    task_ = nullptr;
  }

 private:
  HistPool() {
    const int hw = (int)std::thread::hardware_concurrency();
    const int nt = std::max(0, std::min(47, hw - 1));
    for (int t = 0; t < nt; ++t) std::thread([this] { loop(); }).detach();
  }
  void work() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= n_) break;
      (*task_)(i);
      std::lock_guard<std::mutex> lk(m_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
      }
      work();
    }
  }
  std::mutex m_, batch_mutex_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *task_ = nullptr;
  int n_ = 0, pending_ = 0;
  std::atomic<int> next_{0};
  unsigned long generation_ = 0;
};

}  // namespace

// the pool is shared with the Gaussian fits of lmfit.hip
void origin_host_pool_run(int n, const std::function<void(int)> &task) {
  HistPool::get().run(n, task);
}

extern "C" int origin_o2_histogram_batch(const double *h_data, const long *h_off, int na,
                                         double sigclip, int maxiters, double *h_hist,
                                         double *h_edges, long cap_bins, long *h_nbins) {
  ORIGIN_CHECK_ARG(h_data && h_off && h_hist && h_edges && h_nbins && na >= 0, "bad arguments");
  std::atomic<int> failed(0);
  HistPool::get().run(na, [&](int a) {
    long nk = 0;
    const int rc = origin_o2_histogram(h_data + h_off[a], h_off[a + 1] - h_off[a], sigclip,
                                       maxiters, h_hist + (size_t)a * (cap_bins + 1),
                                       h_edges + (size_t)a * (cap_bins + 1), cap_bins,
                                       h_nbins + a, &nk);
    if (rc != ORIGIN_OK) failed.store(rc);
  });
  if (failed.load() != 0) {
    origin_set_error("origin_o2_histogram failed for at least one area (empty or too many bins)");
    return failed.load();
  }
  return ORIGIN_OK;
}
