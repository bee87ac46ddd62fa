// siren_kmeans.hip — 1-D k-means weight quantisation on the device, no host synchronisation.
//
// Reference being replaced (paths relative to the reference tree):
//   implicit_image/pipeline/quant/kmeans.py:110-150      find_centroids: <= 5 Lloyd iterations over the NON-ZERO weights from a
//                                                        linspace(min, max, 2^bits - 1) guess, 0 prepended, torch.unique,
//                                                        sorted by |c|, labels = argmin of the squared distance (first index
//                                                        wins ties) over ALL weights
//   implicit_image/pipeline/quant/kmeans_helper.py:59-98 kmeans_fit: labels = argmin (x - c)^2; centres = scatter_mean (empty
//                                                        bins -> 0); stop when (sum_k |dc_k|)^2 < 1e-4
// It runs on every forward of the quantise phase, per quantised layer.  The host mirror did it with torch ops that
// materialise an [n, K, 1] distance tensor and synchronise (`.item()`, torch.unique) several times per Lloyd iteration.
// Here: five kernels on the handle's stream, the early exit is a device flag the later iterations test.
//   * distances are the reference's float expression, (x - c) * (x - c), compared with `<` in index order: same labels;
//   * the centroid update is a segmented sum in 64-bit FIXED POINT (integer adds commute: the result does not depend on the
//     order in which workgroups arrive, unlike float atomics), divided in double and rounded once to float - the exact mean,
//     where torch_scatter's sequential float sum carries ~1e-7 of rounding: centroid VALUES stay "parity unpinned" at that
//     third-party boundary, labels are pinned by golden vectors (tests/golden/kmeans_*.npz).
// (included by siren_fit.hip)

namespace sf {

constexpr int kKmMaxK = 512;      // 2^bits <= 512 (the reference's slurm scripts go to bits = 9)

struct KmWs {                     // device workspace of a handle (zeroed at allocation)
  long long sums[kKmMaxK];
  int counts[kKmMaxK];
  int done;                       // 1 once (sum |dc|)^2 < tol: later iterations leave the centres alone
  int n_cent;                     // centroids after {0} U centres -> unique
  double scale;                   // fixed-point unit of the segment sums: 2^e with n * max|w| * 2^e < 2^61
  float sorted[kKmMaxK + 1];
};

// one thread: fixed-point scale from the initial guess (linspace(min, max): its ends bound every weight), state cleared
__global__ void k_km_init(const float* centers, int K, long n, KmWs* ws) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float m = fmaxf(fabsf(centers[0]), fabsf(centers[K - 1]));
    int e = 0;
    frexpf(m > 0.f ? m : 1.f, &e);                               // m < 2^e
    int ln = 0;
    while ((1L << ln) < n + 1) ++ln;                             // n < 2^ln
    ws->scale = ldexp(1.0, 60 - ln - e);
    ws->done = 0;
  }
  for (int i = threadIdx.x; i < kKmMaxK; i += blockDim.x) { ws->sums[i] = 0; ws->counts[i] = 0; }
}

// labels of the non-zero weights against the current centres + fixed-point segment sums
__global__ __launch_bounds__(256) void k_km_assign(const float* __restrict__ w, long n, const float* __restrict__ centers, int K,
                                                   KmWs* ws) {
  if (ws->done) return;
  const double scale = ws->scale;
  __shared__ float sc[kKmMaxK];
  __shared__ unsigned long long ssum[kKmMaxK];
  __shared__ int scnt[kKmMaxK];
  for (int i = threadIdx.x; i < K; i += blockDim.x) { sc[i] = centers[i]; ssum[i] = 0ull; scnt[i] = 0; }
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = w[i];
    if (x == 0.0f) continue;                                     // kmeans.py:123 (zeros are excluded from the fit)
    int best = 0;
    float bd = __fmul_rn(__fsub_rn(x, sc[0]), __fsub_rn(x, sc[0]));
    for (int k = 1; k < K; ++k) {
      const float d = __fsub_rn(x, sc[k]);
      const float dd = __fmul_rn(d, d);
      if (dd < bd) { bd = dd; best = k; }                        // strict: the first index wins ties (torch.argmin)
    }
    atomicAdd(&ssum[best], (unsigned long long)(long long)llrint((double)x * scale));
    atomicAdd(&scnt[best], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    if (scnt[i]) {
      atomicAdd(reinterpret_cast<unsigned long long*>(&ws->sums[i]), ssum[i]);
      atomicAdd(&ws->counts[i], scnt[i]);
    }
  }
}

// new centres = segment means (empty bins -> 0, as scatter_mean), convergence test, accumulators cleared
__global__ __launch_bounds__(kKmMaxK) void k_km_update(float* centers, int K, KmWs* ws, float tol) {
  const double inv_scale = 1.0 / ws->scale;
  __shared__ float sh[kKmMaxK];
  const int t = threadIdx.x;
  const bool done = ws->done != 0;
  float shift = 0.f;
  if (t < K && !done) {
    const int c = ws->counts[t];
    const float old = centers[t];
    const float nw = c > 0 ? (float)((double)ws->sums[t] * inv_scale / (double)c) : 0.0f;
    shift = sqrtf((old - nw) * (old - nw));                      // kmeans_helper.py:91-93, one feature
    centers[t] = nw;
  }
  if (t < kKmMaxK) { ws->sums[t < kKmMaxK ? t : 0] = 0; ws->counts[t < kKmMaxK ? t : 0] = 0; }
  sh[t] = shift;
  __syncthreads();
  if (t == 0 && !done) {
    float s = 0.f;
    for (int i = 0; i < K; ++i) s += sh[i];                      // fixed order
    if (s * s < tol) ws->done = 1;
  }
}

// {0} U centres -> unique (ascending) -> ordered by |c| (ties: the smaller value first) -> centroids_out (zero padded)
__global__ __launch_bounds__(kKmMaxK) void k_km_finish(const float* centers, int K, KmWs* ws, float* cent_out, int n_out, int* n_cent_out) {
  __shared__ float v[kKmMaxK + 1];
  __shared__ float u[kKmMaxK + 1];
  __shared__ int nu;
  const int t = threadIdx.x, m = K + 1;
  for (int i = t; i < m; i += blockDim.x) v[i] = i == 0 ? 0.0f : centers[i - 1];
  __syncthreads();
  // rank sort ascending (distinct positions for equal values by index), m <= 513
  for (int i = t; i < m; i += blockDim.x) {
    const float x = v[i];
    int r = 0;
    for (int j = 0; j < m; ++j) r += (v[j] < x) || (v[j] == x && j < i);
    ws->sorted[r] = x;
  }
  __syncthreads();
  if (t == 0) {                                                  // unique of a sorted list (-0.0 == 0.0 merge, as torch.unique)
    int c = 0;
    for (int i = 0; i < m; ++i) {
      const float x = ws->sorted[i];
      if (c == 0 || !(x == u[c - 1])) u[c++] = x;
    }
    nu = c;
  }
  __syncthreads();
  const int c = nu;
  for (int i = t; i < c; i += blockDim.x) {                      // order by |c| (stable in the ascending-value order)
    const float x = fabsf(u[i]);
    int r = 0;
    for (int j = 0; j < c; ++j) r += (fabsf(u[j]) < x) || (fabsf(u[j]) == x && j < i);
    cent_out[r] = u[i];
  }
  for (int i = c + t; i < n_out; i += blockDim.x) cent_out[i] = 0.0f;
  if (t == 0) { ws->n_cent = c; if (n_cent_out) *n_cent_out = c; ws->done = 0; }
}

// labels of ALL weights against the final centroids (+ the codebook weights), kmeans.py:139-146
__global__ __launch_bounds__(256) void k_km_predict(const float* __restrict__ w, long n, const float* __restrict__ cent, const KmWs* ws,
                                                    long long* labels, float* new_w) {
  __shared__ float sc[kKmMaxK + 1];
  const int c = ws->n_cent;
  for (int i = threadIdx.x; i < c; i += blockDim.x) sc[i] = cent[i];
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = w[i];
    int best = 0;
    float bd = __fmul_rn(__fsub_rn(x, sc[0]), __fsub_rn(x, sc[0]));
    for (int k = 1; k < c; ++k) {
      const float d = __fsub_rn(x, sc[k]);
      const float dd = __fmul_rn(d, d);
      if (dd < bd) { bd = dd; best = k; }
    }
    if (labels) labels[i] = best;
    if (new_w) new_w[i] = sc[best];
  }
}

}  // namespace sf
