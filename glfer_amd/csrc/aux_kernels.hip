// aux_kernels.hip -- the per-column statistics that follow the estimator in the reference:
//   K6  compute_floor            (fft.c:240-294)   -> floor_kernel
//   K5  update_avg_{plain,sumextreme,sumavg} (avg.c:108-298) -> avg_cum_kernel + avg_norm_kernel
// HBM-bound streaming/reduction kernels over PSD rows; no LDS tiling beyond the row itself.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace glfer {

// order-preserving map float -> uint32 (total order, handles -0 and negatives)
__device__ __forceinline__ uint32_t fkey(float v) {
  uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ---------------------------------------------------------------------------
// K6.  One 256-thread block per PSD row.  sig = max bin; peak = first arg max (strict >
// scan from 0.0, fft.c:284-291); floor = (sum of the `m` smallest bins)/0.05/bins where the
// reference sorts all bins (qsort, fft.c:265) and adds sorted[(int)(bins*0.95) ..].  Here the
// m-th smallest is found by a 4x8-bit radix select on the row held in LDS; no sort.
__global__ __launch_bounds__(256) void floor_kernel(const float *__restrict__ psd, int bins, int m,
                                                    float *__restrict__ stats) {
  extern __shared__ float row[];                 // bins floats, then 256 uint32 + scratch
  uint32_t *hist = reinterpret_cast<uint32_t *>(row + bins);
  __shared__ float red_v[256];
  __shared__ int red_i[256];
  __shared__ double red_d[256];
  __shared__ uint32_t sel_prefix, sel_need, wtot[4];

  const int tid = threadIdx.x;
  const float *src = psd + (size_t)blockIdx.x * bins;
  float best = 0.0f;
  int besti = 0;
  for (int i = tid; i < bins; i += 256) {
    const float v = src[i];
    row[i] = v;
    if (v > best) { best = v; besti = i; }       // strided scan keeps the lowest index per thread
  }
  red_v[tid] = best;
  red_i[tid] = besti;
  if (tid == 0) { sel_prefix = 0; sel_need = (uint32_t)m; }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      const float ov = red_v[tid + w];
      const int oi = red_i[tid + w];
      if (ov > red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])) { red_v[tid] = ov; red_i[tid] = oi; }
    }
    __syncthreads();
  }
  const float peak = red_v[0];
  const int peak_i = (peak > 0.0f) ? red_i[0] : 0;

  // radix select: key of the m-th smallest element
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = sel_prefix;
    const uint32_t mask_hi = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < bins; i += 256) {
      const uint32_t k = fkey(row[i]);
      if ((k & mask_hi) == prefix) atomicAdd(&hist[(k >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    // the digit d whose bucket holds the need-th smallest: inclusive scan of the 256 counts (wave
    // shuffles + 4 wave totals), the one thread with excl < need <= incl owns it
    const uint32_t need = sel_need, h = hist[tid];
    uint32_t incl = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
      if ((tid & 63) >= o) incl += up;
    }
    if ((tid & 63) == 63) wtot[tid >> 6] = incl;
    __syncthreads();
    for (int w = 0; w < (tid >> 6); w++) incl += wtot[w];
    if (incl - h < need && need <= incl) {
      sel_prefix = prefix | ((uint32_t)tid << shift);
      sel_need = need - (incl - h);
    }
    __syncthreads();
  }
  const uint32_t kth = sel_prefix;               // key of the m-th smallest
  const uint32_t ties = sel_need;                // how many copies of it belong to the m smallest

  double s = 0.0;
  float kth_val = 0.0f;
  for (int i = tid; i < bins; i += 256) {
    const float v = row[i];
    const uint32_t k = fkey(v);
    if (k < kth) s += (double)v;
    else if (k == kth) kth_val = v;
  }
  red_d[tid] = s;
  red_v[tid] = kth_val;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      red_d[tid] += red_d[tid + w];
      if (red_v[tid + w] != 0.0f) red_v[tid] = red_v[tid + w];
    }
    __syncthreads();
  }
  if (tid == 0) {
    float fl = (float)(red_d[0] + (double)ties * (double)red_v[0]);
    fl = (float)(fl / 0.05);                     // fft.c:274 (float / double)
    fl = fl / (float)bins;                       // fft.c:276
    float *o = stats + (size_t)blockIdx.x * 4;
    o[0] = peak;                                 // *sig_pwr_p = tmp_buf[0] = largest bin (fft.c:279)
    o[1] = fl;
    o[2] = (peak > 0.0f) ? peak : 0.0f;          // *peak_pwr_p
    o[3] = (float)peak_i;
  }
}

// ---------------------------------------------------------------------------
// K5a.  The sliding sum of avg.c:116-127 per bin, as the same double recurrence
//   f <  depth : cum += psd[f]                      (shift register still filling)
//   f >= depth : cum += psd[f] - psd[f-depth]       (avgarray[index][0] is the row f-depth)
// One thread per in-band bin, frames in order; rows are read coalesced across bins and eight rows
// are kept in flight to cover latency.  cum is written into the avg rows.
// The recurrence is sequential in f, and one thread per bin is only ~2000 threads, so the frames
// are cut into chunks of AVG_CHUNK that run in parallel: a chunk starts from the sum of the
// (up to depth) rows before it, added in frame order.  That is the value the recurrence holds at
// that frame whenever its additions are exact -- float terms in a double accumulator: always,
// unless a bin's values within depth frames span more than ~2^26 (78 dB); beyond that the two differ
// in the last bits of a double (the recurrence carries its own rounding history, the restart does
// not).  131 072 rows: 2.1 M rows/s as one chain per bin, HBM-bound in chunks.
constexpr int AVG_CHUNK = 128;
__global__ __launch_bounds__(256) void avg_cum_kernel(const float *__restrict__ psd, long long nframes,
                                                      int bins, int n_out, int depth, int minbin,
                                                      int maxbin, double *__restrict__ avg) {
  const int b = minbin + blockIdx.x * 256 + threadIdx.x;
  if (b >= maxbin) return;
  const long long f0 = (long long)blockIdx.y * AVG_CHUNK;
  const long long f1 = f0 + AVG_CHUNK < nframes ? f0 + AVG_CHUNK : nframes;
  double cum = 0.0;
  for (long long g = f0 > depth ? f0 - depth : 0; g < f0; g++) cum += (double)psd[(size_t)g * bins + b];
  long long f = f0;
  for (; f + 8 <= f1; f += 8) {
    float v[8], old[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      v[u] = psd[(size_t)(f + u) * bins + b];
      old[u] = (f + u >= depth) ? psd[(size_t)(f + u - depth) * bins + b] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (f + u < depth) cum += (double)v[u];
      else cum += (double)v[u] - (double)old[u];
      avg[(size_t)(f + u) * n_out + b] = cum;
    }
  }
  for (; f < f1; f++) {
    const float v = psd[(size_t)f * bins + b];
    if (f < depth) cum += (double)v;
    else cum += (double)v - (double)psd[(size_t)(f - depth) * bins + b];
    avg[(size_t)f * n_out + b] = cum;
  }
}

// K5b.  Per-frame reductions over the band and the output normalisation of the three
// modes (avg.c:129-156, 185-215, 248-294).  One block per frame, in place on the avg row.
// mode: 1 sumavg, 2 plain, 3 sumextreme (glfer.h:56-58).
__global__ __launch_bounds__(256) void avg_norm_kernel(const float *__restrict__ psd, int bins, int n_out,
                                                       int depth, int minbin, int maxbin, int mode,
                                                       int max0, double *__restrict__ avg,
                                                       double *__restrict__ ret) {
  __shared__ double r_sum[256], r_max[256], r_min[256], r_var[256];
  __shared__ int r_idx[256], r_cnt[256];
  const long long f = blockIdx.x;
  const int tid = threadIdx.x;
  double *row = avg + (size_t)f * n_out;
  const int eff = (f + 1 < depth) ? (int)(f + 1) : depth;   // effdepth after this frame
  const double init = (double)psd[(size_t)f * bins + minbin];

  double s = 0.0, mx = -1.0e300, mn = 1.0e300;
  int mi = 0x7fffffff;
  for (int b = minbin + tid; b < maxbin; b += 256) {
    const double c = row[b];
    s += c;
    if (c > mx) { mx = c; mi = b; }
    if (c < mn) mn = c;
  }
  r_sum[tid] = s; r_max[tid] = mx; r_min[tid] = mn; r_idx[tid] = mi;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      r_sum[tid] += r_sum[tid + w];
      if (r_max[tid + w] > r_max[tid] || (r_max[tid + w] == r_max[tid] && r_idx[tid + w] < r_idx[tid])) {
        r_max[tid] = r_max[tid + w];
        r_idx[tid] = r_idx[tid + w];
      }
      if (r_min[tid + w] < r_min[tid]) r_min[tid] = r_min[tid + w];
    }
    __syncthreads();
  }
  // running max starts at psd[minbin] (avg.c:111,163,224) and only a strictly larger sum moves it
  double top = init;
  int peak = -1;
  if (r_max[0] > init) { top = r_max[0]; peak = r_idx[0]; }
  const double low = (r_min[0] < 1.0) ? r_min[0] : 1.0;        // avg.c:165,192-193
  const double span = (double)(maxbin - minbin - 1);
  double spec;
  if (mode == 2) spec = (r_sum[0] - top) / (span * (double)(eff + 1));   // avg.c:147
  else spec = (r_sum[0] - top) / span;                                    // avg.c:199,260
  __syncthreads();

  double var = 0.0;
  int cnt = 0;
  for (int b = tid; b < n_out; b += 256) {
    double out;
    if (b < minbin || b >= maxbin) {
      out = 1e-15;
    } else {
      const double c = row[b];
      if (mode == 2) {
        out = c / (double)(eff + 1);                                      // avg.c:155
      } else if (mode == 3) {
        out = max0 ? (c - low) / (top - low) : c / spec;                  // avg.c:209-212
      } else {
        if (c - spec > 0) {                                               // avg.c:272-284
          out = max0 ? (c - spec) / (top - spec) : c / spec;
          if (b != peak) { var += (c / spec) * (c / spec); cnt++; }
        } else {
          out = 1e-15;
        }
      }
    }
    row[b] = out;
  }
  r_var[tid] = var; r_cnt[tid] = cnt;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) { r_var[tid] += r_var[tid + w]; r_cnt[tid] += r_cnt[tid + w]; }
    __syncthreads();
  }
  if (tid == 0) {
    double *o = ret + (size_t)f * 4;
    o[0] = (mode == 2) ? spec : top / spec;                               // avg.c:158,218,297
    o[1] = (double)peak;
    o[2] = (mode == 1) ? r_var[0] / (double)r_cnt[0] : 0.0;               // avg.c:294
    o[3] = (double)eff;
  }
}

}  // namespace glfer

using namespace glfer;

extern "C" hipError_t glfer_launch_floor(const float *psd, size_t nframes, int bins, int m, float *stats,
                                         hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  const size_t shmem = (size_t)bins * sizeof(float) + 256 * sizeof(uint32_t);
  hipLaunchKernelGGL(floor_kernel, dim3((unsigned)nframes), dim3(256), shmem, st, psd, bins, m, stats);
  return hipGetLastError();
}

extern "C" hipError_t glfer_launch_avg(int mode, const float *psd, size_t nframes, int bins, int n_out,
                                       int depth, int minbin, int maxbin, int max0, double *avg,
                                       double *ret, hipStream_t st) {
  if (nframes == 0) return hipSuccess;
  const int band = maxbin - minbin;
  hipLaunchKernelGGL(avg_cum_kernel, dim3((unsigned)((band + 255) / 256), (unsigned)((nframes + AVG_CHUNK - 1) / AVG_CHUNK)), dim3(256), 0, st, psd,
                     (long long)nframes, bins, n_out, depth, minbin, maxbin, avg);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(avg_norm_kernel, dim3((unsigned)nframes), dim3(256), 0, st, psd, bins, n_out, depth,
                     minbin, maxbin, mode, max0, avg, ret);
  return hipGetLastError();
}
