// host_tables.cpp -- once-per-configuration tables, computed on the host in double.
//
// Product code (not the oracle): the engine's own implementation of what the reference
// does inside fft_init()/mtm_init():
//   * the eight analysis windows, unit power            (compute_window, fft.c:309-360)
//   * the DPSS tapers and eigenvalues by the 32-point Gauss-Legendre discretisation of
//     the sinc kernel, Thomson 1982 App. A               (gl_dpss, g-l_dpss.c:288-347)
// They run once per plan; cost is O(32*N*T) sines (milliseconds), so they stay on the CPU.
#include "host_tables.h"
#include <mutex>
#include <cstring>
#include <cstdint>

#include <cmath>
#include <vector>

namespace glfer {

static const double kPiD = 3.14159265358979323846;

// modified Bessel I0 by the Abramowitz-Stegun 9.8.1 / 9.8.2 polynomials (util.c:222-237)
static double bessel_i0(double x) {
  const double ax = std::fabs(x);
  if (ax < 3.75) {
    double y = (x / 3.75) * (x / 3.75);
    static const double c[] = {1.0, 3.5156229, 3.0899424, 1.2067492, 0.2659732, 0.360768e-01, 0.45813e-02};
    double s = c[6];
    for (int i = 5; i >= 0; i--) s = c[i] + y * s;
    return s;
  }
  const double y = 3.75 / ax;
  static const double d[] = {0.39894228, 0.1328592e-01, 0.225319e-02, -0.157565e-02, 0.916281e-02,
                             -0.2057706e-01, 0.2635537e-01, -0.1647633e-01, 0.392377e-02};
  double s = d[8];
  for (int i = 7; i >= 0; i--) s = d[i] + y * s;
  return (std::exp(ax) / std::sqrt(ax)) * s;
}

// compute_window (fft.c:309-360): shape in double rounded to float, then /sqrt(sum w^2)
// with a float accumulator, exactly as the reference stores and normalises it.
void make_window(int type, int n, float *w) {
  const double nm1 = n - 1.0;
  for (int i = 0; i < n; i++) {
    const double ph = 2.0 * kPiD * i / nm1;
    const double u = (2.0 * i - n + 1.0);
    double v = 1.0;
    switch (type) {
      case 0: v = 0.5 - 0.5 * std::cos(ph); break;                                   // Hanning
      case 1: v = 0.42 - 0.5 * std::cos(ph) + 0.08 * std::cos(4.0 * kPiD * i / nm1); break;  // Blackman
      case 2: v = std::exp(-1.0f * u * u / (nm1 * nm1)); break;                      // Gaussian, alpha = 1
      case 3: v = 1.0 - (u / nm1) * (u / nm1); break;                                // Welch
      case 4: v = 1.0 - std::fabs(u / nm1); break;                                   // Bartlett
      case 6: v = 0.54 - 0.46 * std::cos(ph); break;                                 // Hamming
      case 7: {                                                                      // Kaiser, alpha = 6/t
        const float t = (float)(nm1 / 2.0);
        const float alpha = (float)(6.0 / t);
        const float d = (float)i - t;
        const float arg = t * t - d * d;
        v = bessel_i0(alpha * std::sqrt((double)arg)) / bessel_i0((double)(alpha * t));
        break;
      }
      default: v = 1.0;                                                              // Rectangular
    }
    w[i] = (float)v;
  }
  float pwr = 0.0f;
  for (int i = 0; i < n; i++) pwr += w[i] * w[i];
  const double root = std::sqrt((double)pwr);
  for (int i = 0; i < n; i++) w[i] = (float)(w[i] / root);
}

// ---- 32-point Gauss-Legendre rule on [-1,1] (values as tabulated at g-l_dpss.c:213-282)
static const int GL = 32;
static const double gl_node_half[16] = {
    .048307665687738316235, .144471961582796493485, .239287362252137074545, .331868602282127649780,
    .421351276130635345364, .506899908932229390024, .587715757240762329041, .663044266930215200975,
    .732182118740289680387, .794483795967942406963, .849367613732569970134, .896321155766052123965,
    .934906075937739689171, .964762255587506430774, .985611511545268335400, .997263861849481563545};
static const double gl_weight_half[16] = {
    .096540088514727800567, .095638720079274859419, .093844399080804565639, .091173878695763884713,
    .087652093004403811143, .083311924226946755222, .078193895787070306472, .072345794108848506225,
    .065822222776361846838, .058684093478535547145, .050998059262376176196, .042835898022226680657,
    .034273862913021433103, .025392065309262059456, .016274394730905670605, .007018610009470096600};
static inline double gl_x(int i) { return i < 16 ? -gl_node_half[15 - i] : gl_node_half[i - 16]; }
static inline double gl_w(int i) { return i < 16 ? gl_weight_half[15 - i] : gl_weight_half[i - 16]; }

// Cyclic Jacobi eigen-decomposition of a symmetric GLxGL matrix (row-major, upper triangle
// used, destroyed) -- the classical threshold scheme (Rutishauser; the same algorithm the
// reference takes from GSL, g-l_dpss.c:84-196) followed step for step: the leading DPSS
// eigenvalues are degenerate to ~1e-9, so a different rotation order would turn the tapers
// inside that subspace at the 1e-8 level.  Eigenvectors are the columns of V.
static bool sym_eig_jacobi(std::vector<double> &A, std::vector<double> &lam, std::vector<double> &V) {
  auto rot = [](std::vector<double> &M, int r0, int c0, int r1, int c1, double s, double tau) {
    const double g = M[r0 * GL + c0], h = M[r1 * GL + c1];
    M[r0 * GL + c0] = g - s * (h + g * tau);
    M[r1 * GL + c1] = h + s * (g - h * tau);
  };
  V.assign(GL * GL, 0.0);
  lam.assign(GL, 0.0);
  std::vector<double> base(GL), delta(GL, 0.0);
  for (int i = 0; i < GL; i++) {
    V[i * GL + i] = 1.0;
    base[i] = lam[i] = A[i * GL + i];
  }
  for (int sweep = 1; sweep <= 1000; sweep++) {
    double off = 0.0;
    for (int p = 0; p < GL - 1; p++)
      for (int q = p + 1; q < GL; q++) off += std::fabs(A[p * GL + q]);
    if (off == 0.0) return true;
    const double thresh = sweep < 4 ? 0.2 * off / (GL * GL) : 0.0;
    for (int p = 0; p < GL - 1; p++) {
      for (int q = p + 1; q < GL; q++) {
        const double dp = lam[p], dq = lam[q], apq = A[p * GL + q];
        const double g = 100.0 * std::fabs(apq);
        if (sweep > 4 && std::fabs(dp) + g == std::fabs(dp) && std::fabs(dq) + g == std::fabs(dq)) {
          A[p * GL + q] = 0.0;
          continue;
        }
        if (!(std::fabs(apq) > thresh)) continue;
        double h = dq - dp, t;
        if (std::fabs(h) + g == std::fabs(h)) {
          t = apq / h;
        } else {
          const double theta = 0.5 * h / apq;
          t = 1.0 / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
          if (theta < 0.0) t = -t;
        }
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c, tau = s / (1.0 + c);
        h = t * apq;
        delta[p] -= h;
        delta[q] += h;
        lam[p] = dp - h;
        lam[q] = dq + h;
        A[p * GL + q] = 0.0;
        for (int j = 0; j < p; j++) rot(A, j, p, j, q, s, tau);
        for (int j = p + 1; j < q; j++) rot(A, p, j, j, q, s, tau);
        for (int j = q + 1; j < GL; j++) rot(A, p, j, q, j, s, tau);
        for (int j = 0; j < GL; j++) rot(V, j, p, j, q, s, tau);
      }
    }
    for (int i = 0; i < GL; i++) {
      base[i] += delta[i];
      delta[i] = 0.0;
      lam[i] = base[i];
    }
  }
  return false;
}

// gl_dpss (g-l_dpss.c:288-347).  nw = N*W.  tapers: [kmax+1][n], unit energy; sig = lambda-1.
bool make_dpss(int n, int kmax, double nw, double *tapers, double *sig) {
  const double c = kPiD * nw;
  std::vector<double> K(GL * GL), lam, V;
  for (int i = 0; i < GL; i++)
    for (int j = 0; j < GL; j++) {
      const double d = gl_x(i) - gl_x(j);
      const double k = (i == j) ? c / kPiD : std::sin(c * d) / (kPiD * d);
      K[i * GL + j] = k * std::sqrt(gl_w(i) * gl_w(j));
    }
  if (!sym_eig_jacobi(K, lam, V)) return false;

  // order by |lambda| descending: selection sort with strict > (eigen_symmv_sort, g-l_dpss.c:35-72)
  std::vector<int> ord(GL);
  for (int i = 0; i < GL; i++) ord[i] = i;
  for (int i = 0; i < GL - 1; i++) {
    int best = i;
    for (int j = i + 1; j < GL; j++)
      if (std::fabs(lam[ord[j]]) > std::fabs(lam[ord[best]])) best = j;
    if (best != i) { int t = ord[i]; ord[i] = ord[best]; ord[best] = t; }
  }

  std::vector<double> sw(GL);
  for (int j = 0; j < GL; j++) sw[j] = std::sqrt(gl_w(j));
  for (int k = 0; k <= kmax; k++) {
    const int col = ord[k];
    double *v = tapers + (size_t)k * n;
    for (int i = 0; i < n; i++) {                  // interpolate the eigenfunction to n points
      double acc = 0.0;
      for (int j = 0; j < GL; j++) {
        const double arg = (2.0 * (i + 0.5) / n) - 1.0 - gl_x(j);
        acc += sw[j] * V[j * GL + col] * std::sin(c * arg) / (kPiD * arg);
      }
      v[i] = acc;
    }
    double energy = 0.0;
    for (int i = 0; i < n; i++) energy += v[i] * v[i];
    for (int i = 0; i < n; i++) v[i] /= std::sqrt(energy);
    sig[k] = lam[col] - 1.0;
  }
  return true;
}

// inter-pass twiddles W_N^(t*k1), laid out [k1][t] for coalesced loads (t = lane).
void make_twiddles(int n, int lanes, float *tw_re_im) {
  for (int k1 = 0; k1 < 64; k1++)
    for (int t = 0; t < lanes; t++) {
      const double ang = -2.0 * kPiD * (double)((long long)t * k1 % n) / (double)n;
      tw_re_im[2 * (k1 * lanes + t) + 0] = (float)std::cos(ang);
      tw_re_im[2 * (k1 * lanes + t) + 1] = (float)std::sin(ang);
    }
}

// Radix schedule of spectro16.hip (Plan16): 16, 16, then N/256 (N <= 4096) or 16, N/4096.
int plan16_passes(int logn, int radix[4]) {
  const int n = 1 << logn;
  int np = logn <= 8 ? 2 : (logn <= 12 ? 3 : 4);
  radix[0] = radix[1] = 16;
  if (np == 3) radix[2] = n / 256;
  if (np == 4) { radix[2] = 16; radix[3] = n / 4096; }
  return np;
}

// Per-lane inter-pass twiddles of the Stockham passes, slot-major [slot][T] (cos, sin):
// pass i >= 1, butterfly b, input q >= 1 -> W_(Ls*R)^(k*q) with k = (t + T*b) mod Ls.
int make_twiddles16(int logn, float *tw_re_im) {
  const int n = 1 << logn, T = n / 16;
  int radix[4];
  const int np = plan16_passes(logn, radix);
  int slot = 0, ls = radix[0];
  for (int i = 1; i < np; i++) {
    const int R = radix[i], B = 16 / R;
    for (int b = 0; b < B; b++)
      for (int q = 1; q < R; q++, slot++)
        for (int t = 0; t < T; t++) {
          const long long k = (t + (long long)T * b) % ls;
          const double ang = -2.0 * kPiD * (double)((k * q) % ((long long)ls * R)) / (double)((long long)ls * R);
          if (tw_re_im) {
            tw_re_im[2 * ((size_t)slot * T + t) + 0] = (float)std::cos(ang);
            tw_re_im[2 * ((size_t)slot * T + t) + 1] = (float)std::sin(ang);
          }
        }
    ls *= R;
  }
  return slot;
}

// ---------------------------------------------------------------------------
// set_palette (g_main.c:651-762).  Every palette of the reference is piecewise linear in the
// colour index c: channel = (unsigned char)(a*c + b) on [from, upto).  The reference converts
// the double straight to unsigned char although some segments run below 0 or past 255 (OTD at
// c = 0, BONE's red above c = 262/1.2, COPPER's red near c = 207); on x86-64 that conversion
// is "truncate to int32, keep the low byte", which byte_of() spells out.
namespace {
struct Lin { double a, b; };
struct Seg { int upto; Lin r, g, b; };
constexpr Lin K0{0.0, 0.0}, K255{0.0, 255.0};

const Seg kHsv[] = {{64, K0, {4.0, 0.0}, K255}, {128, K0, K255, {-4.0, 510.0}},
                    {192, {4.0, -510.0}, K255, K0}, {256, K255, {-4.0, 1020.0}, K0}};
const Seg kThresh[] = {{16, K0, K0, K0}, {64, K0, {4.0, 0.0}, K255}, {128, K0, K255, {-4.0, 510.0}},
                       {192, {4.0, -510.0}, K255, K0}, {256, K255, {-4.0, 1020.0}, K0}};
const Seg kCool[] = {{256, {1.0, 0.0}, {-1.0, 255.0}, K255}};
const Seg kHot[] = {{96, {2.66667, 0.5}, K0, K0}, {192, K255, {2.66667, -254.0}, K0},
                    {256, K255, K255, {4.0, -766.0}}};
const Seg kBw[] = {{256, {1.0, 0.0}, {1.0, 0.0}, {1.0, 0.0}}};
const Seg kBone[] = {{96, {0.88889, 0.0}, {0.88889, 0.0}, {1.2, 0.0}},
                     {192, {0.88889, 0.0}, {1.2, -29.0}, {0.88889, 29.0}},
                     {256, {1.2, -60.0}, {0.88889, 29.0}, {0.88889, 29.0}}};
const Seg kCopper[] = {{208, {1.23, 0.0}, {0.78, 0.0}, {0.5, 0.0}}, {256, K255, {0.78, 0.0}, {0.5, 0.0}}};
const Seg kOtd[] = {{128, K0, {2.0, -1.0}, {-2.0, 255.0}}, {256, {2.0, -255.0}, {-2.0, 511.0}, K0}};
const Seg *const kPalettes[] = {kHsv, kThresh, kCool, kHot, kBw, kBone, kCopper, kOtd};

inline unsigned char byte_of(Lin l, int c) {
  const double v = l.a * (double)c + l.b;
  return (unsigned char)(int)v;
}
}  // namespace

void make_palette(int palette, unsigned char tab[768]) {
  const Seg *seg = (palette >= 0 && palette < 8) ? kPalettes[palette] : kBw;   // else branch: black and white
  for (int c = 0; c < 256; c++) {
    while (c >= seg->upto) ++seg;
    tab[3 * c] = byte_of(seg->r, c);
    tab[3 * c + 1] = byte_of(seg->g, c);
    tab[3 * c + 2] = byte_of(seg->b, c);
  }
}

// Tables of the harmonic F-test as mtm_init builds them: U0[j] = sum_i v_j[i] in double
// (mtm.c:76-83), sum_U0_sqr and hn[i] = sum_j U0[j] v_j[i] / sum_U0_sqr with FLOAT accumulators
// (mtm.c:56-60 declares them float, mtm.c:124-136).  tapers: [kmax+1][n].
void make_ftest_tables(int n, int kmax, const double *tapers, double *U0, float *hn, float *sum_U0_sqr) {
  for (int j = 0; j <= kmax; j++) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += tapers[(size_t)j * n + i];
    U0[j] = s;
  }
  float total = 0.0f;
  for (int j = 0; j <= kmax; j++) total += U0[j] * U0[j];            // float += double
  for (int i = 0; i < n; i++) {
    float h = 0.0f;
    for (int j = 0; j <= kmax; j++) h += U0[j] * tapers[(size_t)j * n + i];
    hn[i] = h / total;
  }
  *sum_U0_sqr = total;
}

// levbuf = (short)(10.0 * log10(x)) (g_main.c:1192-1196) truncates a double: which integer comes out is
// a comparison of x with the point where the reference's own expression -- this libm's log10, this
// multiplication -- crosses an integer.  thr[K + k], k = 1..K: the smallest double x with
// 10.0*log10(x) >= k; k = -K..-1: the largest double x with 10.0*log10(x) <= k (the conversion
// truncates towards zero).  Found by bisection over the bit patterns (log10 is monotonic there).
const double *log_thresholds() {
  static double thr[2 * kLogThrK + 1];
  static std::once_flag once;
  std::call_once(once, [] {
    auto bits = [](double d) { uint64_t u; memcpy(&u, &d, 8); return u; };
    auto dbl = [](uint64_t u) { double d; memcpy(&d, &u, 8); return d; };
    thr[kLogThrK] = 1.0;
    for (int k = 1; k <= kLogThrK; k++) {
      // first x with 10 log10 x >= k, between 10^((k-1)/10) and 10^((k+1)/10)
      uint64_t lo = bits(pow(10.0, (k - 1) / 10.0)), hi = bits(pow(10.0, (k + 1) / 10.0));   // f(lo) false, f(hi) true
      while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (10.0 * log10(dbl(mid)) >= (double)k) hi = mid; else lo = mid;
      }
      thr[kLogThrK + k] = dbl(hi);
      // last x with 10 log10 x <= -k
      lo = bits(pow(10.0, (-k - 1) / 10.0));                                                  // f(lo) true
      hi = bits(pow(10.0, (-k + 1) / 10.0));                                                  // f(hi) false
      while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (10.0 * log10(dbl(mid)) <= (double)-k) lo = mid; else hi = mid;
      }
      thr[kLogThrK - k] = dbl(lo);
    }
  });
  return thr;
}

}  // namespace glfer
