/* glfer_oracle.c -- CPU restatement of glfer's spectral-estimation hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see glfer_oracle.h).  Plain C, libm only.
 * Build with -O2 -ffp-contract=off (oracle/Makefile) so the float/double
 * operation order written here is the order executed.
 */
#include "glfer_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ */
/* util.c:222-237  bessel_I0 -- polynomial approximations (A&S 9.8.1/2) */
double go_bessel_i0(double x)
{
  double ax = fabs(x);
  if (ax < 3.75) {
    double y = x / 3.75;
    y *= y;
    return 1.0 + y * (3.5156229 + y * (3.0899424 + y * (1.2067492
           + y * (0.2659732 + y * (0.360768e-01 + y * 0.45813e-02)))));
  } else {
    double y = 3.75 / ax;
    return (exp(ax) / sqrt(ax)) * (0.39894228 + y * (0.1328592e-01
           + y * (0.225319e-02 + y * (-0.157565e-02 + y * (0.916281e-02
           + y * (-0.2057706e-01 + y * (0.2635537e-01 + y * (-0.1647633e-01
           + y * 0.392377e-02))))))));
  }
}

/* ------------------------------------------------------------------ */
/* fft.c:309-360 compute_window */
void go_window(int window_type, int n, float *w)
{
  const double nm1 = n - 1.0;
  for (int i = 0; i < n; i++) {
    switch (window_type) {
    case GO_WIN_HANNING:                                   /* fft.c:321-323 */
      w[i] = 0.5 - 0.5 * cos(2.0 * M_PI * i / nm1);
      break;
    case GO_WIN_BLACKMAN:                                  /* fft.c:324-326 */
      w[i] = 0.42 - 0.5 * cos(2.0 * M_PI * i / nm1) + 0.08 * cos(4.0 * M_PI * i / nm1);
      break;
    case GO_WIN_GAUSSIAN: {                                /* fft.c:327-330 */
      float alpha = 1.0f;
      double c = 2.0 * i - n + 1.0;
      w[i] = exp(-alpha * c * c / (nm1 * nm1));
      break;
    }
    case GO_WIN_WELCH: {                                   /* fft.c:331-333 */
      double c = (2.0 * i - n + 1.0) / nm1;
      w[i] = 1.0 - c * c;
      break;
    }
    case GO_WIN_BARTLETT:                                  /* fft.c:334-336 */
      w[i] = 1.0 - fabs((2.0 * i - n + 1.0) / nm1);
      break;
    case GO_WIN_HAMMING:                                   /* fft.c:340-342 */
      w[i] = 0.54 - 0.46 * cos(2.0 * M_PI * i / nm1);
      break;
    case GO_WIN_KAISER: {                                  /* fft.c:343-347 */
      float t = nm1 / 2.0;            /* float, as in the reference */
      float alpha = 6.0 / t;
      float d = i - t;
      float arg = t * t - d * d;      /* float arithmetic */
      w[i] = go_bessel_i0(alpha * sqrt(arg)) / go_bessel_i0(alpha * t);
      break;
    }
    case GO_WIN_RECTANGULAR:                               /* fft.c:337-339 */
    default:                                               /* fft.c:348-349 */
      w[i] = 1.0;
    }
  }
  /* fft.c:352-359: unit power, accumulator is a float */
  float pwr = 0.0f;
  for (int i = 0; i < n; i++)
    pwr += w[i] * w[i];
  for (int i = 0; i < n; i++)
    w[i] /= sqrt(pwr);
}

/* fft.c:70 */
int go_hop(int n, float overlap)
{
  return (int)(n * (1.0 - overlap));
}

void go_fft_state_init(go_fft_state *st, int n, float overlap, int window_type,
                       float a, int limiter, int sub_mean)
{
  st->n = n;
  st->overlap = overlap;
  st->window_type = window_type;
  st->a = a;
  st->limiter = limiter;
  st->sub_mean = sub_mean;
  st->window = (float *)malloc((size_t)n * sizeof(float));
  st->inbuf_audio = (float *)calloc((size_t)n, sizeof(float));   /* fft.c:178 */
  st->inbuf_fft = (float *)calloc((size_t)n, sizeof(float));     /* fft.c:179 */
  go_window(window_type, n, st->window);                          /* fft.c:184-185 */
}

void go_fft_state_free(go_fft_state *st)
{
  free(st->window);
  free(st->inbuf_audio);
  free(st->inbuf_fft);
  st->window = st->inbuf_audio = st->inbuf_fft = NULL;
}

/* ------------------------------------------------------------------ */
/* fft.c:66-165 prepare_audio */
void go_prepare(go_fft_state *st, float *hop, int first_buffer)
{
  const int n = st->n;
  const int h = go_hop(n, st->overlap);
  const int keep = n - h;
  float *frame = st->inbuf_audio;
  float *out = st->inbuf_fft;

  if (st->sub_mean) {                                      /* fft.c:86-96 */
    float mean = 0.0f;
    for (int i = 0; i < h; i++)
      mean += hop[i];
    mean /= h;
    for (int i = 0; i < h; i++)
      hop[i] -= mean;
  }

  if (!first_buffer) {                                     /* fft.c:99-102 */
    for (int i = 0; i < keep; i++)
      frame[i] = frame[n - keep + i];
  } else {                                                 /* fft.c:103-108 */
    for (int i = 0; i < keep; i++)
      frame[i] = 0.0f;
  }
  for (int i = 0; i < h; i++)                              /* fft.c:111-113 */
    frame[keep + i] = hop[i];

  const int windowed = (st->window_type != GO_WIN_RECTANGULAR);
  if (st->a > 0.0) {                                       /* fft.c:127-136 */
    for (int i = 0; i < n; i++) {
      float x = frame[i];
      out[i] = x / (st->a + x * x);
      if (windowed)
        out[i] *= st->window[i];
    }
  } else if (windowed) {                                   /* fft.c:139-146 */
    for (int i = 0; i < n; i++)
      out[i] = st->window[i] * frame[i];
  } else {                                                 /* fft.c:147-148 */
    for (int i = 0; i < n; i++)
      out[i] = frame[i];
  }

  if (st->limiter == 1) {                                  /* fft.c:151-156 */
    for (int i = 0; i < n; i++) {
      float lg = log(fabs(out[i]));
      out[i] = (out[i] > 0 ? exp(lg * 0.1) : -exp(lg * 0.1));
    }
  }
}

/* ------------------------------------------------------------------ */
/* fft_radix2.c:27-44 */
static int ilog2_exact(size_t n)
{
  int lg = 0;
  size_t k = 1;
  while (k < n) {
    k <<= 1;
    lg++;
  }
  return (k == n) ? lg : -1;
}

/* fft_radix2.c:47-71 -- same permutation (bit reversal), written directly */
static void bit_reverse_permute(float *d, size_t n, int lg)
{
  for (size_t i = 0; i < n; i++) {
    size_t r = 0, v = i;
    for (int b = 0; b < lg; b++) {
      r = (r << 1) | (v & 1);
      v >>= 1;
    }
    if (i < r) {
      float t = d[i];
      d[i] = d[r];
      d[r] = t;
    }
  }
}

/* fft_radix2.c:75-177 fft_real_radix2_transform */
void go_rfft_halfcomplex(float *d, size_t n)
{
  if (n == 1)
    return;
  int lg = ilog2_exact(n);
  if (lg < 0)
    lg = 0;          /* fft_radix2.c:89-93: message only, then zero stages */

  /* the reference permutes with logn derived from n even when n is not a
   * power of two; that case is outside the supported domain, so only the
   * power-of-two permutation is restated */
  if (ilog2_exact(n) >= 0)
    bit_reverse_permute(d, n, lg);

  size_t span = 1;         /* p   */
  size_t groups = n;       /* q   */
  for (int stage = 1; stage <= lg; stage++) {
    const size_t half = span;                    /* p_1 */
    span *= 2;
    groups /= 2;

    for (size_t b = 0; b < groups; b++) {        /* a = 0: fft_radix2.c:113-119 */
      float *g = d + b * span;
      float s0 = g[0] + g[half];
      float s1 = g[0] - g[half];
      g[0] = s0;
      g[half] = s1;
    }

    {                                            /* fft_radix2.c:123-166 */
      float wr = 1.0f, wi = 0.0f;
      const double theta = -2.0 * M_PI / span;
      const float s = sin(theta);
      const float t = sin(theta / 2.0);
      const float s2 = 2.0 * t * t;

      for (size_t a = 1; a < half / 2; a++) {
        {
          const float nr = wr - s * wi - s2 * wr;
          const float ni = wi + s * wr - s2 * wi;
          wr = nr;
          wi = ni;
        }
        for (size_t b = 0; b < groups; b++) {
          float *g = d + b * span;
          float z0r = g[a];
          float z0i = g[half - a];
          float z1r = g[half + a];
          float z1i = g[span - a];

          float t0r = z0r + wr * z1r - wi * z1i;
          float t0i = z0i + wr * z1i + wi * z1r;
          float t1r = z0r - wr * z1r + wi * z1i;
          float t1i = z0i - wr * z1i - wi * z1r;

          g[a] = t0r;
          g[span - a] = t0i;
          g[half - a] = t1r;
          g[half + a] = -t1i;
        }
      }
    }

    if (half > 1) {                              /* fft_radix2.c:168-174 */
      for (size_t b = 0; b < groups; b++)
        d[b * span + span - half / 2] *= -1;
    }
  }
}

/* fft.c:203-226 */
void go_psd(const float *hc, int n, float *psd)
{
  psd[0] = hc[0] * hc[0] / n;
  for (int i = 1; i < (n + 1) / 2; i++)
    psd[i] = (hc[i] * hc[i] + hc[n - i] * hc[n - i]) / n;
  if (n % 2 == 0)
    psd[n / 2] = hc[n / 2] * hc[n / 2] / n;
}

/* fft.c:190-200 then fft.c:203-226 */
void go_fft_frame(go_fft_state *st, float *hop, int first_buffer, float *psd)
{
  go_prepare(st, hop, first_buffer);
  go_rfft_halfcomplex(st->inbuf_fft, (size_t)st->n);
  go_psd(st->inbuf_fft, st->n, psd);
}

/* ------------------------------------------------------------------ */
/* g-l_dpss.c:213-282: 32-point Gauss-Legendre nodes and weights.  Only the
 * 16 positive nodes are tabulated here; the rule is symmetric. */
static const double glq_node_pos[16] = {
  .048307665687738316235, .144471961582796493485, .239287362252137074545,
  .331868602282127649780, .421351276130635345364, .506899908932229390024,
  .587715757240762329041, .663044266930215200975, .732182118740289680387,
  .794483795967942406963, .849367613732569970134, .896321155766052123965,
  .934906075937739689171, .964762255587506430774, .985611511545268335400,
  .997263861849481563545
};
static const double glq_weight_pos[16] = {
  .096540088514727800567, .095638720079274859419, .093844399080804565639,
  .091173878695763884713, .087652093004403811143, .083311924226946755222,
  .078193895787070306472, .072345794108848506225, .065822222776361846838,
  .058684093478535547145, .050998059262376176196, .042835898022226680657,
  .034273862913021433103, .025392065309262059456, .016274394730905670605,
  .007018610009470096600
};
#define GLQ 32
static double glq_x(int i) { return i < 16 ? -glq_node_pos[15 - i] : glq_node_pos[i - 16]; }
static double glq_w(int i) { return i < 16 ? glq_weight_pos[15 - i] : glq_weight_pos[i - 16]; }

/* g-l_dpss.c:75-81: one plane rotation applied to two matrix entries */
#define ROT(M, r0, c0, r1, c1) do { \
    double g_ = M[(r0) * GLQ + (c0)], h_ = M[(r1) * GLQ + (c1)]; \
    M[(r0) * GLQ + (c0)] = g_ - s * (h_ + g_ * tau); \
    M[(r1) * GLQ + (c1)] = h_ + s * (g_ - h_ * tau); } while (0)

/* g-l_dpss.c:84-196 eigen_jacobi (cyclic Jacobi, GSL-0.9 lineage), for the
 * fixed 32x32 case.  A is destroyed; eigenvectors are the COLUMNS of V. */
static int jacobi32(double *A, double *eval, double *V, unsigned max_sweeps)
{
  double bsum[GLQ], zacc[GLQ];
  for (int p = 0; p < GLQ; p++) {
    for (int q = 0; q < GLQ; q++)
      V[p * GLQ + q] = 0.0;
    V[p * GLQ + p] = 1.0;
  }
  for (int p = 0; p < GLQ; p++) {
    zacc[p] = 0.0;
    bsum[p] = eval[p] = A[p * GLQ + p];
  }
  for (unsigned sweep = 1; sweep <= max_sweeps; sweep++) {
    double off = 0.0;
    for (int p = 0; p < GLQ - 1; p++)
      for (int q = p + 1; q < GLQ; q++)
        off += fabs(A[p * GLQ + q]);
    if (off == 0.0)
      return 0;
    const double thresh = (sweep < 4) ? 0.2 * off / (GLQ * GLQ) : 0.0;

    for (int p = 0; p < GLQ - 1; p++) {
      for (int q = p + 1; q < GLQ; q++) {
        const double dp = eval[p], dq = eval[q];
        const double apq = A[p * GLQ + q];
        double g = 100.0 * fabs(apq);
        if (sweep > 4 && fabs(dp) + g == fabs(dp) && fabs(dq) + g == fabs(dq)) {
          A[p * GLQ + q] = 0.0;
        } else if (fabs(apq) > thresh) {
          double h = dq - dp, t;
          if (fabs(h) + g == fabs(h)) {
            t = apq / h;
          } else {
            double theta = 0.5 * h / apq;
            t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
            if (theta < 0.0)
              t = -t;
          }
          const double c = 1.0 / sqrt(1.0 + t * t);
          const double s = t * c;
          const double tau = s / (1.0 + c);
          h = t * apq;
          zacc[p] -= h;
          zacc[q] += h;
          eval[p] = dp - h;
          eval[q] = dq + h;
          A[p * GLQ + q] = 0.0;
          for (int j = 0; j < p; j++)
            ROT(A, j, p, j, q);
          for (int j = p + 1; j < q; j++)
            ROT(A, p, j, j, q);
          for (int j = q + 1; j < GLQ; j++)
            ROT(A, p, j, q, j);
          for (int j = 0; j < GLQ; j++)
            ROT(V, j, p, j, q);
        }
      }
    }
    for (int p = 0; p < GLQ; p++) {
      bsum[p] += zacc[p];
      zacc[p] = 0.0;
      eval[p] = bsum[p];
    }
  }
  return -1;
}

/* g-l_dpss.c:35-72: selection sort on |eigenvalue|, descending */
static void sort_by_magnitude(double *eval, double *V)
{
  for (int i = 0; i < GLQ - 1; i++) {
    int best = i;
    double eb = eval[i];
    for (int j = i + 1; j < GLQ; j++) {
      if (fabs(eval[j]) > fabs(eb)) {
        best = j;
        eb = eval[j];
      }
    }
    if (best != i) {
      double t = eval[i];
      eval[i] = eval[best];
      eval[best] = t;
      for (int r = 0; r < GLQ; r++) {
        t = V[r * GLQ + i];
        V[r * GLQ + i] = V[r * GLQ + best];
        V[r * GLQ + best] = t;
      }
    }
  }
}

/* g-l_dpss.c:288-347 gl_dpss.  nw is N*W (g-l_dpss.c:295-297). */
int go_dpss(int n, int kmax, double nw, double *tapers, double *sig)
{
  const double c = M_PI * nw;
  double K[GLQ * GLQ], V[GLQ * GLQ], ev[GLQ];

  for (int i = 0; i < GLQ; i++) {                          /* g-l_dpss.c:303-313 */
    for (int j = 0; j < GLQ; j++) {
      double kij;
      if (i == j)
        kij = c / M_PI;
      else
        kij = sin(c * (glq_x(i) - glq_x(j))) / (M_PI * (glq_x(i) - glq_x(j)));
      kij *= sqrt(glq_w(i) * glq_w(j));
      K[i * GLQ + j] = kij;
    }
  }
  int err = jacobi32(K, ev, V, 1000);                      /* g-l_dpss.c:315 */
  sort_by_magnitude(ev, V);                                /* g-l_dpss.c:316 */

  for (int i = 0; i < n; i++) {                            /* g-l_dpss.c:319-328 */
    for (int k = 0; k <= kmax; k++) {
      double acc = 0.0;
      for (int j = 0; j < GLQ; j++) {
        double arg = (2.0 * (i + 0.5) / n) - 1.0 - glq_x(j);
        acc += sqrt(glq_w(j)) * V[j * GLQ + k] * sin(c * arg) / (M_PI * arg);
      }
      tapers[(size_t)k * n + i] = acc;
    }
  }
  for (int k = 0; k <= kmax; k++) {                        /* g-l_dpss.c:331-339 */
    double e = 0.0;
    double *v = tapers + (size_t)k * n;
    for (int i = 0; i < n; i++)
      e += v[i] * v[i];
    for (int i = 0; i < n; i++)
      v[i] /= sqrt(e);
  }
  for (int k = 0; k <= kmax; k++)                          /* g-l_dpss.c:342-344 */
    sig[k] = ev[k] - 1.0;
  return err;
}

/* ------------------------------------------------------------------ */
/* mtm.c:154-239 mtm_do, observable part */
void go_mtm_frame(go_fft_state *st, const double *tapers, const double *sig,
                  int kmax, float *hop, int first_buffer, float *psd)
{
  const int n = st->n;
  const int nb = n / 2 + 1;
  float *tmp = (float *)malloc((size_t)nb * sizeof(float));

  go_prepare(st, hop, first_buffer);                       /* mtm.c:162 */
  for (int i = 0; i < nb; i++)                             /* mtm.c:179-186 */
    psd[i] = 0.0f;

  for (int j = 0; j <= kmax; j++) {                        /* mtm.c:189 */
    const double *v = tapers + (size_t)j * n;
    for (int i = 0; i < n; i++)                            /* mtm.c:190-192 */
      st->inbuf_fft[i] = v[i] * st->inbuf_audio[i];
    go_rfft_halfcomplex(st->inbuf_fft, (size_t)n);         /* mtm.c:198 */
    go_psd(st->inbuf_fft, n, tmp);                         /* mtm.c:212 */
    for (int i = 0; i < nb; i++)                           /* mtm.c:214-219 */
      psd[i] += tmp[i] / (1.0 + sig[j]);
  }
  free(tmp);
}

/* ------------------------------------------------------------------ */
/* fft.c:229-238: the comparator returns (a<b); with glibc's merge sort that
 * is a stable descending sort, which is what is restated here. */
static int cmp_desc(const void *pa, const void *pb)
{
  float a = *(const float *)pa, b = *(const float *)pb;
  return (a < b) - (a > b);
}

/* fft.c:240-294 compute_floor */
void go_floor(const float *psd, int n, float *sig_pwr, float *floor_pwr,
              float *peak_pwr, unsigned int *peak_bin)
{
  float *sorted = (float *)malloc((size_t)n * sizeof(float));
  memcpy(sorted, psd, (size_t)n * sizeof(float));
  qsort(sorted, (size_t)n, sizeof(float), cmp_desc);

  float fl = 0.0f;
  for (int i = n * 0.95; i < n; i++)                       /* fft.c:271-272 */
    fl += sorted[i];
  fl /= 0.05;                                              /* fft.c:274 */
  fl /= n;                                                 /* fft.c:276 */

  *sig_pwr = sorted[0];                                    /* fft.c:279 */
  *floor_pwr = fl;
  *peak_pwr = 0.0f;                                        /* fft.c:284-291 */
  *peak_bin = 0;
  for (int i = 0; i < n; i++) {
    if (psd[i] > *peak_pwr) {
      *peak_pwr = psd[i];
      *peak_bin = (unsigned)i;
    }
  }
  free(sorted);
}

/* ------------------------------------------------------------------ */
/* avg.c:38-60 */
void go_avg_alloc(go_avg *a, int width, int depth)
{
  a->width = width;
  a->depth = depth;
  a->effdepth = 0;
  a->avg = (double *)calloc((size_t)width, sizeof(double));
  a->cum = (double *)calloc((size_t)width, sizeof(double));
  a->ring = (double *)calloc((size_t)width * depth, sizeof(double));
}

/* avg.c:62-78 */
void go_avg_free(go_avg *a)
{
  if (a->width) {
    free(a->avg);
    free(a->cum);
    free(a->ring);
  }
  a->width = a->depth = a->effdepth = 0;
}

/* the sliding-sum body shared by all three modes: avg.c:116-127,172-183,235-246 */
static void avg_push(go_avg *a, int bin, float v)
{
  double *reg = a->ring + (size_t)bin * a->depth;
  if (a->effdepth < a->depth) {
    reg[a->effdepth] = v;
    a->cum[bin] += v;
  } else {
    a->cum[bin] += v - reg[0];
    memmove(reg, reg + 1, (size_t)(a->depth - 1) * sizeof(double));
    reg[a->depth - 1] = v;
  }
}

/* avg.c:108-159 */
double go_avg_plain(go_avg *a, int n, const float *psd, int minbin, int maxbin,
                    int *peakbin)
{
  double spec = 0.0, top = psd[minbin];
  for (int b = minbin; b < maxbin; b++) {
    avg_push(a, b, psd[b]);
    if (a->cum[b] > top) {
      top = a->cum[b];
      *peakbin = b;
    }
    spec += a->cum[b];
  }
  if (a->effdepth < a->depth)
    a->effdepth++;
  spec = (spec - top) / ((double)(maxbin - minbin - 1) * (double)(a->effdepth + 1));
  for (int b = 0; b < n; b++) {
    if (b < minbin || b >= maxbin)
      a->avg[b] = 1e-15;
    else
      a->avg[b] = a->cum[b] / (double)(a->effdepth + 1);
  }
  return spec;
}

/* avg.c:161-219 */
double go_avg_sumextreme(go_avg *a, int n, const float *psd, int max0,
                         int minbin, int maxbin, int *peakbin)
{
  double top = psd[minbin], spec = 0.0, low = 1.0;
  for (int b = minbin; b < maxbin; b++) {
    avg_push(a, b, psd[b]);
    spec += a->cum[b];
    if (a->cum[b] > top) {
      top = a->cum[b];
      *peakbin = b;
    }
    if (a->cum[b] < low)
      low = a->cum[b];
  }
  if (a->effdepth < a->depth)
    a->effdepth++;
  spec = (spec - top) / (double)(maxbin - minbin - 1);
  for (int b = 0; b < n; b++) {
    if (b < minbin || b >= maxbin)
      a->avg[b] = 1e-15;
    else if (max0)
      a->avg[b] = (a->cum[b] - low) / (top - low);
    else
      a->avg[b] = a->cum[b] / spec;
  }
  return top / spec;
}

/* avg.c:222-298 */
double go_avg_sumavg(go_avg *a, int n, const float *psd, int max0, int minbin,
                     int maxbin, int *peakbin, double *variance)
{
  double top = psd[minbin], spec = 0.0;
  for (int b = minbin; b < maxbin; b++) {
    avg_push(a, b, psd[b]);
    spec += a->cum[b];
    if (a->cum[b] > top) {
      top = a->cum[b];
      *peakbin = b;
    }
  }
  if (a->effdepth < a->depth)
    a->effdepth++;
  spec = (spec - top) / (double)(maxbin - minbin - 1);
  *variance = 0.0;
  int nvar = 0;
  for (int b = 0; b < n; b++) {
    if (b < minbin || b >= maxbin) {
      a->avg[b] = 1e-15;
    } else {
      double delta = a->cum[b] - spec;
      if (delta > 0) {
        if (max0)
          a->avg[b] = (a->cum[b] - spec) / (top - spec);
        else
          a->avg[b] = a->cum[b] / spec;
        if (b != *peakbin) {
          *variance += (a->cum[b] / spec) * (a->cum[b] / spec);
          nvar++;
        }
      } else {
        a->avg[b] = 1e-15;
      }
    }
  }
  *variance = *variance / (double)nvar;
  return top / spec;
}

/* ------------------------------------------------------------------ */
/* util.c:261-386 compute_svd -- one-sided Jacobi (Nash / Demmel-Veselic):
 * double inner products, float storage. */
int go_svd(float *A, int nrow, int ncol, float *S, float *Q)
{
  if (!A || !S || !Q || nrow == 0 || ncol == 0)
    return -1;
  const double tol = 1.0e-12, dbl_eps = 2.22e-16;
  int sweeps = 0, limit = ncol > 12 ? ncol : 12;
  int pending = 1;

  for (int i = 0; i < ncol; i++) {
    for (int j = 0; j < ncol; j++)
      Q[i * ncol + j] = 0.0f;
    Q[i * ncol + i] = 1.0f;
  }
  while (pending > 0 && sweeps <= limit) {
    pending = ncol * (ncol - 1) / 2;
    for (int j = 0; j < ncol - 1; j++) {
      for (int k = j + 1; k < ncol; k++) {
        double p = 0.0, q = 0.0, r = 0.0;
        for (int i = 0; i < nrow; i++) {
          const double aj = A[i * ncol + j], ak = A[i * ncol + k];
          p += aj * ak;
          q += aj * aj;
          r += ak * ak;
        }
        if (q * r < dbl_eps) {
          pending--;
          continue;
        }
        if (p * p / (q * r) < tol) {
          pending--;
          continue;
        }
        double cs, sn;
        if (q < r) {
          cs = 0.0;
          sn = 1.0;
        } else {
          q -= r;
          double v = sqrt(4.0 * p * p + q * q);
          cs = sqrt((v + q) / (2.0 * v));
          sn = p / (v * cs);
        }
        for (int i = 0; i < nrow; i++) {
          double ak = A[i * ncol + k], aj = A[i * ncol + j];
          A[i * ncol + j] = aj * cs + ak * sn;
          A[i * ncol + k] = -aj * sn + ak * cs;
        }
        for (int i = 0; i < ncol; i++) {
          double qj = Q[i * ncol + j], qk = Q[i * ncol + k];
          Q[i * ncol + j] = qj * cs + qk * sn;
          Q[i * ncol + k] = -qj * sn + qk * cs;
        }
      }
    }
    sweeps++;
  }
  for (int j = 0; j < ncol; j++) {
    double q = 0.0;
    for (int i = 0; i < nrow; i++) {
      double aj = A[i * ncol + j];
      q += aj * aj;
    }
    S[j] = sqrt(q);
    for (int i = 0; i < nrow; i++) {
      double aj = A[i * ncol + j], sj = S[j];
      A[i * ncol + j] = aj / sj;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* hparma.c:74-157 hparma_do (q_e = -1, source.c:375) */
void go_hparma_frame(go_fft_state *st, int t, int p_e, float *hop, int first_buffer, float *psd,
                     float *a_out, int *rank_out)
{
  const int n = st->n, ncol = p_e + 1, q_e = -1;
  /* matrix(0,t,0,p_e) (hparma.c:64): t+1 rows of ncol floats in ONE block, row r at r*ncol */
  float *R = (float *)calloc((size_t)(t + 1) * ncol + 1, sizeof(float));
  float *S = (float *)calloc((size_t)t + 1, sizeof(float));          /* vector(0,t)        hparma.c:66 */
  float *V = (float *)calloc((size_t)ncol * ncol, sizeof(float));    /* matrix(0,p_e,0,p_e) hparma.c:67 */
  float *a = (float *)calloc((size_t)ncol, sizeof(float));           /* vector(0,p_e)      hparma.c:69 */

  go_prepare(st, hop, first_buffer);                                  /* hparma.c:86 */

  for (int i = 0; i <= q_e + t; i++) {                                /* hparma.c:89-95 */
    double s = 0.0;
    for (int k = 0; k < n - i; k++)
      s += st->inbuf_audio[k + i] * st->inbuf_audio[k];
    R[i] = s / (n - i);                /* r_xx[0][i]: runs past column p_e into the next rows */
  }
  for (int i = 1; i < t; i++)                                         /* hparma.c:98-102 */
    for (int j = 0; j <= p_e; j++)
      R[i * ncol + j] = R[abs(j - i)]; /* r_xx[0][|j-i|], possibly a cell this loop already rewrote */

  go_svd(R, t, ncol, S, V);                                           /* hparma.c:104 */

  double sum_sigma2 = 0.0;                                            /* hparma.c:108-111 */
  for (int i = 0; i < ncol; i++)
    sum_sigma2 += S[i] * S[i];
  int p = 4;                                                          /* hparma.c:84 */
  {
    double acc = 0.0;                                                 /* hparma.c:113-122 */
    for (int i = 0; i < ncol; i++) {
      acc += S[i] * S[i];
      double nu = sqrt(acc / sum_sigma2);
      if (nu > 0.995) {
        p = i;
        break;
      }
    }
  }
  for (int i = 0; i <= p_e; i++) {                                    /* hparma.c:125-138 */
    double num = 0.0, den = 0.0;
    for (int k = p + 1; k <= p_e; k++) {
      num += V[0 * ncol + k] * V[i * ncol + k];
      den += V[0 * ncol + k] * V[0 * ncol + k];
    }
    if (p < p_e)
      a[i] = num / den;
    else
      a[i] = (i == 0 ? 1.0 : 0.0);
  }
  for (int i = 0; i <= p_e; i++)                                      /* hparma.c:140-145 */
    st->inbuf_fft[i] = a[i];
  for (int i = p_e + 1; i < n; i++)
    st->inbuf_fft[i] = 0.0f;
  go_rfft_halfcomplex(st->inbuf_fft, (size_t)n);                      /* hparma.c:150 */
  go_psd(st->inbuf_fft, n, psd);                                      /* hparma.c:153 */
  for (int i = 0; i < n / 2; i++)                                     /* hparma.c:154-156 */
    psd[i] = 1.0 / psd[i];

  if (a_out)
    memcpy(a_out, a, (size_t)ncol * sizeof(float));
  if (rank_out)
    *rank_out = p;
  free(R);
  free(S);
  free(V);
  free(a);
}

/* source.c:130-156 over a whole stream, HP-ARMA mode (window forced rectangular, source.c:369) */
void go_spectrogram_hparma(const float *stream, size_t nsamples, int n, float overlap, int t,
                           int p_e, int sub_mean, int history_mode, float *psd_out)
{
  go_fft_state st;
  go_fft_state_init(&st, n, overlap, GO_WIN_RECTANGULAR, 0.0f, 0, sub_mean);
  const int h = go_hop(n, overlap);
  const size_t frames = go_num_frames(nsamples, n, overlap);
  const size_t nb = (size_t)n / 2 + 1;
  float *hop = (float *)malloc((size_t)h * sizeof(float));
  for (size_t f = 0; f < frames; f++) {
    memcpy(hop, stream + f * h, (size_t)h * sizeof(float));
    int first = (history_mode == 1) ? 1 : (f == 0);
    go_hparma_frame(&st, t, p_e, hop, first, psd_out + f * nb, NULL, NULL);
  }
  free(hop);
  go_fft_state_free(&st);
}

/* ------------------------------------------------------------------ */
/* g_main.c:651-762 set_palette.  The reference casts doubles (some negative or > 255) straight
 * to unsigned char; on x86-64 that is a truncating double->int32 conversion whose low byte is
 * kept, which is what (unsigned char)(int) spells out. */
#define PAL(v) ((unsigned char)(int)(v))
void go_palette(int p_n, unsigned char tab[768])
{
  unsigned char *p = tab;
  for (long c = 0; c < 256; c++) {
    long color = c * 256 / 256;
    switch (p_n) {
    case 0: case 1:                                   /* HSV, thresholded HSV */
      if (p_n == 1 && color < 16) { *p++ = 0; *p++ = 0; *p++ = 0; }
      else if (color < 64) { *p++ = 0; *p++ = PAL(color * 4.0); *p++ = 255; }
      else if (color < 128) { *p++ = 0; *p++ = 255; *p++ = PAL(510.0 - color * 4.0); }
      else if (color < 192) { *p++ = PAL(color * 4.0 - 510.0); *p++ = 255; *p++ = 0; }
      else { *p++ = 255; *p++ = PAL(1020.0 - color * 4.0); *p++ = 0; }
      break;
    case 2:                                           /* cool */
      *p++ = (unsigned char)color; *p++ = (unsigned char)(255 - color); *p++ = 255;
      break;
    case 3:                                           /* hot */
      if (color < 96) { *p++ = PAL(color * 2.66667 + 0.5); *p++ = 0; *p++ = 0; }
      else if (color < 192) { *p++ = 255; *p++ = PAL(color * 2.66667 - 254); *p++ = 0; }
      else { *p++ = 255; *p++ = 255; *p++ = PAL(color * 4.0 - 766.0); }
      break;
    case 5:                                           /* bone */
      if (color < 96) { *p++ = PAL(color * 0.88889); *p++ = PAL(color * 0.88889); *p++ = PAL(color * 1.20000); }
      else if (color < 192) { *p++ = PAL(color * 0.88889); *p++ = PAL(color * 1.20000 - 29); *p++ = PAL(color * 0.88889 + 29); }
      else { *p++ = PAL(color * 1.20000 - 60); *p++ = PAL(color * 0.88889 + 29); *p++ = PAL(color * 0.88889 + 29); }
      break;
    case 6:                                           /* copper */
      if (color < 208) { *p++ = PAL(color * 1.23); *p++ = PAL(color * 0.78); *p++ = PAL(color * 0.5); }
      else { *p++ = 255; *p++ = PAL(color * 0.78); *p++ = PAL(color * 0.5); }
      break;
    case 7:                                           /* OTD */
      if (color < 128) { *p++ = 0; *p++ = PAL(2.0 * color - 1.0); *p++ = PAL(2.0 * (127.0 - color) + 1.0); }
      else { *p++ = PAL(2.0 * (color - 127.0) - 1.0); *p++ = PAL(2.0 * (255.0 - color) + 1.0); *p++ = 0; }
      break;
    default:                                          /* BW (4) and anything else */
      *p++ = (unsigned char)color; *p++ = (unsigned char)color; *p++ = (unsigned char)color;
    }
  }
}

/* double -> short / unsigned char the way the reference's implicit conversions behave on x86-64:
 * truncating conversion to int32 (out of range or NaN gives INT_MIN), low bits kept */
static int x86_d2i(double d)
{
  if (!(d > -2147483649.0 && d < 2147483648.0))
    return (int)0x80000000u;
  return (int)d;
}

/* g_main.c:1109-1139 + 1186-1236 */
void go_display_column(go_display_state *st, const float *src_f, const double *src_d, int n,
                       float sig_pwr, float floor_pwr, const unsigned char colortab[768],
                       unsigned char *rgb, short *lev, float levels_out[2])
{
  const float thr_level = st->thr_level / 100.0;                       /* g_main.c:1099 */
  float display_max, display_min;
  if (st->autoscale) {                                                 /* g_main.c:1111-1124 */
    if (st->first_buffer) {
      if (st->overlap > 0.0) {
        sig_pwr /= st->overlap;
        floor_pwr /= st->overlap;
      }
      st->display_max_lvl = sig_pwr;
      st->display_min_lvl = floor_pwr;
      st->first_buffer = 0;
    } else {
      st->display_max_lvl = (1.0 - 0.99) * sig_pwr + 0.99 * st->display_max_lvl;
      st->display_min_lvl = (1.0 - 0.99) * floor_pwr + 0.99 * st->display_min_lvl;
    }
  } else {                                                             /* g_main.c:1125-1130 */
    st->display_max_lvl = pow(10.0, st->max_level_db / 10.0);
    st->display_min_lvl = pow(10.0, st->min_level_db / 10.0);
    st->display_min_lvl = (st->display_max_lvl > st->display_min_lvl ? st->display_min_lvl : st->display_max_lvl / 10.0);
  }
  if (st->scale_log) {                                                 /* g_main.c:1132-1139 */
    display_max = 10.0 * log10(st->display_max_lvl);
    display_min = 10.0 * log10(st->display_min_lvl);
  } else {
    display_max = st->display_max_lvl;
    display_min = st->display_min_lvl;
  }
  levels_out[0] = display_max;
  levels_out[1] = display_min;

  for (int i = 0; i < n; i++) {                                        /* g_main.c:1186-1236 */
    float sig_level;
    if (st->scale_log) {
      /* sig_level = levbuf[..] = 10.0*log10(x): the value of the assignment is the SHORT */
      const double db = src_d ? 10.0 * log10(src_d[n - i - 1]) : 10.0 * log10(src_f[n - i - 1]);
      lev[i] = (short)x86_d2i(db);
      sig_level = lev[i];
    } else {
      sig_level = src_d ? src_d[n - i - 1] : src_f[n - i - 1];
      lev[i] = (short)x86_d2i(10.0 * log10(sig_level));
    }
    const float f = 255 * ((sig_level - display_min) / (display_max - display_min));
    unsigned char v;
    if (f < 255.0 * thr_level)
      v = 0;
    else if (f > 255)
      v = 255;
    else
      v = (unsigned char)x86_d2i((f - 255.0 * thr_level) / (1.0 - thr_level));
    rgb[3 * i] = colortab[3 * v];
    rgb[3 * i + 1] = colortab[3 * v + 1];
    rgb[3 * i + 2] = colortab[3 * v + 2];
  }
}

/* ------------------------------------------------------------------ */
/* wav_fmt.c:104-117 */
void go_pcm_u8_to_float(const unsigned char *in, size_t n, float *out)
{
  for (size_t i = 0; i < n; i++)
    out[i] = ((float)in[i] - 128) / 128;
}

void go_pcm_s16_to_float(const short *in, size_t n, float *out)
{
  for (size_t i = 0; i < n; i++)
    out[i] = (float)in[i] / 32768;
}

/* ------------------------------------------------------------------ */
/* One frame per full hop of H samples.  (A file source also hands over a trailing partial
 * block, wav_fmt.c:102-119: that is go_wav_spectrogram, below.) */
size_t go_num_frames(size_t nsamples, int n, float overlap)
{
  int h = go_hop(n, overlap);
  return h > 0 ? nsamples / (size_t)h : 0;
}

/* source.c:130-144 over a whole stream, FFT mode */
void go_spectrogram_fft(const float *stream, size_t nsamples, int n,
                        float overlap, int window_type, float a, int limiter,
                        int sub_mean, int history_mode, float *psd_out)
{
  go_fft_state st;
  go_fft_state_init(&st, n, overlap, window_type, a, limiter, sub_mean);
  const int h = go_hop(n, overlap);
  const size_t frames = go_num_frames(nsamples, n, overlap);
  const size_t nb = (size_t)n / 2 + 1;
  float *hop = (float *)malloc((size_t)h * sizeof(float));
  for (size_t f = 0; f < frames; f++) {
    memcpy(hop, stream + f * h, (size_t)h * sizeof(float));
    int first = (history_mode == 1) ? 1 : (f == 0);
    go_fft_frame(&st, hop, first, psd_out + f * nb);
  }
  free(hop);
  go_fft_state_free(&st);
}

/* source.c:130-148 over a whole stream, MTM mode (window forced rectangular,
 * source.c:344; a/limiter act on inbuf_fft only, which mtm_do overwrites
 * at mtm.c:190-192, so they cannot change the result) */
void go_spectrogram_mtm(const float *stream, size_t nsamples, int n,
                        float overlap, double nw, int kmax, int sub_mean,
                        int history_mode, float *psd_out)
{
  go_fft_state st;
  go_fft_state_init(&st, n, overlap, GO_WIN_RECTANGULAR, 0.0f, 0, sub_mean);
  double *tapers = (double *)malloc((size_t)(kmax + 1) * n * sizeof(double));
  double *sig = (double *)malloc((size_t)(kmax + 1) * sizeof(double));
  go_dpss(n, kmax, nw, tapers, sig);
  const int h = go_hop(n, overlap);
  const size_t frames = go_num_frames(nsamples, n, overlap);
  const size_t nb = (size_t)n / 2 + 1;
  float *hop = (float *)malloc((size_t)h * sizeof(float));
  for (size_t f = 0; f < frames; f++) {
    memcpy(hop, stream + f * h, (size_t)h * sizeof(float));
    int first = (history_mode == 1) ? 1 : (f == 0);
    go_mtm_frame(&st, tapers, sig, kmax, hop, first, psd_out + f * nb);
  }
  free(hop);
  free(tapers);
  free(sig);
  go_fft_state_free(&st);
}

/* ------------------------------------------------------------------ */
/* wav_fmt.c:81-121 wav_read(): one read() of out_len samples per call into a buffer that is
 * allocated (zeroed) on the first call and never cleared afterwards.  A short last read
 * converts only the samples it delivered -- n_read for 8 bit, n_read/2 for 16 bit (an odd
 * trailing byte is dropped) -- and still reports one block (*n_out = n_read == 0 ? 0 : 1), so
 * the estimator sees the new samples over the STALE tail of the previous block; that tail is
 * whatever the estimator left there (prepare_audio removes the hop's mean in place,
 * fft.c:93-95).  The PCM bytes are handed over in memory instead of through a descriptor. */
void go_wav_open(go_wav *w, const unsigned char *pcm, size_t nbytes, int bits, int out_len)
{
  w->pcm = pcm;
  w->nbytes = nbytes;
  w->pos = 0;
  w->bits = bits;
  w->out_len = out_len;
  w->buff = (float *)calloc((size_t)out_len, sizeof(float));       /* wav_fmt.c:99 */
}

void go_wav_close(go_wav *w)
{
  free(w->buff);
  w->buff = NULL;
}

int go_wav_read(go_wav *w, float **buf_out)
{
  const size_t s_bufsize = (size_t)w->out_len * w->bits / 8;       /* wav_fmt.c:87 */
  size_t n_read = w->nbytes - w->pos;                              /* read(): what is left, at most s_bufsize */
  if (n_read > s_bufsize)
    n_read = s_bufsize;
  const unsigned char *buf = w->pcm + w->pos;
  w->pos += n_read;
  if (w->bits == 8) {                                              /* wav_fmt.c:104-108 */
    for (size_t i = 0; i < n_read; i++)
      w->buff[i] = ((float)buf[i] - 128) / 128;
  } else if (w->bits == 16) {                                      /* wav_fmt.c:109-116 */
    for (size_t i = 0; i < n_read / 2; i++) {
      short v;
      memcpy(&v, buf + 2 * i, 2);
      w->buff[i] = (float)v / 32768;
    }
  }
  *buf_out = w->buff;
  return n_read == 0 ? 0 : 1;                                      /* wav_fmt.c:119 */
}

/* source.c:118-165 over a whole WAV data chunk: wav_read() until it reports no block, one
 * estimator call per block ON THE READER'S OWN BUFFER (audio_buf = buff, source.c:143-148).
 * mode 0: fft_do + fft_psd; mode 1: mtm_do.  Returns the number of rows written (at most
 * max_frames). */
size_t go_wav_spectrogram(const unsigned char *pcm, size_t nbytes, int bits, int mode, int n,
                          float overlap, int window_type, float a, int limiter, int sub_mean,
                          int history_mode, double nw, int kmax, size_t max_frames, float *psd_out)
{
  go_fft_state st;
  go_fft_state_init(&st, n, overlap, mode == 1 ? GO_WIN_RECTANGULAR : window_type, mode == 1 ? 0.0f : a,
                    mode == 1 ? 0 : limiter, sub_mean);
  double *tapers = NULL, *sig = NULL;
  if (mode == 1) {
    tapers = (double *)malloc((size_t)(kmax + 1) * n * sizeof(double));
    sig = (double *)malloc((size_t)(kmax + 1) * sizeof(double));
    go_dpss(n, kmax, nw, tapers, sig);
  }
  const size_t nb = (size_t)n / 2 + 1;
  go_wav w;
  go_wav_open(&w, pcm, nbytes, bits, go_hop(n, overlap));
  size_t f = 0;
  float *hop;
  while (f < max_frames && go_wav_read(&w, &hop)) {
    int first = (history_mode == 1) ? 1 : (f == 0);
    if (mode == 1)
      go_mtm_frame(&st, tapers, sig, kmax, hop, first, psd_out + f * nb);
    else
      go_fft_frame(&st, hop, first, psd_out + f * nb);
    f++;
  }
  go_wav_close(&w);
  free(tapers);
  free(sig);
  go_fft_state_free(&st);
  return f;
}

/* ------------------------------------------------------------------ */
/* lmp.c:59-99 lmp_init, lmp.c:101-181 lmp_do.  The ring of nl periodograms starts zeroed
 * (lmp.c:87-93) and is written round-robin (j_l, lmp.c:176-178); mean and variance per bin are
 * taken over the ring in SLOT order, in double.  prepare_audio's output is overwritten by the
 * raw assembled frame (lmp.c:114-116), so window, a and limiter have no effect (the window is
 * rectangular anyway, source.c:395). */
void go_lmp_init(go_lmp_state *st, int n, float overlap, int nl, int sub_mean)
{
  go_fft_state_init(&st->fft, n, overlap, GO_WIN_RECTANGULAR, 0.0f, 0, sub_mean);
  st->nl = nl;
  st->j_l = 0;
  st->ring = (float *)calloc((size_t)nl * n, sizeof(float));       /* matrix(0,nl-1,0,n-1), cleared */
  st->my = (double *)calloc((size_t)n, sizeof(double));
  st->sy = (double *)calloc((size_t)n, sizeof(double));
}

void go_lmp_free(go_lmp_state *st)
{
  go_fft_state_free(&st->fft);
  free(st->ring);
  free(st->my);
  free(st->sy);
  st->ring = NULL;
  st->my = st->sy = NULL;
}

void go_lmp_frame(go_lmp_state *st, float *hop, int first_buffer, float *psd_buf)
{
  const int n_fft = st->fft.n, nl = st->nl;
  double *my = st->my, *sy = st->sy;
  double v_hat;

  go_prepare(&st->fft, hop, first_buffer);                         /* lmp.c:110 */
  for (int i = 0; i < n_fft; i++)                                  /* lmp.c:114-116 */
    st->fft.inbuf_fft[i] = st->fft.inbuf_audio[i];
  go_rfft_halfcomplex(st->fft.inbuf_fft, (size_t)n_fft);           /* lmp.c:121 */
  go_psd(st->fft.inbuf_fft, n_fft, st->ring + (size_t)st->j_l * n_fft);   /* lmp.c:125 */

  for (int i = 0; i < n_fft / 2 + 1; i++) {                        /* lmp.c:134-140 */
    my[i] = 0.0;
    for (int j = 0; j < nl; j++)
      my[i] += st->ring[(size_t)j * n_fft + i];
    my[i] /= nl;
  }
  for (int i = 0; i < n_fft / 2 + 1; i++) {                        /* lmp.c:143-149 */
    sy[i] = 0.0;
    for (int j = 0; j < nl; j++)
      sy[i] += (st->ring[(size_t)j * n_fft + i] - my[i]) * (st->ring[(size_t)j * n_fft + i] - my[i]);
    sy[i] /= (nl - 1);
  }
  for (int i = 0; i < n_fft / 2 + 1; i++) {                        /* lmp.c:151-159 */
    v_hat = my[i] * my[i] - sy[i];
    if (v_hat < 0.0) v_hat = 0.0;
    v_hat = 0.5 * (my[i] - sqrt(v_hat));
    psd_buf[i] = -sqrt(nl / 2.0) + (nl * my[i]) / (2.0 * sqrt(2.0 * nl) * v_hat);
    if (psd_buf[i] <= 1.0e-3) psd_buf[i] = 1e-3;
  }
  psd_buf[0] = 1e-3;                                               /* lmp.c:160 */

  st->j_l++;                                                       /* lmp.c:176-178 */
  if (st->j_l == nl)
    st->j_l = 0;
}

/* source.c:130-158 over a whole stream, LMP mode */
void go_spectrogram_lmp(const float *stream, size_t nsamples, int n, float overlap, int nl,
                        int sub_mean, int history_mode, float *out)
{
  go_lmp_state st;
  go_lmp_init(&st, n, overlap, nl, sub_mean);
  const int h = go_hop(n, overlap);
  const size_t frames = go_num_frames(nsamples, n, overlap);
  const size_t nb = (size_t)n / 2 + 1;
  float *hop = (float *)malloc((size_t)h * sizeof(float));
  for (size_t f = 0; f < frames; f++) {
    memcpy(hop, stream + f * h, (size_t)h * sizeof(float));
    int first = (history_mode == 1) ? 1 : (f == 0);
    go_lmp_frame(&st, hop, first, out + f * nb);
  }
  free(hop);
  go_lmp_free(&st);
}

/* ------------------------------------------------------------------ */
/* The harmonic F-test side computation of mtm.c.  Tables: mtm.c:76-83 (U0), mtm.c:124-136
 * (sum_U0_sqr and hn, both accumulated in float).  Per frame: mtm.c:165-174 (mu = transform of
 * the hn-windowed frame), mtm.c:203-210 (denominator: sum over tapers of |y_j - mu U0_j|^2, DC
 * and 0 < i < n/2 only: the Nyquist bin's stays 0), mtm.c:222-233 (numerator / denominator).
 * In the reference build that fft_radix2.c serves, the transform at mtm.c:173 runs in place on
 * inbuf_fft and `mu` is never written: mu_live = 0 restates that (mu all zeros, so ftest is 0,
 * or NaN where the denominator is 0); mu_live = 1 is the evident intent (what the FFTW build
 * computes: rfftw_one(plan, inbuf_fft, mu), mtm.c:171). */
void go_ftest_tables(int n, int kmax, const double *tapers, double *U0, float *hn, float *sum_U0_sqr_out)
{
  float sum_U0_sqr;
  for (int j = 0; j <= kmax; j++) {                                /* mtm.c:76-83 */
    U0[j] = 0.0;
    for (int i = 0; i < n; i++)
      U0[j] += tapers[(size_t)j * n + i];
  }
  sum_U0_sqr = 0.0;                                                /* mtm.c:125-128 */
  for (int j = 0; j <= kmax; j++)
    sum_U0_sqr += U0[j] * U0[j];
  for (int i = 0; i < n; i++) {                                    /* mtm.c:130-136 */
    hn[i] = 0.0;
    for (int j = 0; j <= kmax; j++)
      hn[i] += U0[j] * tapers[(size_t)j * n + i];
    hn[i] /= sum_U0_sqr;
  }
  *sum_U0_sqr_out = sum_U0_sqr;
}

void go_mtm_ftest_frame(go_fft_state *st, const double *tapers, const double *sig, int kmax,
                        const double *U0, const float *hn, float sum_U0_sqr, int mu_live,
                        float *hop, int first_buffer, float *psd_buf, float *ftest)
{
  const int n_fft = st->n, k = kmax;
  double tmpr, tmpi, num_ftest;
  float *mu = (float *)calloc((size_t)n_fft, sizeof(float));       /* mtm.c:108 */
  float *psdbuftmp = (float *)malloc(((size_t)n_fft / 2 + 1) * sizeof(float));
  float *outbuf = st->inbuf_fft;                                   /* mtm.c:106 */

  go_prepare(st, hop, first_buffer);                               /* mtm.c:162 */
  for (int i = 0; i < n_fft; i++)                                  /* mtm.c:165-168 */
    st->inbuf_fft[i] = st->inbuf_audio[i] * hn[i];
  go_rfft_halfcomplex(st->inbuf_fft, (size_t)n_fft);               /* mtm.c:173 */
  if (mu_live)
    memcpy(mu, st->inbuf_fft, (size_t)n_fft * sizeof(float));      /* mtm.c:171 (FFTW build) */

  for (int i = 0; i < (n_fft + 1) / 2; i++) {                      /* mtm.c:179-186 */
    psd_buf[i] = 0.0;
    ftest[i] = 0.0;
  }
  if (n_fft % 2 == 0) {
    psd_buf[n_fft / 2] = 0.0;
    ftest[n_fft / 2] = 0.0;
  }
  for (int j = 0; j <= k; j++) {                                   /* mtm.c:189-220 */
    for (int i = 0; i < n_fft; i++)
      st->inbuf_fft[i] = tapers[(size_t)j * n_fft + i] * st->inbuf_audio[i];
    go_rfft_halfcomplex(st->inbuf_fft, (size_t)n_fft);
    tmpr = outbuf[0] - mu[0] * U0[j];
    ftest[0] += tmpr * tmpr;
    for (int i = 1; i < (n_fft + 1) / 2; i++) {
      tmpr = outbuf[i] - mu[i] * U0[j];
      tmpi = outbuf[n_fft - i] - mu[n_fft - i] * U0[j];
      ftest[i] += tmpr * tmpr + tmpi * tmpi;
    }
    go_psd(st->inbuf_fft, n_fft, psdbuftmp);
    for (int i = 0; i < (n_fft + 1) / 2; i++)
      psd_buf[i] += psdbuftmp[i] / (1.0 + sig[j]);
    if (n_fft % 2 == 0)
      psd_buf[n_fft / 2] += psdbuftmp[n_fft / 2] / (1.0 + sig[j]);
  }
  num_ftest = k * (mu[0] * mu[0]) * sum_U0_sqr;                    /* mtm.c:222-233 */
  ftest[0] = num_ftest / ftest[0];
  for (int i = 1; i < (n_fft + 1) / 2; i++) {
    num_ftest = k * (mu[i] * mu[i] + mu[n_fft - i] * mu[n_fft - i]) * sum_U0_sqr;
    ftest[i] = num_ftest / ftest[i];
  }
  if (n_fft % 2 == 0) {
    int i = n_fft / 2;
    num_ftest = k * (mu[i] * mu[i] + mu[n_fft - i] * mu[n_fft - i]) * sum_U0_sqr;
    ftest[i] = num_ftest / ftest[i];
  }
  free(mu);
  free(psdbuftmp);
}

/* source.c:130-148 over a whole stream, MTM mode, with the F statistic of every frame */
void go_spectrogram_mtm_ftest(const float *stream, size_t nsamples, int n, float overlap, double nw,
                              int kmax, int sub_mean, int history_mode, int mu_live, float *psd_out,
                              float *ftest_out)
{
  go_fft_state st;
  go_fft_state_init(&st, n, overlap, GO_WIN_RECTANGULAR, 0.0f, 0, sub_mean);
  double *tapers = (double *)malloc((size_t)(kmax + 1) * n * sizeof(double));
  double *sig = (double *)malloc((size_t)(kmax + 1) * sizeof(double));
  double *U0 = (double *)malloc((size_t)(kmax + 1) * sizeof(double));
  float *hn = (float *)malloc((size_t)n * sizeof(float));
  float sum_U0_sqr;
  go_dpss(n, kmax, nw, tapers, sig);
  go_ftest_tables(n, kmax, tapers, U0, hn, &sum_U0_sqr);
  const int h = go_hop(n, overlap);
  const size_t frames = go_num_frames(nsamples, n, overlap);
  const size_t nb = (size_t)n / 2 + 1;
  float *hop = (float *)malloc((size_t)h * sizeof(float));
  for (size_t f = 0; f < frames; f++) {
    memcpy(hop, stream + f * h, (size_t)h * sizeof(float));
    int first = (history_mode == 1) ? 1 : (f == 0);
    go_mtm_ftest_frame(&st, tapers, sig, kmax, U0, hn, sum_U0_sqr, mu_live, hop, first,
                       psd_out + f * nb, ftest_out + f * nb);
  }
  free(hop);
  free(hn);
  free(U0);
  free(tapers);
  free(sig);
  go_fft_state_free(&st);
}
