/* glfer_oracle.h -- CPU restatement of glfer's spectral-estimation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP engine in
 * glfer_amd/: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing under glfer_amd/ links or calls it.
 *
 * Every function restates one reference routine (file:line into the
 * reference tree, glfer 0.4.2) with the same IEEE arithmetic in the same
 * order, so that on one compiler/flag set the results are bit-identical to
 * the reference objects built by oracle/Makefile into oracle/_ref/.
 *
 * Pinning status:
 *   - go_rfft_halfcomplex, go_dpss, go_avg_*, go_bessel_i0, go_svd, go_wav_read are
 *     checked BIT-EXACT against the reference's own fft_radix2.c, g-l_dpss.c, avg.c,
 *     util.c and wav_fmt.c compiled unmodified (tests/test_oracle_pinning.py).
 *   - go_window, go_prepare, go_psd, go_mtm_frame, go_floor restate fft.c/mtm.c,
 *     which cannot be compiled here (glfer.h needs <gtk/gtk.h>, absent from
 *     this image); they are pinned by independent known answers (numpy rfft
 *     in float64, scipy DPSS, Parseval) and by committed golden vectors.
 */
#ifndef GLFER_ORACLE_H
#define GLFER_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* window ids: same numbering as the enum at fft.h:67 */
enum {
  GO_WIN_HANNING = 0, GO_WIN_BLACKMAN, GO_WIN_GAUSSIAN, GO_WIN_WELCH,
  GO_WIN_BARTLETT, GO_WIN_RECTANGULAR, GO_WIN_HAMMING, GO_WIN_KAISER
};

/* util.c:222-237 */
double go_bessel_i0(double x);

/* fft.c:309-360 compute_window(): shape in double, stored float, then
 * divided by sqrt(sum w^2) with the sum held in float. */
void go_window(int window_type, int n, float *w);

/* hop length: fft.c:70  n_eff = N * (1.0 - overlap)  (double arithmetic on a
 * float overlap, truncated toward zero). */
int go_hop(int n, float overlap);

/* Frame assembler state = the parts of fft_params_t (fft.h:52-63) that
 * prepare_audio() touches. */
typedef struct {
  int n;               /* block size N                                   */
  float overlap;       /* fraction of overlap                            */
  int window_type;     /* GO_WIN_*                                       */
  float a;             /* RA9MB parameter; >0 enables x/(a+x^2)          */
  int limiter;         /* 1 = sign(x)*|x|^0.1                            */
  int sub_mean;        /* subtract the mean of the NEW hop samples       */
  float *window;       /* [n]                                            */
  float *inbuf_audio;  /* [n] assembled frame; persists = overlap history */
  float *inbuf_fft;    /* [n] processed frame, transformed in place      */
} go_fft_state;

void go_fft_state_init(go_fft_state *st, int n, float overlap, int window_type,
                       float a, int limiter, int sub_mean);
void go_fft_state_free(go_fft_state *st);

/* fft.c:66-165 prepare_audio().  hop[] holds go_hop() new samples and is
 * modified in place when sub_mean is set (the reference does the same to
 * the caller's buffer).  first_buffer != 0 zeroes the history instead of
 * sliding it (fft.c:99-108). */
void go_prepare(go_fft_state *st, float *hop, int first_buffer);

/* fft_radix2.c:75-177: in-place float32 radix-2 DIT real->halfcomplex,
 * float trigonometric recurrence for the twiddles. */
void go_rfft_halfcomplex(float *data, size_t n);

/* fft.c:203-226 fft_psd() (power branch; phase is never requested). */
void go_psd(const float *halfcomplex, int n, float *psd);

/* fft.c:190-200 + fft.c:203-226: one periodogram frame. */
void go_fft_frame(go_fft_state *st, float *hop, int first_buffer, float *psd);

/* g-l_dpss.c:288-347 gl_dpss(): tapers [kmax+1][n] (row = taper, unlike the
 * reference's 1-based [n][kmax+1] matrix) and sig[k] = lambda_k - 1. */
int go_dpss(int n, int kmax, double nw, double *tapers, double *sig);

/* mtm.c:154-239 mtm_do() output path (F-test side computation omitted: it
 * has no observable result). */
void go_mtm_frame(go_fft_state *st, const double *tapers, const double *sig,
                  int kmax, float *hop, int first_buffer, float *psd);

/* fft.c:240-294 compute_floor(). */
void go_floor(const float *psd, int n, float *sig_pwr, float *floor_pwr,
              float *peak_pwr, unsigned int *peak_bin);

/* avg.c:28-36 avg_data_t, flat storage. */
typedef struct {
  int width, depth, effdepth;
  double *avg;      /* [width]        */
  double *cum;      /* [width]        */
  double *ring;     /* [width][depth] shift registers */
} go_avg;

void go_avg_alloc(go_avg *a, int width, int depth);           /* avg.c:38-60   */
void go_avg_free(go_avg *a);                                   /* avg.c:62-78   */
double go_avg_plain(go_avg *a, int n, const float *psd, int minbin, int maxbin,
                    int *peakbin);                             /* avg.c:108-159 */
double go_avg_sumextreme(go_avg *a, int n, const float *psd, int max0,
                         int minbin, int maxbin, int *peakbin);/* avg.c:161-219 */
double go_avg_sumavg(go_avg *a, int n, const float *psd, int max0, int minbin,
                     int maxbin, int *peakbin, double *variance); /* avg.c:222-298 */

/* util.c:261-386 compute_svd(): one-sided Jacobi, A is [nrow][ncol] row-major
 * float, Q is [ncol][ncol]. */
int go_svd(float *A, int nrow, int ncol, float *S, float *Q);

/* hparma.c:74-157 hparma_do() for one frame (q_e = -1 as set at source.c:375): autocorrelation
 * lags 0..t-1 written into ROW 0 of the (t+1) x (p_e+1) Numerical-Recipes matrix -- past its
 * p_e+1 columns, i.e. into the following rows (util.c:153-160 lays the rows out contiguously) --
 * Toeplitz fill from those (partly overwritten) cells, one-sided Jacobi SVD, rank by
 * sqrt(cumulative sigma^2 / total) > 0.995, AR vector from the noise subspace, N-point FFT of
 * the zero-padded AR vector, psd[i] = 1/psd[i] for i < N/2 (the Nyquist bin is not inverted).
 * a_out (optional): the p_e+1 AR coefficients; rank_out (optional): the rank p. */
void go_hparma_frame(go_fft_state *st, int t, int p_e, float *hop, int first_buffer, float *psd,
                     float *a_out, int *rank_out);

void go_spectrogram_hparma(const float *stream, size_t nsamples, int n, float overlap, int t,
                           int p_e, int sub_mean, int history_mode, float *psd_out);

/* ---- display mapping (g_main.c:1099-1236, palettes g_main.c:651-762) ---- */
/* set_palette(): 256 RGB triplets; p_n as glfer.h:49 {HSV,THRESH,COOL,HOT,BW,BONE,COPPER,OTD} */
void go_palette(int p_n, unsigned char colortab[768]);

typedef struct {
  int scale_log;        /* opt.scale_type is SCALE_LOG or SCALE_LOG_MAX0 (g_main.c:1132) */
  int autoscale;        /* opt.autoscale                                                    */
  float overlap;        /* opt.data_blocks_overlap (first-buffer correction, g_main.c:1114) */
  float max_level_db, min_level_db;   /* fixed levels when autoscale is off (g_main.c:1126) */
  float thr_level;      /* opt.thr_level, percent                                           */
  int first_buffer;     /* glfer.first_buffer: in/out                                       */
  float display_max_lvl, display_min_lvl;   /* the two statics of main_window_draw: in/out  */
} go_display_state;

/* One waterfall column (g_main.c:1109-1139 level tracking, g_main.c:1186-1236 mapping).
 * src_f (float PSD) or src_d (avgdata.avg, when averaging is on) holds n values; sig/floor are
 * compute_floor's outputs for this column.  rgb: n*3 bytes, row i = bin n-1-i; lev: n shorts
 * (levbuf); levels_out: the display_max/display_min actually used (after the log, if any). */
void go_display_column(go_display_state *st, const float *src_f, const double *src_d, int n,
                       float sig_pwr, float floor_pwr, const unsigned char colortab[768],
                       unsigned char *rgb, short *lev, float levels_out[2]);

/* wav_fmt.c:104-117 sample conversion rules. */
void go_pcm_u8_to_float(const unsigned char *in, size_t n, float *out);
void go_pcm_s16_to_float(const short *in, size_t n, float *out);

/* ---- whole-stream drivers: the loop of source.c:130-158 over a stream ---- */

/* history_mode: 0 = zero history on frame 0 only (autoscale on: the drawer
 * clears glfer.first_buffer after the first frame, g_main.c:1111-1120);
 * 1 = zero history on every frame (autoscale off: nothing clears it). */
size_t go_num_frames(size_t nsamples, int n, float overlap);

void go_spectrogram_fft(const float *stream, size_t nsamples, int n,
                        float overlap, int window_type, float a, int limiter,
                        int sub_mean, int history_mode, float *psd_out);

void go_spectrogram_mtm(const float *stream, size_t nsamples, int n,
                        float overlap, double nw, int kmax, int sub_mean,
                        int history_mode, float *psd_out);

/* ---- file source: wav_fmt.c:81-121 wav_read() + the loop of source.c:118-165 ---- */
typedef struct {
  const unsigned char *pcm;   /* the bytes after the WAV header                             */
  size_t nbytes, pos;
  int bits;                   /* wavhd.bit_p_spl: 8 or 16                                   */
  int out_len;                /* samples per block = the hop (open_wav_file's n)            */
  float *buff;                /* the reader's float block, handed to the estimator as is    */
} go_wav;
void go_wav_open(go_wav *w, const unsigned char *pcm, size_t nbytes, int bits, int out_len);
int go_wav_read(go_wav *w, float **buf_out);       /* returns *n_out: 0 at end of data, else 1 */
void go_wav_close(go_wav *w);
/* mode 0 = FFT (fft_do + fft_psd), 1 = MTM (mtm_do); returns the rows written */
size_t go_wav_spectrogram(const unsigned char *pcm, size_t nbytes, int bits, int mode, int n,
                          float overlap, int window_type, float a, int limiter, int sub_mean,
                          int history_mode, double nw, int kmax, size_t max_frames, float *psd_out);

/* ---- LMP estimator: lmp.c:59-99 lmp_init, lmp.c:101-181 lmp_do ---- */
typedef struct {
  go_fft_state fft;
  int nl;              /* params->avg = opt.lmp_av (source.c:397)                           */
  int j_l;             /* ring slot the next periodogram goes to (lmp.c:104 static)         */
  float *ring;         /* [nl][n] psdbufl                                                   */
  double *my, *sy;     /* [n] per-bin mean and variance over the ring                       */
} go_lmp_state;
void go_lmp_init(go_lmp_state *st, int n, float overlap, int nl, int sub_mean);
void go_lmp_free(go_lmp_state *st);
void go_lmp_frame(go_lmp_state *st, float *hop, int first_buffer, float *psd_buf);
void go_spectrogram_lmp(const float *stream, size_t nsamples, int n, float overlap, int nl,
                        int sub_mean, int history_mode, float *out);

/* ---- MTM harmonic F-test: mtm.c:76-83,124-136 (tables), mtm.c:165-174,203-233 (per frame) ---- */
void go_ftest_tables(int n, int kmax, const double *tapers, double *U0, float *hn, float *sum_U0_sqr_out);
void go_mtm_ftest_frame(go_fft_state *st, const double *tapers, const double *sig, int kmax,
                        const double *U0, const float *hn, float sum_U0_sqr, int mu_live,
                        float *hop, int first_buffer, float *psd_buf, float *ftest);
void go_spectrogram_mtm_ftest(const float *stream, size_t nsamples, int n, float overlap, double nw,
                              int kmax, int sub_mean, int history_mode, int mu_live, float *psd_out,
                              float *ftest_out);

#ifdef __cplusplus
}
#endif
#endif
