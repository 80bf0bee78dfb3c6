/* glfer_compat.h -- the reference's OWN estimator interface, served by the HIP engine.
 *
 * libglfer_compat.so exports exactly the L2 functions that source.c and g_main.c call, with
 * the reference's signatures and struct layouts, so that glfer links against it instead of
 * its fft.o / fft_radix2.o / mtm.o / g-l_dpss.o / avg.o (INTEGRATION.md shows the link line
 * and the six-line glue file).  Each call is one hop, as in the reference's main loop
 * (source.c:130-158): the frame is assembled in params->inbuf_audio exactly as
 * prepare_audio() does, then window/taper, FFT, |X|^2 and the taper sum run on the GPU
 * through the batch C-ABI of glfer_hip.h.  Nothing is computed by a CPU fallback; if HIP
 * is unusable the functions print the error and exit(-1), the reference's own failure
 * behaviour (fft.c:249-252, util.c:106-113).
 *
 * For file sources the batch entry glfer_hip_spectrogram_device() is the fast path; these
 * per-hop shims exist so that g_main.c's waterfall draw is unchanged.
 */
#ifndef GLFER_COMPAT_H
#define GLFER_COMPAT_H

#ifdef __cplusplus
extern "C" {
#endif

/* fft.h:52-63 (the non-FFTW variant: outbuf aliases inbuf_fft, fft.c:180) */
typedef struct {
  float *inbuf_audio;
  float *inbuf_fft;
  float *outbuf;
  int n;
  float *window;
  int window_type;
  float overlap;
  float a;                      /* RA9MB nonlinear processing parameter */
  int limiter;
  int sub_mean;
} fft_params_t;

/* fft.h:67 */
enum {HANNING_WINDOW = 0, BLACKMAN_WINDOW, GAUSSIAN_WINDOW, WELCH_WINDOW, BARTLETT_WINDOW,
      RECTANGULAR_WINDOW, HAMMING_WINDOW, KAISER_WINDOW};

/* mtm.h:36-44 */
typedef struct {
  fft_params_t fft;
  double **window;              /* [1..n][0..kmax], rows 1-based as dmatrix(1,n,0,kmax) (mtm.c:118) */
  double *sig;                  /* [0..kmax] = lambda - 1 */
  float w;                      /* N*W */
  int kmax;
} mtm_params_t;

/* hparma.h:25-32 */
typedef struct {
  fft_params_t fft;
  int t;
  int p_e;
  int q_e;                      /* only -1 is supported (what source.c:375 sets) */
} hparma_params_t;

/* avg.h:28-36 */
typedef struct {
  int    avgwidth;
  int    avgdepth;
  int    effdepth;
  double *avg;
  double *cum;
  double **avgarray;
} avg_data_t;

/* fft.h:77-83 */
void prepare_audio(float *audio_buf, fft_params_t *params);
void fft_init(fft_params_t *params);
void fft_do(float *audio_buf, fft_params_t *params);
void fft_psd(float *psd_buf, float *phase_buf, fft_params_t *params);
void fft_close(fft_params_t *params);
void compute_floor(float *psd_buf, int n, float *sig_pwr_p, float *floor_pwr_p, float *peak_pwr_p,
                   unsigned int *peak_bin_p);

/* mtm.h:47-49 */
void mtm_init(mtm_params_t *params);
void mtm_do(float *audio_buf, float *psd_buf, float *phase_buf, mtm_params_t *params);
void mtm_close(mtm_params_t *params);

/* hparma.h:34-38 */
void hparma_init(hparma_params_t *params);
void hparma_do(float *audio_buf, float *psd_buf, float *phase_buf, hparma_params_t *params);
void hparma_close(hparma_params_t *params);

/* avg.h:38-43 */
void init_avg(avg_data_t *avgdata);
void alloc_avg(avg_data_t *avgdata, int width, int depth);
void delete_avg(avg_data_t *avgdata);
double update_avg_plain(avg_data_t *avgdata, int N, float *psd, int minbin, int maxbin, int *peakbin);
double update_avg_sumextreme(avg_data_t *avgdata, int N, float *psd, int max0, int minbin, int maxbin,
                             int *peakbin);
double update_avg_sumavg(avg_data_t *avgdata, int N, float *psd, int max0, int minbin, int maxbin,
                         int *peakbin, double *variance);

/* The two globals the reference's estimators read directly: opt.autoscale at *_init
 * (fft.c:186, mtm.c:111) and glfer.first_buffer on every hop (fft.c:99).  opt_t / glfer_t
 * cannot be declared here (glfer.h drags in GTK), so the library reads them through these
 * two hooks.  They are WEAK: inside glfer the glue file of INTEGRATION.md overrides them
 * with `return opt.autoscale;` / `return glfer.first_buffer;`; stand-alone users set the
 * variables below. */
int glfer_compat_get_autoscale(void);
int glfer_compat_get_first_buffer(void);
extern int glfer_compat_autoscale;      /* default 1 (glfer.c default for opt.autoscale) */
extern int glfer_compat_first_buffer;   /* default 1 until the caller clears it, as g_main.c:1120 does */

#ifdef __cplusplus
}
#endif
#endif /* GLFER_COMPAT_H */
