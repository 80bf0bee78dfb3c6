/* glfer_compat.h -- the reference's OWN estimator interface, served by the HIP engine.
 *
 * libglfer_compat.so exports exactly the L2 functions that source.c and g_main.c call, with
 * the reference's signatures and struct layouts, so that glfer links against it instead of
 * its fft.o / fft_radix2.o / mtm.o / g-l_dpss.o / avg.o / hparma.o / lmp.o (INTEGRATION.md
 * shows the link line; no source change and no glue file are needed: the library reads glfer's
 * own globals `opt` and `glfer`, see below).  Each call is one hop, as in the reference's main loop
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

/* fft.h:69-75 and fft.c:47-59: the window table the options dialog builds its menu from
 * (g_options.c:47-48, 579-583) -- two DATA symbols of the object this library replaces */
typedef struct _fft_window_t fft_window_t;
struct _fft_window_t {
  char *name;
  int type;
};
extern fft_window_t fft_windows[];      /* "/Hanning" ... "/Kaiser", in the order of the enum above */
extern int num_fft_windows;             /* 8 */

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

/* lmp.h:36-45 */
typedef struct {
  fft_params_t fft;
  int avg;                      /* opt.lmp_av: periodograms in the ring (source.c:397) */
  double **window;              /* unused by lmp.c */
  double *sig;
  float w;
  int kmax;
} lmp_params_t;

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

/* lmp.h:48-50 */
void lmp_init(lmp_params_t *params);
void lmp_do(float *audio_buf, float *psd_buf, float *phase_buf, lmp_params_t *params);
void lmp_close(lmp_params_t *params);

/* wav_fmt.h:24-26 -- the file source.  Exported here so that the file loop of source.c:112-171 runs
 * at the GPU's rate UNCHANGED (link this library instead of wav_fmt.o too): when fft_do / mtm_do /
 * hparma_do are handed the reader's own buffer, untouched, for a file's hops in order, the row comes
 * from a batch the device computed from the file itself (a window of frames at a time, per-hop means in
 * the reference's order); anything else -- another buffer, samples changed after wav_read, a hop
 * skipped or repeated, glfer.first_buffer not following the mode's pattern (cleared after the first
 * column with autoscale, g_main.c:1111-1120; stuck TRUE without), the scope window open, LMP mode --
 * takes the per-hop launch from that hop on.  inbuf_audio and the caller's buffer (mean removed in
 * place, fft.c:93-95) are kept hop by hop either way; inbuf_fft / outbuf hold the last PER-HOP
 * frame only (their one reader, the scope window, turns the read-ahead off while open).
 * The header's RIFF chunks are walked; 8/16-bit PCM as wav_fmt.c:104-117; a short last block
 * counts as a block with its fresh samples over the stale rest of the buffer (wav_fmt.c:102-119). */
int open_wav_file(char *fname, int n, int *speed);
void close_wav_file(void);
void wav_read(float **buf_out, int *n_out);
extern int glfer_compat_readahead;                 /* 1 (default); 0 = every hop through the per-hop path */
extern unsigned long glfer_compat_readahead_served;   /* hops served from a batch so far */

/* avg.h:38-43 */
void init_avg(avg_data_t *avgdata);
void alloc_avg(avg_data_t *avgdata, int width, int depth);
void delete_avg(avg_data_t *avgdata);
double update_avg_plain(avg_data_t *avgdata, int N, float *psd, int minbin, int maxbin, int *peakbin);
double update_avg_sumextreme(avg_data_t *avgdata, int N, float *psd, int max0, int minbin, int maxbin,
                             int *peakbin);
double update_avg_sumavg(avg_data_t *avgdata, int N, float *psd, int max0, int minbin, int maxbin,
                         int *peakbin, double *variance);

/* The two globals the reference's estimators read directly: opt.autoscale at *_init (fft.c:186,
 * mtm.c:111, hparma.c:62, lmp.c:82) and glfer.first_buffer on every hop (fft.c:99).  Inside glfer
 * they are the program's own `opt_t opt` and `glfer_t glfer` (glfer.c:56-57): the library refers
 * to those two symbols WEAKLY and reads the fields through the layout-compatible declarations
 * below (glfer.h:62-139 with the GTK pointer types as void*: same sizes, same offsets), so the
 * relink needs no added source.  A program that defines neither (a stand-alone user of these
 * entry points) sets glfer_compat_autoscale / glfer_compat_first_buffer instead.
 * Skipped when the real glfer.h is in the translation unit. */
#ifndef _GLFER_H_
typedef enum { NO_AVG, AVG_SUMAVG, AVG_PLAIN, AVG_SUMEXTREME } avgmode_t;                 /* glfer.h:54-56 */
typedef enum { SOURCE_NONE, SOUNDCARD_SOURCE, FILE_SOURCE } datasource_t;                 /* glfer.h:49-51 */
typedef struct {                                                                          /* glfer.h:62-120 */
  char *program_name;
  int mode;
  int scale_type;
  int data_block_size;
  float data_blocks_overlap;
  float display_update_time;
  float limiter_a;
  int enable_limiter;
  float mtm_w;
  int mtm_k;
  int hparma_t;
  int hparma_p_e;
  int lmp_av;
  int window_type;
  char *audio_device;
  int sample_rate;
  float dot_time;
  float dfcw_gap_time;
  int tx_mode;
  float dash_dot_ratio;
  float ptt_delay;
  float sidetone_freq;
  int sidetone;
  float dfcw_dot_freq;
  float dfcw_dash_freq;
  int beacon_mode;
  float beacon_pause;
  int beacon_tx_pause;
  char *ctrl_device;
  int device_type;
  float offset_freq;
  float thr_level;
  int autoscale;
  float max_level_db;
  float min_level_db;
  avgmode_t averaging;
  int avgsamples;
  float min_avgband;
  float max_avgband;
  int palette;
} opt_t;
typedef struct {                                                                          /* glfer.h:123-139 */
  void *tt;                     /* GtkTooltips * */
  void *qso_menu_item;          /* GtkWidget *   */
  void *test_menu_item;         /* GtkWidget *   */
  int init_done;
  int first_buffer;
  datasource_t input_source;
  float cpu_usage;
  int current_mode;
  void *scope_window;           /* GtkWidget *   */
  float avgmax;
  double avgvar;
  int avgfill;
  float peakfreq;
  float peakval;
  float avgtime;
} glfer_t;
#endif /* _GLFER_H_ */
int glfer_compat_get_autoscale(void);      /* opt.autoscale when the program defines `opt`, else the variable below */
int glfer_compat_get_first_buffer(void);   /* glfer.first_buffer when the program defines `glfer`, else the variable */
extern int glfer_compat_autoscale;         /* default 1 (glfer.c:275 default for opt.autoscale) */
extern int glfer_compat_first_buffer;      /* default 1 until the caller clears it, as g_main.c:1120 does */

#ifdef __cplusplus
}
#endif
#endif /* GLFER_COMPAT_H */
