/* spectro_params.h -- kernel argument block shared by the launchers and the C-ABI layer. */
#ifndef GLFER_SPECTRO_PARAMS_H
#define GLFER_SPECTRO_PARAMS_H

#include <hip/hip_runtime.h>

enum { GLFER_FMT_F32 = 0, GLFER_FMT_S16 = 1, GLFER_FMT_U8 = 2 };

struct SpectroParams {
  const void *stream;      /* device: sample stream (f32 / s16 / u8)                        */
  long long frame0;        /* index of this launch's first frame in the whole stream        */
  int nframes;             /* frames in this launch                                          */
  int H;                   /* hop = (int)(N*(1.0-overlap))            fft.c:70               */
  int R;                   /* N - H samples of history per frame      fft.c:71               */
  int npairs;              /* ceil(T/2) complex transforms per frame                         */
  int history_mode;        /* 0: zeros before sample 0 only; 1: history zeroed every frame   */
  int fmt;                 /* GLFER_FMT_*                                                    */
  int nonlin;              /* RA9MB / limiter path (periodogram only)                        */
  int limiter;             /* fft.c:151-156                                                  */
  float a;                 /* fft.c:127-136                                                  */
  float post_scale;        /* nonlin path: sqrt(1/(2N)) applied after the limiter            */
  float spec_unscale;      /* factor folded into taper 0 (undone for the spectrum output)    */
  const float *taps;       /* device: [npairs][N][2] taper pairs interleaved, weights and 1/(2N) folded */
  const float2 *tw;        /* device: [slots][N/16] per-lane inter-pass twiddles (cos,sin)    */
  /* real-input (N/2-point) form of a single taper, spectro16h.hip; NULL when not built for this plan */
  const float *htaps;      /* device: [8][N/32][4] window as (w[2n],w[2n+1]) pairs, sqrt(1/(4N)) folded */
  const float2 *htw;       /* device: [slots][N/32] inter-pass twiddles of the N/2-point transform      */
  int htapers;             /* windows in htaps: 0/1 = periodogram; > 1 = multitaper via the real-input form, [htapers] tables */
  const float2 *hrot;      /* device: [N/32] (cos,sin)(2 pi t/N), the lane part of the post twiddle     */
  /* real-input form with wavefront-private 1024-point transforms, spectro16w.hip (N >= 2048); NULL when not built */
  const float *wtaps;      /* device: [wtapers][W][8][64][4] window/taper pairs in lane order, sqrt(1/(4N(1+sig))) folded */
  int wtapers;             /* tables in wtaps: 1 = periodogram window; > 1 = the tapers of mtm_do       */
  const float2 *wtw;       /* device: [27][64] inter-pass twiddles of the 1024-point transform          */
  const float2 *wcomb;     /* device: [IPL*W][64 W] (cos,sin): [i][0] = 2 pi k1/N, [i][w] = 2 pi w k1/M, k1 = u + 64 W i */
  const float2 *bigtw;     /* device: [W][16] (cos,sin)(-2 pi 64 w m / M), spectro_big.hip (N >= 32768); NULL when not built */
  /* odd taper counts, spectro16x.hip: the last taper alone; NULL when not built for this plan */
  const float *xtaps;      /* device: [4][N/16][4] last taper, sqrt(1/(4N(1+sig))) folded               */
  /* odd taper counts with LDS-resident half tables, spectro16xl.hip; NULL when not built */
  const float *ltaps;      /* device: [npairs-1][8][N/16][2] pair halves, then [8][N/16] the last taper */
  float *psd;              /* device: [nframes][pitch], the first N/2+1 floats of a row are its bins */
  int pitch;               /* floats from one PSD row to the next (cfg.psd_pitch; N/2+1 = dense)      */
  float *spec;             /* device, optional: [nframes][N] halfcomplex spectrum            */
  /* harmonic F statistic (mtm.c:165-174, 203-233) inside spectro16_kernel; ftest NULL = off.  taps then holds
     [rounds][2N] tables with ONE taper each (im part zero): hn first when ft_mu_live, then tapers 0..ntap-1 */
  float *ftest;            /* device: [nframes][N/2+1]                                        */
  const double *ft_U0;     /* device: [ntap]                                                  */
  float ft_sum_U0_sqr;
  int ft_mu_live;          /* 0: mu is all zeros (the reference build without FFTW, mtm.c:173) */
  int ft_nseq;             /* > 0: the PAIRED form (round 5): taps holds [ceil(ft_nseq / 2)][2N] tables with TWO real sequences each
                              (re: sequence 2r, im: sequence 2r+1, both scaled by 1/2; the sequences are hn -- when ft_mu_live --
                              then tapers 0..ntap-1), one N-point transform per pair, the two spectra separated through the
                              mirror bins (X_a = Z[k] + conj Z[N-k], X_b = (Z[k] - conj Z[N-k]) / i)                 */
  int mean_inkernel;       /* per-hop mean removal (fft.c:86-96) inside spectro16h.hip: the stream is the RAW one;
                              only where the hop is 2, 4, 8 or 16 sixteenths of N               */
  const float *means;      /* device, optional (with mean_inkernel): means[h] = the mean of hop h of the whole stream (virtual
                              base), taken in the reference's own order (submean_seq.hip, GLFER_SUBMEAN_EXACT); the kernel
                              subtracts these instead of summing the hops itself                */
  /* spectro16h.hip's table form with the means PRODUCED INSIDE THE SAME LAUNCH (round 4): the first nprod workgroups take the
     hop means in the reference's own order (the 64-hops-side-by-side chains of submean_seq.hip) while the others transform;
     a consumer workgroup waits for the chunks its frames' hops lie in.  nprod = 0: the means table was filled by an earlier launch. */
  int nprod;               /* producer workgroups at the head of the grid                                            */
  int prod_chunk;          /* hops per chunk of means_ready (a multiple of 64)                                       */
  float *means_out;        /* = means, writable                                                                       */
  unsigned *means_ready;   /* device: [ceil(prod_nhops / prod_chunk)] hop groups of the chunk that are written (zeroed before the launch) */
  long long prod_hop0;     /* the hops to produce: [prod_hop0, prod_hop0 + prod_nhops) of the whole stream             */
  long long prod_nhops;
  /* round 5, the LOCK-STEPPED fused launch (prod_front_frames > 0): the consumer workgroups walk the stream in eight fronts (one per
     XCD: consumer workgroup w belongs to front w mod 8 and takes the front's next range of frames), producer workgroup j serves front
     j mod 8 and stays at most prod_look hops ahead of what that front's consumers have finished, so that the estimator's read of a
     hop comes out of the Infinity Cache the producers filled a few tens of microseconds earlier.  means_ready then holds one flag
     per 64-hop group (prod_chunk = 64), followed by front_done[8] (consumer workgroups finished per front).                     */
  long long prod_front_frames;   /* frames per front: (consumer workgroups / 8) x frames per workgroup                                */
  int prod_block_frames;         /* frames per consumer workgroup                                                                    */
  int prod_look;                 /* hops a front's producers may run ahead of its finished consumers                                  */
  unsigned *front_done;          /* device: [8]                                                                                     */
  /* update_avg_plain (avg.c:108-159) INSIDE spectro16h.hip's periodogram kernel (round 5): avg != NULL.  A frame slot walks
     consecutive frames and keeps the last depth-1 PSD rows of its bins in registers; the window's sum is taken in double per
     bin and divided by depth+1 as the reference does once its window is full (avg.c:138-139,155).  EVERY frame of the launch
     has its full window: the launcher hands over frames whose depth-1 predecessors are computable (>= the first frame
     that lies inside the stream) and belong to the same averaging state; a slot recomputes them in front of its range. */
  double *avg;             /* device: [nframes][avg_nout] doubles, row i = frame frame0 + i; columns outside the band 1e-15 */
  double *avg_ret;         /* device, optional: [nframes][4] = {band mean (avg.c:147), peak bin or -1, 0, effdepth}        */
  int avg_depth;           /* 1..4                                                                                           */
  int avg_minbin, avg_maxbin, avg_nout;
};

#ifdef __cplusplus
extern "C" {
#endif
hipError_t glfer_launch_spectro16_n8(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16_n9(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16_n10(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16_n11(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16_n12(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16_n13(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16_n14(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16x_n8(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16x_n9(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16x_n10(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16xl_n8(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16xl_n9(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16xl_n10(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16xl_n11(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16y_n12(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16h_n9(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16h_n10(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16h_n11(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16h_n12(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16h_n13(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16h_n14(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16w_n11(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16w_n12(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16w_n13(const SpectroParams *p, hipStream_t st);
hipError_t glfer_launch_spectro16w_n14(const SpectroParams *p, hipStream_t st);
size_t glfer_levels_scratch_floats(size_t nframes);
hipError_t glfer_launch_levels(const float *stats, size_t nframes, int scale_log, int autoscale,
                               int first_buffer, float overlap, float max_lvl0, float min_lvl0,
                               float *levels, float *chunk_state, hipStream_t st);
hipError_t glfer_launch_levels_fixed(size_t nframes, float dmax, float dmin, float max_lvl, float min_lvl,
                                     float *levels, hipStream_t st);
hipError_t glfer_launch_map(const float *psd, const double *avg, size_t nframes, int n, int psd_pitch, int scale_log,
                            double thr255, double one_m_thr, const float *levels,
                            const unsigned char *colortab, const double *log_thr, unsigned char *rgb, short *lev,
                            hipStream_t st);
hipError_t glfer_launch_submean(const void *in, float *out, int H, long long nhops, int fmt,
                                hipStream_t st, const float *means /* NULL: summed by the kernel */);
#ifdef __cplusplus
}
#endif
#endif
