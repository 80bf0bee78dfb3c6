/* glfer_hip.h -- C-ABI of the MI355X spectral-estimation engine (libglfer_hip.so).
 *
 * This is the drop-in boundary for glfer's L2 "spectral estimator" layer: the calls
 * that source.c:141-158 makes once per hop (fft_do + fft_psd, mtm_do) and that
 * g_main.c:1109,1153-1183 makes once per drawn column (compute_floor, update_avg_*).
 * The reference has no FFI; its boundary is that C function interface, so the
 * replacement is (a) a batch API over a whole sample stream (this file) and (b)
 * signature-compatible per-hop shims built on it (glfer_compat.h).
 *
 * Plain C types only: pointers, sizes, ints, floats.  Device pointers are HIP device
 * addresses (hipMalloc / torch tensor data_ptr); `hip_stream` is a hipStream_t passed
 * as void* (NULL = the default stream).  All functions return 0 on success or a
 * negative GLFER_E_* code; glfer_hip_strerror() names it.  Nothing here falls back to
 * the CPU: without a usable HIP device every compute entry point fails with
 * GLFER_E_HIP.
 *
 * Device memory the library takes by itself: what a plan holds (tables; freed by
 * glfer_hip_plan_destroy) and per-call scratch, given back in stream order.  Scratch requests of
 * 16 MiB and more (averaged rows, a mean-corrected stream copy, big-block and LMP / F-test
 * intermediates) come from up to six blocks per device that the library keeps until the process ends
 * -- the stream-ordered allocator costs milliseconds, now and then seconds, per GB-sized request.
 * GLFER_SCRATCH_CACHE=0 in the environment takes them from the stream-ordered pool instead.
 */
#ifndef GLFER_HIP_H
#define GLFER_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* estimator: glfer.h:47  enum {MODE_NONE=-1, MODE_FFT, MODE_MTM, MODE_HPARMA, MODE_LMP} */
#define GLFER_MODE_FFT 0
#define GLFER_MODE_MTM 1
#define GLFER_MODE_HPARMA 2
#define GLFER_MODE_LMP 3

/* window ids: fft.h:67 */
#define GLFER_WIN_HANNING 0
#define GLFER_WIN_BLACKMAN 1
#define GLFER_WIN_GAUSSIAN 2
#define GLFER_WIN_WELCH 3
#define GLFER_WIN_BARTLETT 4
#define GLFER_WIN_RECTANGULAR 5
#define GLFER_WIN_HAMMING 6
#define GLFER_WIN_KAISER 7

/* sample formats of the stream: float [-1,1), or raw PCM converted on the device with
 * the rules of wav_fmt.c:104-117 (u8: (x-128)/128, s16: x/32768) */
#define GLFER_SAMPLES_F32 0
#define GLFER_SAMPLES_S16 1
#define GLFER_SAMPLES_U8 2

/* history_mode: what the first N-H samples of a frame hold (fft.c:98-108).
 * ZERO_FIRST : zeros before the first sample only (glfer.first_buffer cleared after
 *              frame 0 -- what happens with opt.autoscale on, g_main.c:1111-1120)
 * ZERO_ALWAYS: zeros in every frame (glfer.first_buffer never cleared -- what the
 *              reference does with opt.autoscale off)                              */
#define GLFER_HISTORY_ZERO_FIRST 0
#define GLFER_HISTORY_ZERO_ALWAYS 1

/* averaging modes: glfer.h:56-58 avgmode_t */
#define GLFER_AVG_SUMAVG 1
#define GLFER_AVG_PLAIN 2
#define GLFER_AVG_SUMEXTREME 3

#define GLFER_OK 0
#define GLFER_E_ARG (-1)      /* bad argument / unsupported size             */
#define GLFER_E_HIP (-2)      /* HIP runtime error (no device, launch, copy) */
#define GLFER_E_NOMEM (-3)
#define GLFER_E_NUMERIC (-4)  /* DPSS eigen-solve did not converge           */

/* Everything the reference copies from `opt` into fft_params_t / mtm_params_t at
 * change_params() time (source.c:320-325, source.c:343-350) plus the two globals the
 * estimators read directly (opt.autoscale -> sub_mean, fft.c:186; glfer.first_buffer
 * -> history_mode, fft.c:99). */
typedef struct glfer_hip_config {
  int mode;            /* GLFER_MODE_*                                                  */
  int n;               /* opt.data_block_size; power of two, 8..1048576 (HP-ARMA: 32..32768; the halfcomplex spectrum
                          output up to 32768, the F-test up to 16384, the per-column stages up to 32769 bins)  */
  float overlap;       /* opt.data_blocks_overlap, [0,1)                                */
  int window_type;     /* opt.window_type (FFT mode; MTM forces rectangular, source.c:344) */
  float limiter_a;     /* opt.limiter_a  -> fft_params_t.a        (FFT mode only acts)  */
  int enable_limiter;  /* opt.enable_limiter -> fft_params_t.limiter                    */
  int sub_mean;        /* per-hop mean removal, fft.c:86-96: 0 off; 1 (= GLFER_SUBMEAN_EXACT, what fft_init stores:
                          sub_mean = opt.autoscale) the reference's rows; GLFER_SUBMEAN_FAST (2) the opt-in below */
  int history_mode;    /* GLFER_HISTORY_*                                               */
  float mtm_w;         /* opt.mtm_w = N*W time-bandwidth product (g-l_dpss.c:295-297)   */
  int mtm_k;           /* opt.mtm_k = kmax; kmax+1 tapers are used (mtm.c:189)          */
  int sample_format;   /* GLFER_SAMPLES_*                                               */
  int device;          /* HIP device ordinal                                            */
  int hparma_t;        /* opt.hparma_t: number of equations (rows), HP-ARMA mode (source.c:373) */
  int hparma_p_e;      /* opt.hparma_p_e: number of poles (source.c:374); q_e is fixed to -1 (source.c:375) */
  int lmp_av;          /* opt.lmp_av: periodograms in the LMP estimator's ring (source.c:397, lmp.c:85) */
  int psd_pitch;       /* OURS (round 4): floats from one PSD row to the next in the DEVICE entries' d_psd; 0 = dense rows of
                          N/2+1 floats (what every caller of the reference sees: one psd_buf, source.c:317-318).  A multiple
                          of 16 floats -- 2112 for N = 4096 -- puts every row on a 64-byte boundary: dense rows of 2049 floats
                          start 4 bytes further off the cache lines each and cap every row-writing stage at 0.58-0.61 of the
                          HBM peak (profiles/r03_streaming_ceilings.txt); d_psd then holds nframes * psd_pitch floats, the
                          first N/2+1 of a row are its bins, the rest is never written.  glfer_hip_floor_device_pitched and
                          glfer_hip_display.psd_pitch read such rows; update_avg takes the pitch as its `bins` (its band is
                          minbin..maxbin).  Not with GLFER_MODE_LMP; the host / file entries keep dense rows. */
} glfer_hip_config;

/* cfg.sub_mean.  The reference sums a hop sample after sample in a float (fft.c:88-92), and on a hop with a DC level the
 * ORDER of that sum is observable in the rows (up to 6.6e-4 of a row's maximum at |mean| = rms).
 *   GLFER_SUBMEAN_EXACT  (1: what fft_init stores, sub_mean = opt.autoscale -- the DEFAULT meaning of "on" since round 4)
 *                        the reference's rows: the means are accumulated in the reference's own order (submean_seq.hip: one
 *                        lane walks a hop, 64 hops side by side: one more read of the stream, at 6.1 TB/s) and handed to the
 *                        estimator kernels as a table (periodograms -- a form of its own at the plain periodogram's
 *                        occupancy that corrects a hop's samples once, in place --, even taper counts, 5 / 7 ... tapers at
 *                        N = 4096; the other forms read a corrected copy).  To the usual 1e-5 of the oracle whatever the
 *                        input up to N = 4096; above, where the REFERENCE's own recurrence-twiddle transform is 1e-5 and
 *                        more from exact arithmetic on noise-like frames (1.2-1.6e-5 at N = 8192 and 16384), to
 *                        max(1e-5, 1.1 x the oracle's distance from exact), the device itself within 1e-6 of exact
 *                        (tests/test_gpu_round4.py).  Cost against no mean removal on 2^30-sample device-resident f32
 *                        streams: the extra read -- C1 774 against 1 125, C2 269 against 336, C3 67 against 84 M frames/s
 *                        (bench.py's "+mean" rows; against GLFER_SUBMEAN_FAST: 0.69 / 0.94 / 0.83).  Taking the means
 *                        piece by piece so that the estimator's read of a piece comes out of the Infinity Cache was built
 *                        and measured (GLFER_EXACT_PIECE_MB): launches of a few tens of microseconds cost more than the
 *                        cache saves (profiles/r04_piecewise_means.txt) -- one piece is the default.  Behind a PCIe or file
 *                        source (the host / WAV entries) the pass is hidden: it runs per chunk at 60x the link's rate.
 *   GLFER_SUBMEAN_FAST   (2, opt-in) the hop is summed inside the estimator kernels (lane partials, then across the lanes):
 *                        the more accurate sum, 0-3 % over no mean removal (C2: the two-wavefront form, 14 %) -- and not
 *                        the reference's.  Indistinguishable (<= 1e-6 of a row's maximum) while a hop's mean is small
 *                        against its rms, i.e. for AC-coupled audio; on a hop with a DC level the rows differ at the low
 *                        bins by up to ~7e-4 x |mean|/rms of the row maximum (measured at |mean| = rms: 6.6e-4 Hanning
 *                        periodogram, 7e-5 multitaper, N = 4096, 50 % overlap; tests/test_gpu_round3.py).
 * Any other non-zero value is taken as GLFER_SUBMEAN_EXACT.  The per-hop shims (glfer_compat.h) take the reference's order. */
enum { GLFER_SUBMEAN_OFF = 0, GLFER_SUBMEAN_EXACT = 1, GLFER_SUBMEAN_FAST = 2 };
/* ABI BREAK (library 0.8 -> 0.9, round 4): the two non-zero values were swapped -- 0.8 had 1 = the in-kernel sums and 2 = the
 * reference's order.  A caller built against the 0.8 header that passes 2 now gets GLFER_SUBMEAN_FAST.  GLFER_HIP_ABI counts such
 * breaks; glfer_hip_abi_version() returns the number the LIBRARY was built with, so a host can refuse a mismatch at start-up:
 *     if (glfer_hip_abi_version() != GLFER_HIP_ABI) ...                                   ("ABI version") */
#define GLFER_HIP_ABI 5
int glfer_hip_abi_version(void);

/* Cutting a stream into launches, chunks or shards.
 * (1) Cut at frame indices that are multiples of GLFER_FRAME_ALIGN and every frame's PSD is
 *     bit-identical to the one-shot run (the multitaper kernel for odd taper counts works on
 *     aligned groups of up to 32 frames).  Other cuts are still correct, to rounding.
 * (2) A piece that starts at frame f > 0 must hold, to the left of sample f*H, the N-H history
 *     ROUNDED UP TO WHOLE HOPS: ceil((N-H)/H)*H samples.  Whole hops because per-hop mean removal
 *     (cfg.sub_mean, fft.c:86-96) corrects every history sample by the mean of the hop it arrived
 *     in, so the engine reads complete hops back.  (history_mode ZERO_ALWAYS needs no history, and
 *     no kernel loads any: the gathers of that mode start at the frame's own hop -- a piece may
 *     begin at the first byte of its allocation; tests/test_gpu_round3.py.)
 *     LMP mode adds lmp_av-1 hops: the frames its ring still holds are recomputed, not carried.
 *     The device entries take the stream's VIRTUAL base (address of sample 0), so a piece is
 *     passed as  d_piece - begin*sample_size  with frame indices left global.
 * glfer_hip_frame_range() deals a stream out over `world` ranks by these rules (the arithmetic
 * of glfer_amd/shard.py): contiguous ranges, boundaries on multiples of GLFER_FRAME_ALIGN. */
#define GLFER_FRAME_ALIGN 32
void glfer_hip_frame_range(size_t total_frames, unsigned rank, unsigned world, size_t *first, size_t *count);

typedef struct glfer_hip_plan glfer_hip_plan;

/* fft_init (fft.c:168-187) / mtm_init (mtm.c:88-151) / hparma_init (hparma.c:45-71): builds the
 * window or the DPSS tapers + eigenvalues on the host (double), uploads the device tables.
 * HP-ARMA mode (BASELINE config 5): hparma_do (hparma.c:74-157) per frame -- autocorrelation,
 * the t x (p_e+1) matrix with the reference's row-0 overflow, one-sided Jacobi SVD (util.c:261-386),
 * AR spectrum; window forced rectangular, a/limiter without effect (source.c:369-372). */
int glfer_hip_plan_create(const glfer_hip_config *cfg, glfer_hip_plan **plan_out);
/* fft_close (fft.c:297-306) / mtm_close (mtm.c:242-265) */
void glfer_hip_plan_destroy(glfer_hip_plan *plan);

int glfer_hip_hop(const glfer_hip_plan *plan);      /* H = (int)(N*(1.0-overlap)), fft.c:70 */
int glfer_hip_bins(const glfer_hip_plan *plan);     /* N/2+1, source.c:317                  */
int glfer_hip_num_tapers(const glfer_hip_plan *plan);
/* whole hops in nsamples (wav_fmt.c:119 hands out whole blocks only) */
size_t glfer_hip_num_frames(const glfer_hip_plan *plan, size_t nsamples);

/* Host copies of the tables, for inspection and parity tests.
 * window: n floats (compute_window, fft.c:309-360; all ones in MTM mode).
 * tapers: [kmax+1][n] doubles, unit energy (gl_dpss, g-l_dpss.c:288-347);
 * sig   : kmax+1 doubles, lambda_k - 1 (g-l_dpss.c:342-344). */
int glfer_hip_get_window(const glfer_hip_plan *plan, float *window);
int glfer_hip_get_tapers(const glfer_hip_plan *plan, double *tapers, double *sig);

/* The same two table generators without a plan or a device (pure host code): what
 * fft_init()/mtm_init() compute before anything touches the GPU. */
int glfer_hip_make_window(int window_type, int n, float *window);
int glfer_hip_make_dpss(int n, int kmax, double nw, double *tapers, double *sig);

/* THE hot path: frames [first_frame, first_frame+nframes) of a device-resident stream.
 *   d_stream : device pointer to sample 0 of the stream (format = cfg.sample_format)
 *   nsamples : samples in the stream (for bounds: frame f reads [f*H-(N-H), f*H+H))
 *   d_psd    : device, [nframes][N/2+1] floats, row i = frame first_frame+i
 * Equivalent to calling fft_do+fft_psd (source.c:143-144) or mtm_do (source.c:148)
 * once per hop.  Asynchronous on hip_stream.  With sub_mean the per-hop means are
 * removed inside the estimator kernels where the hop is 2, 4, 8 or 16 sixteenths of the block
 * (overlap 87.5 / 75 / 50 / 0 %; a stream's first ceil((N-H)/H) frames and the remaining
 * forms go through a device copy of the hops involved); the reference mutates the caller's
 * buffer (fft.c:93-95), the device stream is left untouched either way.
 * GLFER_MEAN_PREPASS=1 in the environment forces the copy everywhere (A/B runs). */
int glfer_hip_spectrogram_device(glfer_hip_plan *plan, const void *d_stream, size_t nsamples,
                                 size_t first_frame, size_t nframes, float *d_psd,
                                 void *hip_stream);

/* In LMP mode (GLFER_MODE_LMP, lmp.c:101-181) the same entry writes the detection statistic:
 * per frame the rectangular-window periodogram of the assembled frame (lmp.c:114-125), then per
 * bin mean and variance over the ring of the last lmp_av periodograms (zeros before the stream,
 * slot order as lmp.c:134-149) and the clamped statistic of lmp.c:151-160. */

/* Same, also writing the halfcomplex spectrum of each tapered frame in the layout of
 * fft_radix2.c:75-177 (data[k]=Re X_k, data[N-k]=Im X_k).  FFT mode only: this is
 * what fft_do leaves in params->outbuf.  d_spec: [nframes][N] floats. */
int glfer_hip_spectrum_device(glfer_hip_plan *plan, const void *d_stream, size_t nsamples,
                              size_t first_frame, size_t nframes, float *d_psd, float *d_spec,
                              void *hip_stream);

/* prepare_audio (fft.c:66-165) on its own: what it leaves in params->inbuf_fft for every frame
 * (history, RA9MB, window, limiter) -- lmp.c:101-120 and the scope (g_scope.c:194-197) read it.
 * d_frames: [nframes][N] floats. */
int glfer_hip_prepare_device(glfer_hip_plan *plan, const void *d_stream, size_t nsamples,
                             size_t first_frame, size_t nframes, float *d_frames, void *hip_stream);

/* The harmonic F-test that mtm_do computes beside the spectrum (mtm.c:165-174 mu = transform of
 * the hn-windowed frame; mtm.c:203-210 denominator; mtm.c:222-233 F = k |mu|^2 sum(U0^2) / den).
 * MTM plans only.  d_ftest: [nframes][N/2+1] floats.
 *   mu_live = 0: the reference as built without FFTW -- the transform at mtm.c:173 runs in place
 *                and `mu` stays zero, so F is 0 (NaN where the denominator is 0);
 *   mu_live = 1: mu as the FFTW build computes it (mtm.c:171) -- the statistic as intended.
 * Quirks kept: the Nyquist bin's denominator is never accumulated (x/0), its numerator counts
 * mu[N/2] twice; sums in float with double terms, as the reference's declarations give. */
int glfer_hip_mtm_ftest_device(glfer_hip_plan *plan, const void *d_stream, size_t nsamples,
                               size_t first_frame, size_t nframes, float *d_ftest, int mu_live,
                               void *hip_stream);

/* Host-buffer entry: h_stream goes to the device in chunks through a two-deep ring (two pinned
 * sample buffers, two device buffers each way; uploads on one stream, kernels and downloads on two:
 * a chunk goes up while the previous chunk's rows come down), the PSD rows come back; blocks until done.
 * h_psd in pinned memory (glfer_hip_host_alloc) receives its rows by DMA directly, and an h_stream
 * in pinned memory is uploaded from where it lies; any other memory goes through pinned staging
 * and a host copy (the slower way by 3-4x: the host copy sets the pace).
 * *nframes_out receives glfer_hip_num_frames(nsamples). */
int glfer_hip_spectrogram_host(glfer_hip_plan *plan, const void *h_stream, size_t nsamples,
                               float *h_psd, size_t *nframes_out);

/* Pinned host memory for the ring's ends (source.c / wav_fmt.c side buffers). */
/* (pinned with the calling thread on the CPUs of the current device's NUMA node, its affinity restored afterwards;
 * GLFER_NUMA_BIND=0 turns the placement off) */
void *glfer_hip_host_alloc(size_t bytes);
void glfer_hip_host_free(void *p);

/* The same over several GPUs of one node: the frame range is dealt out with
 * glfer_hip_frame_range() over the devices whose bit is set in device_mask (bit d = HIP device
 * d; cfg->device is ignored), one host thread per GPU, each with its own plan, streams and
 * pinned ring; every GPU reads its hops plus the history rule (2) above from h_stream and writes
 * a disjoint row range of h_psd.  No data moves between GPUs (frames are independent:
 * source.c:130-158 is a loop over hops).  Rows are bit-identical to the one-GPU run. */
int glfer_hip_spectrogram_host_multi(const glfer_hip_config *cfg, unsigned device_mask,
                                     const void *h_stream, size_t nsamples, float *h_psd,
                                     size_t *nframes_out);
/* Host-side placement: every worker thread of the *_multi / *_workers entries is bound, before it makes its
 * plan and pinned ring, to the CPUs of its GPU's NUMA node -- /sys/bus/pci/devices/<bus id>/numa_node and
 * /sys/devices/system/node/node<N>/cpulist, intersected with the CPUs this process may use -- so pinned staging memory,
 * and the first touch of the worker's range of the caller's rows, land beside the GPU; the calling thread's own
 * affinity is restored.  Unknown node (-1), a one-worker call or GLFER_NUMA_BIND=0: nothing is bound.  The two
 * look-ups are exported (sysfs_root NULL = "/sys"): the node of a PCI bus id (-1 = unknown) and a node's CPUs as a bit
 * mask (returns how many, -1 on a missing or malformed list). */
int glfer_hip_numa_node_of_bus_id(const char *bus_id, const char *sysfs_root);
int glfer_hip_numa_node_cpus(int node, const char *sysfs_root, unsigned char *mask, size_t mask_bytes);

/* The same with the workers listed: one host thread + plan + streams + pinned ring per entry of
 * devices[0..nworkers); an ordinal may repeat (workers then share that GPU -- how a one-GPU machine
 * runs, and tests, the multi-worker path).  _multi is this with the mask's devices, one worker each. */
int glfer_hip_spectrogram_host_workers(const glfer_hip_config *cfg, const int *devices, int nworkers,
                                       const void *h_stream, size_t nsamples, float *h_psd,
                                       size_t *nframes_out);

/* ---- ingest: the file source of source.c:118-128 / wav_fmt.c:45-121 ------------------------
 * The canonical 44-byte RIFF/WAVE header of wav_fmt.h:34-52, read with fixed-width fields
 * (the reference's struct uses u_long and mis-parses every file on LP64 hosts).  As in the
 * reference only PCM (format 1) with 8 or 16 bits per sample is accepted and the channel
 * count is not interpreted: interleaved channels are treated as one sample stream.  The RIFF
 * chunks are walked ("fmt ", then "data"; "LIST" / "fact" / ... skipped), so a file with other
 * chunks before or after its samples is read correctly -- the reference's fixed 44-byte struct
 * (wav_fmt.h:34-52) would play them as samples; a file that cannot be walked is read its way. */
typedef struct glfer_wav_info {
  int format;            /* 1 = PCM                                  wav_fmt.h:42 */
  int channels;          /* "modus": 1 mono, 2 stereo                wav_fmt.h:43 */
  int sample_rate;       /* sample_fq                                wav_fmt.h:44 */
  int bits_per_sample;   /* bit_p_spl: 8 or 16                       wav_fmt.h:47 */
  size_t data_offset;    /* first byte of the "data" chunk's payload (44 in the reference's fixed layout) */
  size_t nsamples;       /* samples in the data chunk                             */
  size_t data_bytes;     /* bytes in the data chunk (to the end of the file when the chunk size is 0 / ~0 / too long) */
} glfer_wav_info;
int glfer_hip_wav_probe(const char *path, glfer_wav_info *info);

/* Whole-file spectrogram: reads `path` hop block by hop block through two pinned host
 * buffers (the next block is read from the file while the GPU works on the current one),
 * uploads the raw PCM with hipMemcpyAsync -- the conversion of wav_fmt.c:104-117 happens in
 * the kernel's gather -- and writes frame rows to h_psd ([max_frames][N/2+1], host).
 * plan->sample_format must match the file (8 bit: GLFER_SAMPLES_U8, 16 bit: _S16).
 * chunk_frames = frames per upload (0 = default 16384). */
int glfer_hip_spectrogram_wav(glfer_hip_plan *plan, const char *path, float *h_psd, size_t max_frames,
                              size_t *nframes_out, size_t chunk_frames);
/* flags = GLFER_WAV_PARTIAL_TAIL: a file whose data is not a whole number of hop blocks yields
 * one more frame, as in the reference: wav_read (wav_fmt.c:102-119) converts the samples of the
 * short last read over the STALE rest of its buffer -- the previous block as the estimator left it
 * (prepare_audio removes the hop's mean in place, fft.c:93-95) -- and reports a block.  An odd last
 * byte of a 16-bit file is dropped (n_read/2 samples).  Without the flag (and in
 * glfer_hip_spectrogram_wav) whole blocks only.  Not available in LMP mode. */
#define GLFER_WAV_PARTIAL_TAIL 1u
int glfer_hip_spectrogram_wav_ex(glfer_hip_plan *plan, const char *path, float *h_psd, size_t max_frames,
                                 size_t *nframes_out, size_t chunk_frames, unsigned flags);

/* The same for frames [first_frame, first_frame + max_frames) of the file only: rows to h_psd[0 ..).
 * (The per-hop shims' read-ahead walks a file in such windows, glfer_compat.h.) */
int glfer_hip_spectrogram_wav_range(glfer_hip_plan *plan, const char *path, size_t first_frame, size_t max_frames,
                                    float *h_psd, size_t *nframes_out, size_t chunk_frames, unsigned flags);

/* BASELINE config 4 as worded ("1-hour 48 kHz WAV, frame-batch sharded across 8 x MI355X"): the
 * file's frames dealt out over the GPUs named in device_mask (glfer_hip_frame_range: contiguous
 * ranges, boundaries on multiples of GLFER_FRAME_ALIGN), one host thread + plan + pinned ring per
 * GPU, every worker reading its own part of the file -- its hops and the history halo in front of
 * them -- through its own handle (source.c:193, wav_fmt.c:45-121); rows land in disjoint ranges of
 * h_psd [frames][N/2+1]; no collective.  _workers: the workers listed; an ordinal may repeat (several
 * workers then share that GPU -- how a one-GPU box exercises the path).  cfg->sample_format must
 * match the file; flags as glfer_hip_spectrogram_wav_ex (the partial block belongs to the last worker). */
int glfer_hip_spectrogram_wav_multi(const glfer_hip_config *cfg, unsigned device_mask, const char *path, float *h_psd,
                                    size_t max_frames, size_t *nframes_out, unsigned flags);
int glfer_hip_spectrogram_wav_workers(const glfer_hip_config *cfg, const int *devices, int nworkers, const char *path,
                                      float *h_psd, size_t max_frames, size_t *nframes_out, unsigned flags);

/* ---- a persistent set of workers for the *_workers / *_multi entries (round 5) -------------------------------------------------
 * The stateless entries above make everything a worker needs inside the call.  For BASELINE config 4 as worded -- a 1-hour WAV,
 * 346 MB, 10 546 frames, 1.2 ms of kernels -- that set-up WAS the call: DPSS tapers (0.1-0.3 s at N = 16384, kept by the library
 * after the first plan), tables uploaded per plan, 30-40 ms of pinned and device buffers per worker unless the device's one parked
 * ring was free.  A handle keeps, per worker: its plan (tables on its GPU) and ITS OWN chunk ring, sized at creation for jobs of
 * hint_frames frames in all (0 = the default chunk); calls through the handle allocate nothing.  One call at a time per handle
 * (a second one waits).  The stateless entries keep the two most recently used handles themselves (same configuration and worker
 * list -> same handle; GLFER_SCRATCH_CACHE=0 or glfer_hip_scratch_limit(0): nothing is kept, glfer_hip_scratch_trim(device, 0)
 * drops them), so repeated calls are fast there too -- what a handle adds is the set-up OUTSIDE the first call.
 * phases (may be NULL): where the call's time went -- per field the largest value over the workers (they run side by side),
 * seconds: set-up inside the call (ring look-up, any allocation), reading / copying the samples into pinned memory, uploads,
 * kernels (from a chunk's upload end to its kernels' end), downloads -- sums over a worker's chunks, which OVERLAP one another, so
 * the fields add up to more than wall_s -- and the call's wall time; chunks = chunks of all workers. */
typedef struct glfer_hip_phases {
  double setup_s, read_s, h2d_s, kernel_s, d2h_s, wall_s;
  unsigned chunks;
} glfer_hip_phases;
typedef struct glfer_hip_workers glfer_hip_workers;
int glfer_hip_workers_create(const glfer_hip_config *cfg, const int *devices, int nworkers, size_t hint_frames,
                             glfer_hip_workers **out);
void glfer_hip_workers_destroy(glfer_hip_workers *w);
/* glfer_hip_spectrogram_wav_workers / glfer_hip_spectrogram_host_workers through a handle */
int glfer_hip_workers_spectrogram_wav(glfer_hip_workers *w, const char *path, float *h_psd, size_t max_frames,
                                      size_t *nframes_out, unsigned flags, glfer_hip_phases *phases);
int glfer_hip_workers_spectrogram_host(glfer_hip_workers *w, const void *h_stream, size_t nsamples, float *h_psd,
                                       size_t *nframes_out, glfer_hip_phases *phases);

/* K0 on its own: per-hop mean removal (fft.c:86-96).  d_out[i] = sample(d_in[i]) - mean of the
 * hop i belongs to; nhops hops of `hop` samples each.  (The spectrogram entries apply it
 * themselves when cfg.sub_mean is set; this entry serves the per-hop shims, which must hand
 * the corrected hop back to the caller as the reference does.) */
int glfer_hip_submean_device(const void *d_in, float *d_out, int hop, size_t nhops, int sample_format,
                             void *hip_stream);
/* The same with every hop summed in the reference's own order (GLFER_SUBMEAN_EXACT above): what the
 * per-hop shim's prepare_audio() uses. */
int glfer_hip_submean_exact_device(const void *d_in, float *d_out, int hop, size_t nhops, int sample_format,
                                   void *hip_stream);

/* compute_floor (fft.c:240-294) for a batch of PSD rows on the device.
 * d_stats: [nframes][4] floats = {sig (max bin), floor, peak value, peak bin as float}. */
int glfer_hip_floor_device(const float *d_psd, size_t nframes, int bins, float *d_stats,
                           void *hip_stream);
/* the same over rows `pitch` floats apart (cfg.psd_pitch; pitch >= bins) */
int glfer_hip_floor_device_pitched(const float *d_psd, size_t nframes, int bins, int pitch, float *d_stats,
                                   void *hip_stream);

/* update_avg_* (avg.c:108-298) over consecutive PSD rows: the sliding sum over the last
 * `depth` frames per bin in [minbin,maxbin) and the three output normalisations.
 * The state starts empty (alloc_avg, avg.c:38-60) at row 0 of the call.
 *   d_avg  : [nframes][n_out] doubles (avgdata->avg after each frame; 1e-15 out of band)
 *   d_ret  : [nframes][4] doubles = {return value, peak bin, variance (sumavg), effdepth}
 */
int glfer_hip_avg_device(int avg_mode, const float *d_psd, size_t nframes, int bins, int n_out,
                         int depth, int minbin, int maxbin, int max0, double *d_avg,
                         double *d_ret, void *hip_stream);
/* Estimator AND moving average in one call: frames [first_frame, first_frame + nframes) of the stream as
 * glfer_hip_spectrogram_device computes them, followed by update_avg_* over those rows with the averaging state empty at
 * first_frame -- what source.c:141-158 and g_main.c:1153-1183 do per hop.  d_avg [nframes][n_out] and d_ret [nframes][4] as
 * glfer_hip_avg_device writes them (d_ret may be NULL); d_psd [nframes][N/2+1] receives the PSD rows themselves, or NULL: they
 * are then never stored.  n_out >= N/2+1 (the reference's avgdata is N wide, source.c:312).
 * The plain average (GLFER_AVG_PLAIN) over a window of up to four frames (the reference's default depth, glfer.c:295-296) of
 * a periodogram plan (FFT mode, N = 512..4096, no RA9MB / limiter, history from the stream; mean removal off, or the reference's own --
 * cfg.sub_mean = 1 with a hop of 2, 4, 8 or 16 sixteenths of the block: the hop means are taken first and given to the kernel) is taken INSIDE the
 * estimator launch, on the |X|^2 values while they are in registers: per frame 4 H bytes in and 8 n_out bytes out, no PSD row
 * in memory.  Every other case runs the two launches.  d_avg is identical, double for double, to glfer_hip_avg_device over
 * the rows (a window's sum of float bins is exact in a double unless a bin spans more than ~2^26 within the window); the band
 * mean in d_ret [.][0] is the same sum taken over the lanes in another order (equal to ~1e-15 relative), the peak bin equal.
 * GLFER_AVG_FUSED=0 in the environment forces the two launches (A/B runs). */
int glfer_hip_spectrogram_avg_device(glfer_hip_plan *plan, const void *d_stream, size_t nsamples, size_t first_frame,
                                     size_t nframes, int avg_mode, int depth, int minbin, int maxbin, int max0, int n_out,
                                     float *d_psd, double *d_avg, double *d_ret, void *hip_stream);

/* The sliding sums alone: d_cum [nframes][n_out] = avgdata->cum after each frame (avg.c:114-127);
 * columns outside [minbin, maxbin) are left untouched. */
int glfer_hip_avg_cum_device(const float *d_psd, size_t nframes, int bins, int n_out, int depth,
                             int minbin, int maxbin, double *d_cum, void *hip_stream);

/* ---- display mapping: main_window_draw's column loop (g_main.c:1099-1236) over a batch ---- */
enum { GLFER_SCALE_LIN = 0, GLFER_SCALE_LIN_MAX0, GLFER_SCALE_LOG, GLFER_SCALE_LOG_MAX0 };   /* glfer.h:43 */
enum { GLFER_PAL_HSV = 0, GLFER_PAL_THRESH, GLFER_PAL_COOL, GLFER_PAL_HOT, GLFER_PAL_BW,
       GLFER_PAL_BONE, GLFER_PAL_COPPER, GLFER_PAL_OTD };                                      /* glfer.h:47 */

typedef struct {
  int scale_type;          /* opt.scale_type                                                   */
  int autoscale;           /* opt.autoscale                                                    */
  float overlap;           /* opt.data_blocks_overlap (first-buffer correction, g_main.c:1114) */
  float max_level_db;      /* opt.max_level_db / opt.min_level_db: used when autoscale is off  */
  float min_level_db;
  float thr_level;         /* opt.thr_level, percent                                           */
  int palette;             /* opt.palette                                                      */
  /* the state main_window_draw keeps between columns; updated by every call */
  int first_buffer;        /* glfer.first_buffer                                               */
  float display_max_lvl;   /* the two function statics of g_main.c:1081                        */
  float display_min_lvl;
  int psd_pitch;           /* floats from one row of d_psd to the next in glfer_hip_display_device / glfer_hip_waterfall_device /
                              glfer_hip_waterfall_map_device (0 = dense: `bins`); cfg.psd_pitch of the plan that wrote the rows */
} glfer_hip_display;

/* set_palette (g_main.c:651-762): 256 RGB triplets into host memory. */
int glfer_hip_palette(int palette, unsigned char colortab[768]);

/* One call = `nframes` consecutive calls of the mapping part of main_window_draw.
 *   d_psd    : [nframes][bins] float PSD rows (opt.averaging == NO_AVG), or NULL
 *   d_avg    : [nframes][bins] double rows of avgdata.avg (any averaging mode), or NULL
 *              -- exactly one of the two is given
 *   d_stats  : [nframes][4] as written by glfer_hip_floor_device (sig, floor are used)
 *   d_rgb    : [nframes][bins][3] bytes, pixel i of a column = bin bins-1-i (rgbbuf, n_zoom 1)
 *   d_lev    : [nframes][bins] shorts = levbuf column, or NULL
 *   d_levels : [nframes][4] floats = {display_max, display_min, display_max_lvl,
 *              display_min_lvl} used for each column, or NULL
 * disp->first_buffer / display_*_lvl are read as the incoming state and updated to the state
 * after the last column (the call synchronises the stream to read them back). */
int glfer_hip_display_device(glfer_hip_display *disp, const float *d_psd, const double *d_avg,
                             const float *d_stats, size_t nframes, int bins, unsigned char *d_rgb,
                             short *d_lev, float *d_levels, void *hip_stream);

/* compute_floor + update_avg_* + the mapping for a batch of PSD rows in one call (statistics,
 * level tracking, [moving average +] pixel map).  A single pass over a row is not possible: the level
 * tracking is a chain over the columns fed by every column's statistics, so a row is read once for
 * those and once to be mapped.  With averaging the averages are taken INSIDE the mapping kernel (the
 * levels come from compute_floor of the PSD rows, g_main.c:1109-1139, not from the average): no
 * averaged rows in memory at all.  Where that kernel does not apply (bands wider than 33 x 256 bins,
 * windows much deeper than its frame chunks; or GLFER_WATERFALL_FUSED=0) the averaged rows are scratch,
 * at most 4 GiB at a time, and the batch is walked in tiles of that many rows
 * (GLFER_WATERFALL_TILE=<rows> overrides).  avg_mode 0 = NO_AVG (the PSD rows are mapped), else
 * GLFER_AVG_* with depth/minbin/maxbin/max0 as glfer_hip_avg_device (the state starts empty at row
 * 0).  d_stats: [nframes][4] or NULL.  disp carries the level-tracking state in and out; the call
 * synchronises the stream per tile. */
int glfer_hip_waterfall_device(glfer_hip_display *disp, int avg_mode, int depth, int minbin, int maxbin,
                               int max0, const float *d_psd, size_t nframes, int bins,
                               unsigned char *d_rgb, short *d_lev, float *d_stats, void *hip_stream);

/* The two halves of glfer_hip_waterfall_device for a waterfall whose columns live on several GPUs
 * (or are computed piece by piece).  The level tracking of main_window_draw (g_main.c:1111-1124) is
 * ONE chain over all columns, fed by the 16 bytes of compute_floor statistics per column; everything
 * else is per column.  So each GPU computes its rows and their statistics
 * (glfer_hip_spectrogram_device + glfer_hip_floor_device), the statistics meet on the host,
 *   glfer_hip_levels_host(disp, h_stats [n][4], n, h_levels [n][4], device)
 * walks them once (on `device`; disp's carried state in and out, exactly as
 * glfer_hip_display_device would leave it), and each GPU maps its own rows with its slice of the levels:
 *   glfer_hip_waterfall_map_device(disp, avg_mode, depth, minbin, maxbin, max0, d_batch, first, n,
 *                                  bins, d_levels [n][4], d_rgb, d_lev, stream)
 * maps rows [first, first + n) of the device batch d_batch ([.][bins] floats).  With averaging the
 * moving sums of the first rows reach back into the batch's rows BEFORE `first` -- a piece that is
 * not the start of the waterfall carries `depth` recomputed rows in front (recompute,
 * do not exchange); the state is empty at row 0 of the batch (alloc_avg, avg.c:38-60).  disp is
 * not modified. */
int glfer_hip_levels_host(glfer_hip_display *disp, const float *h_stats, size_t nframes, float *h_levels, int device);
int glfer_hip_waterfall_map_device(const glfer_hip_display *disp, int avg_mode, int depth, int minbin, int maxbin,
                                   int max0, const float *d_batch, size_t first, size_t nframes, int bins,
                                   const float *d_levels, unsigned char *d_rgb, short *d_lev, void *hip_stream);

/* Host samples -> waterfall columns: estimator, compute_floor and the display mapping on the
 * device, chunked through the same ring as glfer_hip_spectrogram_host; what comes back is
 * h_rgb [frames][bins][3] (and h_lev [frames][bins] shorts, or NULL): 3-5 bytes per bin over PCIe
 * instead of 4, and pixels instead of PSD rows.  disp carries the level-tracking state. */
int glfer_hip_waterfall_host(glfer_hip_plan *plan, glfer_hip_display *disp, const void *h_stream,
                             size_t nsamples, unsigned char *h_rgb, short *h_lev, size_t *nframes_out);

/* The waterfall over several GPUs (samples from a host array, or a WAV file): h_rgb [frames][bins][3]
 * (+ h_lev) identical to the one-GPU call's.  Three phases: every worker computes its frames' rows,
 * which stay on its GPU, and their compute_floor statistics; the statistics (16 bytes per column)
 * meet on the host and ONE walk gives every column its levels (glfer_hip_levels_host -- the level
 * tracking of g_main.c:1111-1124 is a chain over all columns); every worker maps its own rows with
 * its slice of the levels.  avg_mode != 0: update_avg_* (avg.c:108-298) inside the map; a worker whose
 * range starts at frame f > 0 RECOMPUTES the `depth` rows in front of it instead of receiving them
 *  -- the moving average crosses worker boundaries without an exchange.  disp carries
 * the level-tracking state in and out. */
int glfer_hip_waterfall_host_workers(const glfer_hip_config *cfg, const int *devices, int nworkers,
                                     glfer_hip_display *disp, int avg_mode, int depth, int minbin, int maxbin, int max0,
                                     const void *h_stream, size_t nsamples, unsigned char *h_rgb, short *h_lev,
                                     size_t *nframes_out);
int glfer_hip_waterfall_wav_workers(const glfer_hip_config *cfg, const int *devices, int nworkers,
                                    glfer_hip_display *disp, int avg_mode, int depth, int minbin, int maxbin, int max0,
                                    const char *path, size_t max_frames, unsigned char *h_rgb, short *h_lev,
                                    size_t *nframes_out, unsigned flags);
int glfer_hip_waterfall_wav_multi(const glfer_hip_config *cfg, unsigned device_mask, glfer_hip_display *disp,
                                  int avg_mode, int depth, int minbin, int maxbin, int max0, const char *path,
                                  size_t max_frames, unsigned char *h_rgb, short *h_lev, size_t *nframes_out,
                                  unsigned flags);

/* Device scratch the library keeps between calls.  Per-call scratch of 16 MiB and more (averaged
 * rows of a staged waterfall tile, the mean-corrected copy of a stream, big-block and LMP / F-test
 * spectra) comes from up to six blocks per device that the library allocates with hipMalloc and
 * KEEPS -- the stream-ordered pool costs milliseconds to seconds per GB-sized request on this stack
 * (profiles/r02_scratch_tail.txt).  That memory is invisible to the host application's own
 * allocator, so it is bounded and can be given back:
 *   glfer_hip_scratch_limit(bytes)        cap per device on kept bytes (default 16 GiB, or
 *                                         GLFER_SCRATCH_CAP_MB; 0 keeps nothing between calls).  Idle
 *                                         blocks are freed before a new one would exceed the cap, and
 *                                         all of them when a hipMalloc for a new block fails.
 *   glfer_hip_scratch_trim(device, keep)  frees idle kept blocks of `device`, smallest first, until at
 *                                         most `keep` bytes stay; waits for the work recorded on a
 *                                         block before freeing it; blocks in use by a running call
 *                                         stay.  Returns the bytes freed.  (The reference frees its
 *                                         buffers in fft_close, fft.c:297-306; a GUI host calls this
 *                                         when a waterfall is closed or the block size changes.)
 *   glfer_hip_scratch_held(device)        bytes kept right now.
 * GLFER_SCRATCH_CACHE=0 in the environment disables keeping altogether.
 * Besides the scratch blocks, ONE idle chunk ring of the host / file entries (two pinned sample buffers, two device buffers each
 * way; at most 2 GiB) is parked per device when its plan is destroyed and taken by the next plan that needs one -- the *_multi /
 * *_workers entries make a plan per worker and call, and allocating a ring costs 30-40 ms; glfer_hip_scratch_trim(device, 0)
 * frees it too.  The parked ring obeys the same switches as the blocks: nothing is parked with GLFER_SCRATCH_CACHE=0 or when the
 * ring is larger than the cap, glfer_hip_scratch_limit() below its size frees it, glfer_hip_scratch_held() counts it. */
size_t glfer_hip_scratch_trim(int device, size_t keep_bytes);
size_t glfer_hip_scratch_held(int device);
void glfer_hip_scratch_limit(size_t bytes);

const char *glfer_hip_strerror(int code);
/* text of the last HIP error seen by this thread ("" if none) */
const char *glfer_hip_last_hip_error(void);
/* library / kernel build description, e.g. "glfer_hip 0.1 gfx950" */
const char *glfer_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GLFER_HIP_H */
