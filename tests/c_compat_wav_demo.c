/* The reference's file loop (source.c:112-171: wav_read -> fft_do/fft_psd | mtm_do -> draw) as a plain C
 * program over the entry points of libglfer_compat.so -- reader included (wav_fmt.h:24-26).
 * Built and run by tests/test_gpu_round3.py::test_file_loop_runs_from_the_read_ahead.
 *   usage: c_compat_wav_demo fft|mtm N overlap autoscale readahead in.wav out.f32 [max_hops [touch_hop [keep_rows]]]
 *   (keep_rows: only the first keep_rows rows are written to out.f32 -- the rate runs time the loop, not the disk;
 *    touch_hop: the program changes a sample of that block after wav_read, as a filter in front of the estimator would)
 * prints: hops, seconds in the loop, hops served from the read-ahead, a checksum, seconds until the first column */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "glfer_compat.h"

static double now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
  if (argc < 8)
    return 2;
  const int mtm = strcmp(argv[1], "mtm") == 0;
  const int n = atoi(argv[2]);
  const float overlap = (float)atof(argv[3]);
  const int autoscale = atoi(argv[4]);
  const long max_hops = argc > 8 ? atol(argv[8]) : -1;
  const long touch_hop = argc > 9 ? atol(argv[9]) : -1;
  const long keep_rows = argc > 10 ? atol(argv[10]) : -1;
  double check = 0.0, t_first = 0.0;
  const int n_eff = n * (1.0 - overlap);                    /* source.c:114 */
  int speed = 0, n_blocks = 0;
  long hops = 0;
  float *audio_buf = NULL, *psd = malloc(sizeof(float) * (size_t)(n / 2 + 1));
  FILE *out = fopen(argv[7], "wb");
  fft_params_t fp;
  mtm_params_t mp;
  if (!out || !psd)
    return 3;
  glfer_compat_readahead = atoi(argv[5]);
  glfer_compat_autoscale = autoscale;                       /* opt.autoscale: read by *_init (fft.c:186) */
  glfer_compat_first_buffer = 1;                            /* g_main.c:990 */
  if (mtm) {
    mp.fft.n = n; mp.fft.window_type = RECTANGULAR_WINDOW; mp.fft.overlap = overlap; mp.fft.a = 0.0f; mp.fft.limiter = 0;
    mp.w = 2.5f; mp.kmax = 4;
    mtm_init(&mp);
  } else {
    fp.n = n; fp.window_type = HANNING_WINDOW; fp.overlap = overlap; fp.a = 0.0f; fp.limiter = 0;
    fft_init(&fp);
  }
  open_wav_file(argv[6], n_eff, &speed);                    /* source.c:193 */
  const double t0 = now();
  for (;;) {
    wav_read(&audio_buf, &n_blocks);                        /* source.c:119 */
    if (n_blocks == 0 || hops == max_hops)
      break;
    if (hops == touch_hop)
      audio_buf[3] += 0.25f;
    if (mtm) {
      mtm_do(audio_buf, psd, NULL, &mp);                    /* source.c:148 */
    } else {
      fft_do(audio_buf, &fp);                               /* source.c:143-144 */
      fft_psd(psd, NULL, &fp);
    }
    if (keep_rows < 0 || hops < keep_rows)
      fwrite(psd, sizeof(float), (size_t)(n / 2 + 1), out); /* stands for main_window_draw(psdbuf) */
    check += psd[hops % (n / 2 + 1)];                        /* every row is touched */
    if (hops == 0)
      t_first = now() - t0;                                 /* the first column: plan, ring and first window */
    if (autoscale)
      glfer_compat_first_buffer = 0;                        /* the drawer clears it, with autoscale only (g_main.c:1111-1120) */
    hops++;
  }
  const double dt = now() - t0;
  close_wav_file();
  if (mtm) mtm_close(&mp); else fft_close(&fp);
  fclose(out);
  printf("%ld %.6f %lu %g %.6f\n", hops, dt, glfer_compat_readahead_served, check, t_first);
  return 0;
}
