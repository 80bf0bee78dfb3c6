/* The reference's file loop (source.c:112-171) in a program that, like glfer.c:56-57, defines the globals
 * `opt` and `glfer` itself -- so the library sees glfer.scope_window and glfer.first_buffer as glfer sets them.
 * Two things the plain loop of c_compat_wav_demo.c does not do:
 *   - the scope window open for some hops (g_scope.c:194-197 reads inbuf_fft of every hop: those hops take the
 *     per-hop path, the hops around them come from the read-ahead, and the hand-over must be exact both ways);
 *   - leaving in the middle of the file WITHOUT close_wav_file / fft_close, as /Source/Quit does
 *     (g_main.c:115 -> gtk_main_quit): the process must end with its exit code, not with SIGABRT.
 * Built and run by tests/test_gpu_round4.py.
 *   usage: c_compat_scope_demo N overlap in.wav out.f32 scope_from scope_to [scope_from2 scope_to2 [quit_hop]]
 *   (the scope window is "open" for hops scope_from <= hop < scope_to and scope_from2 <= hop < scope_to2)
 * prints: hops, hops served from the read-ahead */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "glfer_compat.h"

opt_t opt;
glfer_t glfer;

int main(int argc, char **argv)
{
  if (argc < 7)
    return 2;
  const int n = atoi(argv[1]);
  const float overlap = (float)atof(argv[2]);
  const long s0 = atol(argv[5]), s1 = atol(argv[6]);
  const long t0 = argc > 8 ? atol(argv[7]) : -1, t1 = argc > 8 ? atol(argv[8]) : -1;
  const long quit_hop = argc > 9 ? atol(argv[9]) : -1;
  const int n_eff = n * (1.0 - overlap);                    /* source.c:114 */
  int speed = 0, n_blocks = 0, scope_stands_for_a_widget = 0;
  long hops = 0;
  float *audio_buf = NULL, *psd = malloc(sizeof(float) * (size_t)(n / 2 + 1));
  FILE *out = fopen(argv[4], "wb");
  fft_params_t fp;
  if (!out || !psd)
    return 3;
  memset(&opt, 0, sizeof opt);
  memset(&glfer, 0, sizeof glfer);
  opt.autoscale = 1;                                        /* glfer.c:275 */
  glfer.first_buffer = 1;                                   /* g_main.c:990 */
  fp.n = n; fp.window_type = HANNING_WINDOW; fp.overlap = overlap; fp.a = 0.0f; fp.limiter = 0;
  fft_init(&fp);
  open_wav_file(argv[3], n_eff, &speed);                    /* source.c:193 */
  for (;;) {
    wav_read(&audio_buf, &n_blocks);                        /* source.c:119 */
    if (n_blocks == 0)
      break;
    if (hops == quit_hop) {
      printf("%ld %lu\n", hops, glfer_compat_readahead_served);
      fflush(stdout);
      fclose(out);
      exit(0);                                              /* /Source/Quit: no close_audio, no fft_close */
    }
    glfer.scope_window = ((hops >= s0 && hops < s1) || (hops >= t0 && hops < t1)) ? (void *)&scope_stands_for_a_widget : NULL;
    fft_do(audio_buf, &fp);                                 /* source.c:143-144 */
    fft_psd(psd, NULL, &fp);
    fwrite(psd, sizeof(float), (size_t)(n / 2 + 1), out);
    glfer.first_buffer = 0;                                 /* the drawer, with autoscale (g_main.c:1111-1120) */
    hops++;
  }
  close_wav_file();
  fft_close(&fp);
  fclose(out);
  printf("%ld %lu\n", hops, glfer_compat_readahead_served);
  return 0;
}
