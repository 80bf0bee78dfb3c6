/* A stand-in for the reference objects that STAY in glfer when libglfer_compat.so replaces
 * fft.o fft_radix2.o mtm.o g-l_dpss.o avg.o hparma.o lmp.o wav_fmt.o: it references every symbol the kept
 * objects take from the dropped ones -- functions AND data -- the way those files do:
 *   glfer.c:   init_avg alloc_avg delete_avg fft_close                      (glfer.c:143, 327-331)
 *   source.c:  fft_* mtm_* hparma_* lmp_* open_wav_file wav_read close_wav_file alloc_avg delete_avg
 *              (source.c:119, 141-158, 193, 282-404)
 *   g_main.c:  compute_floor update_avg_plain update_avg_sumavg update_avg_sumextreme   (g_main.c:1109, 1153-1183)
 *   g_options.c: fft_windows[] num_fft_windows alloc_avg delete_avg         (g_options.c:47-48, 329-330, 367-368, 579-583)
 * and defines the two globals the library reads (glfer.c:56-57).  Linked with -Wl,--no-undefined by
 * tests/test_host_logic.py::test_kept_objects_link_against_the_library; never run with a GPU call (main returns
 * before any of them unless an argument is given). */
#include <stdio.h>
#include <string.h>
#include "glfer_compat.h"

opt_t opt;
glfer_t glfer;
avg_data_t avgdata;                                        /* glfer.c:62 */

extern fft_window_t fft_windows[];                         /* g_options.c:47-48, verbatim declarations */
extern int num_fft_windows;

int main(int argc, char **argv)
{
  int i, ok = num_fft_windows == 8;
  static const char *want[] = {"/Hanning", "/Blackman", "/Gaussian", "/Welch", "/Bartlett", "/Rectangular", "/Hamming", "/Kaiser"};
  (void)argv;
  for (i = 0; i < num_fft_windows && i < 8; i++)          /* g_options.c:579-583 walks the table like this */
    ok = ok && strcmp(fft_windows[i].name, want[i]) == 0 && fft_windows[i].type == i;
  printf("%d\n", ok);
  if (argc < 2)
    return ok ? 0 : 1;
  /* referenced, not executed in the CPU test */
  {
    fft_params_t fp; mtm_params_t mp; hparma_params_t hp; lmp_params_t lp;
    float *buf = 0, psd[8], sig, flo, peak; unsigned int pb; int n = 0, speed = 0, peakbin = 0; double var = 0;
    memset(&fp, 0, sizeof fp); memset(&mp, 0, sizeof mp); memset(&hp, 0, sizeof hp); memset(&lp, 0, sizeof lp);
    init_avg(&avgdata); alloc_avg(&avgdata, 8, 2);
    update_avg_plain(&avgdata, 8, psd, 0, 8, &peakbin);
    update_avg_sumextreme(&avgdata, 8, psd, 0, 0, 8, &peakbin);
    update_avg_sumavg(&avgdata, 8, psd, 0, 0, 8, &peakbin, &var);
    delete_avg(&avgdata);
    compute_floor(psd, 8, &sig, &flo, &peak, &pb);
    fft_init(&fp); fft_do(buf, &fp); fft_psd(psd, 0, &fp); prepare_audio(buf, &fp); fft_close(&fp);
    mtm_init(&mp); mtm_do(buf, psd, 0, &mp); mtm_close(&mp);
    hparma_init(&hp); hparma_do(buf, psd, 0, &hp); hparma_close(&hp);
    lmp_init(&lp); lmp_do(buf, psd, 0, &lp); lmp_close(&lp);
    open_wav_file(argv[1], 512, &speed); wav_read(&buf, &n); close_wav_file();
  }
  return 0;
}
