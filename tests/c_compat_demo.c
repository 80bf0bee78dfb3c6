/* A plain C caller of the reference's own entry points, linked against libglfer_compat.so:
 * what source.c's read loop does per hop (source.c:141-158).  Built and run by
 * tests/test_gpu_compat.py::test_c_program_against_the_shim.
 *   usage: c_compat_demo fft|mtm N overlap in.f32 out.f32 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "glfer_compat.h"

int main(int argc, char **argv)
{
  if (argc != 6)
    return 2;
  const int mtm = strcmp(argv[1], "mtm") == 0;
  const int n = atoi(argv[2]);
  const float overlap = (float)atof(argv[3]);
  const int hop = (int)(n * (1.0 - overlap));
  FILE *in = fopen(argv[4], "rb"), *out = fopen(argv[5], "wb");
  float *buf = malloc(sizeof(float) * (size_t)hop), *psd = malloc(sizeof(float) * (size_t)(n / 2 + 1));
  fft_params_t fp;
  mtm_params_t mp;
  if (!in || !out || !buf || !psd)
    return 3;
  glfer_compat_autoscale = 1;
  glfer_compat_first_buffer = 1;
  if (mtm) {
    mp.fft.n = n; mp.fft.window_type = RECTANGULAR_WINDOW; mp.fft.overlap = overlap; mp.fft.a = 0.0f; mp.fft.limiter = 0;
    mp.w = 2.5f; mp.kmax = 4;
    mtm_init(&mp);
  } else {
    fp.n = n; fp.window_type = HANNING_WINDOW; fp.overlap = overlap; fp.a = 0.0f; fp.limiter = 0;
    fft_init(&fp);
  }
  while (fread(buf, sizeof(float), (size_t)hop, in) == (size_t)hop) {
    if (mtm) {
      mtm_do(buf, psd, NULL, &mp);
    } else {
      fft_do(buf, &fp);
      fft_psd(psd, NULL, &fp);
    }
    fwrite(psd, sizeof(float), (size_t)(n / 2 + 1), out);
    glfer_compat_first_buffer = 0;            /* the drawer clears it after the first column (g_main.c:1120) */
  }
  if (mtm) mtm_close(&mp); else fft_close(&fp);
  fclose(in);
  fclose(out);
  return 0;
}
