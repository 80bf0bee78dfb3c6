/* BASELINE config 4's entry from plain C: a WAV file through a kept set of workers (include/glfer_hip.h, round 5) -- what a
 * multi-GPU source.c (source.c:193 + the read loop of source.c:112-171 as one call) would do once per file.
 * Built (gcc, C99, no HIP headers) and run by tests/test_gpu_round5.py::test_c_program_over_a_workers_handle.
 *   usage: c_workers_demo mtm|fft N overlap sub_mean workers in.wav out.f32
 * prints: frames, calls, phases of the last call */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "glfer_hip.h"

int main(int argc, char **argv)
{
  if (argc < 8)
    return 2;
  if (glfer_hip_abi_version() != GLFER_HIP_ABI) {           /* INTEGRATION.md, "ABI version" */
    fprintf(stderr, "library ABI %d, header %d\n", glfer_hip_abi_version(), GLFER_HIP_ABI);
    return 3;
  }
  glfer_hip_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.mode = strcmp(argv[1], "mtm") == 0 ? GLFER_MODE_MTM : GLFER_MODE_FFT;
  cfg.n = atoi(argv[2]);
  cfg.overlap = (float)atof(argv[3]);
  cfg.window_type = GLFER_WIN_HANNING;
  cfg.sub_mean = atoi(argv[4]);
  cfg.history_mode = GLFER_HISTORY_ZERO_FIRST;
  cfg.mtm_w = 2.5f;
  cfg.mtm_k = 4;
  cfg.sample_format = GLFER_SAMPLES_S16;
  const int nworkers = atoi(argv[5]);
  int devices[16] = {0};
  glfer_wav_info wi;
  if (nworkers < 1 || nworkers > 16 || glfer_hip_wav_probe(argv[6], &wi) != GLFER_OK || wi.bits_per_sample != 16)
    return 4;
  const int hop = (int)(cfg.n * (1.0 - cfg.overlap));      /* fft.c:70 */
  const size_t frames = wi.nsamples / (size_t)hop, bins = (size_t)cfg.n / 2 + 1;
  glfer_hip_workers *w = NULL;
  int rc = glfer_hip_workers_create(&cfg, devices, nworkers, frames, &w);
  if (rc != GLFER_OK) {
    fprintf(stderr, "workers_create: %s (%s)\n", glfer_hip_strerror(rc), glfer_hip_last_hip_error());
    return 5;
  }
  float *rows = glfer_hip_host_alloc(frames * bins * sizeof(float));      /* pinned: the rows arrive by DMA */
  if (!rows)
    return 6;
  glfer_hip_phases ph;
  size_t got = 0;
  for (int call = 0; call < 3; call++) {                    /* the handle is reused: nothing is made inside these calls */
    memset(rows, 0xff, frames * bins * sizeof(float));
    rc = glfer_hip_workers_spectrogram_wav(w, argv[6], rows, frames, &got, 0, &ph);
    if (rc != GLFER_OK || got != frames) {
      fprintf(stderr, "call %d: rc %d (%s), %zu of %zu frames\n", call, rc, glfer_hip_last_hip_error(), got, frames);
      return 7;
    }
  }
  FILE *out = fopen(argv[7], "wb");
  if (!out || fwrite(rows, sizeof(float), frames * bins, out) != frames * bins)
    return 8;
  fclose(out);
  printf("%zu frames, 3 calls; last: wall %.2f ms, read %.2f, uploads %.2f, kernels %.2f, downloads %.2f, %u chunks\n", frames, ph.wall_s * 1e3,
         ph.read_s * 1e3, ph.h2d_s * 1e3, ph.kernel_s * 1e3, ph.d2h_s * 1e3, ph.chunks);
  glfer_hip_host_free(rows);
  glfer_hip_workers_destroy(w);
  return 0;
}
