/* Compiled (not run) by tests/test_host_logic.py with plain `gcc -std=c99 -Wall -Werror -c`:
 * the two headers are the drop-in boundary and must be usable from the reference's own language, C. */
#include <stddef.h>
#include "glfer_hip.h"
#include "glfer_compat.h"

int probe_batch(const void *d_pcm, size_t nsamples, float *d_psd, float *d_stats, unsigned char *d_rgb, void *stream)
{
  glfer_hip_config cfg = {0};
  glfer_hip_plan *plan = NULL;
  glfer_hip_display disp = {0};
  size_t frames;
  int rc;
  cfg.mode = GLFER_MODE_MTM;
  cfg.n = 4096;
  cfg.overlap = 0.75f;
  cfg.mtm_w = 2.5f;
  cfg.mtm_k = 4;
  cfg.sample_format = GLFER_SAMPLES_S16;
  cfg.history_mode = GLFER_HISTORY_ZERO_FIRST;
  rc = glfer_hip_plan_create(&cfg, &plan);
  if (rc != GLFER_OK)
    return rc;
  frames = glfer_hip_num_frames(plan, nsamples);
  rc = glfer_hip_spectrogram_device(plan, d_pcm, nsamples, 0, frames, d_psd, stream);
  if (rc == GLFER_OK)
    rc = glfer_hip_floor_device(d_psd, frames, glfer_hip_bins(plan), d_stats, stream);
  disp.scale_type = GLFER_SCALE_LOG;
  disp.autoscale = 1;
  disp.first_buffer = 1;
  disp.palette = GLFER_PAL_HSV;
  if (rc == GLFER_OK)
    rc = glfer_hip_display_device(&disp, d_psd, NULL, d_stats, frames, glfer_hip_bins(plan), d_rgb, NULL, NULL, stream);
  glfer_hip_plan_destroy(plan);
  return rc + (int)(frames % GLFER_FRAME_ALIGN) * 0;
}

void probe_per_hop(float *hop, float *psd)
{
  fft_params_t fp;
  mtm_params_t mp;
  fp.n = 1024;
  fp.window_type = HANNING_WINDOW;
  fp.overlap = 0.5f;
  fp.a = 0.0f;
  fp.limiter = 0;
  fft_init(&fp);
  fft_do(hop, &fp);
  fft_psd(psd, NULL, &fp);
  fft_close(&fp);
  mp.fft.n = 1024;
  mp.fft.overlap = 0.0f;
  mp.w = 4.0f;
  mp.kmax = 7;
  mtm_init(&mp);
  mtm_do(hop, psd, NULL, &mp);
  mtm_close(&mp);
}
