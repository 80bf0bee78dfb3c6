/* A program that, like glfer.c:56-57, defines the globals `opt` and `glfer` itself and links
 * libglfer_compat.so with NO glue file: the library must read opt.autoscale / glfer.first_buffer
 * straight out of them (weak references, layout of include/glfer_compat.h).  No GPU call is made:
 * built and run on the CPU by tests/test_host_logic.py. */
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "glfer_compat.h"

opt_t opt;
glfer_t glfer;

int main(void)
{
  memset(&opt, 0, sizeof opt);
  memset(&glfer, 0, sizeof glfer);
  /* the fallback variables say the opposite of the globals: only the globals may be believed */
  glfer_compat_autoscale = 1;
  glfer_compat_first_buffer = 1;
  opt.autoscale = 0;
  glfer.first_buffer = 0;
  if (glfer_compat_get_autoscale() != 0 || glfer_compat_get_first_buffer() != 0) return 1;
  opt.autoscale = 1;
  if (glfer_compat_get_autoscale() != 1) return 2;
  glfer.first_buffer = 1;
  if (glfer_compat_get_first_buffer() != 1) return 3;
  /* neighbours of the two fields must not be what is read */
  opt.autoscale = 0; opt.thr_level = 1.0f; opt.max_level_db = 1.0f;
  glfer.first_buffer = 0; glfer.init_done = 1; glfer.input_source = FILE_SOURCE;
  if (glfer_compat_get_autoscale() != 0 || glfer_compat_get_first_buffer() != 0) return 4;
  printf("%zu %zu %zu %zu\n", sizeof(opt_t), offsetof(opt_t, autoscale), sizeof(glfer_t), offsetof(glfer_t, first_buffer));
  return 0;
}
