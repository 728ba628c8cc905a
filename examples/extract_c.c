/* extract_c.c -- the C ABI from plain C99, no C++, no Python: what a cgo / JNI / FFI binding of another host language would do.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/extract_c.c -Lrefactored_orb_slam2_amd/csrc -lorbfe \
 *       -Wl,-rpath,$PWD/refactored_orb_slam2_amd/csrc -o extract_c
 *   ./extract_c image.png [nfeatures]
 *
 * Reads an 8-bit PNG with the library's own reader, runs ORBextractor::operator() (orbfe_extract) on it and prints the number
 * of keypoints, a checksum of the descriptor bytes and the first keypoints.  Exit code: 0 ok, 3 no HIP device (the library has
 * no CPU fallback), 1 any other error. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "orbfe.h"

static int fail(const char* what, int rc) {
  fprintf(stderr, "%s failed: %d (%s)\n", what, rc, orbfe_last_error());
  return rc == ORBFE_ERR_NO_DEVICE ? 3 : 1;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s image.png [nfeatures]\n", argv[0]);
    return 1;
  }
  int w = 0, h = 0, rc;
  if ((rc = orbfe_png_info(argv[1], &w, &h)) != ORBFE_OK) return fail("orbfe_png_info", rc);
  uint8_t* img = (uint8_t*)malloc((size_t)w * h);
  if (!img) return 1;
  if ((rc = orbfe_png_read_gray(argv[1], img, w, h, &w, &h)) != ORBFE_OK) return fail("orbfe_png_read_gray", rc);

  orbfe_params prm;
  prm.n_features = argc > 2 ? atoi(argv[2]) : 1000;   /* ORBextractor.nFeatures etc. of the TUM settings files */
  prm.scale_factor = 1.2f;
  prm.n_levels = 8;
  prm.ini_th_fast = 20;
  prm.min_th_fast = 7;
  orbfe_extractor* ex = NULL;
  if ((rc = orbfe_extractor_create(&prm, -1, &ex)) != ORBFE_OK) return fail("orbfe_extractor_create", rc);
  int cap = 0;
  if ((rc = orbfe_extractor_max_keypoints(ex, w, h, &cap)) != ORBFE_OK) return fail("orbfe_extractor_max_keypoints", rc);
  orbfe_keypoint* kps = (orbfe_keypoint*)malloc(sizeof(orbfe_keypoint) * (size_t)cap);
  uint8_t* desc = (uint8_t*)malloc((size_t)cap * 32);
  int n = 0;
  if ((rc = orbfe_extract(ex, img, w, h, w, kps, desc, cap, &n)) != ORBFE_OK) return fail("orbfe_extract", rc);
  uint32_t sum = 0;
  for (int i = 0; i < n * 32; i++) sum = sum * 16777619u ^ desc[i];
  printf("%dx%d: %d keypoints, descriptor checksum %08x\n", w, h, n, sum);
  for (int i = 0; i < n && i < 5; i++)
    printf("  kp %d: (%.1f, %.1f) octave %d angle %.3f response %.0f\n", i, kps[i].x, kps[i].y, kps[i].octave, kps[i].angle,
           kps[i].response);
  orbfe_extractor_destroy(ex);
  free(kps); free(desc); free(img);
  return 0;
}
