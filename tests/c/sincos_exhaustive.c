/* Compares the oracle's restated glibc sinf/cosf with libm over every float in [0, hi]. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
float oo_sinf(float), oo_cosf(float);
int main(int argc, char** argv) {
  float hi = argc > 1 ? (float)atof(argv[1]) : 6.3f;
  uint32_t uh;
  memcpy(&uh, &hi, 4);
  long bad = 0;
  for (uint32_t u = 0; u <= uh; u++) {
    float x, a, b;
    memcpy(&x, &u, 4);
    a = sinf(x); b = oo_sinf(x);
    if (memcmp(&a, &b, 4)) bad++;
    a = cosf(x); b = oo_cosf(x);
    if (memcmp(&a, &b, 4)) bad++;
  }
  printf("%u %ld\n", uh + 1, bad);
  return bad != 0;
}
