/* Compares the oracle's restated glibc logf with libm over every float with bit pattern in [lo, hi] (hex arguments;
 * default: every positive normal float). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
float oo_logf(float);
int main(int argc, char** argv) {
  uint32_t lo = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 16) : 0x00800000u;
  uint32_t hi = argc > 2 ? (uint32_t)strtoul(argv[2], 0, 16) : 0x7f7fffffu;
  long bad = 0;
  for (uint32_t u = lo;; u++) {
    float x, a, b;
    memcpy(&x, &u, 4);
    a = logf(x); b = oo_logf(x);
    if (memcmp(&a, &b, 4)) bad++;
    if (u == hi) break;
  }
  printf("%lu %ld\n", (unsigned long)hi - lo + 1, bad);
  return bad != 0;
}
