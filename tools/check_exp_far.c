/* exp far path: glibc specialcase with normal result == formula used on the device, vs libm */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../bs_call_amd/csrc/bsmath_tables.h"
#include "../bs_call_amd/csrc/bsmath.h"
static double far_form(double x) {
  const double kd0 = BSM_FMA(x, BSM_EXP_INVLN2N, BSM_EXP_SHIFT);
  const uint64_t ki = bsm_bits(kd0);
  const double kd = kd0 - BSM_EXP_SHIFT;
  const double r = BSM_FMA(kd, BSM_EXP_NEGLN2LON, BSM_FMA(kd, BSM_EXP_NEGLN2HIN, x));
  const uint64_t idx = 2u * (ki & 127u);
  const double tail = bsm_from_bits(bsm_exp_tab[idx]);
  uint64_t sbits = bsm_exp_tab[idx + 1] + ((uint64_t)((uint32_t)ki << 13) << 32);
  const double r2 = r * r;
  const double p23 = BSM_FMA(r, BSM_EXP_C3, BSM_EXP_C2);
  const double p45 = BSM_FMA(r, BSM_EXP_C5, BSM_EXP_C4);
  const double t = BSM_FMA(p23, r2, tail + r);
  const double tmp = BSM_FMA(r2 * r2, p45, t);
  const double sc2 = bsm_from_bits(sbits + (1022ull << 52));
  const double st = sc2 * tmp;
  return 0x1p-1022 * (sc2 + st);
}
int main(void) {
  uint64_t s = 88172645463325252ull, bad = 0, n = 0, diff_main = 0;
  for (long i = 0; i < 200000000L; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double u = (double)(s >> 11) * 0x1p-53;
    double x = -512.0 - u * 188.0; /* [-700, -512] */
    double a = far_form(x), b = exp(x);
    if (memcmp(&a, &b, 8)) bad++;
    n++;
  }
  /* edges */
  double xs[] = {-512.0, -700.0, -699.99999999999989, -512.00000000000011, -600.5, -708.0};
  for (int i = 0; i < 6; i++) { double a = far_form(xs[i]), b = exp(xs[i]); printf("%a %a %a %s\n", xs[i], a, b, memcmp(&a,&b,8)?"DIFF":"ok"); }
  printf("%llu of %llu differ from libm\n", (unsigned long long)bad, (unsigned long long)n);
  return bad != 0;
}
