// cda_libm.hpp - the two libm functions numpy's random generator calls on the host (`log1p` in the ziggurat tail, `exp` in
// its wedge test: numpy/random/src/distributions/distributions.c random_standard_normal), restated from glibc 2.35's
// published algorithms so that the device draws the SAME normals bit for bit.  Plain IEEE double arithmetic, built with
// -ffp-contract=off: a fused multiply-add appears only where it is written.  Host and device (the host build lets the
// CPU test suite compare the restatement with the machine's own libm on 10^7 arguments, tests/test_oracle_arith.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CDA_HD __host__ __device__ inline
#define CDA_EXP_QUAL __device__ const
#include "cda_exp_table.h"
#undef CDA_EXP_QUAL
namespace cda_host { 
#define CDA_EXP_QUAL static const
#include "cda_exp_table.h"
#undef CDA_EXP_QUAL
}

namespace cda {

CDA_HD double f64_from_bits(unsigned long long b) { union { unsigned long long u; double d; } c; c.u = b; return c.d; }
CDA_HD unsigned long long f64_bits(double x) { union { unsigned long long u; double d; } c; c.d = x; return c.u; }
// log1p as glibc 2.35 computes it (sysdeps/ieee754/dbl-64/s_log1p.c: the fdlibm algorithm with the
// split polynomial evaluation), restated so that the ziggurat tail `r + xx` of numpy - which calls the
// host libm - is reproduced bit for bit on the device (checked against glibc on 2e7 inputs in the build
// container; built with -ffp-contract=off).  Finite x > -1 only.
CDA_HD int32_t f64_hi(double x) { return (int32_t)(f64_bits(x) >> 32); }
CDA_HD double f64_set_hi(double x, int32_t h) {
    unsigned long long b = f64_bits(x);
    b = (b & 0xffffffffull) | ((unsigned long long)(uint32_t)h << 32);
    return f64_from_bits(b);
}
CDA_HD double glibc_log1p(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    double hfsq, f = 0.0, c = 0.0, s, z, R, u;
    int32_t k = 1, hx = f64_hi(x), hu = 0, ax = hx & 0x7fffffff;
    if (hx < 0x3FDA827A) {
        if (ax >= 0x3ff00000) return x == -1.0 ? -f64_from_bits(0x7ff0000000000000ULL) : f64_from_bits(0x7ff8000000000000ULL);
        if (ax < 0x3e200000) return ax < 0x3c900000 ? x : x - x * x * 0.5;
        if (hx > 0 || hx <= (int32_t)0xbfd2bec3) { k = 0; f = x; hu = 1; }
    }
    if (k != 0) {
        if (hx < 0x43400000) {
            u = 1.0 + x; hu = f64_hi(u); k = (hu >> 20) - 1023;
            c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
            c /= u;
        } else { u = x; hu = f64_hi(u); k = (hu >> 20) - 1023; c = 0.0; }
        hu &= 0x000fffff;
        if (hu < 0x6a09e) u = f64_set_hi(u, hu | 0x3ff00000);
        else { k += 1; u = f64_set_hi(u, hu | 0x3fe00000); hu = (0x00100000 - hu) >> 2; }
        f = u - 1.0;
    }
    hfsq = 0.5 * f * f;
    if (hu == 0) {
        if (f == 0.0) { if (k == 0) return 0.0; c += k * ln2_lo; return k * ln2_hi + c; }
        R = hfsq * (1.0 - 0.66666666666666666 * f);
        if (k == 0) return f - R;
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    s = f / (2.0 + f); z = s * s;
    double R1 = z * Lp1, z2 = z * z, R2 = Lp2 + z * Lp3, z4 = z2 * z2, R3 = Lp4 + z * Lp5, z6 = z4 * z2, R4 = Lp6 + z * Lp7;
    R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}
// exp as glibc 2.35 computes it on x86-64 (sysdeps/ieee754/dbl-64/e_exp.c, the table-driven algorithm with N = 128 and a
// degree-5 polynomial): x = k ln2/N + r, exp(x) = 2^(k/N) exp(r) ~= scale + scale * (tail + r + r^2 (C2 + r C3) + r^4 (C4 + r C5)).
// glibc selects its FMA build (__exp_fma) on every CPU with FMA3 - all current x86-64 hosts, the build container and
// the GPU box's EPYC included - where the compiler contracts each a*b + c of that source into one fused operation; the
// contractions are written out below (without them 0.07 % of the arguments differ in the last bit).  Domain of this
// restatement: |x| < 512 (the ziggurat wedge asks for exp(-x^2/2), |x| < 3.66); checked against the host libm on 2e7
// arguments.  The 2^(k/N) table is generated from its definition by tools/gen_exp_table.py.
CDA_HD double glibc_exp(double x) {
    const double InvLn2N = 0x1.71547652b82fep0 * 128, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47,
                 Shift = 0x1.8p52, C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    const unsigned abstop = (unsigned)(f64_bits(x) >> 52) & 0x7ffu;
    if (abstop < 0x3c9u) return 1.0 + x;                     // |x| < 2^-54 (and +-0)
    const double z = InvLn2N * x;
    double kd = z + Shift;
    const unsigned long long ki = f64_bits(kd);
    kd -= Shift;
    const double r = __builtin_fma(kd, NegLn2loN, __builtin_fma(kd, NegLn2hiN, x));
    const unsigned idx = 2u * (unsigned)(ki % 128u);
    const unsigned long long top = ki << (52 - 7);
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long t0 = cda_exp_tab[idx], t1 = cda_exp_tab[idx + 1];
#else
    const unsigned long long t0 = cda_host::cda_exp_tab[idx], t1 = cda_host::cda_exp_tab[idx + 1];
#endif
    const double tail = f64_from_bits(t0);
    const double scale = f64_from_bits(t1 + top);
    const double r2 = r * r;
    const double p1 = __builtin_fma(r, C3, C2), p2 = __builtin_fma(r, C5, C4);
    const double tmp = __builtin_fma(r2 * r2, p2, __builtin_fma(r2, p1, tail + r));
    return __builtin_fma(scale, tmp, scale);
}

}  // namespace cda
