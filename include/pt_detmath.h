/* pt_detmath.h — deterministic single-precision elementary functions.
 *
 * Why this exists: the path tracer's result is a chaotic function of every
 * rounded bit (one flipped lobe decision changes a whole path), and the GPU's
 * device libm (ocml) and the host's glibc round sinf/cosf/acosf/atan2f/logf/powf
 * differently.  These functions use only +,-,*,/ and sqrtf — all correctly
 * rounded by IEEE-754 on both x86-64 (SSE) and gfx950 (hipcc default, no
 * fast-math) — so a kernel and a CPU checker that both include this header and
 * are both compiled with -ffp-contract=off produce IDENTICAL BITS.
 *
 * The reference (bipul-mohanto/OptixPathTracer) calls CUDA's sinf/cosf/acosf/
 * atan2/logf/powf under nvcc --use_fast_math (CMakeLists.txt:181), whose bits
 * nobody can reproduce off NVIDIA hardware; any ~1-ulp implementation is within
 * the reference's own semantics.  Polynomials are the classic Cephes
 * single-precision minimax sets (public domain, S. Moshier).
 *
 * Plain C99; usable from C, C++ and HIP device code.  Compile every translation
 * unit that includes it with -ffp-contract=off.
 */
#ifndef PT_DETMATH_H
#define PT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ static inline
#else
#include <math.h>
#include <string.h>
#define PT_HD static inline
#endif

#define PT_PIF 3.14159265358979323846f
#define PT_PIO2F 1.57079632679489661923f
#define PT_PIO4F 0.78539816339744830962f

PT_HD float pt_bits2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}
PT_HD uint32_t pt_f2bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
#endif
}
PT_HD float pt_fabsf(float x) { return pt_bits2f(pt_f2bits(x) & 0x7fffffffu); }

/* sin and cos of x together; |x| < 8192 keeps ~1 ulp (the path tracer only
 * passes [0, 2*pi]).  Cody–Waite 3-part reduction by pi/4 octants. */
PT_HD void pt_sincosf(float xx, float* s_out, float* c_out) {
    const float FOPI = 1.27323954473516f;
    const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    float x = pt_fabsf(xx);
    int sign_s = (pt_f2bits(xx) >> 31) ? -1 : 1;
    int sign_c = 1;
    int j = (int)(FOPI * x);
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    j &= 7;
    if (j > 3) {
        sign_s = -sign_s;
        sign_c = -sign_c;
        j -= 4;
    }
    if (j > 1) sign_c = -sign_c;
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x;
    ps = ps + x;
    float s, c;
    if (j == 1 || j == 2) {
        s = pc;
        c = ps;
    } else {
        s = ps;
        c = pc;
    }
    *s_out = (sign_s < 0) ? -s : s;
    *c_out = (sign_c < 0) ? -c : c;
}
PT_HD float pt_sinf(float x) {
    float s, c;
    pt_sincosf(x, &s, &c);
    return s;
}
PT_HD float pt_cosf(float x) {
    float s, c;
    pt_sincosf(x, &s, &c);
    return c;
}

/* asin on [-1,1] */
PT_HD float pt_asinf(float xx) {
    float a = pt_fabsf(xx);
    int neg = (int)(pt_f2bits(xx) >> 31);
    float x, z;
    int flag = 0;
    if (a > 1.0f) return 0.0f;
    if (a < 1.0e-4f) {
        z = a;
    } else {
        if (a > 0.5f) {
            z = 0.5f * (1.0f - a);
            x = sqrtf(z);
            flag = 1;
        } else {
            x = a;
            z = x * x;
        }
        z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z +
             1.6666752422E-1f) *
                z * x +
            x;
        if (flag) {
            z = z + z;
            z = PT_PIO2F - z;
        }
    }
    return neg ? -z : z;
}

/* acos on [-1,1] */
PT_HD float pt_acosf(float x) {
    if (x < -1.0f) x = -1.0f;
    if (x > 1.0f) x = 1.0f;
    if (x < -0.5f) return PT_PIF - 2.0f * pt_asinf(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * pt_asinf(sqrtf(0.5f * (1.0f - x)));
    return PT_PIO2F - pt_asinf(x);
}

PT_HD float pt_atanf(float xx) {
    float x = pt_fabsf(xx), y;
    int neg = (int)(pt_f2bits(xx) >> 31);
    if (x > 2.414213562373095f) {
        y = PT_PIO2F;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = PT_PIO4F;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x;
    return neg ? -y : y;
}

/* atan2(y, x), result in (-pi, pi] */
PT_HD float pt_atan2f(float y, float x) {
    int code = 0;
    if (x < 0.0f) code = 2;
    if (y < 0.0f) code |= 1;
    if (x == 0.0f) {
        if (code & 1) return -PT_PIO2F;
        if (y == 0.0f) return 0.0f;
        return PT_PIO2F;
    }
    if (y == 0.0f) return (code & 2) ? PT_PIF : 0.0f;
    float w = (code == 2) ? PT_PIF : ((code == 3) ? -PT_PIF : 0.0f);
    return w + pt_atanf(y / x);
}

/* natural log, x > 0 and normal (callers pass a^2 >= 1e-6 or sRGB inputs >= 0.0031) */
PT_HD float pt_logf(float x) {
    uint32_t u = pt_f2bits(x);
    int e = (int)((u >> 23) & 0xff) - 126; /* x = m * 2^e, m in [0.5,1) */
    float m = pt_bits2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = ((((((((7.0376836292E-2f * m - 1.1514610310E-1f) * m + 1.1676998740E-1f) * m - 1.2420140846E-1f) * m +
                   1.4249322787E-1f) *
                      m -
                  1.6668057665E-1f) *
                     m +
                 2.0000714765E-1f) *
                    m -
                2.4999993993E-1f) *
                   m +
               3.3333331174E-1f) *
              m * z;
    float fe = (float)e;
    if (e) y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    z = m + y;
    if (e) z += 0.693359375f * fe;
    return z;
}

/* e^x for |x| < 80 */
PT_HD float pt_expf(float x) {
    float t = 1.44269504088896341f * x + 0.5f;
    int n = (int)t;
    if ((float)n > t) n -= 1; /* floor */
    float z = (float)n;
    x -= z * 0.693359375f;
    x -= z * -2.12194440e-4f;
    z = x * x;
    z = (((((1.9875691500E-4f * x + 1.3981999507E-3f) * x + 8.3334519073E-3f) * x + 4.1665795894E-2f) * x +
          1.6666665459E-1f) *
             x +
         5.0000001201E-1f) *
            z +
        x + 1.0f;
    /* ldexp: n in [-126,127] for our inputs */
    return z * pt_bits2f((uint32_t)(n + 127) << 23);
}

/* x^y for x >= 0 (the sRGB transfer only): exp(y*log(x)), the same
 * decomposition CUDA's fast-math powf uses (exp2(y*log2 x)). */
PT_HD float pt_powf(float x, float y) {
    if (!(x > 0.0f)) return 0.0f;
    return pt_expf(y * pt_logf(x));
}

#endif /* PT_DETMATH_H */
