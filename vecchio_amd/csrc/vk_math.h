// vk_math.h — arithmetic shared bit-for-bit between host C++ and gfx950 device code.
//
// Why this exists: the reference (Rust) draws every random number from rand::thread_rng()
// (unseeded; 25 call sites, e.g. main.rs:184,187-188, util.rs:33,44,54-55,179,
// hittable.rs:125-126,287-288,431,473, material.rs:53-54,167,198,477) and calls libm
// (sin, cos, atan2, asin, ln, powf).  Neither is reproducible on a GPU, and a 1-ulp
// difference in a hit/miss or rejection decision changes a whole path.  So the generator
// is replaced by a counter-based one keyed (seed, pixel, sample) and the transcendental
// functions are implemented here from +,-,*,/ only (IEEE, no FMA contraction: every
// translation unit that includes this file is compiled with -ffp-contract=off), which makes
// host and device results identical by construction.  tests/ checks them against libm.
//
// The float mappings of the draws follow rand 0.7.3 (Cargo.lock; third-party, not vendored,
// restated from its published algorithm):
//   gen::<f32>()          = (u32 >> 8) * 2^-24                       (Standard, 24 bits)
//   gen_range(lo,hi):f32  = ((u32 >> 9 | 0x3F800000 as f32) - 1) * (hi-lo) + lo, redraw if >= hi
//   gen_range(0,n):u32    = widening-multiply rejection (UniformInt::sample_single)
#ifndef VK_MATH_H
#define VK_MATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define VK_HD __host__ __device__ __forceinline__
// The f64 transcendental kernels (and the non-solid texture path, vk_trace.h) are called once per bounce at most.  Rounds 1-2 kept
// them out of line so that their double constants were not hoisted into the megakernel's persistent loop.  With the phases re-reading
// their constants through laundered pointers (vk_kernels.h) that no longer happens, and a called function costs more than it saves:
// every value that is live across the call has to sit in a callee-saved register, which the out-of-line shading phase then has to
// save and restore itself.  Inline since round 3: the everything-variants' shading function has 30 scratch stores instead of 55, the
// sphere-only LDS kernel 72 VGPRs instead of 76 (C3 +1 %, C2 / C4 +0.4 %).
#define VK_COLD __host__ __device__ __forceinline__
#else
#define VK_HD inline
#define VK_COLD inline
#endif

namespace vk {

// ---------------------------------------------------------------------------------------
// bit casts
VK_HD uint32_t f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
VK_HD float bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
VK_HD uint64_t f64_bits(double f) { return __builtin_bit_cast(uint64_t, f); }
VK_HD double bits_f64(uint64_t u) { return __builtin_bit_cast(double, u); }

// ---------------------------------------------------------------------------------------
// Counter-based generator.  One stream per (seed, pixel, sample): the 64-bit key is a SplitMix64 hash
// of the triple, draw i of the stream is a 32-bit hash of (key, i) (next_u32 below).  Regenerating a
// sample on any lane, wave or GPU therefore reproduces the same stream.
struct Rng {
    uint64_t key;
    uint32_t ctr;
};

VK_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

VK_HD Rng rng_for_sample(uint64_t seed, uint32_t pixel, uint32_t sample) {
    Rng r;
    uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ull);
    r.key = mix64(h ^ (((uint64_t)pixel << 32) | (uint64_t)sample));
    r.ctr = 0;
    return r;
}

// Per-sample stream, drawn from on the device once per random decision: a 32-bit Weyl sequence
// through a two-multiply finaliser (the "lowbias32" constants), keyed by both halves of the
// sample's 64-bit key.  A draw is 3 v_mul_lo_u32 + 8 simple ops; SplitMix64 per draw (two
// 64-bit multiplies = ~16 quarter-rate integer multiplies on gfx950) made the generator the
// largest single cost of the shading phase.  Statistical checks: tests/test_math_rng.py.
VK_HD uint32_t next_u32(Rng &r) {
    r.ctr += 1u;
    uint32_t x = (uint32_t)r.key + r.ctr * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x21F0AAADu;
    x ^= x >> 15; x ^= (uint32_t)(r.key >> 32); x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}

// The stream for host-side scene construction (scene builders, BVH axis choice, Perlin tables):
// SplitMix64 in counter mode.  A separate type so that scenes do not depend on the per-sample generator.
struct BuildRng { uint64_t key; uint32_t ctr; };
VK_HD BuildRng rng_for_stream(uint64_t seed, uint64_t stream) {
    BuildRng r;
    uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ull);
    r.key = mix64(h ^ mix64(stream + 0xD1B54A32D192ED03ull));
    r.ctr = 0;
    return r;
}
VK_HD uint32_t next_u32(BuildRng &r) {
    r.ctr += 1u;
    uint64_t z = r.key + (uint64_t)r.ctr * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(mix64(z) >> 32);
}

// rand 0.7.3 Standard for f32: 24 random bits scaled by 2^-24 -> [0,1)
template <class R> VK_HD float gen_f32(R &r) { return (float)(next_u32(r) >> 8) * (1.0f / 16777216.0f); }

// rand 0.7.3 UniformFloat<f32>::sample_single
template <class R> VK_HD float gen_range(R &r, float lo, float hi) {
    float scale = hi - lo;
    // The reference asserts lo < hi (panics otherwise) and redraws while res >= hi, which for lo < hi
    // happens with probability ~2^-24 per draw.  A degenerate range (lo >= hi, NaN, hi-lo overflowing)
    // that slipped past validation is answered up front so that it can never hang a GPU wave; the
    // test folds away at the call sites with constant bounds and costs no loop-carried register
    // (a draw counter in the loop pushed the sphere-only kernel over its 80-VGPR budget).
    if (!(scale > 0.0f) || scale > 3.0e38f) return lo;
    for (;;) {
        float v12 = bits_f32((next_u32(r) >> 9) | 0x3F800000u);
        float v01 = v12 - 1.0f;
        float res = v01 * scale + lo;
        if (res < hi) return res;
    }
}

// gen_range(r, -1.0f, 1.0f) without its loop: the redraw test can never fire for this range.  v01 = k 2^-23 with k < 2^23; v01 * 2 is
// exact; 2 v01 - 1 is a multiple of 2^-22 of magnitude <= 1, hence exact too, and at most 1 - 2^-22 < hi.  (tests/test_math_rng.py
// walks all 2^23 values.)  The rejection samplers built on it (util.rs:31-52) then have ONE loop, not one per coordinate inside it.
template <class R> VK_HD float gen_pm1(R &r) {
    float v12 = bits_f32((next_u32(r) >> 9) | 0x3F800000u);
    float v01 = v12 - 1.0f;
    return v01 * 2.0f + -1.0f;
}

// gen_range(r, 0.0f, hi) for a CONSTANT hi for which the redraw test can never fire either: fl(v01 * hi) < hi for every v01 <= 1 - 2^-23
// when (1 - 2^-23) hi lies more than half an ulp below hi — true unless hi is a power of two (its lower neighbour is half as far);
// used for hi = 2 pi (6.2831855: 1.57 ulps below).  tests/test_math_rng.py walks all 2^23 values.
template <class R> VK_HD float gen_0_to(R &r, float hi) {
    float v12 = bits_f32((next_u32(r) >> 9) | 0x3F800000u);
    float v01 = v12 - 1.0f;
    return v01 * hi + 0.0f;
}

// rand 0.7.3 UniformInt<u32>::sample_single(0, n) as used by SliceRandom::choose/gen_index
template <class R> VK_HD uint32_t gen_index(R &r, uint32_t n) {
    uint32_t range = n;
    uint32_t zone = (range << __builtin_clz(range)) - 1u;
    for (;;) {
        uint32_t v = next_u32(r);
        uint64_t m = (uint64_t)v * (uint64_t)range;
        uint32_t lo = (uint32_t)m;
        if (lo <= zone) return (uint32_t)(m >> 32);
    }
}

// ---------------------------------------------------------------------------------------
// Rust `as usize` from f32: saturating, NaN -> 0 (material.rs:286-287,399-401)
VK_HD uint32_t sat_u32(float f) {
    if (!(f > 0.0f)) return 0u;           // negative, -0, NaN
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

// low 8 bits of Rust's `f as usize` (64-bit usize, saturating; material.rs:399-401 then
// `(i + di) & 255` with wrapping add): all that Perlin::noise uses of the cast.
VK_HD uint32_t usize_low8(float f) {
    if (!(f > 0.0f)) return 0u;                       // negative, -0, NaN -> 0
    if (f >= 18446744073709551616.0f) return 255u;    // saturates to usize::MAX
    if (f >= 4294967296.0f) return 0u;                // multiples of 512 here
    return ((uint32_t)f) & 255u;
}

// ---------------------------------------------------------------------------------------
// transcendental functions: evaluated in f64 with +,-,*,/ only, rounded once to f32.
// (f32 results are within ~0.5000001 ulp of the true value; they agree with a correctly
// rounded libm except on the rare argument that falls within 1e-9 ulp of a rounding tie.)

// reduce x to r in [-pi/4, pi/4], returns quadrant (low 2 bits valid)
VK_HD int rem_pio2(double x, double &r) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double PIO2_1T = 6.07710050650619224932e-11;  // pi/2 - PIO2_1
    const double MAGIC = 6755399441055744.0;            // 1.5 * 2^52: round-to-nearest-int trick
    double kd = x * TWO_OVER_PI + MAGIC;
    kd = kd - MAGIC;
    r = (x - kd * PIO2_1) - kd * PIO2_1T;
    return (int)(long long)kd;
}

VK_HD double k_sin(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return r + (r * z) * (S1 + z * p);
}

VK_HD double k_cos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return (1.0 - 0.5 * z) + z * p;
}

#if defined(VK_MATH_LIBM) && !defined(__HIP_DEVICE_COMPILE__)
// The INDEPENDENT variant (oracle/Makefile target `libm`, tests/test_oracle_libm.py): the seven functions from glibc's f32
// libm instead of the kernels below.  Everything that is compiled into oracle, emulator and device alike shares these kernels, so an
// error in them would be common-mode; this build of the oracle bounds how many paths a last-bit difference in them moves.
VK_COLD float sinf_(float x) { return ::sinf(x); }
VK_COLD float cosf_(float x) { return ::cosf(x); }
struct SinCos { float s, c; };
VK_COLD SinCos sincosf_(float x) { SinCos o; o.s = ::sinf(x); o.c = ::cosf(x); return o; }
VK_COLD float logf_(float x) { return ::logf(x); }
VK_COLD float atan2f_(float y, float x) { return ::atan2f(y, x); }
VK_COLD float asinf_(float x) { return ::asinf(x); }
VK_HD float pow5f_(float x) { return ::powf(x, 5.0f); }
#else
VK_COLD float sinf_(float xf) {
    double x = (double)xf;
    if (!(x > -1.0e9 && x < 1.0e9)) return xf - xf;  // inf/NaN -> NaN; |x|>=1e9 unsupported -> 0
    double r;
    int q = rem_pio2(x, r);
    double v = (q & 1) ? k_cos(r) : k_sin(r);
    if (q & 2) v = -v;
    return (float)v;
}

VK_COLD float cosf_(float xf) {
    double x = (double)xf;
    if (!(x > -1.0e9 && x < 1.0e9)) return xf - xf;
    double r;
    int q = rem_pio2(x, r);
    double v = (q & 1) ? k_sin(r) : k_cos(r);
    if ((q + 1) & 2) v = -v;
    return (float)v;
}

// sin and cos of the same angle, bit-identical to sinf_(x) and cosf_(x) (one reduction, one
// evaluation of each kernel polynomial)
struct SinCos { float s, c; };
VK_COLD SinCos sincosf_(float xf) {
    SinCos o;
    double x = (double)xf;
    if (!(x > -1.0e9 && x < 1.0e9)) { o.s = xf - xf; o.c = xf - xf; return o; }
    double r;
    int q = rem_pio2(x, r);
    double ks = k_sin(r), kc = k_cos(r);
    double vs = (q & 1) ? kc : ks;
    double vc = (q & 1) ? ks : kc;
    if (q & 2) vs = -vs;
    if ((q + 1) & 2) vc = -vc;
    o.s = (float)vs;
    o.c = (float)vc;
    return o;
}

// natural log (hittable.rs:473); x is a 24-bit draw in [0,1) there, general f32 supported
VK_COLD float logf_(float xf) {
    if (xf != xf) return xf;
    if (xf < 0.0f) return (xf - xf) / (xf - xf);  // NaN
    if (xf == 0.0f) return -INFINITY;
    if (xf == INFINITY) return xf;
    const double LN2 = 6.93147180559945286227e-01;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double x = (double)xf;  // every finite f32 (denormals included) is a normal f64
    uint64_t b = f64_bits(x);
    int e = (int)((b >> 52) & 0x7FF) - 1023;
    uint64_t mant = b & 0x000FFFFFFFFFFFFFull;
    double m = bits_f64(mant | 0x3FF0000000000000ull);  // [1,2)
    if (m > 1.41421356237309504880) {
        m = m * 0.5;
        e += 1;
    }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double l1p = f - (hfsq - s * (hfsq + R));
    return (float)((double)e * LN2 + l1p);
}

VK_HD double k_atan(double x) {  // x >= 0 finite or +inf
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    double hi, lo;
    int id;
    if (x < 0.4375) {
        id = -1; hi = 0.0; lo = 0.0;
    } else if (x < 0.6875) {
        id = 0; hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17;
        x = (2.0 * x - 1.0) / (2.0 + x);
    } else if (x < 1.1875) {
        id = 1; hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17;
        x = (x - 1.0) / (x + 1.0);
    } else if (x < 2.4375) {
        id = 2; hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17;
        x = (x - 1.5) / (1.0 + 1.5 * x);
    } else {
        id = 3; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17;
        x = -1.0 / x;
    }
    double z = x * x;
    double w = z * z;
    double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    return hi - ((x * (s1 + s2) - lo) - x);
}

VK_HD double atan2_d(double y, double x) {
    const double PI = 3.14159265358979311600e+00;
    const double PI_LO = 1.2246467991473531772e-16;
    if (x != x || y != y) return x + y;
    bool yneg = (f64_bits(y) >> 63) != 0;
    bool xneg = (f64_bits(x) >> 63) != 0;
    if (y == 0.0) {
        double v = xneg ? PI : 0.0;
        return yneg ? -v : v;
    }
    if (x == 0.0) return yneg ? -0.5 * PI : 0.5 * PI;
    double ay = yneg ? -y : y;
    double ax = xneg ? -x : x;
    double a;
    if (ax == INFINITY && ay == INFINITY) a = 0.25 * PI;
    else a = k_atan(ay / ax);
    if (xneg) a = PI - (a - PI_LO);
    return yneg ? -a : a;
}

VK_COLD float atan2f_(float y, float x) { return (float)atan2_d((double)y, (double)x); }

// asin(x) = atan2(x, sqrt((1-x)(1+x))); |x| > 1 -> NaN (hittable.rs:57)
VK_COLD float asinf_(float xf) {
    double x = (double)xf;
    double c2 = (1.0 - x) * (1.0 + x);
    if (!(c2 >= 0.0)) return (float)((x - x) / (x - x));  // NaN
    // Newton-free sqrt: f32 sqrt is correctly rounded on both sides but too coarse near
    // |x|=1, so refine it in f64 with two Heron steps (deterministic +,-,*,/ only).
    double s;
    if (c2 == 0.0) {
        s = 0.0;
    } else {
        // scale into the f32 normal range first (c2 may be as small as ~1e-14 .. 1)
        s = (double)sqrtf((float)c2);
        s = 0.5 * (s + c2 / s);
        s = 0.5 * (s + c2 / s);
    }
    return (float)atan2_d(x, s);
}

// x.powf(5.0) in schlick (util.rs:28): restated as exact-order multiplies
VK_HD float pow5f_(float x) {
    double d = (double)x;
    double d2 = d * d;
    return (float)((d2 * d2) * d);   // f64 product rounded once: agrees with a correctly rounded powf
}
#endif      // VK_MATH_LIBM

}  // namespace vk
#endif
