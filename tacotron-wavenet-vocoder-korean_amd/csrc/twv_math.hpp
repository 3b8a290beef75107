// twv_math.hpp -- gfx950 device implementations of the arithmetic contract's elementary functions
// (DESIGN.md "AC-2").  Every fusion is an explicit fma; the translation unit is built with
// -ffp-contract=off, IEEE division (exp / log / float64 forms; tanh and the logistic carry no division since round 5),
// f32 subnormals kept -- so these return the same bits as the CPU checker.  Replaces tf.tanh / tf.sigmoid / tf.exp / tf.log as called by
// wavenet/model.py:86 and wavenet/mixture.py:103-111 and np.log/np.exp/np.logaddexp of
// generate.py:219-222 (implementations live in TensorFlow/Eigen/numpy, un-vendored).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace twv {

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float div_(float a, float b) { return __fdiv_rn(a, b); }

// ---- tanh / sigmoid: Eigen-3.3-style rationals (odd P / even Q), AC-2 as of round 5 ---------------------------
// The same rational P(x^2) x / Q(x^2) (+ 0.5 for the logistic) with the same coefficients as rounds 1-4, evaluated for a lone wave
// of the generation chain, which is ISSUE-bound (a layer is ~117 instructions at ~4.5 clocks each: profiles/r05_chain_contract_ubench.txt):
// numerator and denominator by Estrin's scheme (t = x^2, t^2, t^4), and instead of the IEEE division (v_div_scale x2, v_rcp_f32,
// five fmas, v_div_fmas, v_div_fixup: 12 instructions behind BOTH polynomials) a software-specified reciprocal of the denominator
// that runs next to the numerator: integer seed kRcpMagic - bits(Q) (within 5.1 % for every positive normal Q), one cubic and one
// quadratic Newton step in fma, then ONE fma(x P, r, half).  18 instructions instead of 24, dependency depth 12 instead of 21,
// every one an IEEE operation gcc reproduces (the CPU checker holds the same sequence).  Q is positive and normal on the clamped domain.
// Unified form used by the gated unit: the filter half-wave evaluates tanh, the gate half-wave the logistic, with the SAME
// instruction stream and per-lane coefficients (zero coefficients reproduce the shorter polynomials exactly: fma(t, +0, c) == c).
constexpr unsigned kRcpMagic = 0x7EF311C7u;
struct ActCoef {
    float clampv, a1, a3, a5, a7, a9, a11, a13, b0, b2, b4, b6, b8, b10, half;
};
__device__ __forceinline__ ActCoef act_coef(bool sig)
{
    ActCoef c;
    c.clampv = sig ? 18.0f : 9.0f;
    c.a13 = sig ? 0.0f : -2.76076847742355e-16f;
    c.a11 = sig ? 0.0f : 2.00018790482477e-13f;
    c.a9 = sig ? 4.37031012579801e-11f : -8.60467152213735e-11f;
    c.a7 = sig ? 1.15627324459942e-07f : 5.12229709037114e-08f;
    c.a5 = sig ? 6.08574864600143e-05f : 1.48572235717979e-05f;
    c.a3 = sig ? 8.51377133304701e-03f : 6.37261928875436e-04f;
    c.a1 = sig ? 2.48287947061529e-01f : 4.89352455891786e-03f;
    c.b10 = sig ? 6.10247389755681e-13f : 0.0f;
    c.b8 = sig ? 5.76102136993427e-09f : 0.0f;
    c.b6 = sig ? 6.29106785017040e-06f : 1.19825839466702e-06f;
    c.b4 = sig ? 1.70198817374094e-03f : 1.18534705686654e-04f;
    c.b2 = sig ? 1.16817656904453e-01f : 2.26843463243900e-03f;
    c.b0 = sig ? 9.93151921023180e-01f : 4.89352518554385e-03f;
    c.half = sig ? 0.5f : -0.0f;          // x + (-0) == x for every x, signed zeros included
    return c;
}
// the reciprocal and the final fma (shared tail of every form)
__device__ __forceinline__ float act_tail(float xp, float Q, float half)
{
    float r = __uint_as_float(kRcpMagic - __float_as_uint(Q));
    float e = fma_(-Q, r, 1.0f);
    const float s = fma_(e, e, e);
    r = fma_(r, s, r);                    // cubic step: r (1 + e + e^2)
    e = fma_(-Q, r, 1.0f);
    r = fma_(r, e, r);                    // quadratic step
    return fma_(xp, r, half);
}
// scalar form: throughput code (many independent evaluations per thread: Tacotron attention scores, GRU gates)
__device__ __forceinline__ float act_eval(const ActCoef& c, float x)
{
    x = x < -c.clampv ? -c.clampv : (x > c.clampv ? c.clampv : x);
    const float t = x * x, t2 = t * t, t4 = t2 * t2;
    const float p01 = fma_(c.a3, t, c.a1), p23 = fma_(c.a7, t, c.a5), p45 = fma_(c.a11, t, c.a9);
    const float q01 = fma_(c.b2, t, c.b0), q23 = fma_(c.b6, t, c.b4), q45 = fma_(c.b10, t, c.b8);
    const float p456 = fma_(c.a13, t2, p45);
    const float p03 = fma_(p23, t2, p01), q03 = fma_(q23, t2, q01);
    const float P = fma_(p456, t4, p03), Q = fma_(q45, t4, q03);
    return act_tail(x * P, Q, c.half);
}
// Latency-oriented form for the generation chain wave (one evaluation per layer on a lone wave, where instruction count is what
// matters): the steps of P and Q that share a multiplier run packed (v_pk_fma_f32) -- the same fmas, fewer instructions.
typedef float f32x2m __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float act_eval_pk_clamped(const ActCoef& c, float x)
{
    const float t = x * x;
    const f32x2m tt = {t, t};
    const f32x2m pq01 = __builtin_elementwise_fma(f32x2m{c.a3, c.b2}, tt, f32x2m{c.a1, c.b0});
    const f32x2m pq23 = __builtin_elementwise_fma(f32x2m{c.a7, c.b6}, tt, f32x2m{c.a5, c.b4});
    const f32x2m pq45 = __builtin_elementwise_fma(f32x2m{c.a11, c.b10}, tt, f32x2m{c.a9, c.b8});
    const float t2 = t * t;
    const float p456 = fma_(c.a13, t2, pq45[0]);
    const float t4 = t2 * t2;
    const f32x2m pq03 = __builtin_elementwise_fma(pq23, f32x2m{t2, t2}, pq01);
    const f32x2m PQ = __builtin_elementwise_fma(f32x2m{p456, pq45[1]}, f32x2m{t4, t4}, pq03);
    return act_tail(x * PQ[0], PQ[1], c.half);
}
__device__ __forceinline__ float act_eval_pk(const ActCoef& c, float x)
{
    x = x < -c.clampv ? -c.clampv : (x > c.clampv ? c.clampv : x);
    return act_eval_pk_clamped(c, x);
}
// the same with the clamp as ONE v_med3_f32; identical for every non-NaN input
__device__ __forceinline__ float act_eval_pk_med3(const ActCoef& c, float x)
{
    return act_eval_pk_clamped(c, __builtin_amdgcn_fmed3f(x, -c.clampv, c.clampv));
}
// Throughput form: TWO evaluations per lane, element by element the operations of act_eval in packed instructions (v_pk_mul_f32 /
// v_pk_fma_f32 round each half as the scalar instructions do) -- half the issue slots where a workgroup is bound by them (the Tacotron
// attention scores: 8 tanh per thread behind each other)
__device__ __forceinline__ f32x2m act_eval2(const ActCoef& c, f32x2m x)
{
    x[0] = x[0] < -c.clampv ? -c.clampv : (x[0] > c.clampv ? c.clampv : x[0]);
    x[1] = x[1] < -c.clampv ? -c.clampv : (x[1] > c.clampv ? c.clampv : x[1]);
#define TWV_SPLAT(v) f32x2m{(v), (v)}
    const f32x2m t = x * x, t2 = t * t, t4 = t2 * t2;
    const f32x2m p01 = __builtin_elementwise_fma(TWV_SPLAT(c.a3), t, TWV_SPLAT(c.a1)), p23 = __builtin_elementwise_fma(TWV_SPLAT(c.a7), t, TWV_SPLAT(c.a5)),
                 p45 = __builtin_elementwise_fma(TWV_SPLAT(c.a11), t, TWV_SPLAT(c.a9));
    const f32x2m q01 = __builtin_elementwise_fma(TWV_SPLAT(c.b2), t, TWV_SPLAT(c.b0)), q23 = __builtin_elementwise_fma(TWV_SPLAT(c.b6), t, TWV_SPLAT(c.b4)),
                 q45 = __builtin_elementwise_fma(TWV_SPLAT(c.b10), t, TWV_SPLAT(c.b8));
    const f32x2m p456 = __builtin_elementwise_fma(TWV_SPLAT(c.a13), t2, p45);
    const f32x2m p03 = __builtin_elementwise_fma(p23, t2, p01), q03 = __builtin_elementwise_fma(q23, t2, q01);
    const f32x2m P = __builtin_elementwise_fma(p456, t4, p03), Q = __builtin_elementwise_fma(q45, t4, q03);
    const f32x2m xp = x * P;
    f32x2m r = {__uint_as_float(kRcpMagic - __float_as_uint(Q[0])), __uint_as_float(kRcpMagic - __float_as_uint(Q[1]))};
    const f32x2m one = {1.0f, 1.0f};
    f32x2m e = __builtin_elementwise_fma(-Q, r, one);
    const f32x2m s = __builtin_elementwise_fma(e, e, e);
    r = __builtin_elementwise_fma(r, s, r);
    e = __builtin_elementwise_fma(-Q, r, one);
    r = __builtin_elementwise_fma(r, e, r);
    return __builtin_elementwise_fma(xp, r, TWV_SPLAT(c.half));
#undef TWV_SPLAT
}
__device__ __forceinline__ float tanh_e(float x) { return act_eval(act_coef(false), x); }
__device__ __forceinline__ f32x2m tanh_e2(f32x2m x) { return act_eval2(act_coef(false), x); }
__device__ __forceinline__ float sigmoid_e(float x) { return act_eval(act_coef(true), x); }

// ---- exp / log: Cephes single-precision forms as vectorised in Eigen 3.3 ---------------------------
// the body after the clamp to [exp_lo, exp_hi]
__device__ __forceinline__ float exp_clamped_e(float x)
{
    const float LOG2EF = 1.44269504088896341f;
    const float C1 = 0.693359375f, C2 = -2.12194440e-4f;
    const float p0 = 1.9875691500E-4f, p1 = 1.3981999507E-3f, p2 = 8.3334519073E-3f, p3 = 4.1665795894E-2f,
                p4 = 1.6666665459E-1f, p5 = 5.0000001201E-1f;
    const float fx = __builtin_floorf(fma_(x, LOG2EF, 0.5f));
    float tmp = fx * C1, z0 = fx * C2;
    x = x - tmp;
    x = x - z0;
    const float z = x * x;
    float y = p0;
    y = fma_(y, x, p1);
    y = fma_(y, x, p2);
    y = fma_(y, x, p3);
    y = fma_(y, x, p4);
    y = fma_(y, x, p5);
    y = fma_(y, z, x);
    y = y + 1.0f;
    const int32_t n = (int32_t)fx;
    const float pow2n = __uint_as_float((uint32_t)(n + 0x7f) << 23);
    return y * pow2n;
}
__device__ __forceinline__ float exp_e(float x0)
{
    const float exp_hi = 88.3762626647950f, exp_lo = -88.3762626647949f;
    float x = x0 < exp_hi ? x0 : exp_hi;
    x = x > exp_lo ? x : exp_lo;
    return exp_clamped_e(x);
}

// the body for a positive, normal, finite argument
__device__ __forceinline__ float log_normal_e(float x)
{
    const float SQRTHF = 0.707106781186547524f;
    const float p0 = 7.0376836292E-2f, p1 = -1.1514610310E-1f, p2 = 1.1676998740E-1f, p3 = -1.2420140846E-1f,
                p4 = +1.4249322787E-1f, p5 = -1.6668057665E-1f, p6 = +2.0000714765E-1f, p7 = -2.4999993993E-1f,
                p8 = +3.3333331174E-1f;
    const float q1 = -2.12194440e-4f, q2 = 0.693359375f;
    const uint32_t ix = __float_as_uint(x);
    const int32_t emm0 = (int32_t)(ix >> 23) - 0x7f;
    x = __uint_as_float((ix & ~0x7f800000u) | 0x3f000000u);
    float e = (float)emm0 + 1.0f;
    if (x < SQRTHF) { e = e - 1.0f; x = (x - 1.0f) + x; } else { x = x - 1.0f; }
    const float x2 = x * x, x3 = x2 * x;
    float y = fma_(p0, x, p1), y1 = fma_(p3, x, p4), y2 = fma_(p6, x, p7);
    y = fma_(y, x, p2);
    y1 = fma_(y1, x, p5);
    y2 = fma_(y2, x, p8);
    y = fma_(y, x3, y1);
    y = fma_(y, x3, y2);
    y = y * x3;
    y1 = e * q1;
    const float tmp = x2 * 0.5f;
    y = y + y1;
    x = x - tmp;
    y2 = e * q2;
    x = x + y;
    x = x + y2;
    return x;
}
__device__ __forceinline__ float log_e(float x)
{
    if (x != x || x < 0.0f) return __uint_as_float(0x7fc00000u);
    if (x == 0.0f) return __uint_as_float(0xff800000u);
    const float min_norm = __uint_as_float(0x00800000u);
    if (x < min_norm) x = min_norm;
    return log_normal_e(x);
}

__device__ __forceinline__ float log1p_e(float x)
{
    const float u = 1.0f + x;
    if (u == 1.0f) return x;
    return log_e(u) * div_(x, u - 1.0f);
}
// ---- float64 (model.py:243 softmax in float64; np.random.choice's float64 cdf) ---------------------
__device__ __forceinline__ double exp64_e(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return __longlong_as_double(0x7ff0000000000000LL);
    if (x < -745.13321910194110842) return 0.0;
    const double kf = __builtin_floor(x * invln2 + 0.5);
    const int k = (int)kf;
    const double hi = x - kf * ln2HI, lo = kf * ln2LO;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k >= -1021 && k <= 1023) return y * __longlong_as_double((long long)(k + 1023) << 52);
    if (k > 1023) return y * __longlong_as_double((long long)(k - 1 + 1023) << 52) * 2.0;
    return y * __longlong_as_double((long long)(k + 1000 + 1023) << 52) * __longlong_as_double((long long)(-1000 + 1023) << 52);
}

// exp64_e for a non-positive finite argument (the float64 softmax of model.py:243 evaluates exp(logit - max)), as ONE straight line:
// the same operations in the same order on every path such an argument can take -- the overflow exit is unreachable, the underflow
// exit and the subnormal scaling become selects -- so four of them interleave in one instruction stream instead of running as four
// branchy calls one after the other.  Bit-identical to exp64_e on [-inf, 0] (tests/test_wavenet_gpu.py::test_elementwise64_bit_exact).
__device__ __forceinline__ double exp64_nonpos_e(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const bool tiny = x < -745.13321910194110842;
    const double xc = tiny ? -745.0 : x;                           // keeps k in range; the result is replaced below
    const double kf = __builtin_floor(xc * invln2 + 0.5);
    const int k = (int)kf;
    const double hi = xc - kf * ln2HI, lo = kf * ln2LO;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const bool normal = k >= -1021;
    const int k1 = normal ? k : k + 1000;
    const double scaled = y * __longlong_as_double((long long)(k1 + 1023) << 52);
    const double res = normal ? scaled : scaled * __longlong_as_double((long long)(-1000 + 1023) << 52);
    return tiny ? 0.0 : res;
}

__device__ __forceinline__ double log64_e(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (x != x || x < 0.0) return __longlong_as_double(0x7ff8000000000000LL);
    if (x == 0.0) return __longlong_as_double(0xfff0000000000000LL);
    if (x == __longlong_as_double(0x7ff0000000000000LL)) return x;
    int k = 0;
    unsigned long long ix = (unsigned long long)__double_as_longlong(x);
    if ((ix >> 52) == 0) { x *= 18014398509481984.0; ix = (unsigned long long)__double_as_longlong(x); k -= 54; }
    k += (int)(ix >> 52) - 1023;
    ix = (ix & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)ix);
    if (m > 1.4142135623730951) { m *= 0.5; k += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

}  // namespace twv
