/* mathfn.c -- CPU ORACLE (test infrastructure): elementary functions of the arithmetic contract.
 *
 * The reference calls tf.tanh / tf.sigmoid / tf.exp / tf.log (wavenet/model.py:86,
 * wavenet/mixture.py:103-111) and np.log / np.exp / np.logaddexp (generate.py:219-222); their
 * implementations live in TensorFlow/Eigen/numpy, absent from /root/reference and unpinned.  The
 * contract (DESIGN.md "AC-2") fixes them as the single-precision rational / Cephes-polynomial forms
 * Eigen 3.3 ships for CPU [recalled, parity unpinned], evaluated with fused multiply-adds (tanh / logistic since round 5: Estrin
 * order and a software-specified reciprocal instead of the division, see below), so the same bits come out of gcc here and of
 * the gfx950 kernels.
 *
 * Build with -ffp-contract=off: every fusion below is an explicit fmaf().
 */
#include <math.h>
#include <string.h>
#include "twv_oracle.h"

static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* AC-2 (round 5): the rational P(x^2) x / Q(x^2) of Eigen 3.3's ptanh / plogistic -- the same coefficients as rounds 1-4 -- evaluated
 * for a lone GPU wave, where the number of instructions and the depth of the dependency chain are what a layer of the generation
 * chain costs (profiles/r05_chain_contract_ubench.txt): numerator and denominator by Estrin's scheme (t = x^2, t^2, t^4: depth 4
 * instead of 6 Horner steps), and the division replaced by a SOFTWARE-SPECIFIED reciprocal of the denominator that runs next to the
 * numerator -- integer seed 0x7EF311C7 - bits(Q) (within 5.1 % for every positive normal Q), one cubic and one quadratic Newton
 * step in fmaf (relative error of r <= 7.7e-8) -- and the result is ONE fma: fma(x P, r, half), half = 0.5 for the logistic and -0
 * for tanh (x + -0 == x, signed zeros included).  Every step is an IEEE operation a C compiler reproduces; max abs error against
 * float64: tanh 3.2e-7, logistic 1.9e-7 (rounds 1-4's Horner + IEEE division: 2.3e-7 / 1.4e-7; tests/test_cpu.py).
 * Q is positive and normal on the clamped domain (tanh: [4.9e-3, 1.61], logistic: [0.99, 497]): no special cases. */
typedef struct { float cl, a1, a3, a5, a7, a9, a11, a13, b0, b2, b4, b6, b8, b10, half; } act_coef;
static const act_coef TANH_C = { 9.0f, 4.89352455891786e-03f, 6.37261928875436e-04f, 1.48572235717979e-05f, 5.12229709037114e-08f,
                                 -8.60467152213735e-11f, 2.00018790482477e-13f, -2.76076847742355e-16f,
                                 4.89352518554385e-03f, 2.26843463243900e-03f, 1.18534705686654e-04f, 1.19825839466702e-06f, 0.0f, 0.0f, -0.0f };
static const act_coef SIGM_C = { 18.0f, 2.48287947061529e-01f, 8.51377133304701e-03f, 6.08574864600143e-05f, 1.15627324459942e-07f,
                                 4.37031012579801e-11f, 0.0f, 0.0f,
                                 9.93151921023180e-01f, 1.16817656904453e-01f, 1.70198817374094e-03f, 6.29106785017040e-06f,
                                 5.76102136993427e-09f, 6.10247389755681e-13f, 0.5f };
#define TWVO_RCP_MAGIC 0x7EF311C7u
static float act_eval(const act_coef* c, float x)
{
    x = clampf(x, -c->cl, c->cl);
    const float t = x * x, t2 = t * t, t4 = t2 * t2;
    const float p01 = fmaf(c->a3, t, c->a1), p23 = fmaf(c->a7, t, c->a5), p45 = fmaf(c->a11, t, c->a9);
    const float q01 = fmaf(c->b2, t, c->b0), q23 = fmaf(c->b6, t, c->b4), q45 = fmaf(c->b10, t, c->b8);
    const float p456 = fmaf(c->a13, t2, p45);
    const float p03 = fmaf(p23, t2, p01), q03 = fmaf(q23, t2, q01);
    const float P = fmaf(p456, t4, p03), Q = fmaf(q45, t4, q03);
    const float xp = x * P;
    float r = bits2f(TWVO_RCP_MAGIC - f2bits(Q));
    float e = fmaf(-Q, r, 1.0f);
    const float s = fmaf(e, e, e);
    r = fmaf(r, s, r);                          /* cubic step: r (1 + e + e^2) */
    e = fmaf(-Q, r, 1.0f);
    r = fmaf(r, e, r);                          /* quadratic step */
    return fmaf(xp, r, c->half);
}
/* tanh: odd 13th-degree / even 6th-degree rational, input clamped to [-9, 9] */
float twvo_tanh(float x) { return act_eval(&TANH_C, x); }
/* logistic: odd 9th-degree / even 10th-degree rational + 0.5, input clamped to [-18, 18] */
float twvo_sigmoid(float x) { return act_eval(&SIGM_C, x); }

/* exp: Cephes expf as vectorised in Eigen 3.3 (range reduction by ln2 split C1+C2, degree-5 polynomial) */
float twvo_exp(float x0)
{
    const float exp_hi = 88.3762626647950f, exp_lo = -88.3762626647949f, LOG2EF = 1.44269504088896341f;
    const float C1 = 0.693359375f, C2 = -2.12194440e-4f;
    const float p0 = 1.9875691500E-4f, p1 = 1.3981999507E-3f, p2 = 8.3334519073E-3f, p3 = 4.1665795894E-2f,
                p4 = 1.6666665459E-1f, p5 = 5.0000001201E-1f;
    float x = x0 < exp_hi ? x0 : exp_hi;
    x = x > exp_lo ? x : exp_lo;
    float fx = floorf(fmaf(x, LOG2EF, 0.5f));
    const float tmp = fx * C1, z0 = fx * C2;
    x = x - tmp;
    x = x - z0;
    const float z = x * x;
    float y = p0;
    y = fmaf(y, x, p1);
    y = fmaf(y, x, p2);
    y = fmaf(y, x, p3);
    y = fmaf(y, x, p4);
    y = fmaf(y, x, p5);
    y = fmaf(y, z, x);
    y = y + 1.0f;
    const int32_t n = (int32_t)fx;
    const float pow2n = bits2f((uint32_t)(n + 0x7f) << 23);
    return y * pow2n;
}

/* log: Cephes logf as vectorised in Eigen 3.3.  x <= 0 is outside the path's domain (returns -inf / NaN like libm). */
float twvo_log(float x)
{
    const float SQRTHF = 0.707106781186547524f;
    const float p0 = 7.0376836292E-2f, p1 = -1.1514610310E-1f, p2 = 1.1676998740E-1f, p3 = -1.2420140846E-1f,
                p4 = +1.4249322787E-1f, p5 = -1.6668057665E-1f, p6 = +2.0000714765E-1f, p7 = -2.4999993993E-1f,
                p8 = +3.3333331174E-1f;
    const float q1 = -2.12194440e-4f, q2 = 0.693359375f;
    if (x != x || x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    const float min_norm = bits2f(0x00800000u);
    if (x < min_norm) x = min_norm; /* denormals are cut off */
    uint32_t ix = f2bits(x);
    int32_t emm0 = (int32_t)(ix >> 23) - 0x7f;
    x = bits2f((ix & ~0x7f800000u) | 0x3f000000u); /* mantissa in [0.5, 1) */
    float e = (float)emm0 + 1.0f;
    if (x < SQRTHF) { e = e - 1.0f; x = (x - 1.0f) + x; } else { x = x - 1.0f; }
    const float x2 = x * x, x3 = x2 * x;
    float y = fmaf(p0, x, p1), y1 = fmaf(p3, x, p4), y2 = fmaf(p6, x, p7);
    y = fmaf(y, x, p2);
    y1 = fmaf(y1, x, p5);
    y2 = fmaf(y2, x, p8);
    y = fmaf(y, x3, y1);
    y = fmaf(y, x3, y2);
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

/* log1p for the float32 np.logaddexp of generate.py:221: argument is exp(-|a-b|) in [0, 1]. */
float twvo_log1p(float x)
{
    /* log1p(x) = log(u) * x / (u - 1) with u = 1 + x (classic correction), all in f32 */
    const float u = 1.0f + x;
    if (u == 1.0f) return x;
    return twvo_log(u) * (x / (u - 1.0f));
}

/* ---- float64: model.py:243 casts the logits to float64 before tf.nn.softmax; np.random.choice works in float64 ---- */
static inline double bits2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
static inline uint64_t d2bits(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }

/* exp: fdlibm e_exp.c algorithm (argument reduction by ln2 hi/lo, degree-5 Remez in r^2), fma-free */
double twvo_exp64(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return INFINITY;
    if (x < -745.13321910194110842) return 0.0;
    const double kf = floor(x * invln2 + 0.5);
    const int k = (int)kf;
    const double hi = x - kf * ln2HI, lo = kf * ln2LO;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    /* scale by 2^k in two steps so that subnormal results round once */
    if (k >= -1021 && k <= 1023) return y * bits2d((uint64_t)(k + 1023) << 52);
    if (k > 1023) return y * bits2d((uint64_t)(k - 1 + 1023) << 52) * 2.0;
    return y * bits2d((uint64_t)(k + 1000 + 1023) << 52) * bits2d((uint64_t)(-1000 + 1023) << 52);
}

/* log: fdlibm e_log.c algorithm */
double twvo_log64(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (x != x || x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    int k = 0;
    uint64_t ix = d2bits(x);
    if ((ix >> 52) == 0) { x *= 18014398509481984.0; ix = d2bits(x); k -= 54; } /* subnormal */
    k += (int)(ix >> 52) - 1023;
    ix = (ix & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL; /* m in [1,2) */
    double m = bits2d(ix);
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

/* chunked dot product of the arithmetic contract (DESIGN.md "AC-1"):
 * consecutive chunks of 32 terms; inside a chunk FOUR interleaved partial sums s_j (j = k mod 4), each an fma chain
 * from +0 in increasing k; chunk value (s0 + s1) + (s2 + s3); chunk values added in order. */
float twvo_cdot(const float* w, int wstride, const float* x, int K)
{
    float r = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 32) {
        const int k1 = k0 + 32 < K ? k0 + 32 : K;
        float s[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (int k = k0; k < k1; ++k) s[(k - k0) & 3] = fmaf(w[(size_t)k * wstride], x[k], s[(k - k0) & 3]);
        const float a = (s[0] + s[1]) + (s[2] + s[3]);
        r = (k0 == 0) ? a : r + a;
    }
    return r;
}

/* The same chunked dot product for `ncols` outputs at once: out[j] = twvo_cdot(w + j, ncols, x, K) for w row-major
 * (K, ncols).  Per output the order of operations is IDENTICAL to twvo_cdot; looping k outside / j inside only lets
 * gcc vectorise across outputs. */
void twvo_cdot_rows(const float* w, int ncols, const float* x, int K, float* out)
{
    float s[4][1024];
    for (int j0 = 0; j0 < ncols; j0 += 1024) {
        const int n = ncols - j0 < 1024 ? ncols - j0 : 1024;
        for (int k0 = 0; k0 < K; k0 += 32) {
            const int k1 = k0 + 32 < K ? k0 + 32 : K;
            for (int q = 0; q < 4; ++q) for (int j = 0; j < n; ++j) s[q][j] = 0.0f;
            for (int k = k0; k < k1; ++k) {
                const float xv = x[k];
                const float* wr = w + (size_t)k * ncols + j0;
                float* sq = s[(k - k0) & 3];
                for (int j = 0; j < n; ++j) sq[j] = fmaf(wr[j], xv, sq[j]);
            }
            for (int j = 0; j < n; ++j) {
                const float a = (s[0][j] + s[1][j]) + (s[2][j] + s[3][j]);
                out[j0 + j] = (k0 == 0) ? a : out[j0 + j] + a;
            }
        }
    }
}

/* AC-1b (round 5), for the two contractions ON the generation chain's sample-to-sample path (model.py:68-69 conv_filter|conv_gate,
 * model.py:89 dense): the contraction's LAST 32-term chunk does not start its chain 0 from +0 but from the ADDEND -- everything
 * that is added to the contraction anyway and is known before the chunk's operand exists (the earlier chunks, the bias, the gc and lc
 * projections) -- so no add follows the dot product on the dependency chain.
 *   twvo_cdot_rows_head: r[j] = the chunks before the last one, added in order (AC-1); returns 0 when there is none (K <= 32)
 *   twvo_cdot_rows_tail: out[j] = (s0 + s1) + (s2 + s3) of the last chunk with s0 started from addend[j] and s1..s3 from their FIRST
 *                        PRODUCT (= an fma chain from -0): the kernels need no zero-initialised accumulators for them */
int twvo_cdot_rows_head(const float* w, int ncols, const float* x, int K, float* r)
{
    const int klast = ((K - 1) / 32) * 32;
    if (klast <= 0) return 0;
    twvo_cdot_rows(w, ncols, x, klast, r);
    return 1;
}
void twvo_cdot_rows_tail(const float* w, int ncols, const float* x, int K, const float* addend, float* out)
{
    /* per output the order of operations is the one stated above; looping k outside / j inside only lets gcc vectorise across outputs
     * (as twvo_cdot_rows does) */
    const int k0 = ((K - 1) / 32) * 32;
    float s[4][1024];
    for (int j0 = 0; j0 < ncols; j0 += 1024) {
        const int n = ncols - j0 < 1024 ? ncols - j0 : 1024;
        for (int j = 0; j < n; ++j) { s[0][j] = addend[j0 + j]; s[1][j] = -0.0f; s[2][j] = -0.0f; s[3][j] = -0.0f; }   /* fmaf(w, x, -0) == w * x */
        for (int k = k0; k < K; ++k) {
            const float xv = x[k];
            const float* wr = w + (size_t)k * ncols + j0;
            float* sq = s[(k - k0) & 3];
            for (int j = 0; j < n; ++j) sq[j] = fmaf(wr[j], xv, sq[j]);
        }
        for (int j = 0; j < n; ++j) out[j0 + j] = (s[0][j] + s[1][j]) + (s[2][j] + s[3][j]);
    }
}
