// twv_categorical.hpp -- the one-hot model's sampler on ONE wave (arithmetic contract AC-5).
//
// Replaces wavenet/model.py:243 (float64 softmax -> float32) + generate.py:219-231 (temperature rescale through
// np.logaddexp.reduce, np.random.choice = float64 cumsum / last / searchsorted 'right').  Class i sits in lane (i mod 64),
// block (i / 64); every sum over the classes is "per lane over the blocks in block order, then scan64 over the lanes" -- the
// same tree the CPU checker walks, so the drawn class is bit-identical.  Nothing in here is sequential
// over the classes: round 3's literal left-to-right logaddexp (255 dependent exp + log1p evaluations, 32 us per draw) pinned
// nothing -- numpy's own exp/log differ from the contract's by more than the order of the reduce does
// (tests/test_cpu.py::test_categorical_sampler_contract_against_numpy).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "twv_math.hpp"

namespace twv {

template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, BOUND);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, BOUND);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes: row_shr 1, 2, 4, 8 inside each row of 16 lanes, then lane 15 -> the next row
// (rows 1 and 3), then lane 31 -> the upper half.  Lanes without a source add +0 (the operands are never negative zero).
__device__ __forceinline__ double scan64_wave(double v)
{
    v = v + dpp_f64<0x111, 0xf, true>(v);      // row_shr:1
    v = v + dpp_f64<0x112, 0xf, true>(v);      // row_shr:2
    v = v + dpp_f64<0x114, 0xf, true>(v);      // row_shr:4
    v = v + dpp_f64<0x118, 0xf, true>(v);      // row_shr:8
    v = v + dpp_f64<0x142, 0xa, false>(v);     // row_bcast:15 -> rows 1, 3
    v = v + dpp_f64<0x143, 0xc, false>(v);     // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// maximum over the wave, in every lane (max is exact: the order does not matter)
__device__ __forceinline__ float wave_max_f32(float x)
{
#define TWV_ROR_(c_) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), (c_), 0xf, 0xf, false))
    x = __builtin_fmaxf(x, TWV_ROR_(0x128));   // row_ror:8
    x = __builtin_fmaxf(x, TWV_ROR_(0x124));   // row_ror:4
    x = __builtin_fmaxf(x, TWV_ROR_(0x122));   // row_ror:2
    x = __builtin_fmaxf(x, TWV_ROR_(0x121));   // row_ror:1
#undef TWV_ROR_
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
    return __builtin_fmaxf(__builtin_fmaxf(r0, r1), __builtin_fmaxf(r2, r3));
}

// y[k] = logit of class lane + 64 k (k < NB; classes >= Q are ignored).  Returns the drawn class (wave-uniform);
// sp_out (optional, NB floats per lane) receives the scaled probabilities of generate.py:222.
// stamps (optional, tuning aid): eight 64-bit slots that receive the chip-wide clock at the phase boundaries marked TWV_CSTAMP
// nan_out (optional): set to true when the probabilities are not numbers (a NaN or +Inf logit: every class's cdf is NaN and the search
// below would return class Q - 1) -- np.random.choice (generate.py:231) raises "ValueError: probabilities contain NaN" there, so the
// callers turn the flag into an error status instead of a sample
#define TWV_CSTAMP(i_) do { if (stamps != nullptr && lane == 0) stamps[i_] = wall_clock64(); } while (0)
template <int NB>
__device__ __forceinline__ int categorical_sample(const float (&y)[NB], const int Q, const int lane, const float temp32, const double u,
                                                  float* sp_out = nullptr, unsigned long long* stamps = nullptr, bool* nan_out = nullptr)
{
    const float ninf = __uint_as_float(0xff800000u);
    // ---- model.py:243 softmax in float64
    float mx = ninf;
    bool notnum = false;                                           // a NaN or +Inf logit: softmax is NaN in every class (model.py:243)
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (64 * k < Q) { const float v = (lane + 64 * k < Q) ? y[k] : ninf; mx = v > mx ? v : mx; notnum = notnum || !(v <= 3.402823466e38f); }
    mx = wave_max_f32(mx);
    const double m64 = (double)mx;
    double e[NB];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        e[k] = 0.0;
        if (64 * k < Q) {
            e[k] = (lane + 64 * k < Q) ? exp64_nonpos_e((double)y[k] - m64) : 0.0;
            s = (k == 0) ? e[k] : s + e[k];
        }
    }
    TWV_CSTAMP(0);                                                 // max + float64 exps done
    const double sum = readlane_f64(scan64_wave(s), 63);
    TWV_CSTAMP(1);                                                 // softmax denominator
    // ---- generate.py:220 np.log(prediction) / temperature (float32)
    const bool t_is_one = __builtin_amdgcn_readfirstlane(__float_as_int(temp32)) == 0x3f800000;
    float lp[NB];
    float m2 = ninf;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        lp[k] = ninf;
        if (64 * k < Q) {
            const float p32 = (float)(e[k] / sum);                 // tf.cast(softmax(float64), float32)
            const float lg = log_e(p32);
            const float l = t_is_one ? lg : div_(lg, temp32);      // x / 1.0f == x exactly: the IEEE sequence is skipped, not approximated
            const bool in = lane + 64 * k < Q;
            lp[k] = l;
            m2 = (in && l > m2) ? l : m2;
        }
    }
    m2 = wave_max_f32(m2);
    TWV_CSTAMP(2);                                                 // p = e / sum, log p / T, their maximum
    // ---- generate.py:221 log sum exp, max-shifted, summed in float64 in the contract's tree
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        if (64 * k < Q) {
            const double x = (lane + 64 * k < Q) ? (double)exp_e(lp[k] - m2) : 0.0;
            s2 = (k == 0) ? x : s2 + x;
        }
    }
    const float lse = m2 + log_e((float)readlane_f64(scan64_wave(s2), 63));
    TWV_CSTAMP(3);                                                 // log-sum-exp
    // ---- generate.py:221-222 scaled probabilities; RandomState.choice: float64 cdf in class order
    double c[NB];
    double base = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        c[k] = 0.0;
        if (64 * k < Q) {
            const float sp = (lane + 64 * k < Q) ? exp_e(lp[k] - lse) : 0.0f;
            if (sp_out) sp_out[k] = sp;
            const double sc = scan64_wave((double)sp);
            c[k] = (k == 0) ? sc : base + sc;
            base = readlane_f64(c[k], 63);
        }
    }
    const double last = base;
    // (the straight-line float64 exp above maps a NaN argument to a number: the logits themselves are looked at; all logits -Inf is the
    // third way to no distribution)
    if (nan_out != nullptr) *nan_out = __any(notnum) != 0 || !(last == last) || mx == ninf;
    TWV_CSTAMP(4);                                                 // scaled probabilities + float64 cdf
    // ---- cdf /= cdf[-1]; searchsorted(u, side='right'): the first class whose normalised cdf exceeds u
    // The quotient is only needed where it decides the comparison: with ul = fl(u * last), c > ul (1 + 2^-50) implies fl(c / last) > u
    // and c < ul (1 - 2^-50) implies fl(c / last) <= u (each rounding moves a value by at most 2^-53 relative); the division itself is
    // done -- for the whole wave -- only when some class falls inside that band: about once in 10^12 draws.  Same answer, always.
    const double ul = u * last;
    const double band_hi = ul * (1.0 + 0x1p-50), band_lo = ul * (1.0 - 0x1p-50);
    bool close = false;
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (64 * k < Q) close = close || (lane + 64 * k < Q && !(c[k] > band_hi) && !(c[k] < band_lo));
    const bool exact_div = __any(close) != 0;
    unsigned long long mask[NB];
    if (exact_div) {                                                // (a real branch: the divisions must not be speculated)
#pragma unroll
        for (int k = 0; k < NB; ++k) mask[k] = (64 * k < Q) ? __ballot((lane + 64 * k < Q) && (c[k] / last > u)) : 0ull;
    } else {
#pragma unroll
        for (int k = 0; k < NB; ++k) mask[k] = (64 * k < Q) ? __ballot((lane + 64 * k < Q) && (c[k] > band_hi)) : 0ull;
    }
    int idx = Q - 1;
#pragma unroll
    for (int k = NB - 1; k >= 0; --k)
        if (mask[k] != 0ull) idx = 64 * k + (int)__ffsll((long long)mask[k]) - 1;      // the lowest block with a hit wins
    TWV_CSTAMP(5);                                                 // search
    return idx;
}
#undef TWV_CSTAMP

}  // namespace twv
