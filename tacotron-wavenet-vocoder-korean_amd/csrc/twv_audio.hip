// twv_audio.hip -- MI355X (gfx950) spectrogram -> waveform path of the reference synthesizer + its C-ABI (include/twv_amd.h).
//
// Replaces, for hccho2/Tacotron-Wavenet-Vocoder-Korean (citations into /root/reference):
//   synthesizer.py:258 audio_out = inv_linear_spectrogram(wav.T, hparams)
//   utils/audio.py:77-92   inv_linear_spectrogram: _denormalize (:222-227) -> _db_to_amp (:205-206) -> ** power -> _griffin_lim -> inv_preemphasis
//   utils/audio.py:127-137 _griffin_lim: random initial phase, griffin_lim_iters x {librosa.stft -> unit phase -> librosa.istft}
//   utils/audio.py:27-30   inv_preemphasis = lfilter([1], [1, -k])
// The FFTs are plain library transforms (hipFFT, batched over every frame of every utterance); framing, windowing,
// overlap-add with the window sum-of-squares normalisation, reflect padding, phase projection, magnitude shaping and the
// de-emphasis recurrence are hand-written kernels.  The spectrogram stays in Tacotron's own (utterance, frame, bin) layout,
// which is already the batched-FFT layout.  Floating-point work: parity is by tolerance against the float64 numpy restatement.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <math.h>
#include <stdint.h>
#include <string>
#include "../../include/twv_amd.h"
#include "twv_dev.hpp"

#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return twv_fail(TWV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)
#define FFTCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipfftResult r_ = (expr);                                                                                 \
        if (r_ != HIPFFT_SUCCESS) return twv_fail(TWV_E_HIP, std::string(#expr) + ": hipfft status " + std::to_string((int)r_)); \
    } while (0)

struct twv_griffin_lim {
    int n_fft, hop, win, frames, batch, nbin, len;        // len = hop * (frames - 1) samples per utterance
    hipfftHandle c2r, r2c;
    bool have_plans;
};

#define GA_STRIDE(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)
static inline int ga_grid(long long n) { long long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g)); }

// periodic Hann of win samples, zero-padded symmetrically to n_fft  [librosa.filters.get_window + util.pad_center]
__device__ __forceinline__ float ga_window(int k, int n_fft, int win)
{
    const int j = k - (n_fft - win) / 2;
    if (j < 0 || j >= win) return 0.0f;
    return 0.5f - 0.5f * cospif(2.0f * (float)j / (float)win);
}
// utils/audio.py:222-227 + :205-206 + "** power": magnitude the Griffin-Lim iterations keep fixed
__global__ void ga_mag_kernel(const float* lin, float* mag, long long n, float max_abs, float min_db, float ref_db, float power)
{
    GA_STRIDE(i, n) {
        float x = lin[i];
        x = x < -max_abs ? -max_abs : (x > max_abs ? max_abs : x);
        const float db = ((x + max_abs) * -min_db / (2.0f * max_abs)) + min_db;
        const float s = powf(10.0f, (db + ref_db) * 0.05f);
        mag[i] = powf(s, power);
    }
}
// angles = exp(2j*pi*rand) (utils/audio.py:131): spec = mag * angles
__global__ void ga_phase_init_kernel(const float* mag, const float* u, float2* spec, long long n)
{
    GA_STRIDE(i, n) {
        float s, c;
        sincospif(2.0f * u[i], &s, &c);
        spec[i] = make_float2(mag[i] * c, mag[i] * s);
    }
}
// angles = exp(1j * angle(D)) (utils/audio.py:135): spec = mag * D / |D|   (angle(0) = 0)
__global__ void ga_phase_kernel(const float* mag, const float2* D, float2* spec, long long n)
{
    GA_STRIDE(i, n) {
        const float2 d = D[i];
        const float a = hypotf(d.x, d.y);
        const float m = mag[i];
        spec[i] = a > 0.0f ? make_float2(m * (d.x / a), m * (d.y / a)) : make_float2(m, 0.0f);
    }
}
// librosa.istft after the inverse FFTs: y[n] = sum_i w[k] * frame_i[k] / sum_i w[k]^2, k = n + n_fft/2 - i*hop  (already trimmed)
__global__ void ga_ola_kernel(const float* ft, float* y, int batch, int frames, int n_fft, int hop, int win, int len)
{
    const long long total = (long long)batch * len;
    const float inv_n = 1.0f / (float)n_fft;               // hipFFT's C2R is unnormalised, numpy's irfft divides by n
    GA_STRIDE(idx, total) {
        const int b = (int)(idx / len), n = (int)(idx - (long long)b * len);
        const int np_ = n + n_fft / 2;
        int i0 = (np_ - n_fft + hop) / hop; if (np_ - n_fft + 1 <= 0) i0 = 0;      // ceil((np - n_fft + 1) / hop) for positives
        int i1 = np_ / hop; if (i1 > frames - 1) i1 = frames - 1;
        float acc = 0.0f, wss = 0.0f;
        for (int i = i0; i <= i1; ++i) {
            const int k = np_ - i * hop;
            if (k < 0 || k >= n_fft) continue;
            const float w = ga_window(k, n_fft, win);
            acc += w * (ft[((long long)b * frames + i) * n_fft + k] * inv_n);
            wss += w * w;
        }
        y[idx] = wss > 1.17549435e-38f ? acc / wss : acc;
    }
}
// librosa.stft before the forward FFTs: frame_i[k] = w[k] * reflect_pad(y)[i*hop + k]
__global__ void ga_frame_kernel(const float* y, float* fr, int batch, int frames, int n_fft, int hop, int win, int len)
{
    const long long total = (long long)batch * frames * n_fft;
    GA_STRIDE(idx, total) {
        const int k = (int)(idx % n_fft);
        const long long bi = idx / n_fft;
        const int i = (int)(bi % frames), b = (int)(bi / frames);
        const float w = ga_window(k, n_fft, win);
        float v = 0.0f;
        if (w != 0.0f) {
            int j = i * hop + k - n_fft / 2;
            if (j < 0) j = -j;
            if (j >= len) j = 2 * (len - 1) - j;
            v = w * y[(long long)b * len + j];
        }
        fr[idx] = v;
    }
}
// utils/audio.py:27-30 lfilter([1], [1, -k]): y[n] = x[n] + k*y[n-1].  The impulse response k^m is below 1e-13 after 1024 samples
// (k = 0.97), so each thread restarts the recurrence 1024 samples before its 2048-sample chunk with a zero state.
__global__ void ga_deemph_kernel(const float* x, float* y, int batch, int len, float k, int warm)
{
    const int chunk = 2048;
    const int nchunk = (len + chunk - 1) / chunk;
    const long long total = (long long)batch * nchunk;
    GA_STRIDE(idx, total) {
        const int b = (int)(idx / nchunk), c = (int)(idx - (long long)b * nchunk);
        const float* xs = x + (long long)b * len;
        float* ys = y + (long long)b * len;
        const int n0 = c * chunk, n1 = n0 + chunk < len ? n0 + chunk : len;
        float acc = 0.0f;
        for (int n = n0 - warm > 0 ? n0 - warm : 0; n < n0; ++n) acc = xs[n] + k * acc;
        for (int n = n0; n < n1; ++n) { acc = xs[n] + k * acc; ys[n] = acc; }
    }
}

extern "C" int twv_griffin_lim_create(int n_fft, int hop, int win_length, int n_frames, int batch, twv_griffin_lim** out)
{
    if (!out || n_fft < 8 || (n_fft & 1) || hop < 1 || win_length < 1 || win_length > n_fft || n_frames < 2 || batch < 1)
        return twv_fail(TWV_E_INVALID, "bad argument");
    if ((long long)hop * (n_frames - 1) <= n_fft / 2) return twv_fail(TWV_E_INVALID, "signal shorter than the reflect padding (n_fft/2)");
    twv_griffin_lim* h = new twv_griffin_lim();
    h->n_fft = n_fft; h->hop = hop; h->win = win_length; h->frames = n_frames; h->batch = batch; h->nbin = n_fft / 2 + 1;
    h->len = hop * (n_frames - 1);
    h->have_plans = false;
    *out = h;
    return TWV_OK;
}
extern "C" void twv_griffin_lim_destroy(twv_griffin_lim* h)
{
    if (h && h->have_plans) { hipfftDestroy(h->c2r); hipfftDestroy(h->r2c); }
    delete h;
}
extern "C" int twv_griffin_lim_samples(const twv_griffin_lim* h) { return h->len; }
extern "C" size_t twv_griffin_lim_workspace_bytes(const twv_griffin_lim* h)
{
    const long long bf = (long long)h->batch * h->frames;
    return (size_t)(bf * h->nbin * 4 + bf * h->nbin * 8 * 2 + bf * h->n_fft * 4 + (long long)h->batch * h->len * 4 + 4096);
}

extern "C" int twv_inv_linear_spectrogram(twv_griffin_lim* h, const float* lin, const float* uniforms, int iters, double power, double ref_level_db,
                                          double max_abs_value, double min_level_db, double preemphasis, void* workspace, float* out, void* stream)
{
    if (!h || !lin || !uniforms || !workspace || !out || iters < 0) return twv_fail(TWV_E_INVALID, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long long bf = (long long)h->batch * h->frames, nspec = bf * h->nbin;
    if (!h->have_plans) {
        int n[1] = {h->n_fft};
        FFTCHK(hipfftPlanMany(&h->c2r, 1, n, nullptr, 1, h->nbin, nullptr, 1, h->n_fft, HIPFFT_C2R, (int)bf));
        FFTCHK(hipfftPlanMany(&h->r2c, 1, n, nullptr, 1, h->n_fft, nullptr, 1, h->nbin, HIPFFT_R2C, (int)bf));
        h->have_plans = true;
    }
    FFTCHK(hipfftSetStream(h->c2r, st));
    FFTCHK(hipfftSetStream(h->r2c, st));
    char* w = (char*)workspace;
    float* mag = (float*)w; w += (nspec * 4 + 255) / 256 * 256;
    float2* spec = (float2*)w; w += (nspec * 8 + 255) / 256 * 256;
    float2* D = (float2*)w; w += (nspec * 8 + 255) / 256 * 256;
    float* ft = (float*)w; w += (bf * h->n_fft * 4 + 255) / 256 * 256;
    float* y = (float*)w;
    hipLaunchKernelGGL(ga_mag_kernel, dim3(ga_grid(nspec)), dim3(256), 0, st, lin, mag, nspec, (float)max_abs_value, (float)min_level_db,
                       (float)ref_level_db, (float)power);
    hipLaunchKernelGGL(ga_phase_init_kernel, dim3(ga_grid(nspec)), dim3(256), 0, st, mag, uniforms, spec, nspec);
    for (int it = 0; it <= iters; ++it) {
        // librosa.istft: inverse FFT of every frame (the C2R transform may overwrite its input: spec is rebuilt each round), overlap-add
        FFTCHK(hipfftExecC2R(h->c2r, (hipfftComplex*)spec, ft));
        hipLaunchKernelGGL(ga_ola_kernel, dim3(ga_grid((long long)h->batch * h->len)), dim3(256), 0, st, ft, y, h->batch, h->frames, h->n_fft, h->hop,
                           h->win, h->len);
        if (it == iters) break;
        // librosa.stft: reflect-padded windowed frames, forward FFT, keep only the phase
        hipLaunchKernelGGL(ga_frame_kernel, dim3(ga_grid(bf * h->n_fft)), dim3(256), 0, st, y, ft, h->batch, h->frames, h->n_fft, h->hop, h->win, h->len);
        FFTCHK(hipfftExecR2C(h->r2c, ft, (hipfftComplex*)D));
        hipLaunchKernelGGL(ga_phase_kernel, dim3(ga_grid(nspec)), dim3(256), 0, st, mag, D, spec, nspec);
    }
    const int nchunk = (h->len + 2047) / 2048;
    hipLaunchKernelGGL(ga_deemph_kernel, dim3(ga_grid((long long)h->batch * nchunk)), dim3(256), 0, st, y, out, h->batch, h->len, (float)preemphasis, 1024);
    HIPCHK(hipGetLastError());
    return TWV_OK;
}
