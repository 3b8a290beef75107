/* twv_amd.h -- C-ABI of the MI355X (gfx950) WaveNet-vocoder hot path.
 *
 * Drop-in boundary for hccho2/Tacotron-Wavenet-Vocoder-Korean's WaveNet generation path.  The reference has
 * no FFI of its own: the path sits behind Python callables on a TensorFlow session.
 * Each entry point below names the reference callable it replaces (file:line in /root/reference); the
 * Python host in tacotron-wavenet-vocoder-korean_amd/ binds them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions: plain C, raw DEVICE pointers + explicit sizes, no framework types.  Every buffer is owned by
 * the caller; the handle owns host-side metadata only.  `stream` is a hipStream_t passed as void* (NULL =
 * default stream).  All calls are asynchronous on `stream` unless stated.  Return 0 = TWV_OK, else a TWV_E_*
 * code with text in twv_last_error() (thread-local).  One handle per GPU; thread-compatible, not thread-safe.
 */
#ifndef TWV_AMD_H
#define TWV_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TWV_MAX_LAYERS 64

enum { TWV_OK = 0, TWV_E_INVALID = 1, TWV_E_UNSUPPORTED = 2, TWV_E_HIP = 3, TWV_E_KERNEL = 4,
       TWV_E_BUSY = 5 /* a persistent kernel found the device occupied by other work: nothing was done, retry later */ };

/* WaveNetModel constructor arguments (wavenet/model.py:8-10) + hparams.upsample_factor (hparams.py:79) */
typedef struct {
    int32_t n_layers;
    int32_t dilations[TWV_MAX_LAYERS];   /* hparams.dilations */
    int32_t residual_channels;           /* R, must be 32 */
    int32_t dilation_channels;           /* D, must be 32 */
    int32_t skip_channels;               /* S, multiple of 64, <= 1024 */
    int32_t quantization_channels;       /* Q */
    int32_t out_channels;                /* MoL parameter count (3*nr_mix <= 63); ignored when !scalar_input */
    int32_t scalar_input;                /* 1: raw/mulaw scalar input + MoL output; 0: one-hot input + Q-way softmax */
    int32_t initial_filter_width;        /* <= 64 */
    int32_t use_biases;
    int32_t gc_channels;                 /* 0 = no global conditioning, else <= 64 */
    int32_t gc_cardinality;              /* 0 with gc_channels > 0: no gc_embedding table; every `gc_ids` argument then points at (B, gc_channels)
                                            float embeddings instead of int32 ids (_embed_gc's second branch, model.py:199-207) */
    int32_t lc_channels;                 /* num_mels; 0 = no local conditioning, else <= 128 */
    int32_t n_upsample;                  /* len(upsample_factor) <= 4 */
    int32_t upsample_factor[4];
} twv_wavenet_dims;

typedef struct twv_wavenet twv_wavenet;   /* opaque, host-side */

const char* twv_last_error(void);
const char* twv_version(void);

/* WaveNetModel.__init__ (model.py:8-30): validates the dims, computes layouts.  No device memory is touched. */
int twv_wavenet_create(const twv_wavenet_dims* dims, twv_wavenet** out);
void twv_wavenet_destroy(twv_wavenet* h);

/* WaveNetModel.calculate_receptive_field (model.py:31-39) */
int twv_wavenet_receptive_field(const twv_wavenet* h);
int twv_wavenet_hop_size(const twv_wavenet* h);              /* prod(upsample_factor) */

/* sizes (bytes unless noted) of the caller-owned device buffers */
size_t twv_wavenet_blob_floats(const twv_wavenet* h);        /* canonical checkpoint blob, in floats */
size_t twv_wavenet_packed_bytes(const twv_wavenet* h);       /* streaming-layout weights */
size_t twv_wavenet_state_bytes(const twv_wavenet* h, int batch);              /* delay lines etc. (model.py:49-64) */
size_t twv_wavenet_cond_bytes(const twv_wavenet* h, int batch, int n_steps);  /* hoisted lc/gc projections */

/* tf.train.Saver.restore equivalent (generate.py:157-161): re-lays the canonical blob (checkpoint tensors,
 * TF layouts, order of DESIGN.md) into the streaming layout.  blob and packed are device pointers. */
int twv_wavenet_pack(const twv_wavenet* h, const float* blob, void* packed, void* stream);

/* net.queue_initializer (model.py:64, generate.py:163): zero every delay line of every stream. */
int twv_wavenet_reset_state(const twv_wavenet* h, void* state, int batch, void* stream);

/* WaveNetModel.create_upsample (model.py:102-111): mel (B, T_mel, lc) -> out (B, T_mel*hop, lc).
 * scratch: device buffer of at least the output size (ping-pong for the transposed-conv stages). */
int twv_wavenet_upsample(const twv_wavenet* h, const void* packed, const float* mel, int batch, int t_mel,
                         float* out, float* scratch, void* stream);

/* Hoists the per-step 1x1 projections of _create_dilation_layer (model.py:71-83) out of the sample loop:
 * lc_filter/lc_gate of every layer for all n_steps frames of `upsampled` (B, n_steps, lc), and gc_filter/gc_gate
 * of gc_embedding[gc_ids] (model.py:181-212).  cond is twv_wavenet_cond_bytes(batch, n_steps). */
int twv_wavenet_condition(const twv_wavenet* h, const void* packed, const float* upsampled, const int32_t* gc_ids,
                          int batch, int n_steps, void* cond, void* stream);

/* Fused conditioning: when the XCD-per-stream kernel serves this (handle, batch) -- see the "xcd"
 * option -- create_upsample (model.py:102-111) and the lc projections (model.py:75-83) run INSIDE the generation launch, row by
 * row, ahead of the sample loop: neither the (B, T, lc) upsampled tensor nor a projection table is ever materialised.
 *   twv_wavenet_fused_conditioning : 1 if available for this handle and batch, else 0
 *   twv_wavenet_cond_bytes_mel     : size of `cond` for twv_wavenet_condition_mel
 *   twv_wavenet_condition_mel      : mel (B, t_mel, lc) as generate.py:151 hands it over, gc_ids (B) -> cond; the following
 *                                    twv_wavenet_generate call may run up to t_mel*hop steps (row t = frame pushed at step t).
 * Results are bit-identical to twv_wavenet_upsample + twv_wavenet_condition. */
int twv_wavenet_fused_conditioning(const twv_wavenet* h, int batch);
/* the generation kernel twv_wavenet_generate launches for this (handle, batch, options) on the current device:
 * "wn_xcd_generate_kernel", "wn_xcd_many_kernel" or "wn_generate_kernel" (static string; measurement label, bench.py) */
const char* twv_wavenet_kernel_name(const twv_wavenet* h, int batch);
size_t twv_wavenet_cond_bytes_mel(const twv_wavenet* h, int batch, int t_mel);
int twv_wavenet_condition_mel(const twv_wavenet* h, const void* packed, const float* mel, const int32_t* gc_ids,
                              int batch, int t_mel, void* cond, void* stream);

/* predict_proba_incremental (model.py:215-245) iterated by the generate.py:202-233 host loop, as ONE persistent
 * kernel launch per call: n_steps autoregressive steps for `batch` independent streams.
 *   first_input : (B) the sample fed at step 0 (generate.py:204 waveform[:,-1:]); float32 when scalar_input,
 *                 int32 class ids otherwise.  Later steps feed back the sample just drawn (generate.py:233).
 *   uniforms    : scalar_input: (B, n_steps, nr_mix+1) float32 in [1e-5, 1-1e-5] -- the two tf.random_uniform draws
 *                 of mixture.py:103,110;  one-hot: (B, n_steps) float64 in [0,1) -- np.random.choice's draw
 *                 (generate.py:231).  Injected so that results are reproducible (the reference is unseeded).
 *   temperature : generate.py:219-222 (one-hot only).
 *   out         : (B, n_steps) float32 samples in [-1,1] (scalar_input) or int32 class ids.
 *   status      : device int32[4]; [0] = 0 on success, else an internal code (watchdogs; 31 = NaN class probabilities; checked by twv_wavenet_status).
 * State carries over between calls (n_steps=1 reproduces a single sess.run of generate.py:211).
 * cond must cover the same n_steps as this call (row t = frame pushed at step t). */
int twv_wavenet_generate(const twv_wavenet* h, const void* packed, void* state, const void* cond,
                         const void* first_input, const void* uniforms, double temperature,
                         int batch, int n_steps, void* out, int32_t* status, float* debug, int debug_steps,
                         void* stream);

/* Priming (generate.py:168-180, --wav_seed): feeds `inputs` (B, n_steps; float32 | int32 ids) one per step WITHOUT sampling,
 * exactly like the reference's priming loop that discards next_sample: only the causal queue and the delay lines advance.
 * cond as for twv_wavenet_generate (the reference primes with an all-zero local condition). */
int twv_wavenet_prime(const twv_wavenet* h, const void* packed, void* state, const void* cond, const void* inputs,
                      int batch, int n_steps, int32_t* status, void* stream);

/* synchronises `stream` and converts a non-zero status word into an error: TWV_E_BUSY (the generation kernel's role workgroups could
 * not all become resident because other kernels hold CUs -- the launch did nothing, the state is unchanged, retry), TWV_E_INVALID (the
 * conditioning buffer was built for another kernel selection) or TWV_E_KERNEL (a wait inside the kernel ran out: watchdog code). */
int twv_wavenet_status(const int32_t* status, void* stream);

/* launch geometry knobs (performance only, results are bit-identical): "xcd" = 1 (default): the XCD-per-stream kernel
 * (stream b on XCD b % 8, up to four streams per XCD; every weight register-resident across the XCD's CUs, fused conditioning)
 * whenever the model is the hparams-default MoL shape (scalar input, initial_filter_width 32, skip_channels 512,
 * out_channels <= 32, <= 50 layers: hparams.py's default stack), batch <= 32 (<= 16 with more than 30 layers) and the device has
 * 256 CUs; 0 = always the generic kernel.  Set it (and "groups") BEFORE sizing / resetting the
 * state and conditioning buffers.  "groups" = workgroups per stream of the generic kernel (0 auto; an explicit value also selects
 * the generic kernel), "workers" =
 * worker waves per stream workgroup (4 | 3), "helpers" = 1 (default: conv1d_1 and conv1d_2's chunk partials run in one helper
 * workgroup per stream slice when twice the workgroups are co-resident) | 2 (conv1d_1 only) | 0 (none). */
int twv_wavenet_set_option(twv_wavenet* h, const char* name, int value);

/* optional in-kernel phase timestamps (tuning aid): device uint64[steps][80]; per step of stream 0's chain wave:
 * [0] step start, [1] causal layer done, [2] residual stack done, [3..6] the four workgroup barriers (skip sum, conv1d_1,
 * conv1d_2, sample), [7] wall clock (100 MHz), [8+l] layer l done.  Shader-clock ticks (s_memtime).  NULL disables. */
int twv_wavenet_set_profile_buffer(twv_wavenet* h, void* dev_u64, int steps);

/* wavenet/ops.py:22-33 mu_law_encode, ops.py:36-47 mu_law_decode (quantization True / False) */
int twv_mu_law_encode(const float* audio, int64_t n, int quantization_channels, int32_t* out, void* stream);
int twv_mu_law_decode(const int32_t* q, int64_t n, int quantization_channels, float* out, void* stream);
int twv_mu_law_expand(const float* y, int64_t n, int quantization_channels, float* out, void* stream);

/* utils/audio.py:14-17 save_wav's peak normalisation, per row: out = int16(wav * float32(32767 / max(0.01, max|wav|))).
 * wav (rows, n) float, out (rows, n) int16, scratch: rows*64 floats. */
int twv_wav_to_int16(const float* wav, int rows, int64_t n, int16_t* out, float* scratch, void* stream);

/* generate.py:219-231 on rows of logits in device memory: model.py:243 float64 softmax -> float32, the temperature rescale
 * (np.log(p) / T, minus its log-sum-exp, np.exp) and legacy np.random.choice (float64 cumsum / last / searchsorted 'right')
 * with the uniform draw injected.  logits (rows, Q) float, uniforms (rows) double in [0,1), out (rows) int32 class ids,
 * proba (rows, Q) float = generate.py:222's scaled_prediction, or NULL.  The generation kernels draw with the same code.
 * A row whose probabilities are not numbers (a NaN or infinite logit) gets class id -1: np.random.choice raises
 * "ValueError: probabilities contain NaN" on it.  Inside twv_wavenet_generate the same condition sets status code 31, which
 * twv_wavenet_status turns into TWV_E_KERNEL with that explanation. */
int twv_sample_categorical(const float* logits, int64_t rows, int quantization_channels, double temperature, const double* uniforms,
                           int32_t* out, float* proba, void* stream);

/* elementary functions of the arithmetic contract, evaluated on the device (parity tests pin them bit for bit) */
int twv_eval_elementwise(int fn /*0 tanh,1 sigmoid,2 exp,3 log,4 log1p*/, const float* x, int64_t n, float* out, void* stream);
int twv_eval_elementwise64(int fn /*0 exp,1 log,2 exp for x <= 0 (the sampler's straight-line form)*/, const double* x, int64_t n, double* out, void* stream);

/* ======================================= Tacotron text -> mel inference =======================================
 * Replaces the graph synthesizer.py:56 builds with Tacotron.initialize(inputs, input_lengths, num_speakers, speaker_id,
 * rnn_decoder_test_mode=True) (tacotron/tacotron.py:36-235) and runs in one sess.run (synthesizer.py:160); default
 * hparams path: model_type 'deepvoice' with num_speakers > 1, attention_type 'bah_mon_norm'.  Fields = hparams.py:126-165. */
typedef struct {
    int32_t n_symbols, embedding_size, num_speakers, speaker_embedding_size;
    int32_t enc_prenet_sizes[2], enc_bank_size, enc_bank_channel_size, enc_proj_sizes[2], enc_proj_width, enc_highway_depth, enc_rnn_size;
    int32_t attention_size, attention_state_size;
    int32_t dec_prenet_sizes[2], dec_layer_num, dec_rnn_size;
    int32_t post_bank_size, post_bank_channel_size, post_proj_sizes[2], post_proj_width, post_highway_depth, post_rnn_size;
    int32_t num_mels, reduction_factor, num_freq, max_iters;
    int32_t model_simple;   /* 0: hparams.model_type 'deepvoice' (hparams.py:123, the default); 1: 'simple' (tacotron.py:85-90: the speaker embedding is
                             * concatenated inside the decoder, rnn_wrappers.py:425-432 / 455-463); read only when num_speakers > 1 */
} twv_tacotron_dims;
typedef struct twv_tacotron twv_tacotron;

int twv_tacotron_create(const twv_tacotron_dims* dims, twv_tacotron** out);          /* Tacotron(hparams) */
void twv_tacotron_destroy(twv_tacotron* h);
size_t twv_tacotron_blob_floats(const twv_tacotron* h);     /* canonical blob: checkpoint tensors in the order of weights.tacotron_specs,
                                                               batch-norm (gamma,beta,mean,var) replaced by the derived (inv, shift) pair */
size_t twv_tacotron_packed_bytes(const twv_tacotron* h);
size_t twv_tacotron_workspace_bytes(const twv_tacotron* h, int batch, int t_in);
int twv_tacotron_pack(const twv_tacotron* h, const float* blob, void* packed, void* stream);     /* saver.restore, synthesizer.py:69-70 */
/* one synthesize() pass (synthesizer.py:126-160): tokens (B,T_in) int32 (0 pad, 1 EOS), input_lengths (B), speaker ids (B) ->
 * mel (B, max_iters*r, num_mels), linear (B, max_iters*r, num_freq) or NULL, alignments (B, T_in, max_iters) or NULL. */
int twv_tacotron_infer(const twv_tacotron* h, const void* packed, const int32_t* tokens, const int32_t* lengths,
                       const int32_t* speaker_ids, int batch, int t_in, void* workspace, float* mel, float* linear,
                       float* alignments, int32_t* status, void* stream);

/* launch geometry (performance only, results are bit-identical): "decoder_groups" = 0 auto (the XCD-resident decoder kernel wherever it
 * fits -- batch <= 32, t_in <= 512, 256 CUs, decoder widths divisible by 4, not model_type 'simple' -- else the split kernel with 16 / 8 / 4
 * workgroups per utterance), 1/2/4/8/16 = the split kernel with that many workgroups per utterance, 32 = the XCD-resident kernel or an
 * error, -1 = the single-workgroup kernel.  "gemm_group" / "highway_stack" / "gemm_valu": 0 / 0 / 1 select the older launch forms of the
 * dense layers (one launch per problem, one per highway layer, the VALU kernel) for A/B runs and cross-checks. */
int twv_tacotron_set_option(twv_tacotron* h, const char* name, int value);
/* optional decoder phase timestamps (tuning aid): device uint64[max_iters][16], s_memtime ticks of utterance 0's workgroup at the
 * phase boundaries of every decoder step (prenet, attention GRU, query, score, recurrence, context, projection, residual GRUs,
 * output).  NULL disables. */
int twv_tacotron_set_profile_buffer(twv_tacotron* h, void* dev_u64);
/* measurement aid for bench.py's `tacotron.roofline` (no reference counterpart): after twv_tacotron_set_option(h, "gemm_timing", 1)
 * every dense contraction of the pass (CBHG conv banks / projections / highways of tacotron/modules.py:25-74, the attention keys and the
 * linear projection) is bracketed by HIP events on its stream; this returns the useful FLOPs (2*rows*K*N), the summed kernel time
 * and the launch count since then. */
int twv_tacotron_gemm_stats(twv_tacotron* h, double* flop, double* ms, int64_t* launches);
/* the decoder kernel twv_tacotron_infer launches for this (handle, batch, t_in, options) on the current device:
 * "tc_decoder_x_kernel" (XCD-resident), "tc_decoder_g_kernel" (split) or "tc_decoder_kernel" (static string; measurement label, bench.py) */
const char* twv_tacotron_decoder_kernel_name(const twv_tacotron* h, int batch, int t_in);

/* ======================================= WaveNet teacher-forced training step =======================================
 * Replaces one `sess.run([net.loss, net.optimize])` of train_vocoder.py:155-181 for the scalar-input (MoL) model:
 * add_loss (wavenet/model.py:247-312: drop last sample, create_upsample, 'valid' convolution network with the
 * front-sliced local condition, discretized_mix_logistic_loss(num_class=2**16) mean) and add_optimizer
 * (model.py:314-346: Adam with TF defaults, then ExponentialMovingAverage(0.9999).apply).  Parameters and gradients are
 * flat float32 device arrays in the canonical checkpoint order (twv_wavenet_blob_floats / weights.tensor_specs).
 * Data-parallel training all-reduces `grads` between the two calls (host side, RCCL). */
typedef struct twv_wavenet_trainer twv_wavenet_trainer;
/* batch = hparams.wavenet_batch_size, n_samples = crop length fed by DataFeederWavenet (a multiple of the hop size) */
int twv_wavenet_train_create(const twv_wavenet_dims* dims, int batch, int n_samples, twv_wavenet_trainer** out);
void twv_wavenet_train_destroy(twv_wavenet_trainer* h);
size_t twv_wavenet_train_param_floats(const twv_wavenet_trainer* h);
size_t twv_wavenet_train_workspace_bytes(const twv_wavenet_trainer* h);
/* forget which workspace has been cleared: the next twv_wavenet_train_loss_grad clears the one it is given.  Call it when a workspace
 * was freed and re-allocated (a caching allocator may hand out the same address) or written by anything else between two steps. */
int twv_wavenet_train_reset_workspace(twv_wavenet_trainer* h);
int twv_wavenet_train_output_width(const twv_wavenet_trainer* h);          /* n_samples - receptive_field (model.py:135) */
/* loss (device float[1]) and d loss / d params (device float[param_floats], overwritten).
 * audio (B, n_samples) float in [-1,1]; lc (B, n_samples/hop, lc_channels); gc_ids (B) int32.
 * workspace: workspace_bytes of device memory that belongs to the trainer between calls (a pointer seen for the first time is
 * cleared once; parts of it -- activation rows in front of a layer's receptive offset -- are never written afterwards and read as zeros). */
int twv_wavenet_train_loss_grad(twv_wavenet_trainer* h, const float* params, const float* audio, const float* lc,
                                const int32_t* gc_ids, void* workspace, float* loss, float* grads, void* stream);
/* model.py:300-312 optional L2 term over the non-bias variables (after loss_grad): loss += strength*sum(w^2)/2, grads += strength*w.
 * workspace: the trainer's workspace (the call uses a scratch region of its own at the end of it: nothing loss_grad keeps there is touched). */
int twv_wavenet_train_l2(twv_wavenet_trainer* h, const float* params, double strength, void* workspace, float* loss, float* grads,
                         void* stream);
/* model.py:330-331 tf.clip_by_global_norm on the flat gradient buffer: grads <- grads*pre_scale * clip_norm / max(||grads*pre_scale||, clip_norm).
 * scratch: n + 1024 floats of the caller's -- NOT the training workspace (rows of that one must stay as loss_grad left them). */
int twv_clip_by_global_norm(float* grads, int64_t n, double pre_scale, double clip_norm, void* scratch, void* stream);
/* tf.train.AdamOptimizer.apply_gradients (t = 1-based update count) on grads*grad_scale, then the EMA shadow update. */
int twv_adam_ema_step(float* params, const float* grads, float* m, float* v, float* ema, int64_t n, double lr, double beta1,
                      double beta2, double eps, int64_t t, double ema_decay, double grad_scale, void* stream);

/* ======================================= spectrogram -> waveform (Griffin-Lim) =======================================
 * Replaces synthesizer.py:258 `inv_linear_spectrogram(wav.T, hparams)` (utils/audio.py:77-92, 127-146, 27-30): denormalise,
 * dB -> amplitude, ** power, Griffin-Lim with librosa's stft/istft conventions, inverse pre-emphasis.  FFTs by hipFFT. */
typedef struct twv_griffin_lim twv_griffin_lim;
/* n_fft = hparams.fft_size, hop = hop_size, win_length = win_size; n_frames spectrogram frames per utterance */
int twv_griffin_lim_create(int n_fft, int hop, int win_length, int n_frames, int batch, twv_griffin_lim** out);
void twv_griffin_lim_destroy(twv_griffin_lim* h);
int twv_griffin_lim_samples(const twv_griffin_lim* h);                 /* hop * (n_frames - 1) samples per utterance */
size_t twv_griffin_lim_workspace_bytes(const twv_griffin_lim* h);
/* lin (batch, n_frames, n_fft/2+1): the normalised linear spectrogram exactly as twv_tacotron_infer emits it; uniforms (same shape)
 * in [0,1) replace np.random.rand of utils/audio.py:131; out (batch, samples).  iters = hparams.griffin_lim_iters. */
int twv_inv_linear_spectrogram(twv_griffin_lim* h, const float* lin, const float* uniforms, int iters, double power, double ref_level_db,
                               double max_abs_value, double min_level_db, double preemphasis, void* workspace, float* out, void* stream);

/* cross-lane primitive self-test (device float[256]); used by the gpu tests to pin v_permlane32_swap / v_readlane semantics */
int twv_selftest(float* out256, void* stream);
/* test aid: `blocks` workgroups that each hold `lds_bytes` of LDS and spin for `milliseconds` on `stream` -- stands in for "another
 * kernel is using the device" in the co-residency test of the persistent generation kernel (TWV_E_BUSY). */
int twv_debug_occupy(int blocks, int lds_bytes, double milliseconds, void* stream);

/* ---- checkpoint helper (host only) ----
 * CRC-32C of `n` bytes, continuing from `crc` (0 to start): the per-tensor / per-block checksum of the TensorFlow
 * tensor-bundle files that tf.train.Saver writes (train_vocoder.py:133,176 / generate.py:158-161); used by
 * checkpoint.py to verify bundles on read and to stamp the ones it writes.  No device work. */
uint32_t twv_crc32c(const void* data, size_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif
