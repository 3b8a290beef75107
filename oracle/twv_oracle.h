/* twv_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's WaveNet hot path
 * (/root/reference/wavenet/model.py, ops.py, mixture.py, generate.py).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.
 *
 * PARITY UNPINNED: the reference is Python on TensorFlow 1.x (un-vendored,
 * version unpinned, absent from this image) and ships no tests or golden
 * vectors.  This restatement follows the reference
 * SOURCE TEXT; every arithmetic detail TensorFlow/Eigen leaves unspecified
 * (reduction order, tanh/sigmoid/exp/log implementation) is fixed by the
 * "arithmetic contract" in DESIGN.md so that CPU and MI355X results can be
 * compared bit for bit.
 */
#ifndef TWV_ORACLE_H
#define TWV_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TWVO_MAX_LAYERS 64

typedef struct {
    int n_layers;
    int dilations[TWVO_MAX_LAYERS];
    int R;            /* residual_channels   (hparams.py:71) */
    int D;            /* dilation_channels   (hparams.py:72) */
    int S;            /* skip_channels       (hparams.py:75) */
    int Q;            /* quantization_channels (hparams.py:73) */
    int O;            /* network output channels: out_channels (scalar_input) or Q */
    int scalar_input; /* hparams.py:63 */
    int ifw;          /* initial_filter_width (hparams.py:78); 2 (=filter_width) when !scalar_input */
    int use_bias;     /* hparams.py:76 */
    int G;            /* gc_channels, 0 = no global conditioning */
    int gc_card;      /* global_condition_cardinality */
    int L;            /* local_condition_channels (num_mels), 0 = none */
    int n_up;         /* len(upsample_factor) */
    int up[4];        /* upsample_factor (hparams.py:79) */
} twvo_dims;

/* canonical blob = TF checkpoint tensors, row-major, in the order of DESIGN.md "canonical blob" */
size_t twvo_blob_floats(const twvo_dims* d);
int    twvo_receptive_field(const twvo_dims* d);          /* model.py:31-39 */
int    twvo_hop(const twvo_dims* d);

/* elementary functions of the arithmetic contract (exported so tests can pin them) */
float  twvo_tanh(float x);
float  twvo_sigmoid(float x);
float  twvo_exp(float x);
float  twvo_log(float x);
double twvo_exp64(double x);
double twvo_log64(double x);
float  twvo_log1p(float x);
float  twvo_cdot(const float* w, int wstride, const float* x, int K);
void   twvo_cdot_rows(const float* w, int ncols, const float* x, int K, float* out); /* out[j] = twvo_cdot(w+j, ncols, x, K) */
int    twvo_cdot_rows_head(const float* w, int ncols, const float* x, int K, float* r);                       /* AC-1b: chunks before the last */
void   twvo_cdot_rows_tail(const float* w, int ncols, const float* x, int K, const float* addend, float* out); /* AC-1b: last chunk from the addend */

/* ops.py:22-47 */
void twvo_mu_law_encode(const float* audio, int n, int Q, int32_t* out);
void twvo_mu_law_decode(const int32_t* q, int n, int Q, float* out);          /* quantization=True  */
void twvo_mu_law_expand(const float* y, int n, int Q, float* out);            /* quantization=False */

/* model.py:102-111 create_upsample: mel (B,Tm,L) -> out (B,Tm*hop,L) */
void twvo_upsample(const twvo_dims* d, const float* blob, const float* mel, int B, int Tm, float* out);

typedef struct twvo_state twvo_state;
twvo_state* twvo_state_new(const twvo_dims* d, int B);    /* model.py:49-64 _create_queue (zeros) */
void        twvo_state_reset(twvo_state* s);              /* generate.py:163 queue_initializer */
void        twvo_state_free(twvo_state* s);

/* one incremental step (model.py:215-245 up to the raw network output).
 * in_scalar (B) floats when scalar_input else in_q (B) int32 class ids; lc (B,L) or NULL; gc_ids (B) or NULL.
 * raw_out (B,O).  dbg (optional, per stream b: n_layers*(D) gated outputs z then n_layers*R layer outputs) */
void twvo_step(const twvo_dims* d, const float* blob, twvo_state* s,
               const float* in_scalar, const int32_t* in_q, const float* lc, const int32_t* gc_ids,
               float* raw_out, float* dbg_z, float* dbg_x);

/* mixture.py:84-114 with injected uniforms: y (O=3*nr_mix), u (nr_mix+1): u[0..nr_mix) gumbel, u[nr_mix] logistic */
float twvo_sample_mol(const float* y, int nr_mix, const float* u);

/* generate.py:199-233 host loop, scalar_input (MoL).  U (B,T,L) upsampled lc; seed (B) = waveform[:,-1] before the loop;
 * u (B,T,nr_mix+1) injected uniforms in [1e-5,1-1e-5]; out (B,T). */
/* threads for twvo_generate_mol (one stream per thread); twvo_max_threads = host cores */
void twvo_set_threads(int n);
int twvo_max_threads(void);
void twvo_generate_mol(const twvo_dims* d, const float* blob, twvo_state* s, const float* U, const int32_t* gc_ids,
                       const float* seed, const float* u, int B, int T, float* out);

/* generate.py:199-233, one-hot input: float64 softmax (model.py:243), temperature rescale (generate.py:219-222),
 * legacy np.random.choice (generate.py:231) == searchsorted(cumsum(p)/last, u, 'right').  u (B,T) float64 in [0,1). */
void twvo_generate_mulaw(const twvo_dims* d, const float* blob, twvo_state* s, const float* U, const int32_t* gc_ids,
                         const int32_t* seed, const double* u, double temperature, int B, int T, int32_t* out);
/* the categorical sampler alone: logits (Q) -> class id */
int  twvo_sample_categorical(const float* logits, int Q, double temperature, double u, float* proba_out);
/* round 3's all-sequential form of the same sampler (comparison test only) and the scan primitive of the current one */
int  twvo_sample_categorical_sequential(const float* logits, int Q, double temperature, double u, float* proba_out);
void twvo_scan64(double v[64]);

/* train-mode (full convolution) forward, model.py:112-167 with train_mode=True.
 * input (B,Tin) scalar or int32 ids, lc_up (B,Tlc,L) or NULL (sliced from the front, model.py:79-80), raw_out (B,Tin-RF+1,O). */
void twvo_forward_full(const twvo_dims* d, const float* blob, int B, int Tin, const float* in_scalar, const int32_t* in_q,
                       const float* lc_up, int Tlc, const int32_t* gc_ids, float* raw_out);

/* ---------------- Tacotron text -> mel inference (tacotron.c) ---------------- */
typedef struct {
    int n_symbols, emb, n_speakers, spk_emb;        /* 80, 256, >=2, 16 */
    int enc_prenet[2];                              /* 256, 128 */
    int enc_bank, enc_bank_ch, enc_proj[2], enc_proj_w, enc_hw_depth, enc_rnn;   /* 16,128,[128,128],3,4,128 */
    int att, att_state;                             /* 256, 256 */
    int dec_prenet[2], dec_layers, dec_rnn;         /* [256,128], 2, 256 */
    int post_bank, post_bank_ch, post_proj[2], post_proj_w, post_hw_depth, post_rnn;  /* 8,128,[256,80],3,4,128 */
    int num_mels, r, num_freq, max_iters;           /* 80, 5, 1025, 200 */
    int model_simple;                               /* 0: hparams.model_type 'deepvoice' (default); 1: 'simple' (tacotron.py:85-90) -- only read when n_speakers > 1 */
} twvo_taco_dims;
size_t twvo_taco_blob_floats(const twvo_taco_dims* d);
void twvo_taco_infer(const twvo_taco_dims* d, const float* blob, const int32_t* tokens, const int32_t* lengths,
                     const int32_t* speaker_ids, int N, int T, float* mel_out, float* linear_out, float* align_out);

#ifdef __cplusplus
}
#endif
#endif
