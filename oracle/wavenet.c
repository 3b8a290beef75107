/* wavenet.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference WaveNet hot path.  Each function cites the reference
 * lines it follows.  PARITY UNPINNED (see twv_oracle.h): no TensorFlow here, no reference goldens.
 *
 * Layout conventions are TensorFlow's: conv kernels are (width, in, out) row-major, activations are
 * (batch, time, channels) row-major; conv1d is cross-correlation y[t] = sum_k w[k] x[t + k*dilation].
 * Every dot product goes through twvo_cdot (arithmetic contract AC-1).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "twv_oracle.h"

typedef struct {
    size_t wf, bf, wg, bg, gcf, gcg, lcf, lcg, wd, bd, ws, bs;
} layer_offs;
typedef struct {
    size_t causal, gc_emb;
    layer_offs layer[TWVO_MAX_LAYERS];
    size_t w1, b1, w2, b2, up[4], total;
} blob_offs;

/* canonical blob order: the TF checkpoint tensors of SURVEY.md section 8a */
static void offsets(const twvo_dims* d, blob_offs* o)
{
    size_t p = 0;
    o->causal = p; p += d->scalar_input ? (size_t)d->ifw * 1 * d->R : (size_t)2 * d->Q * d->R; /* wavenet/conv1d/kernel */
    o->gc_emb = p; if (d->G) p += (size_t)d->gc_card * d->G;                                   /* wavenet/gc_embedding */
    for (int i = 0; i < d->n_layers; ++i) {
        layer_offs* l = &o->layer[i];
        l->wf = p; p += (size_t)2 * d->R * d->D;              /* conv_filter/kernel (2,R,D) */
        l->bf = p; if (d->use_bias) p += d->D;
        l->wg = p; p += (size_t)2 * d->R * d->D;              /* conv_gate/kernel */
        l->bg = p; if (d->use_bias) p += d->D;
        l->gcf = p; if (d->G) p += (size_t)d->G * d->D;       /* gc_filter/kernel (1,G,D) */
        l->gcg = p; if (d->G) p += (size_t)d->G * d->D;
        l->lcf = p; if (d->L) p += (size_t)d->L * d->D;       /* lc_filter/kernel (1,L,D) */
        l->lcg = p; if (d->L) p += (size_t)d->L * d->D;
        l->wd = p; p += (size_t)d->D * d->R;                  /* dense/kernel (1,D,R) */
        l->bd = p; if (d->use_bias) p += d->R;
        l->ws = p; p += (size_t)d->D * d->S;                  /* skip/kernel (1,D,S) */
        l->bs = p; if (d->use_bias) p += d->S;
    }
    o->w1 = p; p += (size_t)d->S * d->S;                      /* wavenet/conv1d_1/kernel (1,S,S) */
    o->b1 = p; if (d->use_bias) p += d->S;
    o->w2 = p; p += (size_t)d->S * d->O;                      /* wavenet/conv1d_2/kernel (1,S,O) */
    o->b2 = p; if (d->use_bias) p += d->O;
    for (int i = 0; i < d->n_up; ++i) { o->up[i] = p; p += (size_t)d->up[i] * 2; } /* upsample{i}/kernel (f,2,1,1) */
    o->total = p;
}

size_t twvo_blob_floats(const twvo_dims* d) { blob_offs o; offsets(d, &o); return o.total; }

int twvo_hop(const twvo_dims* d) { int h = 1; for (int i = 0; i < d->n_up; ++i) h *= d->up[i]; return h; }

/* model.py:31-39 calculate_receptive_field (filter_width = 2) */
int twvo_receptive_field(const twvo_dims* d)
{
    int sum = 0;
    for (int i = 0; i < d->n_layers; ++i) sum += d->dilations[i];
    int rf = (2 - 1) * sum + 1;
    rf += d->scalar_input ? d->ifw - 1 : 2 - 1;
    return rf;
}

/* ops.py:22-33 mu_law_encode */
void twvo_mu_law_encode(const float* audio, int n, int Q, int32_t* out)
{
    const float mu = (float)(Q - 1);
    const float log1p_mu = twvo_log1p(mu);
    for (int i = 0; i < n; ++i) {
        const float a = fabsf(audio[i]);
        const float safe = a < 1.0f ? a : 1.0f;
        const float mag = twvo_log1p(mu * safe) / log1p_mu;
        const float sgn = (audio[i] > 0.0f) - (audio[i] < 0.0f);
        const float signal = sgn * mag;
        out[i] = (int32_t)((signal + 1.0f) / 2.0f * mu + 0.5f); /* tf.to_int32 truncates */
    }
}

/* ops.py:36-47 mu_law_decode, quantization=False branch: signal already in [-1,1] */
void twvo_mu_law_expand(const float* y, int n, int Q, float* out)
{
    const int mu = Q - 1;
    for (int i = 0; i < n; ++i) {
        const float s = y[i];
        /* (1+mu)**|s| as exp(|s| * log(1+mu)) in the contract's exp/log (AC-2) */
        const float mag = (1.0f / (float)mu) * (twvo_exp(fabsf(s) * twvo_log((float)(1 + mu))) - 1.0f);
        const float sgn = (s > 0.0f) - (s < 0.0f);
        out[i] = sgn * mag;
    }
}

/* ops.py:36-47 mu_law_decode, quantization=True */
void twvo_mu_law_decode(const int32_t* q, int n, int Q, float* out)
{
    const int mu = Q - 1;
    for (int i = 0; i < n; ++i) {
        const float s = 2.0f * ((float)q[i] / (float)mu) - 1.0f;
        twvo_mu_law_expand(&s, 1, Q, &out[i]);
    }
}

/* model.py:102-111 create_upsample.  conv2d_transpose(filters=1, kernel=(f,2), strides=(f,1), 'same', no bias)
 * on (B,T,L,1): out[t*f + a, m] = K[a,0]*in[t,m] + K[a,1]*in[t,m-1]   (in[t,-1] = 0)
 * ['same' alignment of the width-2 frequency tap is [RECALLED-TF]: the forward SAME conv pads on the right,
 *  so its transpose reaches one bin to the left.] */
void twvo_upsample(const twvo_dims* d, const float* blob, const float* mel, int B, int Tm, float* out)
{
    blob_offs o; offsets(d, &o);
    const int L = d->L;
    size_t T = (size_t)Tm;
    float* cur = (float*)malloc(sizeof(float) * (size_t)B * T * L);
    memcpy(cur, mel, sizeof(float) * (size_t)B * T * L);
    for (int i = 0; i < d->n_up; ++i) {
        const int f = d->up[i];
        const float* K = blob + o.up[i]; /* (f,2) */
        float* nxt = (float*)malloc(sizeof(float) * (size_t)B * T * f * L);
        for (int b = 0; b < B; ++b)
            for (size_t t = 0; t < T; ++t)
                for (int a = 0; a < f; ++a)
                    for (int m = 0; m < L; ++m) {
                        float x[2] = { cur[((size_t)b * T + t) * L + m], m > 0 ? cur[((size_t)b * T + t) * L + m - 1] : 0.0f };
                        nxt[((size_t)b * T * f + t * f + a) * L + m] = twvo_cdot(K + (size_t)a * 2, 1, x, 2);
                    }
        free(cur); cur = nxt; T *= f;
    }
    memcpy(out, cur, sizeof(float) * (size_t)B * T * L);
    free(cur);
}

/* ------------------------------------------------------------------------------------------------ */
struct twvo_state {
    twvo_dims d;
    int B;
    int cq_rows, cq_cols;
    float* causal_q;                    /* (B, ifw, 1) | (B, 2, Q)      model.py:52-54 */
    float* lc_q;                        /* (B, 2, L)                    model.py:56 */
    float* dil_q[TWVO_MAX_LAYERS];      /* (B, d+1, R)                  model.py:58-61 */
};

twvo_state* twvo_state_new(const twvo_dims* d, int B)
{
    twvo_state* s = (twvo_state*)calloc(1, sizeof(*s));
    s->d = *d; s->B = B;
    s->cq_rows = d->scalar_input ? d->ifw : 2;
    s->cq_cols = d->scalar_input ? 1 : d->Q;
    s->causal_q = (float*)calloc((size_t)B * s->cq_rows * s->cq_cols, sizeof(float));
    s->lc_q = (float*)calloc((size_t)B * 2 * (d->L ? d->L : 1), sizeof(float));
    for (int i = 0; i < d->n_layers; ++i)
        s->dil_q[i] = (float*)calloc((size_t)B * (d->dilations[i] + 1) * d->R, sizeof(float));
    return s;
}
void twvo_state_reset(twvo_state* s)
{
    const twvo_dims* d = &s->d;
    memset(s->causal_q, 0, sizeof(float) * (size_t)s->B * s->cq_rows * s->cq_cols);
    memset(s->lc_q, 0, sizeof(float) * (size_t)s->B * 2 * (d->L ? d->L : 1));
    for (int i = 0; i < d->n_layers; ++i)
        memset(s->dil_q[i], 0, sizeof(float) * (size_t)s->B * (d->dilations[i] + 1) * d->R);
}
void twvo_state_free(twvo_state* s)
{
    if (!s) return;
    free(s->causal_q); free(s->lc_q);
    for (int i = 0; i < s->d.n_layers; ++i) free(s->dil_q[i]);
    free(s);
}

/* queue <- concat(queue[:,1:,:], new) (model.py:122,125,145: tf.scatter_update of the whole queue) */
static void push(float* q, int rows, int cols, const float* row)
{
    memmove(q, q + cols, sizeof(float) * (size_t)(rows - 1) * cols);
    memcpy(q + (size_t)(rows - 1) * cols, row, sizeof(float) * cols);
}

/* model.py:66-101 _create_dilation_layer at ONE output position.
 * x_old / x_new: the two taps (R each); lc, emb may be NULL.  z (D), transformed (R), skip (S). */
static void dilation_layer_at(const twvo_dims* d, const float* blob, const layer_offs* l,
                              const float* x_old, const float* x_new, const float* lc, const float* emb,
                              float* z, float* transformed, float* skip)
{
    const int R = d->R, D = d->D, S = d->S;
    float taps[2 * 512], f[512], g[512], t1[512], t2[512];
    memcpy(taps, x_old, sizeof(float) * R);
    memcpy(taps + R, x_new, sizeof(float) * R);
    /* model.py:68-69 conv_filter / conv_gate: k=2, dilation d, 'valid', bias; model.py:71-73 gc, model.py:75-83 lc.
     * Statement order of the reference: ((conv + bias) + gc) + lc.  AC-1b: the conv's last chunk (with R = 32: the tap that reads
     * x[t], the only operand on the sample-to-sample path) is started FROM the addend ((earlier chunks + bias) + gc) + lc. */
    float af[512], ag[512];
    const int hf = twvo_cdot_rows_head(blob + l->wf, D, taps, 2 * R, af);
    const int hg = twvo_cdot_rows_head(blob + l->wg, D, taps, 2 * R, ag);
    if (!hf) for (int j = 0; j < D; ++j) af[j] = 0.0f;
    if (!hg) for (int j = 0; j < D; ++j) ag[j] = 0.0f;
    if (d->use_bias) for (int j = 0; j < D; ++j) { af[j] = af[j] + blob[l->bf + j]; ag[j] = ag[j] + blob[l->bg + j]; }
    if (emb) { /* model.py:71-73 */
        twvo_cdot_rows(blob + l->gcf, D, emb, d->G, t1);
        twvo_cdot_rows(blob + l->gcg, D, emb, d->G, t2);
        for (int j = 0; j < D; ++j) { af[j] = af[j] + t1[j]; ag[j] = ag[j] + t2[j]; }
    }
    if (lc) { /* model.py:75-83 */
        twvo_cdot_rows(blob + l->lcf, D, lc, d->L, t1);
        twvo_cdot_rows(blob + l->lcg, D, lc, d->L, t2);
        for (int j = 0; j < D; ++j) { af[j] = af[j] + t1[j]; ag[j] = ag[j] + t2[j]; }
    }
    twvo_cdot_rows_tail(blob + l->wf, D, taps, 2 * R, af, f);
    twvo_cdot_rows_tail(blob + l->wg, D, taps, 2 * R, ag, g);
    for (int j = 0; j < D; ++j) z[j] = twvo_tanh(f[j]) * twvo_sigmoid(g[j]); /* model.py:86 */
    /* model.py:89 dense 1x1 + bias: the bias (after the earlier chunks, if D > 32) is the start value of the last chunk (AC-1b) */
    {
        float ad[512];
        if (!twvo_cdot_rows_head(blob + l->wd, R, z, D, ad)) for (int r = 0; r < R; ++r) ad[r] = 0.0f;
        if (d->use_bias) for (int r = 0; r < R; ++r) ad[r] = ad[r] + blob[l->bd + r];
        twvo_cdot_rows_tail(blob + l->wd, R, z, D, ad, transformed);
    }
    if (skip) { /* model.py:96 skip 1x1 */
        twvo_cdot_rows(blob + l->ws, S, z, D, skip);
        if (d->use_bias) for (int s = 0; s < S; ++s) skip[s] = skip[s] + blob[l->bs + s];
    }
}

/* model.py:150-165 postprocessing at one position: total (S) -> out (O) */
static void postprocess_at(const twvo_dims* d, const float* blob, const blob_offs* o, const float* total, float* out)
{
    const int S = d->S, O = d->O;
    float* h1 = (float*)malloc(sizeof(float) * S), *h2 = (float*)malloc(sizeof(float) * S);
    for (int s = 0; s < S; ++s) h1[s] = total[s] > 0.0f ? total[s] : 0.0f;        /* model.py:157 relu */
    twvo_cdot_rows(blob + o->w1, S, h1, S, h2);                                     /* model.py:158 */
    for (int s = 0; s < S; ++s) {
        float v = h2[s];
        if (d->use_bias) v = v + blob[o->b1 + s];
        h2[s] = v > 0.0f ? v : 0.0f;                                               /* model.py:160 */
    }
    twvo_cdot_rows(blob + o->w2, O, h2, S, out);                                    /* model.py:161-165 */
    if (d->use_bias) for (int c = 0; c < O; ++c) out[c] = out[c] + blob[o->b2 + c];
    free(h1); free(h2);
}

/* model.py:215-245 predict_proba_incremental up to raw_output, via _create_network's train_mode==False branch */
/* one stream of twvo_step (private scratch: streams are independent, so callers may run them on different threads) */
static void step_one(const twvo_dims* d, const float* blob, const blob_offs* op, twvo_state* s, int b,
                     const float* in_scalar, const int32_t* in_q, const float* lc, const int32_t* gc_ids,
                     float* raw_out, float* dbg_z, float* dbg_x)
{
    const blob_offs o = *op;
    const int R = d->R, D = d->D, S = d->S, L = d->L;
    float* total = (float*)malloc(sizeof(float) * S), *skip = (float*)malloc(sizeof(float) * S);
    float x[512], z[512], tr[512];
    {
        /* model.py:122 causal queue shift+append */
        float* cq = s->causal_q + (size_t)b * s->cq_rows * s->cq_cols;
        if (d->scalar_input) {
            push(cq, s->cq_rows, 1, &in_scalar[b]);
        } else { /* model.py:226 one_hot */
            float* onehot = (float*)calloc(d->Q > 0 ? d->Q : 1, sizeof(float));
            if (in_q[b] >= 0 && in_q[b] < d->Q) onehot[in_q[b]] = 1.0f;
            push(cq, 2, d->Q, onehot);
            free(onehot);
        }
        /* model.py:125-126 lc queue */
        float* lq = s->lc_q + (size_t)b * 2 * (L ? L : 1);
        if (L && lc) push(lq, 2, L, lc + (size_t)b * L);
        const float* lc_used = (L && lc) ? lq : NULL; /* model.py:79-80: slice from the FRONT -> queue[0] = previous step's frame */
        /* model.py:191-207: ids looked up in gc_embedding, or (global_condition_cardinality is None) the embedding itself, passed
         * through the same pointer as (B, G) floats */
        const float* emb = !(d->G && gc_ids) ? NULL : d->gc_card > 0 ? blob + o.gc_emb + (size_t)gc_ids[b] * d->G
                                                                      : (const float*)gc_ids + (size_t)b * d->G;
        /* model.py:131 / 41-46 causal layer: conv1d valid, no bias, over the whole queue */
        for (int j = 0; j < R; ++j)
            x[j] = twvo_cdot(blob + o.causal + j, R, cq, s->cq_rows * s->cq_cols);
        for (int i = 0; i < d->n_layers; ++i) {
            const int dil = d->dilations[i];
            float* q = s->dil_q[i] + (size_t)b * (dil + 1) * R;
            push(q, dil + 1, R, x); /* model.py:145: the queue stores the layer INPUT */
            dilation_layer_at(d, blob, &o.layer[i], q, q + (size_t)dil * R, lc_used, emb, z, tr, skip);
            for (int c = 0; c < S; ++c) total[c] = (i == 0) ? skip[c] : total[c] + skip[c]; /* model.py:154 sum(outputs) */
            for (int r = 0; r < R; ++r) x[r] = q[(size_t)dil * R + r] + tr[r]; /* model.py:98-101 */
            if (dbg_z) memcpy(dbg_z + ((size_t)b * d->n_layers + i) * D, z, sizeof(float) * D);
            if (dbg_x) memcpy(dbg_x + ((size_t)b * d->n_layers + i) * R, x, sizeof(float) * R);
        }
        postprocess_at(d, blob, &o, total, raw_out + (size_t)b * d->O);
    }
    free(total); free(skip);
}

/* model.py:215-245 predict_proba_incremental up to raw_output, via _create_network's train_mode==False branch */
void twvo_step(const twvo_dims* d, const float* blob, twvo_state* s,
               const float* in_scalar, const int32_t* in_q, const float* lc, const int32_t* gc_ids,
               float* raw_out, float* dbg_z, float* dbg_x)
{
    blob_offs o; offsets(d, &o);
    for (int b = 0; b < s->B; ++b) step_one(d, blob, &o, s, b, in_scalar, in_q, lc, gc_ids, raw_out, dbg_z, dbg_x);
}

/* threads used by twvo_generate_mol (streams are independent: one stream per thread, no barrier inside the sample loop).
 * 1 = the scalar port; the arithmetic of a stream does not depend on the thread count. */
static int g_threads = 1;
void twvo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int twvo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* mixture.py:84-114 sample_from_discretized_mix_logistic, uniforms injected */
float twvo_sample_mol(const float* y, int nr, const float* u)
{
    const float log_scale_min = (float)-32.23619130191664; /* float(np.log(1e-14)), mixture.py:84 */
    int k = 0; float best = 0.0f;
    for (int i = 0; i < nr; ++i) { /* mixture.py:103 argmax(logit - log(-log(u))) : first maximum */
        const float g = y[i] - twvo_log(-twvo_log(u[i]));
        if (i == 0 || g > best) { best = g; k = i; }
    }
    const float mean = y[nr + k];                                                   /* mixture.py:105 */
    const float ls = y[2 * nr + k] > log_scale_min ? y[2 * nr + k] : log_scale_min;  /* mixture.py:107 */
    const float uu = u[nr];
    const float t = twvo_log(uu) - twvo_log(1.0f - uu);                               /* mixture.py:111 */
    const float e = twvo_exp(ls);
    const float prod = e * t;
    float xs = mean + prod;
    xs = xs > -1.0f ? xs : -1.0f;                                                   /* mixture.py:113 */
    xs = xs < 1.0f ? xs : 1.0f;
    return xs;
}

/* generate.py:199-233 (scalar_input branch); the batch lanes of the reference's sess.run are independent streams */
void twvo_generate_mol(const twvo_dims* d, const float* blob, twvo_state* s, const float* U, const int32_t* gc_ids,
                       const float* seed, const float* u, int B, int T, float* out)
{
    const int nr = d->O / 3, nu = nr + 1, L = d->L;
    blob_offs o; offsets(d, &o);
    float* in = (float*)malloc(sizeof(float) * B), *raw = (float*)malloc(sizeof(float) * (size_t)B * d->O);
    float* lc = (float*)malloc(sizeof(float) * (size_t)B * (L ? L : 1));
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (int b = 0; b < B; ++b) {
        in[b] = seed[b];
        for (int t = 0; t < T; ++t) {
            if (L) memcpy(lc + (size_t)b * L, U + ((size_t)b * T + t) * L, sizeof(float) * L); /* generate.py:211 upsampled[:,step,:] */
            step_one(d, blob, &o, s, b, in, NULL, L ? lc : NULL, gc_ids, raw, NULL, NULL);
            const float smp = twvo_sample_mol(raw + (size_t)b * d->O, nr, u + ((size_t)b * T + t) * nu);
            out[(size_t)b * T + t] = smp;
            in[b] = smp; /* generate.py:204,233: the next window is the sample just appended */
        }
    }
    free(in); free(raw); free(lc);
}

/* np.logaddexp for float32 (numpy npy_logaddexpf) */
static float logaddexpf_np(float x, float y)
{
    if (x == y) return x + 0.693147180559945309417232121458176568f;
    const float tmp = x - y;
    if (tmp > 0) return x + twvo_log1p(twvo_exp(-tmp));
    else if (tmp <= 0) return y + twvo_log1p(twvo_exp(tmp));
    return tmp; /* NaN */
}

/* ROUND 3's form of the categorical sampler, kept for tests/test_cpu.py::test_categorical_sampler_contract_against_numpy only:
 * every reduction sequential in class order (float64 softmax sum, float32 np.logaddexp.reduce left to right, float64
 * cumsum).  Measured against numpy itself (the one piece of the reference path that runs in this image) the literal
 * left-to-right reduce pins nothing: numpy's libm/SIMD exp/log already differ from the contract's by a few 1e-6 in the
 * scaled probabilities, so any fixed reduction order with <= 1e-6 error is as faithful.  Not used by any kernel. */
int twvo_sample_categorical_sequential(const float* logits, int Q, double temperature, double u, float* proba_out)
{
    double* e = (double*)malloc(sizeof(double) * Q);
    float* lp = (float*)malloc(sizeof(float) * Q);
    double m = (double)logits[0];
    for (int i = 1; i < Q; ++i) if ((double)logits[i] > m) m = (double)logits[i];
    double sum = 0.0;
    for (int i = 0; i < Q; ++i) { e[i] = twvo_exp64((double)logits[i] - m); sum += e[i]; }
    const float temp32 = (float)temperature;
    for (int i = 0; i < Q; ++i) {
        const float p32 = (float)(e[i] / sum);
        lp[i] = twvo_log(p32) / temp32;
    }
    float lse = lp[0];
    for (int i = 1; i < Q; ++i) lse = logaddexpf_np(lse, lp[i]);
    double c = 0.0;
    for (int i = 0; i < Q; ++i) {
        const float sp = twvo_exp(lp[i] - lse);
        if (proba_out) proba_out[i] = sp;
        c += (double)sp;
        e[i] = c;
    }
    const double last = e[Q - 1];
    int idx = Q - 1;
    for (int i = 0; i < Q; ++i) if (e[i] / last > u) { idx = i; break; }
    free(e); free(lp);
    return idx;
}

/* AC-5: the ONE reduction primitive of the categorical sampler -- an inclusive prefix sum of 64
 * doubles in the order a 64-lane wave produces with four row_shr steps inside each row of 16 lanes, then the two
 * row-broadcast steps (lane 15 -> next row, lane 31 -> upper half).  v[63] is the total.  (Addition is commutative, so
 * only the TREE matters, not which operand a lane holds.) */
void twvo_scan64(double v[64])
{
    double t[64];
    for (int off = 1; off <= 8; off <<= 1) {
        memcpy(t, v, sizeof(t));
        for (int l = 0; l < 64; ++l) if ((l & 15) >= off) v[l] = t[l] + t[l - off];
    }
    const double r0 = v[15], r2 = v[47];
    for (int l = 16; l < 32; ++l) v[l] = v[l] + r0;
    for (int l = 48; l < 64; ++l) v[l] = v[l] + r2;
    const double h = v[31];
    for (int l = 32; l < 64; ++l) v[l] = v[l] + h;
}

/* model.py:243 float64 softmax -> float32; generate.py:219-222 temperature rescale; generate.py:231 np.random.choice
 * (legacy RandomState.choice: cdf = p.cumsum() in float64, cdf /= cdf[-1], searchsorted(u, 'right')).
 *
 * AC-5: class i sits in lane (i mod 64), block (i / 64); classes >= Q count as probability zero.  Every sum over the
 * classes is "per lane over the blocks in block order, then twvo_scan64 over the lanes":
 *   e_i   = exp64(y_i - max y)                                   float64   (model.py:243)
 *   p_i   = (float)(e_i / SUM e)                                 float64 division, one rounding to float32
 *   lp_i  = log(p_i) / (float)temperature                        float32   (generate.py:220)
 *   lse   = M + log((float) SUM (double)exp(lp_i - M)),  M = max lp   (generate.py:221 np.logaddexp.reduce: the same
 *           quantity, log sum exp; numpy's left-to-right pairwise form is replaced, see above)
 *   sp_i  = exp(lp_i - lse)                                      float32   (generate.py:221-222)
 *   cdf_i = prefix sum of (double)sp in class order: block base (sum of the earlier blocks' totals, in block order)
 *           + scan64 inside the block;  drawn class = first i with cdf_i / cdf_last > u                                */
int twvo_sample_categorical(const float* logits, int Q, double temperature, double u, float* proba_out)
{
    const int NB = (Q + 63) / 64;
    double* e = (double*)calloc((size_t)NB * 64, sizeof(double));
    float* lp = (float*)calloc((size_t)NB * 64, sizeof(float));
    double lane[64];
    float m = logits[0];
    for (int i = 1; i < Q; ++i) if (logits[i] > m) m = logits[i];
    for (int i = 0; i < Q; ++i) e[i] = twvo_exp64((double)logits[i] - (double)m);
    for (int l = 0; l < 64; ++l) { lane[l] = e[l]; for (int k = 1; k < NB; ++k) lane[l] = lane[l] + e[l + 64 * k]; }
    twvo_scan64(lane);
    const double sum = lane[63];
    const float temp32 = (float)temperature;
    float m2 = 0.0f;
    for (int i = 0; i < Q; ++i) {
        const float p32 = (float)(e[i] / sum);           /* tf.cast(softmax(float64), float32) */
        lp[i] = twvo_log(p32) / temp32;                  /* generate.py:220 np.log(prediction) / temperature (float32) */
        if (i == 0 || lp[i] > m2) m2 = lp[i];
    }
    for (int i = 0; i < NB * 64; ++i) e[i] = i < Q ? (double)twvo_exp(lp[i] - m2) : 0.0;
    for (int l = 0; l < 64; ++l) { lane[l] = e[l]; for (int k = 1; k < NB; ++k) lane[l] = lane[l] + e[l + 64 * k]; }
    twvo_scan64(lane);
    const float lse = m2 + twvo_log((float)lane[63]);    /* generate.py:221 */
    int idx = Q - 1, found = 0;
    double base = 0.0, last = 0.0;
    for (int k = 0; k < NB; ++k) {
        for (int l = 0; l < 64; ++l) {
            const int i = l + 64 * k;
            const float sp = i < Q ? twvo_exp(lp[i] - lse) : 0.0f;   /* generate.py:221-222 */
            if (proba_out && i < Q) proba_out[i] = sp;
            lane[l] = (double)sp;
        }
        twvo_scan64(lane);
        for (int l = 0; l < 64; ++l) e[l + 64 * k] = k == 0 ? lane[l] : base + lane[l];
        base = e[63 + 64 * k];
        last = base;
    }
    for (int i = 0; i < Q && !found; ++i) if (e[i] / last > u) { idx = i; found = 1; } /* cdf /= cdf[-1]; searchsorted(u, 'right') */
    free(e); free(lp);
    return idx;
}

/* generate.py:199-233 (one-hot branch) */
void twvo_generate_mulaw(const twvo_dims* d, const float* blob, twvo_state* s, const float* U, const int32_t* gc_ids,
                         const int32_t* seed, const double* u, double temperature, int B, int T, int32_t* out)
{
    const int L = d->L;
    blob_offs o; offsets(d, &o);
    int32_t* in = (int32_t*)malloc(sizeof(int32_t) * B);
    float* raw = (float*)malloc(sizeof(float) * (size_t)B * d->O);
    float* lc = (float*)malloc(sizeof(float) * (size_t)B * (L ? L : 1));
    /* the batch lanes are independent streams: one stream per thread (twvo_set_threads), same arithmetic per stream */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (int b = 0; b < B; ++b) {
        in[b] = seed[b];
        for (int t = 0; t < T; ++t) {
            if (L) memcpy(lc + (size_t)b * L, U + ((size_t)b * T + t) * L, sizeof(float) * L);
            step_one(d, blob, &o, s, b, NULL, in, L ? lc : NULL, gc_ids, raw, NULL, NULL);
            const int q = twvo_sample_categorical(raw + (size_t)b * d->O, d->O, temperature, u[(size_t)b * T + t], NULL);
            out[(size_t)b * T + t] = q;
            in[b] = q;
        }
    }
    free(in); free(raw); free(lc);
}

/* model.py:112-167 _create_network with train_mode=True (full 'valid' convolutions) */
void twvo_forward_full(const twvo_dims* d, const float* blob, int B, int Tin, const float* in_scalar, const int32_t* in_q,
                       const float* lc_up, int Tlc, const int32_t* gc_ids, float* raw_out)
{
    blob_offs o; offsets(d, &o);
    const int R = d->R, D = d->D, S = d->S, L = d->L;
    const int rf = twvo_receptive_field(d);
    const int out_w = Tin - rf + 1; /* model.py:135 */
    const int kc = d->scalar_input ? d->ifw : 2;
    const int cin = d->scalar_input ? 1 : d->Q;
    float* total = (float*)malloc(sizeof(float) * (size_t)out_w * S);
    float* skip = (float*)malloc(sizeof(float) * S);
    float z[512], tr[512];
    for (int b = 0; b < B; ++b) {
        /* input as (Tin, cin) */
        float* inp = (float*)calloc((size_t)Tin * cin, sizeof(float));
        for (int t = 0; t < Tin; ++t) {
            if (d->scalar_input) inp[t] = in_scalar[(size_t)b * Tin + t];
            else inp[(size_t)t * cin + in_q[(size_t)b * Tin + t]] = 1.0f;
        }
        int Tc = Tin - kc + 1;
        float* cur = (float*)malloc(sizeof(float) * (size_t)Tc * R);
        for (int p = 0; p < Tc; ++p)
            for (int j = 0; j < R; ++j)
                cur[(size_t)p * R + j] = twvo_cdot(blob + o.causal + j, R, inp + (size_t)p * cin, kc * cin); /* model.py:131 */
        const float* emb = !(d->G && gc_ids) ? NULL : d->gc_card > 0 ? blob + o.gc_emb + (size_t)gc_ids[b] * d->G : (const float*)gc_ids + (size_t)b * d->G;
        for (int i = 0; i < d->n_layers; ++i) {
            const int dil = d->dilations[i];
            const int Tn = Tc - dil;
            float* nxt = (float*)malloc(sizeof(float) * (size_t)Tn * R);
            const int skip_cut = Tn - out_w; /* model.py:94 */
            for (int p = 0; p < Tn; ++p) {
                const float* lc = (L && lc_up && p < Tlc) ? lc_up + ((size_t)b * Tlc + p) * L : NULL; /* model.py:79-80 front slice */
                dilation_layer_at(d, blob, &o.layer[i], cur + (size_t)p * R, cur + (size_t)(p + dil) * R, lc, emb,
                                  z, tr, p >= skip_cut ? skip : NULL);
                if (p >= skip_cut) {
                    float* tot = total + (size_t)(p - skip_cut) * S;
                    for (int c = 0; c < S; ++c) tot[c] = (i == 0) ? skip[c] : tot[c] + skip[c];
                }
                for (int r = 0; r < R; ++r) nxt[(size_t)p * R + r] = cur[(size_t)(p + dil) * R + r] + tr[r]; /* model.py:98-101 */
            }
            free(cur); cur = nxt; Tc = Tn;
        }
        for (int p = 0; p < out_w; ++p)
            postprocess_at(d, blob, &o, total + (size_t)p * S, raw_out + ((size_t)b * out_w + p) * d->O);
        free(cur); free(inp);
    }
    free(total); free(skip);
    (void)D;
}
