/* tacotron.c -- CPU ORACLE (test infrastructure, NOT product code): Tacotron text -> mel inference.
 *
 * Restates /root/reference/tacotron/tacotron.py:36-235 (initialize, inference branch: rnn_decoder_test_mode=True,
 * linear_targets=None), tacotron/modules.py:10-96, tacotron/rnn_wrappers.py:282-467, tacotron/helpers.py:10-41 for the
 * default hparams path: model_type 'deepvoice' with num_speakers > 1 and speaker_embedding_size != 1
 * (hparams.py:123-124), or a single speaker (tacotron.py:97-104; synthesizer.py:375's default), attention_type
 * 'bah_mon_norm' (hparams.py:140).
 *
 * PARITY UNPINNED, twice over: (1) no TensorFlow here and no reference goldens (see twv_oracle.h); (2) the pieces that
 * live inside tf.contrib / tf.layers -- GRUCell gate order and update rule, bidirectional_dynamic_rnn's handling of
 * sequence_length, 'same' padding, batch_normalization's inference formula, BahdanauMonotonicAttention(normalize=True,
 * mode='parallel') with _maybe_mask_score / safe_cumprod, OutputProjectionWrapper / ResidualWrapper -- are restated from
 * memory of TensorFlow 1.x ([RECALLED-TF]) and marked below.  Arithmetic follows the contract of DESIGN.md.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "twv_oracle.h"

/* every dense / conv on this path: out[rows][N] = act(cdot(in_row, W[K][N]) + b) */
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3, ACT_SOFTSIGN = 4 };
static float act_apply(float v, int act)
{
    switch (act) {
        case ACT_RELU: return v > 0.0f ? v : 0.0f;
        case ACT_TANH: return twvo_tanh(v);
        case ACT_SIGMOID: return twvo_sigmoid(v);
        case ACT_SOFTSIGN: return v / (fabsf(v) + 1.0f);     /* tf.nn.softsign, tacotron.py:76 */
        default: return v;
    }
}
static void dense_rows(const float* in, int rows, int K, const float* W, const float* b, int N, int act, float* out)
{
    for (int r = 0; r < rows; ++r) {
        twvo_cdot_rows(W, N, in + (size_t)r * K, K, out + (size_t)r * N);
        for (int n = 0; n < N; ++n) {
            float v = out[(size_t)r * N + n];
            if (b) v = v + b[n];
            out[(size_t)r * N + n] = act_apply(v, act);
        }
    }
}

/* modules.py:92-96 conv1d: tf.layers.conv1d(k, 'same', activation) THEN batch_normalization (inference).
 * 'same', stride 1: pad_left = (k-1)/2, pad_right = k-1-pad_left [RECALLED-TF]; cross-correlation over (tap, channel).
 * BN inference [RECALLED-TF tf.nn.batch_normalization]: inv = rsqrt(var + eps) * gamma; y = x*inv + (beta - mean*inv);
 * inv / shift are precomputed per channel by the caller (tests do it in float32 numpy). */
static void conv1d_bn(const float* in, int T, int Cin, const float* W, const float* b, int k, int Cout, int act,
                      const float* bn_inv, const float* bn_shift, float* out)
{
    const int pl = (k - 1) / 2;
    float* win = (float*)malloc(sizeof(float) * (size_t)k * Cin);
    for (int t = 0; t < T; ++t) {
        for (int i = 0; i < k; ++i) {
            const int ts = t + i - pl;
            if (ts >= 0 && ts < T) memcpy(win + (size_t)i * Cin, in + (size_t)ts * Cin, sizeof(float) * Cin);
            else memset(win + (size_t)i * Cin, 0, sizeof(float) * Cin);
        }
        float* o = out + (size_t)t * Cout;
        twvo_cdot_rows(W, Cout, win, k * Cin, o);
        for (int c = 0; c < Cout; ++c) {
            float v = o[c] + b[c];
            v = act_apply(v, act);
            const float y = v * bn_inv[c];
            o[c] = y + bn_shift[c];
        }
    }
    free(win);
}

/* tf.contrib.rnn.GRUCell [RECALLED-TF rnn_cell_impl.GRUCell.call]:
 *   value = sigmoid([x, h] . Wg + bg);  r, u = split(value, 2);  c = tanh([x, r*h] . Wc + bc);  h' = u*h + (1-u)*c */
static void gru_step(const float* x, int nin, const float* h, int nu, const float* Wg, const float* bg, const float* Wc,
                     const float* bc, float* hnew)
{
    float cat[1024], g[1024], c[512];
    memcpy(cat, x, sizeof(float) * nin);
    memcpy(cat + nin, h, sizeof(float) * nu);
    twvo_cdot_rows(Wg, 2 * nu, cat, nin + nu, g);
    for (int j = 0; j < 2 * nu; ++j) g[j] = twvo_sigmoid(g[j] + bg[j]);
    for (int j = 0; j < nu; ++j) cat[nin + j] = g[j] * h[j];
    twvo_cdot_rows(Wc, nu, cat, nin + nu, c);
    for (int j = 0; j < nu; ++j) {
        const float cj = twvo_tanh(c[j] + bc[j]);
        const float u = g[nu + j];
        const float t1 = u * h[j];
        const float t2 = 1.0f - u;
        const float t3 = t2 * cj;
        hnew[j] = t1 + t3;
    }
}



/* AC-6: inclusive prefix sum of 64 floats in the order a 64-lane wave produces with four row_shr steps inside each row of 16
 * lanes, then lane 15 -> the next row, then lane 31 -> the upper half (the float32 twin of twvo_scan64, wavenet.c) */
static void scan64f(float v[64])
{
    float t[64];
    for (int off = 1; off <= 8; off <<= 1) {
        memcpy(t, v, sizeof(t));
        for (int l = 0; l < 64; ++l) if ((l & 15) >= off) v[l] = t[l] + t[l - off];
    }
    const float r0 = v[15], r2 = v[47];
    for (int l = 16; l < 32; ++l) v[l] = v[l] + r0;
    for (int l = 48; l < 64; ++l) v[l] = v[l] + r2;
    const float h = v[31];
    for (int l = 32; l < 64; ++l) v[l] = v[l] + h;
}
/* cumulative sum of x[0..T) in blocks of 64 (zero padded): block values = scan64f, plus the total of the earlier blocks
 * (carry + value; the first block has no carry).  inclusive = 0: out[t] = the inclusive value of t-1 (0 for t = 0). */
static void scan_blocks(const float* x, int T, int inclusive, float* out)
{
    float carry = 0.0f;
    for (int base = 0; base < T; base += 64) {
        float v[64];
        for (int l = 0; l < 64; ++l) v[l] = base + l < T ? x[base + l] : 0.0f;
        scan64f(v);
        for (int l = 0; l < 64; ++l) if (base > 0) v[l] = carry + v[l];
        for (int l = 0; l < 64 && base + l < T; ++l)
            out[base + l] = inclusive ? v[l] : (l == 0 ? (base == 0 ? 0.0f : carry) : v[l - 1]);
        carry = v[63];
    }
}

/* ---- canonical blob: tensors in the order of tacotron_specs() in oracle.py / the product's weights module ---- */
typedef struct { const float* p; } cur_t;
static const float* take(cur_t* c, size_t n) { const float* r = c->p; c->p += n; return r; }

typedef struct {
    const float *W[20], *b[20], *inv[20], *shift[20];   /* conv bank */
    const float *pW[2], *pb[2], *pinv[2], *pshift[2];   /* projections */
    const float *dW, *db;                               /* optional dense before the highways */
    const float *hH[8], *hHb[8], *hT[8], *hTb[8];
    const float *gWg[2], *gbg[2], *gWc[2], *gbc[2];     /* fw, bw GRU */
} cbhg_w;

static void take_cbhg(cur_t* c, cbhg_w* w, int Cin, int bank, int bank_ch, const int* proj, int proj_w, int depth, int rnn, int need_dense)
{
    for (int k = 1; k <= bank; ++k) {
        w->W[k] = take(c, (size_t)k * Cin * bank_ch); w->b[k] = take(c, bank_ch);
        w->inv[k] = take(c, bank_ch); w->shift[k] = take(c, bank_ch);
    }
    int cin = bank * bank_ch;
    for (int i = 0; i < 2; ++i) {
        w->pW[i] = take(c, (size_t)proj_w * cin * proj[i]); w->pb[i] = take(c, proj[i]);
        w->pinv[i] = take(c, proj[i]); w->pshift[i] = take(c, proj[i]);
        cin = proj[i];
    }
    if (need_dense) { w->dW = take(c, (size_t)proj[1] * rnn); w->db = take(c, rnn); } else { w->dW = w->db = NULL; }
    for (int i = 0; i < depth; ++i) {
        w->hH[i] = take(c, (size_t)rnn * rnn); w->hHb[i] = take(c, rnn);
        w->hT[i] = take(c, (size_t)rnn * rnn); w->hTb[i] = take(c, rnn);
    }
    for (int d = 0; d < 2; ++d) {
        w->gWg[d] = take(c, (size_t)2 * rnn * 2 * rnn); w->gbg[d] = take(c, 2 * rnn);
        w->gWc[d] = take(c, (size_t)2 * rnn * rnn); w->gbc[d] = take(c, rnn);
    }
}

/* modules.py:25-74 cbhg for ONE sequence: in (T, Cin) -> out (T, 2*rnn).  len <= T: sequence_length of the biGRU
 * (len = T when input_lengths is None, as for the post net).  before_hw: (rnn) or NULL; init_fw/bw: (rnn) or NULL. */
static void cbhg_one(const cbhg_w* w, const float* in, int T, int len, int Cin, int bank, int bank_ch, const int* proj, int proj_w,
                     int depth, int rnn, const float* before_hw, const float* init_fw, const float* init_bw, float* out)
{
    const int CB = bank * bank_ch;
    float* bankout = (float*)malloc(sizeof(float) * (size_t)T * CB);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)T * (CB > 1024 ? CB : 1024));
    for (int k = 1; k <= bank; ++k) {   /* modules.py:30-35 conv bank, concatenated on the channel axis */
        conv1d_bn(in, T, Cin, w->W[k], w->b[k], k, bank_ch, ACT_RELU, w->inv[k], w->shift[k], tmp);
        for (int t = 0; t < T; ++t) memcpy(bankout + (size_t)t * CB + (size_t)(k - 1) * bank_ch, tmp + (size_t)t * bank_ch, sizeof(float) * bank_ch);
    }
    /* modules.py:38 max_pooling1d(pool 2, stride 1, 'same'): out[t] = max(x[t], x[t+1]), the pad position ignored [RECALLED-TF] */
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < CB; ++c) {
            const float a = bankout[(size_t)t * CB + c];
            const float b = t + 1 < T ? bankout[(size_t)(t + 1) * CB + c] : a;
            tmp[(size_t)t * CB + c] = a > b ? a : b;
        }
    /* modules.py:41-44 two projections (relu on the first only), each followed by batch-norm */
    float* p1 = (float*)malloc(sizeof(float) * (size_t)T * proj[0]);
    float* p2 = (float*)malloc(sizeof(float) * (size_t)T * proj[1]);
    conv1d_bn(tmp, T, CB, w->pW[0], w->pb[0], proj_w, proj[0], ACT_RELU, w->pinv[0], w->pshift[0], p1);
    conv1d_bn(p1, T, proj[0], w->pW[1], w->pb[1], proj_w, proj[1], ACT_NONE, w->pinv[1], w->pshift[1], p2);
    /* modules.py:47-53 residual: proj_out + inputs (+ before_highway) */
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < proj[1]; ++c) {
            float v = p2[(size_t)t * proj[1] + c] + in[(size_t)t * Cin + c];
            if (before_hw) v = v + before_hw[c];
            p2[(size_t)t * proj[1] + c] = v;
        }
    float* hw = (float*)malloc(sizeof(float) * (size_t)T * rnn);
    float* hh = (float*)malloc(sizeof(float) * (size_t)T * rnn);
    float* ht = (float*)malloc(sizeof(float) * (size_t)T * rnn);
    if (w->dW) dense_rows(p2, T, proj[1], w->dW, w->db, rnn, ACT_NONE, hw);   /* modules.py:56-57 */
    else memcpy(hw, p2, sizeof(float) * (size_t)T * rnn);
    for (int i = 0; i < depth; ++i) {   /* modules.py:83-89 highwaynet: H*T + x*(1-T) */
        dense_rows(hw, T, rnn, w->hH[i], w->hHb[i], rnn, ACT_RELU, hh);
        dense_rows(hw, T, rnn, w->hT[i], w->hTb[i], rnn, ACT_SIGMOID, ht);
        for (size_t e = 0; e < (size_t)T * rnn; ++e) {
            const float a = hh[e] * ht[e];
            const float b = 1.0f - ht[e];
            const float c = hw[e] * b;
            hw[e] = a + c;
        }
    }
    /* modules.py:66-74 bidirectional_dynamic_rnn(GRUCell, GRUCell, sequence_length): outputs past `len` are zero, the
     * backward direction runs over the first `len` steps reversed [RECALLED-TF] */
    float h[512], hn[512];
    memset(out, 0, sizeof(float) * (size_t)T * 2 * rnn);
    if (init_fw) memcpy(h, init_fw, sizeof(float) * rnn); else memset(h, 0, sizeof(float) * rnn);
    for (int t = 0; t < len; ++t) {
        gru_step(hw + (size_t)t * rnn, rnn, h, rnn, w->gWg[0], w->gbg[0], w->gWc[0], w->gbc[0], hn);
        memcpy(h, hn, sizeof(float) * rnn);
        memcpy(out + (size_t)t * 2 * rnn, h, sizeof(float) * rnn);
    }
    if (init_bw) memcpy(h, init_bw, sizeof(float) * rnn); else memset(h, 0, sizeof(float) * rnn);
    for (int t = len - 1; t >= 0; --t) {
        gru_step(hw + (size_t)t * rnn, rnn, h, rnn, w->gWg[1], w->gbg[1], w->gWc[1], w->gbc[1], hn);
        memcpy(h, hn, sizeof(float) * rnn);
        memcpy(out + (size_t)t * 2 * rnn + rnn, h, sizeof(float) * rnn);
    }
    free(bankout); free(tmp); free(p1); free(p2); free(hw); free(hh); free(ht);
}

/* tacotron.py:36-235, inference.  tokens (N,T) int32 (0 = pad, 1 = EOS), lengths (N), speaker_ids (N).
 * mel_out (N, max_iters*r, num_mels), linear_out (N, max_iters*r, num_freq) or NULL, align_out (N, T, max_iters) or NULL. */
void twvo_taco_infer(const twvo_taco_dims* d, const float* blob, const int32_t* tokens, const int32_t* lengths,
                     const int32_t* speaker_ids, int N, int T, float* mel_out, float* linear_out, float* align_out)
{
    cur_t c = { blob };
    const int E = d->emb, SE = d->spk_emb, P0 = d->enc_prenet[0], P1 = d->enc_prenet[1], RN = d->enc_rnn, A = d->att,
              AS = d->att_state, DR = d->dec_rnn, M = d->num_mels, R = d->r, ENC = 2 * RN;
    const float* emb = take(&c, (size_t)d->n_symbols * E);                 /* tacotron.py:51 'embedding' */
    /* tacotron.py:62-104: the speaker tensors exist only when num_speakers > 1; otherwise before_highway and every initial
     * state is None (zero_state) */
    const int multi = d->n_speakers > 1;
    /* tacotron.py:69-75: with speaker_embedding_size == 1 the five speaker-dependent vectors are embedding tables of their own
     * (modules.py:10-12 get_embed), looked up by speaker id; otherwise (:76-82) dense(softsign) layers of ONE speaker embedding */
    /* tacotron.py:85-90 model_type 'simple': no speaker-dependent initial states at all; the speaker embedding itself is concatenated to
     * the decoder prenet's output (rnn_wrappers.py:425-432) and to [output, attention] in front of the first projection (:455-463) */
    const int simple = multi && d->model_simple && SE != 1;
    const int tables = multi && SE == 1;
    const float* semb = (multi && !tables) ? take(&c, (size_t)d->n_speakers * SE) : NULL;   /* tacotron.py:67 'speaker_embedding' */
    /* tacotron.py:76-82 deep_dense (softsign): before_highway, encoder rnn init, attention rnn init, decoder rnn inits */
    const float *dW[8], *db[8], *tab[8];
    const int dn[8] = { P1, 2 * RN, AS, DR, DR, DR, DR, DR };
    const int ndense = (multi && !simple) ? 3 + d->dec_layers : 0;
    const int SEc = simple ? SE : 0;                                       /* width of the embedding concatenated inside the decoder */
    for (int i = 0; i < ndense; ++i) {
        if (tables) tab[i] = take(&c, (size_t)d->n_speakers * dn[i]);
        else { dW[i] = take(&c, (size_t)SE * dn[i]); db[i] = take(&c, dn[i]); }
    }
    const float* pW1 = take(&c, (size_t)E * P0); const float* pb1 = take(&c, P0);      /* prenet, modules.py:15-23 */
    const float* pW2 = take(&c, (size_t)P0 * P1); const float* pb2 = take(&c, P1);
    cbhg_w enc;
    take_cbhg(&c, &enc, P1, d->enc_bank, d->enc_bank_ch, d->enc_proj, d->enc_proj_w, d->enc_hw_depth, RN, d->enc_proj[1] != RN);
    /* attention (tacotron.py:130): memory_layer, query_layer, attention_v, attention_g, attention_b, attention_score_bias */
    const float* Wm = take(&c, (size_t)ENC * A);
    const float* Wq = take(&c, (size_t)AS * A);
    const float* av = take(&c, A); const float* ag = take(&c, 1); const float* ab = take(&c, A); const float* asb = take(&c, 1);
    const float* dpW1 = take(&c, (size_t)M * d->dec_prenet[0]); const float* dpb1 = take(&c, d->dec_prenet[0]);   /* decoder_prenet */
    const float* dpW2 = take(&c, (size_t)d->dec_prenet[0] * d->dec_prenet[1]); const float* dpb2 = take(&c, d->dec_prenet[1]);
    const int DP = d->dec_prenet[1];
    const int ain = DP + SEc + ENC;                                        /* attention GRU input: [prenet_out, (speaker embed,) context] */
    const float* aWg = take(&c, (size_t)(ain + AS) * 2 * AS); const float* abg = take(&c, 2 * AS);
    const float* aWc = take(&c, (size_t)(ain + AS) * AS); const float* abc = take(&c, AS);
    const float* cW = take(&c, (size_t)(AS + ENC + SEc) * DR); const float* cb = take(&c, DR);    /* OutputProjectionWrapper -> dec_rnn */
    const float *rWg[4], *rbg[4], *rWc[4], *rbc[4];
    for (int i = 0; i < d->dec_layers; ++i) {
        rWg[i] = take(&c, (size_t)2 * DR * 2 * DR); rbg[i] = take(&c, 2 * DR);
        rWc[i] = take(&c, (size_t)2 * DR * DR); rbc[i] = take(&c, DR);
    }
    const float* oW = take(&c, (size_t)DR * M * R); const float* ob = take(&c, M * R);              /* OutputProjectionWrapper -> M*r */
    cbhg_w post;
    take_cbhg(&c, &post, M, d->post_bank, d->post_bank_ch, d->post_proj, d->post_proj_w, d->post_hw_depth, d->post_rnn,
              d->post_proj[1] != d->post_rnn);
    const float* lW = take(&c, (size_t)2 * d->post_rnn * d->num_freq); const float* lb = take(&c, d->num_freq);   /* tacotron.py:219 */

    /* normed_v = g * v * rsqrt(sum(v^2)) [RECALLED-TF _bahdanau_score, normalize=True] */
    float nv[1024];
    {
        const float s = twvo_cdot(av, 1, av, A);
        const float rs = 1.0f / sqrtf(s);
        for (int j = 0; j < A; ++j) { const float gv = ag[0] * av[j]; nv[j] = gv * rs; }
    }
    const int iters = d->max_iters, TO = iters * R;
    float* x0 = (float*)malloc(sizeof(float) * (size_t)T * E);
    float* x1 = (float*)malloc(sizeof(float) * (size_t)T * P0);
    float* x2 = (float*)malloc(sizeof(float) * (size_t)T * P1);
    float* memo = (float*)malloc(sizeof(float) * (size_t)T * ENC);
    float* keys = (float*)malloc(sizeof(float) * (size_t)T * A);
    float* post_out = (float*)malloc(sizeof(float) * (size_t)TO * 2 * d->post_rnn);
    float* score_t = (float*)malloc(sizeof(float) * T * 6);
    for (int n = 0; n < N; ++n) {
        const int len = lengths[n];
        /* tacotron.py:51-60 embedding with row 0 forced to zeros */
        for (int t = 0; t < T; ++t) {
            const int tok = tokens[(size_t)n * T + t];
            if (tok == 0) memset(x0 + (size_t)t * E, 0, sizeof(float) * E);
            else memcpy(x0 + (size_t)t * E, emb + (size_t)tok * E, sizeof(float) * E);
        }
        float init[8][512];
        memset(init, 0, sizeof(init));                                      /* single speaker: zero states (tacotron.py:97-104) */
        if (tables) {
            for (int i = 0; i < ndense; ++i) memcpy(init[i], tab[i] + (size_t)speaker_ids[n] * dn[i], sizeof(float) * dn[i]);   /* embedding_lookup */
        } else if (multi && !simple) {
            const float* se = semb + (size_t)speaker_ids[n] * SE;
            for (int i = 0; i < ndense; ++i) dense_rows(se, 1, SE, dW[i], db[i], dn[i], ACT_SOFTSIGN, init[i]);
        }
        const int has_init = multi && !simple;                                /* 'simple' and single speaker: None -> zero states, no before_highway */
        const float* sev = simple ? semb + (size_t)speaker_ids[n] * SE : NULL;
        dense_rows(x0, T, E, pW1, pb1, P0, ACT_RELU, x1);                 /* tacotron.py:108 prenet (dropout rate 0) */
        dense_rows(x1, T, P0, pW2, pb2, P1, ACT_RELU, x2);
        /* tacotron.py:113 encoder cbhg; encoder_rnn_init_state split into fw | bw (modules.py:66) */
        cbhg_one(&enc, x2, T, len, P1, d->enc_bank, d->enc_bank_ch, d->enc_proj, d->enc_proj_w, d->enc_hw_depth, RN,
                 has_init ? init[0] : NULL, has_init ? init[1] : NULL, has_init ? init[1] + RN : NULL, memo);
        /* [RECALLED-TF _prepare_memory]: memory zeroed past input_lengths (the biGRU already outputs zeros there) */
        for (int t = len; t < T; ++t) memset(memo + (size_t)t * ENC, 0, sizeof(float) * ENC);
        dense_rows(memo, T, ENC, Wm, NULL, A, ACT_NONE, keys);              /* keys = memory_layer(memory) */

        float* align = score_t;            /* previous alignments: one-hot at 0 [RECALLED-TF initial_alignments] */
        float* p = score_t + T, *cp = score_t + 2 * T, *lg = score_t + 3 * T, *nal = score_t + 4 * T, *sc = score_t + 5 * T;
        for (int t = 0; t < T; ++t) align[t] = t == 0 ? 1.0f : 0.0f;
        float ctx[1024], ah[512], rh[4][512], frame[256];
        memset(ctx, 0, sizeof(float) * ENC);                                /* AttentionWrapper zero_state: attention = 0 */
        memcpy(ah, init[2], sizeof(float) * AS);                            /* initial_cell_state = attention_rnn_init_state */
        for (int i = 0; i < d->dec_layers; ++i) memcpy(rh[i], init[3 + i], sizeof(float) * DR);   /* tacotron.py:184-195 */
        memset(frame, 0, sizeof(float) * M);                                /* helpers.py:90-92 go frame */
        for (int it = 0; it < iters; ++it) {
            float q1[512], q2[512], cin[2048], hn[512], pq[1024], outp[2048];
            dense_rows(frame, 1, M, dpW1, dpb1, d->dec_prenet[0], ACT_RELU, q1);       /* rnn_wrappers.py:425 decoder prenet */
            dense_rows(q1, 1, d->dec_prenet[0], dpW2, dpb2, DP, ACT_RELU, q2);
            memcpy(cin, q2, sizeof(float) * DP);
            if (SEc) memcpy(cin + DP, sev, sizeof(float) * SEc);            /* rnn_wrappers.py:429-430 concat([prenet_out, embed_to_concat]) */
            memcpy(cin + DP + SEc, ctx, sizeof(float) * ENC);               /* rnn_wrappers.py:310 concat([inputs, state.attention]) */
            gru_step(cin, ain, ah, AS, aWg, abg, aWc, abc, hn);             /* rnn_wrappers.py:312 attention GRU */
            memcpy(ah, hn, sizeof(float) * AS);
            /* rnn_wrappers.py:369-398 + [RECALLED-TF BahdanauMonotonicAttention.__call__] */
            dense_rows(ah, 1, AS, Wq, NULL, A, ACT_NONE, pq);
            for (int t = 0; t < T; ++t) {
                float th[1024];
                for (int j = 0; j < A; ++j) { const float s1 = keys[(size_t)t * A + j] + pq[j]; th[j] = twvo_tanh(s1 + ab[j]); }
                sc[t] = twvo_cdot(nv, 1, th, A) + asb[0];
                p[t] = t < len ? twvo_sigmoid(sc[t]) : 0.0f;                /* _maybe_mask_score(-inf) -> sigmoid = 0 */
            }
            /* monotonic_attention, mode 'parallel': p * cumprod_excl(1-p) * cumsum(prev / clip(cumprod, 1e-10, 1));
             * safe_cumprod = exp(cumsum_excl(log(clip(1-p, tiny, 1)))).  TensorFlow leaves the order of the two cumulative sums
             * open; the contract (AC-6) runs them in blocks of 64 time steps -- scan64 tree in float32
             * inside a block (zero padded), the earlier blocks' total added in front -- see scan_blocks below. */
            {
                for (int t = 0; t < T; ++t) {
                    float om = 1.0f - p[t];
                    const float tiny = 1.17549435e-38f;
                    om = om < tiny ? tiny : (om > 1.0f ? 1.0f : om);
                    lg[t] = twvo_log(om);
                }
                scan_blocks(lg, T, 0, cp);                                   /* exclusive cumsum of the logs */
                for (int t = 0; t < T; ++t) {
                    cp[t] = twvo_exp(cp[t]);
                    float den = cp[t];
                    den = den < 1e-10f ? 1e-10f : (den > 1.0f ? 1.0f : den);
                    lg[t] = align[t] / den;
                }
                scan_blocks(lg, T, 1, nal);                                  /* inclusive cumsum */
                for (int t = 0; t < T; ++t) { const float pc = p[t] * cp[t]; nal[t] = pc * nal[t]; }
                memcpy(align, nal, sizeof(float) * T);
            }
            if (align_out) for (int t = 0; t < T; ++t) align_out[((size_t)n * T + t) * iters + it] = align[t];   /* tacotron.py:223 */
            twvo_cdot_rows(memo, ENC, align, T, ctx);                       /* rnn_wrappers.py:390 context = alignments . values */
            /* rnn_wrappers.py:463 concat(output, attention) -> OutputProjectionWrapper(dec_rnn) [RECALLED-TF: linear + bias] */
            memcpy(cin, ah, sizeof(float) * AS); memcpy(cin + AS, ctx, sizeof(float) * ENC);
            if (SEc) memcpy(cin + AS + ENC, sev, sizeof(float) * SEc);      /* rnn_wrappers.py:458-460 concat([output, attention, embed_to_concat]) */
            float y[512];
            dense_rows(cin, 1, AS + ENC + SEc, cW, cb, DR, ACT_NONE, y);
            for (int i = 0; i < d->dec_layers; ++i) {                       /* tacotron.py:167 ResidualWrapper(GRUCell): y + GRU(y) */
                gru_step(y, DR, rh[i], DR, rWg[i], rbg[i], rWc[i], rbc[i], hn);
                memcpy(rh[i], hn, sizeof(float) * DR);
                for (int j = 0; j < DR; ++j) y[j] = y[j] + hn[j];
            }
            dense_rows(y, 1, DR, oW, ob, M * R, ACT_NONE, outp);            /* tacotron.py:173 */
            memcpy(mel_out + ((size_t)n * TO + (size_t)it * R) * M, outp, sizeof(float) * M * R);   /* tacotron.py:204 reshape */
            memcpy(frame, outp + (size_t)M * (R - 1), sizeof(float) * M);   /* helpers.py:40 last frame fed back */
        }
        if (linear_out) {
            /* tacotron.py:209 post cbhg on the mel outputs (no lengths, zero initial states), tacotron.py:219 linear dense */
            cbhg_one(&post, mel_out + (size_t)n * TO * M, TO, TO, M, d->post_bank, d->post_bank_ch, d->post_proj, d->post_proj_w,
                     d->post_hw_depth, d->post_rnn, NULL, NULL, NULL, post_out);
            dense_rows(post_out, TO, 2 * d->post_rnn, lW, lb, d->num_freq, ACT_NONE, linear_out + (size_t)n * TO * d->num_freq);
        }
    }
    free(x0); free(x1); free(x2); free(memo); free(keys); free(post_out); free(score_t);
}

size_t twvo_taco_blob_floats(const twvo_taco_dims* d)
{
    /* walk the same order with a NULL base */
    size_t n = 0;
    const int E = d->emb, SE = d->spk_emb, P0 = d->enc_prenet[0], P1 = d->enc_prenet[1], RN = d->enc_rnn, A = d->att,
              AS = d->att_state, DR = d->dec_rnn, M = d->num_mels, R = d->r, ENC = 2 * RN;
    n += (size_t)d->n_symbols * E;
    const int dn[8] = { P1, 2 * RN, AS, DR, DR, DR, DR, DR };
    const int simple = d->n_speakers > 1 && d->model_simple && SE != 1;
    const int SEc = simple ? SE : 0;
    if (simple) {
        n += (size_t)d->n_speakers * SE;
    } else if (d->n_speakers > 1 && SE == 1) {
        for (int i = 0; i < 3 + d->dec_layers; ++i) n += (size_t)d->n_speakers * dn[i];
    } else if (d->n_speakers > 1) {
        n += (size_t)d->n_speakers * SE;
        for (int i = 0; i < 3 + d->dec_layers; ++i) n += (size_t)SE * dn[i] + dn[i];
    }
    n += (size_t)E * P0 + P0 + (size_t)P0 * P1 + P1;
#define CBHG_N(Cin, bank, bch, proj, pw, depth, rnn)                                                              \
    do {                                                                                                          \
        for (int k = 1; k <= (bank); ++k) n += (size_t)k * (Cin) * (bch) + 3 * (size_t)(bch);                     \
        n += (size_t)(pw) * (bank) * (bch) * (proj)[0] + 3 * (size_t)(proj)[0];                                   \
        n += (size_t)(pw) * (proj)[0] * (proj)[1] + 3 * (size_t)(proj)[1];                                        \
        if ((proj)[1] != (rnn)) n += (size_t)(proj)[1] * (rnn) + (rnn);                                           \
        n += (size_t)(depth) * 2 * ((size_t)(rnn) * (rnn) + (rnn));                                               \
        n += 2 * ((size_t)2 * (rnn) * 2 * (rnn) + 2 * (rnn) + (size_t)2 * (rnn) * (rnn) + (rnn));                 \
    } while (0)
    CBHG_N(P1, d->enc_bank, d->enc_bank_ch, d->enc_proj, d->enc_proj_w, d->enc_hw_depth, RN);
    n += (size_t)ENC * A + (size_t)AS * A + A + 1 + A + 1;
    n += (size_t)M * d->dec_prenet[0] + d->dec_prenet[0] + (size_t)d->dec_prenet[0] * d->dec_prenet[1] + d->dec_prenet[1];
    const int ain = d->dec_prenet[1] + SEc + ENC;
    n += (size_t)(ain + AS) * 2 * AS + 2 * AS + (size_t)(ain + AS) * AS + AS;
    n += (size_t)(AS + ENC + SEc) * DR + DR;
    for (int i = 0; i < d->dec_layers; ++i) n += (size_t)2 * DR * 2 * DR + 2 * DR + (size_t)2 * DR * DR + DR;
    n += (size_t)DR * M * R + M * R;
    CBHG_N(M, d->post_bank, d->post_bank_ch, d->post_proj, d->post_proj_w, d->post_hw_depth, d->post_rnn);
    n += (size_t)2 * d->post_rnn * d->num_freq + d->num_freq;
    return n;
}
