/*
 * vits_oracle.c — CPU restatement of the VITS inference graph that phoonnx runs
 * through onnxruntime (`phoonnx/voice.py:374-377`), i.e. a restatement of
 * `SynthesizerTrn.infer` (`phoonnx_train/vits/models.py:681-722`) as traced by
 * `phoonnx_train/export_onnx.py:250-327`.
 *
 * TEST INFRASTRUCTURE.  This file is the *checker*: only tests/, the smoke test
 * and bench.py's cpu_baseline leg may load it.  The product (phoonnx_amd/) never
 * links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every stage output of
 * this file against fixtures produced by the reference's own PyTorch module
 * (oracle/gen_golden.py imports /root/reference in the build container) for three
 * tiny presets; oracle/validate_big.py repeats that for the full-size "medium",
 * "high" and multi-speaker presets in-container.  The reference's own tests hold
 * no vectors for this path, and onnxruntime itself is not
 * installable here, so the PyTorch definition that the .onnx is traced from is the
 * anchor.
 *
 * All tensors are float32, layout [B, C, T] row-major, like the reference.
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/phoonnx_train/vits/).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ model store */

typedef struct {
    char name[128];
    float *data;
    int nd;
    int64_t d[4];
} vo_tensor;

typedef struct {
    char key[128];
    int64_t v;
} vo_int;

typedef struct vo_model {
    vo_tensor *t;
    int nt, capt;
    vo_int *ints;
    int ni, capi;
    vo_tensor *res;
    int nres, capres;
    char err[512];
} vo_model;

static void vo_fail(vo_model *m, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(m->err, sizeof m->err, fmt, ap);
    va_end(ap);
}

vo_model *vo_new(void) { return (vo_model *)calloc(1, sizeof(vo_model)); }

static void vo_clear_results(vo_model *m) {
    for (int i = 0; i < m->nres; i++) free(m->res[i].data);
    m->nres = 0;
}

void vo_free(vo_model *m) {
    if (!m) return;
    for (int i = 0; i < m->nt; i++) free(m->t[i].data);
    vo_clear_results(m);
    free(m->t);
    free(m->ints);
    free(m->res);
    free(m);
}

const char *vo_error(vo_model *m) { return m->err; }

static int64_t numel(int nd, const int64_t *d) {
    int64_t n = 1;
    for (int i = 0; i < nd; i++) n *= d[i];
    return n;
}

int vo_set_tensor(vo_model *m, const char *name, const float *data, int nd, const int64_t *dims) {
    if (nd > 4) return -1;
    if (m->nt == m->capt) {
        m->capt = m->capt ? m->capt * 2 : 256;
        m->t = (vo_tensor *)realloc(m->t, sizeof(vo_tensor) * m->capt);
    }
    vo_tensor *t = &m->t[m->nt++];
    memset(t, 0, sizeof *t);
    snprintf(t->name, sizeof t->name, "%s", name);
    t->nd = nd;
    for (int i = 0; i < nd; i++) t->d[i] = dims[i];
    int64_t n = numel(nd, dims);
    t->data = (float *)malloc(sizeof(float) * (n ? n : 1));
    memcpy(t->data, data, sizeof(float) * n);
    return 0;
}

int vo_set_int(vo_model *m, const char *key, int64_t v) {
    if (m->ni == m->capi) {
        m->capi = m->capi ? m->capi * 2 : 128;
        m->ints = (vo_int *)realloc(m->ints, sizeof(vo_int) * m->capi);
    }
    snprintf(m->ints[m->ni].key, sizeof m->ints[m->ni].key, "%s", key);
    m->ints[m->ni++].v = v;
    return 0;
}

static const vo_tensor *T_opt(vo_model *m, const char *fmt, ...) {
    char name[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    for (int i = 0; i < m->nt; i++)
        if (!strcmp(m->t[i].name, name)) return &m->t[i];
    return NULL;
}

static int g_missing; /* set when a required tensor is absent */

static const vo_tensor *T_req(vo_model *m, const char *fmt, ...) {
    static vo_tensor dummy;
    char name[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    for (int i = 0; i < m->nt; i++)
        if (!strcmp(m->t[i].name, name)) return &m->t[i];
    vo_fail(m, "missing tensor %s", name);
    g_missing = 1;
    return &dummy;
}

static int64_t I_get(vo_model *m, int64_t dflt, const char *fmt, ...) {
    char name[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    for (int i = 0; i < m->ni; i++)
        if (!strcmp(m->ints[i].key, name)) return m->ints[i].v;
    return dflt;
}

static float *put_result(vo_model *m, const char *name, int nd, const int64_t *dims) {
    if (m->nres == m->capres) {
        m->capres = m->capres ? m->capres * 2 : 32;
        m->res = (vo_tensor *)realloc(m->res, sizeof(vo_tensor) * m->capres);
    }
    vo_tensor *t = &m->res[m->nres++];
    memset(t, 0, sizeof *t);
    snprintf(t->name, sizeof t->name, "%s", name);
    t->nd = nd;
    for (int i = 0; i < nd; i++) t->d[i] = dims[i];
    int64_t n = numel(nd, dims);
    t->data = (float *)calloc(n ? n : 1, sizeof(float));
    return t->data;
}

const float *vo_result(vo_model *m, const char *name, int *nd, int64_t *dims) {
    for (int i = 0; i < m->nres; i++)
        if (!strcmp(m->res[i].name, name)) {
            *nd = m->res[i].nd;
            for (int k = 0; k < m->res[i].nd; k++) dims[k] = m->res[i].d[k];
            return m->res[i].data;
        }
    return NULL;
}

static float *falloc(int64_t n) { return (float *)calloc(n > 0 ? n : 1, sizeof(float)); }

/* ------------------------------------------------------------------ primitive ops */

/* torch.nn.Conv1d forward, stride 1, zero padding `pad` on both sides, dilation,
 * groups (only groups==1 or groups==Cin==Cout are used by the graph).
 * weight [Cout, Cin/groups, K]; out length = T + 2*pad - dil*(K-1).
 * Used by every Conv1d in models.py / modules.py / attentions.py. */
void vo_conv1d(const float *x, int B, int Cin, int T, const float *w, const float *bias,
               int Cout, int K, int dil, int pad_l, int pad_r, int groups, float *out) {
    int To = T + pad_l + pad_r - dil * (K - 1);
    if (To < 0) To = 0;
    int cig = Cin / groups, cog = Cout / groups;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int co = 0; co < Cout; co++) {
            float *o = out + ((int64_t)b * Cout + co) * To;
            float bv = bias ? bias[co] : 0.f;
            for (int t = 0; t < To; t++) o[t] = bv;
            int g = co / cog;
            for (int ci = 0; ci < cig; ci++) {
                const float *xi = x + ((int64_t)b * Cin + g * cig + ci) * T;
                const float *wk = w + ((int64_t)co * cig + ci) * K;
                for (int k = 0; k < K; k++) {
                    float wv = wk[k];
                    int off = k * dil - pad_l; /* input index = t + off */
                    int t0 = off < 0 ? -off : 0;
                    int t1 = To;
                    if (t1 + off > T) t1 = T - off;
                    for (int t = t0; t < t1; t++) o[t] += wv * xi[t + off];
                }
            }
        }
}

/* torch.nn.ConvTranspose1d forward (models.py:321-332): weight [Cin, Cout, K],
 * stride u, padding p; out length = (T-1)*u - 2p + K. */
void vo_conv_transpose1d(const float *x, int B, int Cin, int T, const float *w, const float *bias,
                         int Cout, int K, int stride, int pad, float *out) {
    int To = (T - 1) * stride - 2 * pad + K;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int co = 0; co < Cout; co++) {
            float *o = out + ((int64_t)b * Cout + co) * To;
            float bv = bias ? bias[co] : 0.f;
            for (int t = 0; t < To; t++) o[t] = bv;
            for (int ci = 0; ci < Cin; ci++) {
                const float *xi = x + ((int64_t)b * Cin + ci) * T;
                const float *wk = w + ((int64_t)ci * Cout + co) * K;
                for (int k = 0; k < K; k++) {
                    float wv = wk[k];
                    /* out index = i*stride - pad + k */
                    for (int i = 0; i < T; i++) {
                        int t = i * stride - pad + k;
                        if (t >= 0 && t < To) o[t] += wv * xi[i];
                    }
                }
            }
        }
}

/* modules.py:14-26 LayerNorm over the channel axis of [B,C,T], eps 1e-5, biased var */
static void layer_norm_c(float *x, int B, int C, int T, const float *gamma, const float *beta) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++) {
            float *p = x + (int64_t)b * C * T + t;
            float mean = 0.f;
            for (int c = 0; c < C; c++) mean += p[(int64_t)c * T];
            mean /= (float)C;
            float var = 0.f;
            for (int c = 0; c < C; c++) {
                float d = p[(int64_t)c * T] - mean;
                var += d * d;
            }
            var /= (float)C;
            float rs = 1.0f / sqrtf(var + 1e-5f);
            for (int c = 0; c < C; c++)
                p[(int64_t)c * T] = (p[(int64_t)c * T] - mean) * rs * gamma[c] + beta[c];
        }
}

static inline float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
static inline float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
static inline float softplusf_(float v) { return v > 20.0f ? v : log1pf(expf(v)); } /* F.softplus, threshold 20 */
static inline float lrelu(float v, float s) { return v > 0.f ? v : v * s; }

static void mul_mask(float *x, int B, int C, int T, const int64_t *len) {
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++) {
            float *p = x + ((int64_t)b * C + c) * T;
            for (int t = 0; t < T; t++)
                if (t >= len[b]) p[t] = 0.f;
        }
}

/* conv through named module "<prefix>.weight"/".bias" with same padding */
static float *conv_named(vo_model *m, const char *prefix, const float *x, int B, int T, int dil, int pad_l,
                         int pad_r, int groups, int *Cout_out) {
    const vo_tensor *w = T_req(m, "%s.weight", prefix);
    const vo_tensor *bs = T_opt(m, "%s.bias", prefix);
    if (g_missing) return NULL;
    int Cout = (int)w->d[0], cig = (int)w->d[1], K = (int)w->d[2];
    int Cin = cig * groups;
    int To = T + pad_l + pad_r - dil * (K - 1);
    float *out = falloc((int64_t)B * Cout * To);
    vo_conv1d(x, B, Cin, T, w->data, bs ? bs->data : NULL, Cout, K, dil, pad_l, pad_r, groups, out);
    if (Cout_out) *Cout_out = Cout;
    return out;
}

/* ------------------------------------------------------------------ text encoder */

/* attentions.py:215-272 (MultiHeadAttention.forward/attention) with window_size
 * relative embeddings shared across heads (heads_share=True :196-205).
 * The pad/slice/skew helpers (:292-348) reduce to
 *   scores[i][j] += q_i . E_k[j-i+w]      for |j-i| <= w
 *   out_i        += sum_{|j-i|<=w} p[i][j] * E_v[j-i+w]
 * (verified against the module incl. T < w+1 where the reference slices, :295-297). */
/* core of attentions.py:225-272 on already-projected q,k,v [B,C,T]; o [B,C,T] */
void vo_attention_core(const float *q, const float *k, const float *v, const float *ek, const float *ev, int B,
                       int C, int T, int dk, int win, const int64_t *len, float *o) {
    int H = C / dk;
    float scale = sqrtf((float)dk);
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; b++)
        for (int h = 0; h < H; h++) {
            const float *qh = q + ((int64_t)b * C + h * dk) * T; /* [dk][T] */
            const float *kh = k + ((int64_t)b * C + h * dk) * T;
            const float *vh = v + ((int64_t)b * C + h * dk) * T;
            float *oh = o + ((int64_t)b * C + h * dk) * T;
            float *sc = (float *)malloc(sizeof(float) * T);
            for (int i = 0; i < T; i++) {
                /* scores = (q/sqrt(dk)) k^T  (:232) */
                for (int j = 0; j < T; j++) {
                    float s = 0.f;
                    for (int d = 0; d < dk; d++) s += (qh[(int64_t)d * T + i] / scale) * kh[(int64_t)d * T + j];
                    int r = j - i + win;
                    if (r >= 0 && r <= 2 * win) { /* :237-242 */
                        float rl = 0.f;
                        for (int d = 0; d < dk; d++) rl += (qh[(int64_t)d * T + i] / scale) * ek[(int64_t)r * dk + d];
                        s += rl;
                    }
                    /* attn_mask = x_mask[i]*x_mask[j]; masked_fill(mask==0, -1e4) (:247, :61) */
                    if (i >= len[b] || j >= len[b]) s = -1e4f;
                    sc[j] = s;
                }
                float mx = sc[0];
                for (int j = 1; j < T; j++) mx = sc[j] > mx ? sc[j] : mx;
                float sum = 0.f;
                for (int j = 0; j < T; j++) {
                    sc[j] = expf(sc[j] - mx);
                    sum += sc[j];
                }
                for (int j = 0; j < T; j++) sc[j] /= sum; /* :258 */
                for (int d = 0; d < dk; d++) {
                    float acc = 0.f;
                    for (int j = 0; j < T; j++) acc += sc[j] * vh[(int64_t)d * T + j]; /* :260 */
                    float rel = 0.f;
                    for (int r = 0; r <= 2 * win; r++) { /* :261-268 */
                        int j = i + r - win;
                        if (j >= 0 && j < T) rel += sc[j] * ev[(int64_t)r * dk + d];
                    }
                    oh[(int64_t)d * T + i] = acc + rel;
                }
            }
            free(sc);
        }
}

static float *mha(vo_model *m, const char *pfx, const float *x, int B, int C, int T, const int64_t *len) {
    const vo_tensor *ek = T_req(m, "%s.emb_rel_k", pfx);
    const vo_tensor *ev = T_req(m, "%s.emb_rel_v", pfx);
    if (g_missing) return NULL;
    int win = ((int)ek->d[1] - 1) / 2, dk = (int)ek->d[2];
    char nm[200];
    snprintf(nm, sizeof nm, "%s.conv_q", pfx);
    float *q = conv_named(m, nm, x, B, T, 1, 0, 0, 1, NULL);
    snprintf(nm, sizeof nm, "%s.conv_k", pfx);
    float *k = conv_named(m, nm, x, B, T, 1, 0, 0, 1, NULL);
    snprintf(nm, sizeof nm, "%s.conv_v", pfx);
    float *v = conv_named(m, nm, x, B, T, 1, 0, 0, 1, NULL);
    if (g_missing) return NULL;
    float *o = falloc((int64_t)B * C * T);
    vo_attention_core(q, k, v, ek->data, ev->data, B, C, T, dk, win, len, o);
    free(q);
    free(k);
    free(v);
    snprintf(nm, sizeof nm, "%s.conv_o", pfx);
    float *y = conv_named(m, nm, o, B, T, 1, 0, 0, 1, NULL);
    free(o);
    return y;
}

/* attentions.py:386-407 FFN.forward (non-causal, relu): conv(pad(x*mask)) -> relu ->
 * conv(pad(h*mask)) -> *mask; same padding (k-1)//2 left, k//2 right (:419-427) */
static float *ffn(vo_model *m, const char *pfx, const float *x, int B, int C, int T, const int64_t *len) {
    char nm[200];
    const vo_tensor *w1 = T_req(m, "%s.conv_1.weight", pfx);
    if (g_missing) return NULL;
    int K = (int)w1->d[2], Fc = (int)w1->d[0];
    float *xm = falloc((int64_t)B * C * T);
    memcpy(xm, x, sizeof(float) * (int64_t)B * C * T);
    mul_mask(xm, B, C, T, len);
    snprintf(nm, sizeof nm, "%s.conv_1", pfx);
    float *h = conv_named(m, nm, xm, B, T, 1, (K - 1) / 2, K / 2, 1, NULL);
    free(xm);
    if (!h) return NULL;
    for (int64_t i = 0; i < (int64_t)B * Fc * T; i++) h[i] = h[i] > 0.f ? h[i] : 0.f;
    mul_mask(h, B, Fc, T, len);
    snprintf(nm, sizeof nm, "%s.conv_2", pfx);
    float *y = conv_named(m, nm, h, B, T, 1, (K - 1) / 2, K / 2, 1, NULL);
    free(h);
    if (!y) return NULL;
    mul_mask(y, B, C, T, len);
    return y;
}

/* models.py:198-209 TextEncoder.forward + attentions.py:60-74 Encoder.forward */
static int text_encoder(vo_model *m, const int64_t *ids, const int64_t *len, int B, int T, float **x_out,
                        float **mp_out, float **logs_out, int *Hc, int *Cc) {
    const vo_tensor *emb = T_req(m, "enc_p.emb.weight");
    if (g_missing) return -1;
    int H = (int)emb->d[1], V = (int)emb->d[0];
    float *x = falloc((int64_t)B * H * T);
    float sq = (float)sqrt((double)H); /* math.sqrt(hidden) then fp32 multiply (:199) */
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++) {
            int64_t id = ids[(int64_t)b * T + t];
            if (id < 0 || id >= V) {
                vo_fail(m, "phoneme id %lld out of range [0,%d)", (long long)id, V);
                free(x);
                return -1;
            }
            for (int c = 0; c < H; c++)
                x[((int64_t)b * H + c) * T + t] = t < len[b] ? emb->data[id * H + c] * sq : 0.f;
        }
    { /* tap "emb": the integer gather itself (bit-exact row lookup, one fp32 multiply), before any float pipeline */
        int64_t de[3] = {B, H, T};
        memcpy(put_result(m, "emb", 3, de), x, sizeof(float) * B * H * T);
    }
    int L = 0;
    while (T_opt(m, "enc_p.encoder.attn_layers.%d.conv_q.weight", L)) L++;
    for (int l = 0; l < L; l++) {
        char pfx[160];
        snprintf(pfx, sizeof pfx, "enc_p.encoder.attn_layers.%d", l);
        float *y = mha(m, pfx, x, B, H, T, len);
        if (!y) { free(x); return -1; }
        for (int64_t i = 0; i < (int64_t)B * H * T; i++) x[i] += y[i];
        free(y);
        layer_norm_c(x, B, H, T, T_req(m, "enc_p.encoder.norm_layers_1.%d.gamma", l)->data,
                     T_req(m, "enc_p.encoder.norm_layers_1.%d.beta", l)->data);
        snprintf(pfx, sizeof pfx, "enc_p.encoder.ffn_layers.%d", l);
        y = ffn(m, pfx, x, B, H, T, len);
        if (!y) { free(x); return -1; }
        for (int64_t i = 0; i < (int64_t)B * H * T; i++) x[i] += y[i];
        free(y);
        layer_norm_c(x, B, H, T, T_req(m, "enc_p.encoder.norm_layers_2.%d.gamma", l)->data,
                     T_req(m, "enc_p.encoder.norm_layers_2.%d.beta", l)->data);
        if (g_missing) { free(x); return -1; }
    }
    mul_mask(x, B, H, T, len);
    int C2;
    float *stats = conv_named(m, "enc_p.proj", x, B, T, 1, 0, 0, 1, &C2);
    if (!stats) { free(x); return -1; }
    mul_mask(stats, B, C2, T, len);
    int C = C2 / 2;
    float *mp = falloc((int64_t)B * C * T), *lg = falloc((int64_t)B * C * T);
    for (int b = 0; b < B; b++) {
        memcpy(mp + (int64_t)b * C * T, stats + (int64_t)b * C2 * T, sizeof(float) * C * T);
        memcpy(lg + (int64_t)b * C * T, stats + ((int64_t)b * C2 + C) * T, sizeof(float) * C * T);
    }
    free(stats);
    *x_out = x;
    *mp_out = mp;
    *logs_out = lg;
    *Hc = H;
    *Cc = C;
    return 0;
}

/* ------------------------------------------------------------------ duration predictors */

/* modules.py:117-129 DDSConv.forward; x is updated in place. g (optional) is added first. */
static int ddsconv(vo_model *m, const char *pfx, float *x, const float *g, int B, int C, int T,
                   const int64_t *len) {
    int64_t n = (int64_t)B * C * T;
    if (g)
        for (int64_t i = 0; i < n; i++) x[i] += g[i];
    for (int l = 0; T_opt(m, "%s.convs_sep.%d.weight", pfx, l); l++) {
        const vo_tensor *w = T_req(m, "%s.convs_sep.%d.weight", pfx, l);
        int K = (int)w->d[2];
        int dil = (int)I_get(m, -1, "%s.convs_sep.%d.dilation", pfx, l);
        if (dil < 0) { /* dilation = kernel_size**i (:101) */
            dil = 1;
            for (int i = 0; i < l; i++) dil *= K;
        }
        int pad = (K * dil - dil) / 2; /* :102 */
        float *xm = falloc(n);
        memcpy(xm, x, sizeof(float) * n);
        mul_mask(xm, B, C, T, len);
        char nm[200];
        snprintf(nm, sizeof nm, "%s.convs_sep.%d", pfx, l);
        float *y = conv_named(m, nm, xm, B, T, dil, pad, pad, C, NULL);
        free(xm);
        if (!y) return -1;
        layer_norm_c(y, B, C, T, T_req(m, "%s.norms_1.%d.gamma", pfx, l)->data,
                     T_req(m, "%s.norms_1.%d.beta", pfx, l)->data);
        for (int64_t i = 0; i < n; i++) y[i] = gelu_erf(y[i]);
        snprintf(nm, sizeof nm, "%s.convs_1x1.%d", pfx, l);
        float *y2 = conv_named(m, nm, y, B, T, 1, 0, 0, 1, NULL);
        free(y);
        if (!y2) return -1;
        layer_norm_c(y2, B, C, T, T_req(m, "%s.norms_2.%d.gamma", pfx, l)->data,
                     T_req(m, "%s.norms_2.%d.beta", pfx, l)->data);
        for (int64_t i = 0; i < n; i++) x[i] += gelu_erf(y2[i]);
        free(y2);
        if (g_missing) return -1;
    }
    mul_mask(x, B, C, T, len);
    return 0;
}

/* transforms.py:50-98 + 101-191, inverse branch, tails="linear", tail_bound=5, for one
 * element.  uw/uh have nb entries (already divided by sqrt(filter), modules.py:505-508),
 * ud has nb-1 entries. */
static float rqs_inverse(float x, const float *uw, const float *uh, const float *ud, int nb) {
    const float tb = 5.0f, minw = 1e-3f, minh = 1e-3f, mind = 1e-3f;
    if (!(x >= -tb && x <= tb)) return x; /* :62,75 */
    float w[32], h[32], cw[33], ch[33], d[33];
    /* widths = softmax(uw); widths = minw + (1 - minw*nb)*widths (:125-126) */
    float mx = uw[0];
    for (int i = 1; i < nb; i++) mx = uw[i] > mx ? uw[i] : mx;
    float s = 0.f;
    for (int i = 0; i < nb; i++) { w[i] = expf(uw[i] - mx); s += w[i]; }
    for (int i = 0; i < nb; i++) w[i] = minw + (1.0f - minw * nb) * (w[i] / s);
    cw[0] = 0.f;
    for (int i = 0; i < nb; i++) cw[i + 1] = cw[i] + w[i]; /* cumsum, pad (1,0) (:127-128) */
    for (int i = 0; i <= nb; i++) cw[i] = (tb - (-tb)) * cw[i] + (-tb); /* :129 */
    cw[0] = -tb; cw[nb] = tb;                                            /* :130-132 */
    for (int i = 0; i < nb; i++) w[i] = cw[i + 1] - cw[i];               /* :133 */
    /* derivatives: pad (1,1) with log(exp(1-mind)-1) (:69-73), then mind + softplus (:135) */
    float cst = (float)log(exp(1.0 - 1e-3) - 1.0);
    d[0] = mind + softplusf_(cst);
    d[nb] = mind + softplusf_(cst);
    for (int i = 1; i < nb; i++) d[i] = mind + softplusf_(ud[i - 1]);
    mx = uh[0];
    for (int i = 1; i < nb; i++) mx = uh[i] > mx ? uh[i] : mx;
    s = 0.f;
    for (int i = 0; i < nb; i++) { h[i] = expf(uh[i] - mx); s += h[i]; }
    for (int i = 0; i < nb; i++) h[i] = minh + (1.0f - minh * nb) * (h[i] / s);
    ch[0] = 0.f;
    for (int i = 0; i < nb; i++) ch[i + 1] = ch[i] + h[i];
    for (int i = 0; i <= nb; i++) ch[i] = (tb - (-tb)) * ch[i] + (-tb);
    ch[0] = -tb; ch[nb] = tb;
    for (int i = 0; i < nb; i++) h[i] = ch[i + 1] - ch[i];
    /* searchsorted(cumheights, x): last knot += 1e-6 (:44-47,148) */
    int bin = -1;
    for (int i = 0; i <= nb; i++) {
        float loc = ch[i] + (i == nb ? 1e-6f : 0.f);
        if (x >= loc) bin++;
    }
    if (bin < 0) bin = 0;
    if (bin > nb - 1) bin = nb - 1;
    float icw = cw[bin], ibw = w[bin], ich = ch[bin], ih = h[bin];
    float delta = h[bin] / w[bin], dd = d[bin], dp1 = d[bin + 1];
    /* :165-177 */
    float a = (x - ich) * (dd + dp1 - 2.0f * delta) + ih * (delta - dd);
    float b = ih * dd - (x - ich) * (dd + dp1 - 2.0f * delta);
    float c = -delta * (x - ich);
    float disc = b * b - 4.0f * a * c;
    float root = (2.0f * c) / (-b - sqrtf(disc));
    return root * ibw + icw;
}

/* modules.py:496-527 ConvFlow.forward(reverse=True); z is [B,2,T], updated in place.
 * cond = the SDP's conditioning h [B,C,T]. */
static int convflow_reverse(vo_model *m, const char *pfx, float *z, const float *cond, int B, int C, int T,
                            const int64_t *len) {
    const vo_tensor *pw = T_req(m, "%s.pre.weight", pfx); /* [C,1,1] */
    const vo_tensor *pb = T_req(m, "%s.pre.bias", pfx);
    if (g_missing) return -1;
    float *h = falloc((int64_t)B * C * T);
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int t = 0; t < T; t++)
                h[((int64_t)b * C + c) * T + t] = pw->data[c] * z[((int64_t)b * 2 + 0) * T + t] + pb->data[c];
    char nm[200];
    snprintf(nm, sizeof nm, "%s.convs", pfx);
    if (ddsconv(m, nm, h, cond, B, C, T, len)) { free(h); return -1; }
    int P;
    snprintf(nm, sizeof nm, "%s.proj", pfx);
    float *pr = conv_named(m, nm, h, B, T, 1, 0, 0, 1, &P);
    free(h);
    if (!pr) return -1;
    mul_mask(pr, B, P, T, len);
    int nb = (P + 1) / 3; /* P = 3*nb - 1 (:491) */
    float isq = sqrtf((float)C); /* / math.sqrt(filter_channels) (:505-508) */
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++) {
            float uw[32], uh[32], ud[32];
            for (int i = 0; i < nb; i++) {
                uw[i] = pr[((int64_t)b * P + i) * T + t] / isq;
                uh[i] = pr[((int64_t)b * P + nb + i) * T + t] / isq;
            }
            for (int i = 0; i < nb - 1; i++) ud[i] = pr[((int64_t)b * P + 2 * nb + i) * T + t];
            float *x0 = &z[((int64_t)b * 2 + 0) * T + t], *x1 = &z[((int64_t)b * 2 + 1) * T + t];
            float y1 = rqs_inverse(*x1, uw, uh, ud, nb);
            float mk = t < len[b] ? 1.f : 0.f; /* cat([x0,x1]) * x_mask (:521) */
            *x0 = *x0 * mk;
            *x1 = y1 * mk;
        }
    free(pr);
    return 0;
}

/* models.py:63-70,108-117 StochasticDurationPredictor.forward(reverse=True) */
static float *sdp_reverse(vo_model *m, const float *x, const float *gcond /*[B,gin] or NULL*/, int gin, int B,
                          int H, int T, const int64_t *len, const float *noise /*[B,2,T] or NULL*/,
                          float noise_w) {
    int C;
    float *h = conv_named(m, "dp.pre", x, B, T, 1, 0, 0, 1, &C);
    if (!h) return NULL;
    if (gcond) { /* x = x + self.cond(g) (:66-68) */
        const vo_tensor *cw = T_req(m, "dp.cond.weight"), *cb = T_req(m, "dp.cond.bias");
        if (g_missing) { free(h); return NULL; }
        for (int b = 0; b < B; b++)
            for (int c = 0; c < C; c++) {
                float s = cb->data[c];
                for (int k = 0; k < gin; k++) s += cw->data[(int64_t)c * gin + k] * gcond[(int64_t)b * gin + k];
                for (int t = 0; t < T; t++) h[((int64_t)b * C + c) * T + t] += s;
            }
    }
    if (ddsconv(m, "dp.convs", h, NULL, B, C, T, len)) { free(h); return NULL; }
    float *cond = conv_named(m, "dp.proj", h, B, T, 1, 0, 0, 1, NULL);
    free(h);
    if (!cond) return NULL;
    mul_mask(cond, B, C, T, len);
    float *z = falloc((int64_t)B * 2 * T);
    if (noise)
        for (int64_t i = 0; i < (int64_t)B * 2 * T; i++) z[i] = noise[i] * noise_w; /* :111 */
    /* flows = reversed(self.flows); drop flows[-2] (:109-110).  self.flows =
     * [EA0, CF1, Flip2, CF3, Flip4, CF5, Flip6, CF7, Flip8]  (:35-40, n_flows=4)
     * reversed -> [Flip8, CF7, Flip6, CF5, Flip4, CF3, Flip2, CF1, EA0]; removing the
     * one before last (CF1) -> [Flip, CF7, Flip, CF5, Flip, CF3, Flip, EA0]. */
    int order[3] = {7, 5, 3};
    for (int f = 0; f < 3; f++) {
        /* Flip (modules.py:384-391): swap the two channels, no mask */
        for (int b = 0; b < B; b++)
            for (int t = 0; t < T; t++) {
                float a = z[((int64_t)b * 2) * T + t];
                z[((int64_t)b * 2) * T + t] = z[((int64_t)b * 2 + 1) * T + t];
                z[((int64_t)b * 2 + 1) * T + t] = a;
            }
        char pfx[64];
        snprintf(pfx, sizeof pfx, "dp.flows.%d", order[f]);
        if (convflow_reverse(m, pfx, z, cond, B, C, T, len)) { free(z); free(cond); return NULL; }
    }
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++) {
            float a = z[((int64_t)b * 2) * T + t];
            z[((int64_t)b * 2) * T + t] = z[((int64_t)b * 2 + 1) * T + t];
            z[((int64_t)b * 2 + 1) * T + t] = a;
        }
    /* ElementwiseAffine reverse (modules.py:407-409): (x - m) * exp(-logs) * mask */
    const vo_tensor *em = T_req(m, "dp.flows.0.m"), *el = T_req(m, "dp.flows.0.logs");
    if (g_missing) { free(z); free(cond); return NULL; }
    float *logw = falloc((int64_t)B * T);
    for (int b = 0; b < B; b++)
        for (int t = 0; t < T; t++) {
            float mk = t < len[b] ? 1.f : 0.f;
            logw[(int64_t)b * T + t] = (z[((int64_t)b * 2) * T + t] - em->data[0]) * expf(-el->data[0]) * mk;
        }
    free(z);
    free(cond);
    (void)H;
    return logw;
}

/* models.py:151-165 DurationPredictor.forward (use_sdp=False) */
static float *dp_plain(vo_model *m, const float *x, const float *gcond, int gin, int B, int H, int T,
                       const int64_t *len) {
    int64_t n = (int64_t)B * H * T;
    float *xi = falloc(n);
    memcpy(xi, x, sizeof(float) * n);
    if (gcond) { /* x = x + self.cond(g) (:153-155) */
        const vo_tensor *cw = T_req(m, "dp.cond.weight"), *cb = T_req(m, "dp.cond.bias");
        if (g_missing) { free(xi); return NULL; }
        for (int b = 0; b < B; b++)
            for (int c = 0; c < H; c++) {
                float s = cb->data[c];
                for (int k = 0; k < gin; k++) s += cw->data[(int64_t)c * gin + k] * gcond[(int64_t)b * gin + k];
                for (int t = 0; t < T; t++) xi[((int64_t)b * H + c) * T + t] += s;
            }
    }
    mul_mask(xi, B, H, T, len);
    const vo_tensor *w1 = T_req(m, "dp.conv_1.weight");
    if (g_missing) { free(xi); return NULL; }
    int K = (int)w1->d[2], Fc;
    float *h = conv_named(m, "dp.conv_1", xi, B, T, 1, K / 2, K / 2, 1, &Fc);
    free(xi);
    if (!h) return NULL;
    for (int64_t i = 0; i < (int64_t)B * Fc * T; i++) h[i] = h[i] > 0.f ? h[i] : 0.f;
    layer_norm_c(h, B, Fc, T, T_req(m, "dp.norm_1.gamma")->data, T_req(m, "dp.norm_1.beta")->data);
    mul_mask(h, B, Fc, T, len);
    float *h2 = conv_named(m, "dp.conv_2", h, B, T, 1, K / 2, K / 2, 1, NULL);
    free(h);
    if (!h2) return NULL;
    for (int64_t i = 0; i < (int64_t)B * Fc * T; i++) h2[i] = h2[i] > 0.f ? h2[i] : 0.f;
    layer_norm_c(h2, B, Fc, T, T_req(m, "dp.norm_2.gamma")->data, T_req(m, "dp.norm_2.beta")->data);
    mul_mask(h2, B, Fc, T, len);
    float *o = conv_named(m, "dp.proj", h2, B, T, 1, 0, 0, 1, NULL);
    free(h2);
    if (!o) return NULL;
    mul_mask(o, B, 1, T, len);
    return o;
}

/* ------------------------------------------------------------------ flow */

/* modules.py:184-209 WN.forward; weights are the weight-norm-folded ones (App. B) */
static float *wn_forward(vo_model *m, const char *pfx, const float *xin, const float *gcond, int gin, int B,
                         int H, int T, const int64_t *len) {
    int64_t n = (int64_t)B * H * T;
    float *x = falloc(n), *out = falloc(n);
    memcpy(x, xin, sizeof(float) * n);
    int L = 0;
    while (T_opt(m, "%s.in_layers.%d.weight", pfx, L)) L++;
    float *gc = NULL; /* cond_layer(g): [B, 2*H*L] (:188-189) */
    if (gcond) {
        const vo_tensor *cw = T_req(m, "%s.cond_layer.weight", pfx), *cb = T_req(m, "%s.cond_layer.bias", pfx);
        if (g_missing) { free(x); free(out); return NULL; }
        int Cc = (int)cw->d[0];
        gc = falloc((int64_t)B * Cc);
        for (int b = 0; b < B; b++)
            for (int c = 0; c < Cc; c++) {
                float s = cb->data[c];
                for (int k = 0; k < gin; k++) s += cw->data[(int64_t)c * gin + k] * gcond[(int64_t)b * gin + k];
                gc[(int64_t)b * Cc + c] = s;
            }
    }
    for (int i = 0; i < L; i++) {
        const vo_tensor *w = T_req(m, "%s.in_layers.%d.weight", pfx, i);
        int K = (int)w->d[2];
        int dil = (int)I_get(m, 1, "%s.in_layers.%d.dilation", pfx, i);
        int pad = (K * dil - dil) / 2; /* :163 */
        char nm[200];
        snprintf(nm, sizeof nm, "%s.in_layers.%d", pfx, i);
        float *a = conv_named(m, nm, x, B, T, dil, pad, pad, 1, NULL); /* [B,2H,T] */
        if (!a) { free(x); free(out); free(gc); return NULL; }
        float *acts = falloc(n);
        /* fused_add_tanh_sigmoid_multiply (commons.py:99-106) */
        for (int b = 0; b < B; b++)
            for (int c = 0; c < H; c++) {
                float g0 = gc ? gc[(int64_t)b * 2 * H * L + (int64_t)i * 2 * H + c] : 0.f;
                float g1 = gc ? gc[(int64_t)b * 2 * H * L + (int64_t)i * 2 * H + H + c] : 0.f;
                const float *pa = a + ((int64_t)b * 2 * H + c) * T, *pb = a + ((int64_t)b * 2 * H + H + c) * T;
                float *po = acts + ((int64_t)b * H + c) * T;
                for (int t = 0; t < T; t++) po[t] = tanhf(pa[t] + g0) * sigmoidf_(pb[t] + g1);
            }
        free(a);
        int RC;
        snprintf(nm, sizeof nm, "%s.res_skip_layers.%d", pfx, i);
        float *rs = conv_named(m, nm, acts, B, T, 1, 0, 0, 1, &RC);
        free(acts);
        if (!rs) { free(x); free(out); free(gc); return NULL; }
        if (i < L - 1) { /* :203-206 */
            for (int b = 0; b < B; b++)
                for (int c = 0; c < H; c++)
                    for (int t = 0; t < T; t++) {
                        int64_t o = ((int64_t)b * H + c) * T + t;
                        float mk = t < len[b] ? 1.f : 0.f;
                        x[o] = (x[o] + rs[((int64_t)b * RC + c) * T + t]) * mk;
                        out[o] += rs[((int64_t)b * RC + H + c) * T + t];
                    }
        } else {
            for (int64_t o = 0; o < n; o++) out[o] += rs[o]; /* :208 */
        }
        free(rs);
    }
    mul_mask(out, B, H, T, len); /* :209 */
    free(x);
    free(gc);
    return out;
}

/* models.py:247-254 ResidualCouplingBlock.forward(reverse=True) +
 * modules.py:447-466 ResidualCouplingLayer.forward(reverse=True), mean_only=True */
static int flow_reverse(vo_model *m, float *z, const float *gcond, int gin, int B, int C, int T,
                        const int64_t *len) {
    int half = C / 2;
    int nfl = 0;
    while (T_opt(m, "flow.flows.%d.pre.weight", 2 * nfl)) nfl++;
    float *tmp = falloc((int64_t)B * C * T);
    for (int f = nfl - 1; f >= 0; f--) {
        /* Flip (modules.py:386): reverse channel order */
        for (int b = 0; b < B; b++)
            for (int c = 0; c < C; c++)
                memcpy(tmp + ((int64_t)b * C + c) * T, z + ((int64_t)b * C + (C - 1 - c)) * T, sizeof(float) * T);
        memcpy(z, tmp, sizeof(float) * (int64_t)B * C * T);
        char pfx[96], nm[160];
        snprintf(pfx, sizeof pfx, "flow.flows.%d", 2 * f);
        float *x0 = falloc((int64_t)B * half * T);
        for (int b = 0; b < B; b++)
            memcpy(x0 + (int64_t)b * half * T, z + (int64_t)b * C * T, sizeof(float) * half * T);
        int H;
        snprintf(nm, sizeof nm, "%s.pre", pfx);
        float *h = conv_named(m, nm, x0, B, T, 1, 0, 0, 1, &H);
        free(x0);
        if (!h) { free(tmp); return -1; }
        mul_mask(h, B, H, T, len); /* :449 */
        snprintf(nm, sizeof nm, "%s.enc", pfx);
        float *e = wn_forward(m, nm, h, gcond, gin, B, H, T, len);
        free(h);
        if (!e) { free(tmp); return -1; }
        int PC;
        snprintf(nm, sizeof nm, "%s.post", pfx);
        float *st = conv_named(m, nm, e, B, T, 1, 0, 0, 1, &PC);
        free(e);
        if (!st) { free(tmp); return -1; }
        mul_mask(st, B, PC, T, len); /* :451 */
        /* x1 = (x1 - m) * exp(-logs) * mask, logs = 0 for mean_only (:455-456,464) */
        for (int b = 0; b < B; b++)
            for (int c = 0; c < half; c++)
                for (int t = 0; t < T; t++) {
                    int64_t o = ((int64_t)b * C + half + c) * T + t;
                    float mk = t < len[b] ? 1.f : 0.f;
                    float lg = PC == 2 * half ? st[((int64_t)b * PC + half + c) * T + t] : 0.f;
                    z[o] = (z[o] - st[((int64_t)b * PC + c) * T + t]) * expf(-lg) * mk;
                }
        free(st);
    }
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------ HiFi-GAN generator */

static void lrelu_copy(float *dst, const float *src, int64_t n, float slope) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) dst[i] = lrelu(src[i], slope);
}

/* modules.py:301-314 ResBlock1.forward / modules.py:355-364 ResBlock2.forward, x_mask=None */
static float *resblock(vo_model *m, int idx, const float *xin, int B, int C, int T) {
    int64_t n = (int64_t)B * C * T;
    float *x = falloc(n), *xt = falloc(n);
    memcpy(x, xin, sizeof(float) * n);
    char nm[128];
    if (T_opt(m, "dec.resblocks.%d.convs1.0.weight", idx)) {
        for (int j = 0; T_opt(m, "dec.resblocks.%d.convs1.%d.weight", idx, j); j++) {
            const vo_tensor *w = T_req(m, "dec.resblocks.%d.convs1.%d.weight", idx, j);
            int K = (int)w->d[2];
            int d = (int)I_get(m, 1, "dec.resblocks.%d.convs1.%d.dilation", idx, j);
            int pad = (K * d - d) / 2; /* commons.py:17-18 */
            lrelu_copy(xt, x, n, 0.1f);
            snprintf(nm, sizeof nm, "dec.resblocks.%d.convs1.%d", idx, j);
            float *c1 = conv_named(m, nm, xt, B, T, d, pad, pad, 1, NULL);
            if (!c1) { free(x); free(xt); return NULL; }
            lrelu_copy(c1, c1, n, 0.1f);
            snprintf(nm, sizeof nm, "dec.resblocks.%d.convs2.%d", idx, j);
            const vo_tensor *w2 = T_req(m, "dec.resblocks.%d.convs2.%d.weight", idx, j);
            int K2 = (int)w2->d[2];
            float *c2 = conv_named(m, nm, c1, B, T, 1, (K2 - 1) / 2, (K2 - 1) / 2, 1, NULL);
            free(c1);
            if (!c2) { free(x); free(xt); return NULL; }
            for (int64_t i = 0; i < n; i++) x[i] = c2[i] + x[i];
            free(c2);
        }
    } else {
        for (int j = 0; T_opt(m, "dec.resblocks.%d.convs.%d.weight", idx, j); j++) {
            const vo_tensor *w = T_req(m, "dec.resblocks.%d.convs.%d.weight", idx, j);
            int K = (int)w->d[2];
            int d = (int)I_get(m, 1, "dec.resblocks.%d.convs.%d.dilation", idx, j);
            int pad = (K * d - d) / 2;
            lrelu_copy(xt, x, n, 0.1f);
            snprintf(nm, sizeof nm, "dec.resblocks.%d.convs.%d", idx, j);
            float *c1 = conv_named(m, nm, xt, B, T, d, pad, pad, 1, NULL);
            if (!c1) { free(x); free(xt); return NULL; }
            for (int64_t i = 0; i < n; i++) x[i] = c1[i] + x[i];
            free(c1);
        }
    }
    free(xt);
    return x;
}

/* models.py:348-368 Generator.forward; z is [B,C,F] (already masked by the caller) */
float *vo_generator(vo_model *m, const float *z, const float *gcond, int gin, int B, int C, int F, int *S_out) {
    g_missing = 0;
    int C0;
    float *x = conv_named(m, "dec.conv_pre", z, B, F, 1, 3, 3, 1, &C0);
    if (!x) return NULL;
    if (gcond) { /* x = x + self.cond(g) (:350-351) */
        const vo_tensor *cw = T_req(m, "dec.cond.weight"), *cb = T_req(m, "dec.cond.bias");
        if (g_missing) { free(x); return NULL; }
        for (int b = 0; b < B; b++)
            for (int c = 0; c < C0; c++) {
                float s = cb->data[c];
                for (int k = 0; k < gin; k++) s += cw->data[(int64_t)c * gin + k] * gcond[(int64_t)b * gin + k];
                for (int t = 0; t < F; t++) x[((int64_t)b * C0 + c) * F + t] += s;
            }
    }
    int nups = 0;
    while (T_opt(m, "dec.ups.%d.weight", nups)) nups++;
    int nrb = 0;
    while (T_opt(m, "dec.resblocks.%d.convs1.0.weight", nrb) || T_opt(m, "dec.resblocks.%d.convs.0.weight", nrb)) nrb++;
    int nk = nrb / nups; /* num_kernels (:313) */
    int T = F, Cc = C0;
    for (int i = 0; i < nups; i++) {
        const vo_tensor *w = T_req(m, "dec.ups.%d.weight", i); /* [Cin,Cout,K] */
        const vo_tensor *bs = T_opt(m, "dec.ups.%d.bias", i);
        int K = (int)w->d[2], Co = (int)w->d[1];
        int u = (int)I_get(m, -1, "dec.ups.%d.stride", i);
        if (u < 0) { vo_fail(m, "missing int dec.ups.%d.stride", i); free(x); return NULL; }
        int pad = (K - u) / 2; /* :329 */
        int64_t n = (int64_t)B * Cc * T;
        lrelu_copy(x, x, n, 0.1f); /* :354 */
        int To = (T - 1) * u - 2 * pad + K;
        float *y = falloc((int64_t)B * Co * To);
        vo_conv_transpose1d(x, B, Cc, T, w->data, bs ? bs->data : NULL, Co, K, u, pad, y);
        free(x);
        T = To;
        Cc = Co;
        n = (int64_t)B * Cc * T;
        float *xs = NULL;
        for (int j = 0; j < nk; j++) { /* :356-363 */
            float *r = resblock(m, i * nk + j, y, B, Cc, T);
            if (!r) { free(y); free(xs); return NULL; }
            if (!xs) xs = r;
            else {
                for (int64_t q = 0; q < n; q++) xs[q] += r[q];
                free(r);
            }
        }
        free(y);
        for (int64_t q = 0; q < n; q++) xs[q] = xs[q] / (float)nk;
        x = xs;
    }
    int64_t n = (int64_t)B * Cc * T;
    lrelu_copy(x, x, n, 0.01f); /* F.leaky_relu default slope (:364) */
    float *o = conv_named(m, "dec.conv_post", x, B, T, 1, 3, 3, 1, NULL);
    free(x);
    if (!o) return NULL;
    for (int64_t i = 0; i < (int64_t)B * T; i++) o[i] = tanhf(o[i]);
    *S_out = T;
    (void)C;
    return o;
}

/* ------------------------------------------------------------------ top level */

/* models.py:681-722 SynthesizerTrn.infer wrapped by export_onnx.py:250-278.
 * noise_dp [B,2,T] or NULL (zeros); noise_z [B,C,noise_z_stride] or NULL (zeros).
 * Results (vo_result): x, m_p, logs_p, logw, w_ceil, y_lengths, z_p, z, output. */
int vo_infer(vo_model *m, const int64_t *ids, const int64_t *lens, int B, int T, const float *scales,
             const int64_t *sid, const float *noise_dp, const float *noise_z, int64_t noise_z_stride) {
    vo_clear_results(m);
    g_missing = 0;
    m->err[0] = 0;
    if (B <= 0 || T <= 0) { vo_fail(m, "empty batch or sequence"); return -1; }
    for (int b = 0; b < B; b++)
        if (lens[b] < 0 || lens[b] > T) { vo_fail(m, "input_lengths[%d]=%lld outside [0,%d]", b, (long long)lens[b], T); return -1; }
    float noise_scale = scales[0], length_scale = scales[1], noise_w = scales[2];
    float *x, *m_p, *logs_p;
    int H, C;
    if (text_encoder(m, ids, lens, B, T, &x, &m_p, &logs_p, &H, &C)) return -1;
    /* g = emb_g(sid) (:692-696) */
    const vo_tensor *eg = T_opt(m, "emb_g.weight");
    float *g = NULL;
    int gin = 0;
    if (eg) {
        if (!sid) { vo_fail(m, "Missing speaker id"); free(x); free(m_p); free(logs_p); return -1; }
        gin = (int)eg->d[1];
        g = falloc((int64_t)B * gin);
        for (int b = 0; b < B; b++) {
            if (sid[b] < 0 || sid[b] >= eg->d[0]) { vo_fail(m, "sid out of range"); return -1; }
            memcpy(g + (int64_t)b * gin, eg->data + sid[b] * gin, sizeof(float) * gin);
        }
    }
    float *logw;
    if (T_opt(m, "dp.flows.0.m"))
        logw = sdp_reverse(m, x, g, gin, B, H, T, lens, noise_dp, noise_w);
    else
        logw = dp_plain(m, x, g, gin, B, H, T, lens);
    if (!logw) { free(x); free(m_p); free(logs_p); free(g); return -1; }
    int64_t dBT[2] = {B, T}, dB[1] = {B};
    int64_t dx[3] = {B, H, T}, dm[3] = {B, C, T};
    memcpy(put_result(m, "x", 3, dx), x, sizeof(float) * B * H * T);
    memcpy(put_result(m, "m_p", 3, dm), m_p, sizeof(float) * B * C * T);
    memcpy(put_result(m, "logs_p", 3, dm), logs_p, sizeof(float) * B * C * T);
    int64_t dlw[3] = {B, 1, T};
    memcpy(put_result(m, "logw", 3, dlw), logw, sizeof(float) * B * T);
    /* :702-704 */
    float *wc = put_result(m, "w_ceil", 2, dBT);
    float *yl = put_result(m, "y_lengths", 1, dB);
    int64_t *ylen = (int64_t *)calloc(B, sizeof(int64_t));
    int64_t F = 0;
    for (int b = 0; b < B; b++) {
        float sum = 0.f;
        for (int t = 0; t < T; t++) {
            float mk = t < lens[b] ? 1.f : 0.f;
            float w = expf(logw[(int64_t)b * T + t]) * mk * length_scale;
            float c = ceilf(w);
            wc[(int64_t)b * T + t] = c;
            sum += c;
        }
        if (sum < 1.f) sum = 1.f; /* clamp_min(.., 1) */
        ylen[b] = (int64_t)sum;
        yl[b] = (float)ylen[b];
        if (ylen[b] > F) F = ylen[b];
    }
    free(logw);
    free(x);
    if (noise_z && noise_z_stride < F) {
        vo_fail(m, "noise_z has %lld frames per row but %lld are needed", (long long)noise_z_stride, (long long)F);
        free(m_p); free(logs_p); free(g); free(ylen);
        return -1;
    }
    /* generate_path + expansion (commons.py:116-129, models.py:705-718): frame f of item b
     * belongs to token i iff cum[i-1] <= f < cum[i]; frames >= y_len or masked tokens get 0. */
    int64_t dz[3] = {B, C, F};
    float *z_p = put_result(m, "z_p", 3, dz);
    for (int b = 0; b < B; b++) {
        int *tok = (int *)malloc(sizeof(int) * (F ? F : 1));
        for (int64_t f = 0; f < F; f++) tok[f] = -1;
        float cum = 0.f;
        for (int t = 0; t < T; t++) {
            float prev = cum;
            cum += wc[(int64_t)b * T + t];
            if (t >= lens[b]) continue; /* attn_mask has x_mask */
            for (int64_t f = (int64_t)prev; f < (int64_t)cum && f < ylen[b]; f++) tok[f] = t;
        }
        for (int c = 0; c < C; c++)
            for (int64_t f = 0; f < F; f++) {
                float mp = 0.f, lp = 0.f;
                if (tok[f] >= 0) {
                    mp = m_p[((int64_t)b * C + c) * T + tok[f]];
                    lp = logs_p[((int64_t)b * C + c) * T + tok[f]];
                }
                float e = noise_z ? noise_z[((int64_t)b * C + c) * noise_z_stride + f] : 0.f;
                z_p[((int64_t)b * C + c) * F + f] = mp + e * expf(lp) * noise_scale; /* :718 */
            }
        free(tok);
    }
    free(m_p);
    free(logs_p);
    float *z = put_result(m, "z", 3, dz);
    memcpy(z, z_p, sizeof(float) * B * C * F);
    if (flow_reverse(m, z, g, gin, B, C, (int)F, ylen)) { free(g); free(ylen); return -1; }
    /* o = dec((z * y_mask), g) (:720) */
    float *zm = falloc((int64_t)B * C * F);
    memcpy(zm, z, sizeof(float) * B * C * F);
    mul_mask(zm, B, C, (int)F, ylen);
    int S;
    float *o = vo_generator(m, zm, g, gin, B, C, (int)F, &S);
    free(zm);
    free(g);
    free(ylen);
    if (!o) return -1;
    int64_t dout[4] = {B, 1, 1, S}; /* export_onnx.py:276 */
    memcpy(put_result(m, "output", 4, dout), o, sizeof(float) * B * S);
    free(o);
    return 0;
}

/* vocoder-only entry: z [B,C,F] (masked by caller), optional sid; result "output" */
int vo_vocoder(vo_model *m, const float *z, int B, int C, int F, const int64_t *sid) {
    vo_clear_results(m);
    m->err[0] = 0;
    const vo_tensor *eg = T_opt(m, "emb_g.weight");
    float *g = NULL;
    int gin = 0;
    if (eg && sid) {
        gin = (int)eg->d[1];
        g = falloc((int64_t)B * gin);
        for (int b = 0; b < B; b++) memcpy(g + (int64_t)b * gin, eg->data + sid[b] * gin, sizeof(float) * gin);
    }
    int S;
    float *o = vo_generator(m, z, g, gin, B, C, F, &S);
    free(g);
    if (!o) return -1;
    int64_t dout[4] = {B, 1, 1, S};
    memcpy(put_result(m, "output", 4, dout), o, sizeof(float) * B * S);
    free(o);
    return 0;
}

int vo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
