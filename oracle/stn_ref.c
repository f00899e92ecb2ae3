/* ORACLE — test infrastructure, never shipped, never on the product path.
 *
 * stn_ref.c: plain-C fp32 CPU restatement of the four neural stages that sit behind the
 * reference's `Ort::Session::Run` sites:
 *     duration predictor   /root/reference/cpp/helper.cpp:512-526
 *     text encoder         /root/reference/cpp/helper.cpp:545-556
 *     vector estimator     /root/reference/cpp/helper.cpp:590-659   (one Euler step per call)
 *     vocoder              /root/reference/cpp/helper.cpp:662-679
 * Tensor names, dtypes, layouts and the step-counter convention (float [B], 0-based
 * current_step, output fed back as the next noisy_latent) follow those call sites.
 *
 * PARITY UNPINNED (neural part): the algorithm itself lives in four ONNX graphs executed
 * by ONNX Runtime (C++ install unversioned, cpp/CMakeLists.txt:22-51; Python pin
 * onnxruntime==1.23.1, py/requirements.txt:1).  Neither the graphs (Hugging Face
 * Supertone/supertonic-2, README.md:97-105) nor ONNX Runtime exist in /root/reference
 * or in this image, and the reference holds no golden vectors.  The layer
 * stack below restates the published architecture (arXiv:2503.23108; LARoPE
 * arXiv:2509.11084) through include/stn_arch.h on deterministic synthetic weights.
 * It is the checker for the HIP engine, not a claim of equality with ORT outputs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Activation layout inside: row-major [rows = b*len + t][channels].
 */
#define _GNU_SOURCE
#include "stn_ref.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* deterministic synthetic weights                                                       */
/* ------------------------------------------------------------------------------------ */

enum { K_W = 0, K_BIAS = 1, K_LN_G = 2, K_LN_B = 3, K_LSCALE = 4, K_EMB = 5 };

typedef struct {
    char name[64];
    int64_t n;       /* elements */
    int rows, cols;  /* for 2-D [rows][cols] */
    float* w;        /* canonical layout */
    float* wt;       /* transposed [cols][rows] for K_W matrices, else NULL */
} tensor;

struct stnref_model {
    stn_arch a;
    uint64_t seed;
    tensor* t;
    int nt, cap;
    int64_t params;
};

static uint64_t fnv1a(const char* s) {
    uint64_t h = 1469598103934665603ULL;
    for (; *s; ++s) { h ^= (unsigned char)*s; h *= 1099511628211ULL; }
    return h;
}
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* u in [-1,1): 24 random bits, exactly representable in fp32 */
static float unit(uint64_t seed, uint64_t tid, uint64_t idx) {
    uint64_t h = mix64(mix64(seed ^ tid) + idx);
    return (float)(h >> 40) * (1.0f / 8388608.0f) - 1.0f;
}

static tensor* declare(stnref_model* m, int kind, int rows, int cols, float gain, const char* fmt, ...) {
    if (m->nt == m->cap) { m->cap = m->cap ? 2 * m->cap : 256; m->t = realloc(m->t, sizeof(tensor) * m->cap); }
    tensor* t = &m->t[m->nt++];
    va_list ap; va_start(ap, fmt); vsnprintf(t->name, sizeof t->name, fmt, ap); va_end(ap);
    t->rows = rows; t->cols = cols; t->n = (int64_t)rows * cols;
    t->w = malloc(sizeof(float) * t->n); t->wt = NULL;
    uint64_t tid = fnv1a(t->name);
    float scale, offs = 0.f;
    switch (kind) {
        case K_W:      scale = sqrtf(3.0f / (float)cols) * gain; break;  /* cols = fan-in */
        case K_BIAS:   scale = 0.05f; break;
        case K_LN_G:   scale = 0.1f; offs = 1.0f; break;
        case K_LN_B:   scale = 0.05f; break;
        case K_LSCALE: scale = 0.1f; offs = 0.2f; break;
        default:       scale = sqrtf(3.0f); break; /* K_EMB: unit variance */
    }
    for (int64_t i = 0; i < t->n; ++i) t->w[i] = offs + scale * unit(m->seed, tid, (uint64_t)i);
    if (kind == K_W && rows > 1) {
        t->wt = malloc(sizeof(float) * t->n);
        for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) t->wt[(int64_t)c * rows + r] = t->w[(int64_t)r * cols + c];
    }
    m->params += t->n;
    return t;
}

static const tensor* get(const stnref_model* m, const char* fmt, ...) {
    char name[64];
    va_list ap; va_start(ap, fmt); vsnprintf(name, sizeof name, fmt, ap); va_end(ap);
    for (int i = 0; i < m->nt; ++i) if (!strcmp(m->t[i].name, name)) return &m->t[i];
    fprintf(stderr, "stn_ref: unknown tensor %s\n", name); abort();
}

static void declare_linear(stnref_model* m, const char* p, int out, int in, float gain) {
    declare(m, K_W, out, in, gain, "%s.w", p);
    declare(m, K_BIAS, 1, out, 1.f, "%s.b", p);
}
static void declare_ln(stnref_model* m, const char* p, int c) {
    declare(m, K_LN_G, 1, c, 1.f, "%s.g", p);
    declare(m, K_LN_B, 1, c, 1.f, "%s.b", p);
}
static void declare_convnext(stnref_model* m, const char* p, int c, int hid, int k) {
    char q[64];
    declare(m, K_W, c, k, 1.f, "%s.dw.w", p);   /* depthwise [C][k], fan-in k */
    declare(m, K_BIAS, 1, c, 1.f, "%s.dw.b", p);
    snprintf(q, sizeof q, "%s.ln", p);  declare_ln(m, q, c);
    snprintf(q, sizeof q, "%s.pw1", p); declare_linear(m, q, hid, c, 1.f);
    snprintf(q, sizeof q, "%s.pw2", p); declare_linear(m, q, c, hid, 1.f);
    declare(m, K_LSCALE, 1, c, 1.f, "%s.gamma", p);
}
static void declare_attn(stnref_model* m, const char* p, int c, int cctx) {
    char q[64];
    snprintf(q, sizeof q, "%s.ln", p); declare_ln(m, q, c);
    snprintf(q, sizeof q, "%s.q", p);  declare_linear(m, q, c, c, 1.f);
    snprintf(q, sizeof q, "%s.k", p);  declare_linear(m, q, c, cctx, 1.f);
    snprintf(q, sizeof q, "%s.v", p);  declare_linear(m, q, c, cctx, 1.f);
    snprintf(q, sizeof q, "%s.o", p);  declare_linear(m, q, c, c, 1.f);
}

static void declare_all(stnref_model* m) {
    const stn_arch* a = &m->a; char p[64];
    /* duration predictor */
    declare(m, K_EMB, a->vocab_size, a->dp_dim, 1.f, "dp.emb");
    for (int i = 0; i < a->dp_conv_blocks; ++i) { snprintf(p, sizeof p, "dp.conv%d", i); declare_convnext(m, p, a->dp_dim, a->dp_hidden, a->dp_kernel); }
    declare_attn(m, "dp.st", a->dp_dim, a->d_style_dp);
    declare_ln(m, "dp.out_ln", a->dp_dim);
    declare_linear(m, "dp.fc1", a->dp_dim, a->dp_dim, 1.f);
    declare_linear(m, "dp.fc2", 1, a->dp_dim, 1.f);
    /* text encoder */
    declare(m, K_EMB, a->vocab_size, a->te_dim, 1.f, "te.emb");
    for (int i = 0; i < a->te_conv_blocks; ++i) { snprintf(p, sizeof p, "te.conv%d", i); declare_convnext(m, p, a->te_dim, a->te_hidden, a->te_kernel); }
    for (int i = 0; i < a->te_attn_blocks; ++i) {
        snprintf(p, sizeof p, "te.sa%d", i); declare_attn(m, p, a->te_dim, a->te_dim);
        snprintf(p, sizeof p, "te.sa%d.ffn_ln", i); declare_ln(m, p, a->te_dim);
        snprintf(p, sizeof p, "te.sa%d.ffn1", i); declare_linear(m, p, a->te_ffn, a->te_dim, 1.f);
        snprintf(p, sizeof p, "te.sa%d.ffn2", i); declare_linear(m, p, a->te_dim, a->te_ffn, 1.f);
    }
    for (int i = 0; i < a->te_style_blocks; ++i) { snprintf(p, sizeof p, "te.st%d", i); declare_attn(m, p, a->te_dim, a->d_style_ttl); }
    declare_ln(m, "te.out_ln", a->te_dim);
    declare_linear(m, "te.proj", a->te_out_dim, a->te_dim, 1.f);
    /* vector estimator */
    int D = a->latent_dim * a->chunk_compress_factor;
    declare_linear(m, "ve.in", a->ve_dim, D, 1.f);
    declare_linear(m, "ve.t1", a->ve_dim, a->ve_time_dim, 1.f);
    declare_linear(m, "ve.t2", a->ve_dim, a->ve_dim, 1.f);
    for (int b = 0; b < a->ve_main_blocks; ++b) {
        for (int j = 0; j < a->ve_dilated; ++j) { snprintf(p, sizeof p, "ve.m%d.dil%d", b, j); declare_convnext(m, p, a->ve_dim, a->ve_hidden, a->ve_kernel); }
        snprintf(p, sizeof p, "ve.m%d.time", b); declare_linear(m, p, a->ve_dim, a->ve_dim, 1.f);
        snprintf(p, sizeof p, "ve.m%d.cn_a", b); declare_convnext(m, p, a->ve_dim, a->ve_hidden, a->ve_kernel);
        snprintf(p, sizeof p, "ve.m%d.text", b); declare_attn(m, p, a->ve_dim, a->te_out_dim);
        snprintf(p, sizeof p, "ve.m%d.cn_b", b); declare_convnext(m, p, a->ve_dim, a->ve_hidden, a->ve_kernel);
        snprintf(p, sizeof p, "ve.m%d.style", b); declare_attn(m, p, a->ve_dim, a->d_style_ttl);
    }
    for (int j = 0; j < a->ve_tail_blocks; ++j) { snprintf(p, sizeof p, "ve.tail%d", j); declare_convnext(m, p, a->ve_dim, a->ve_hidden, a->ve_kernel); }
    declare_ln(m, "ve.out_ln", a->ve_dim);
    declare_linear(m, "ve.out", D, a->ve_dim, 1.f);
    /* vocoder */
    declare(m, K_W, a->vo_dim, a->latent_dim * a->vo_in_kernel, 1.f, "vo.in.w"); /* [Cout][Cin][k] */
    declare(m, K_BIAS, 1, a->vo_dim, 1.f, "vo.in.b");
    for (int i = 0; i < a->vo_blocks; ++i) { snprintf(p, sizeof p, "vo.blk%d", i); declare_convnext(m, p, a->vo_dim, a->vo_hidden, a->vo_kernel); }
    declare_ln(m, "vo.out_ln", a->vo_dim);
    declare_linear(m, "vo.head", a->base_chunk_size, a->vo_dim, a->head_gain);
}

/* bound the OpenMP team: a container's CPU quota is usually far below the host's core count */
void stnref_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int stnref_get_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

stnref_model* stnref_create(const stn_arch* a, uint64_t seed) {
    stnref_model* m = calloc(1, sizeof *m);
    m->a = *a; m->seed = seed;
    declare_all(m);
    return m;
}
void stnref_destroy(stnref_model* m) {
    if (!m) return;
    for (int i = 0; i < m->nt; ++i) { free(m->t[i].w); free(m->t[i].wt); }
    free(m->t); free(m);
}
int64_t stnref_param_count(const stnref_model* m) { return m->params; }
int stnref_num_tensors(const stnref_model* m) { return m->nt; }
const char* stnref_tensor_name(const stnref_model* m, int i) { return m->t[i].name; }
int64_t stnref_tensor(const stnref_model* m, const char* name, float* out, int64_t cap) {
    for (int i = 0; i < m->nt; ++i) if (!strcmp(m->t[i].name, name)) {
        if (out) memcpy(out, m->t[i].w, sizeof(float) * (size_t)(m->t[i].n < cap ? m->t[i].n : cap));
        return m->t[i].n;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------ */
/* primitive ops                                                                         */
/* ------------------------------------------------------------------------------------ */

static float* falloc(int64_t n) { float* p = malloc(sizeof(float) * (size_t)(n > 0 ? n : 1)); if (!p) abort(); return p; }

/* GELU: the erf form (torch.nn.GELU()) or, when the graphs being checked spell it with Tanh (stnref_set_gelu_tanh), the tanh
 * approximation 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))).  Process-wide switch of this test library. */
static int g_gelu_tanh = 0;
void stnref_set_gelu_tanh(int on) { g_gelu_tanh = on != 0; }
static float gelu(float x) {
    if (g_gelu_tanh) return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
static float silu(float x) { return x / (1.0f + expf(-x)); }

/* Y[M][N] = X[M][K] . W[N][K]^T + b   (k-ordered fp32 accumulation) */
void stnref_linear(const float* X, int64_t M, int K, const float* Wt /*[K][N]*/, const float* bias, int N, float* Y) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < M; ++r) {
        float* y = Y + r * N; const float* x = X + r * K;
        if (bias) memcpy(y, bias, sizeof(float) * N); else memset(y, 0, sizeof(float) * N);
        for (int k = 0; k < K; ++k) {
            const float xv = x[k]; const float* w = Wt + (int64_t)k * N;
            for (int n = 0; n < N; ++n) y[n] += xv * w[n];
        }
    }
}
static void linear_t(const stnref_model* m, const char* p, const float* X, int64_t M, float* Y) {
    const tensor* w = get(m, "%s.w", p); const tensor* b = get(m, "%s.b", p);
    stnref_linear(X, M, w->cols, w->wt ? w->wt : w->w /* 1-row: same */, b->w, w->rows, Y);
}

void stnref_layernorm(const float* X, int64_t M, int C, const float* g, const float* b, float eps, float* Y) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < M; ++r) {
        const float* x = X + r * C; float* y = Y + r * C;
        double s = 0; for (int c = 0; c < C; ++c) s += x[c];
        float mean = (float)(s / C);
        double v = 0; for (int c = 0; c < C; ++c) { float d = x[c] - mean; v += (double)d * d; }
        float rstd = 1.0f / sqrtf((float)(v / C) + eps);
        for (int c = 0; c < C; ++c) y[c] = (x[c] - mean) * rstd * g[c] + b[c];
    }
}
static void ln_t(const stnref_model* m, const char* p, const float* X, int64_t M, int C, float* Y) {
    stnref_layernorm(X, M, C, get(m, "%s.g", p)->w, get(m, "%s.b", p)->w, m->a.ln_eps, Y);
}

/* depthwise 'same' conv along t inside each of B sequences of length L; zero outside [0,L) */
void stnref_dwconv(const float* X, int B, int L, int C, const float* w /*[C][k]*/, const float* bias, int k, int dil, float* Y) {
    const int half = (k - 1) / 2;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)B * L; ++r) {
        int t = (int)(r % L); int64_t base = r - t;
        float* y = Y + r * C;
        for (int c = 0; c < C; ++c) y[c] = bias[c];
        for (int j = 0; j < k; ++j) {
            int tt = t + (j - half) * dil;
            if (tt < 0 || tt >= L) continue;
            const float* x = X + (base + tt) * C;
            for (int c = 0; c < C; ++c) y[c] += w[c * k + j] * x[c];
        }
    }
}

static void mask_rows(float* X, int B, int L, int C, const int* len) {
    if (!len) return;
    for (int b = 0; b < B; ++b) for (int t = len[b]; t < L; ++t) memset(X + ((int64_t)b * L + t) * C, 0, sizeof(float) * C);
}

/* ConvNeXt block, in place on x[B*L][C]; len==NULL -> unmasked */
static void convnext(const stnref_model* m, const char* p, float* x, int B, int L, int C, int hid, int k, int dil, const int* len) {
    int64_t M = (int64_t)B * L;
    float* h = falloc(M * C); float* u = falloc(M * hid); float* v = falloc(M * C);
    stnref_dwconv(x, B, L, C, get(m, "%s.dw.w", p)->w, get(m, "%s.dw.b", p)->w, k, dil, h);
    char q[64]; snprintf(q, sizeof q, "%s.ln", p); ln_t(m, q, h, M, C, h);
    snprintf(q, sizeof q, "%s.pw1", p); linear_t(m, q, h, M, u);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M * hid; ++i) u[i] = gelu(u[i]);
    snprintf(q, sizeof q, "%s.pw2", p); linear_t(m, q, u, M, v);
    const float* g = get(m, "%s.gamma", p)->w;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < M; ++r) for (int c = 0; c < C; ++c) x[r * C + c] += g[c] * v[r * C + c];
    mask_rows(x, B, L, C, len);
    free(h); free(u); free(v);
}

/* rotary, half-split pairing (x[i], x[i+dh/2]); angle = pos * base^(-2i/dh) */
static void rope_rows(float* X, int B, int L, int C, int H, const int* len, int mode, float base, float gamma) {
    int dh = C / H, hd2 = dh / 2;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)B * L; ++r) {
        int b = (int)(r / L), t = (int)(r % L);
        float pos = (mode == 1) ? gamma * (float)t / (float)(len[b] > 0 ? len[b] : 1) : (float)t;
        for (int h = 0; h < H; ++h) {
            float* x = X + r * C + h * dh;
            for (int i = 0; i < hd2; ++i) {
                float inv = expf(-logf(base) * (float)(2 * i) / (float)dh);
                float ang = pos * inv, c = cosf(ang), s = sinf(ang);
                float a0 = x[i], a1 = x[i + hd2];
                x[i] = a0 * c - a1 * s; x[i + hd2] = a1 * c + a0 * s;
            }
        }
    }
}

/* softmax(q k^T / sqrt(dh)) v per (b, head); keys j >= klen[b] are excluded */
void stnref_attention_core(const float* Q, const float* Kx, const float* V, int B, int Lq, int Lk, int C, int H,
                           const int* klen, float* O) {
    int dh = C / H; float sc = 1.0f / sqrtf((float)dh);
#pragma omp parallel for schedule(static) collapse(2)
    for (int b = 0; b < B; ++b) for (int t = 0; t < Lq; ++t) {
        int nk = klen ? klen[b] : Lk;
        float* s = malloc(sizeof(float) * (size_t)(Lk > 0 ? Lk : 1));
        for (int h = 0; h < H; ++h) {
            const float* q = Q + ((int64_t)b * Lq + t) * C + h * dh;
            float mx = -INFINITY;
            for (int j = 0; j < nk; ++j) {
                const float* kk = Kx + ((int64_t)b * Lk + j) * C + h * dh;
                float d = 0; for (int i = 0; i < dh; ++i) d += q[i] * kk[i];
                s[j] = d * sc; if (s[j] > mx) mx = s[j];
            }
            float sum = 0; for (int j = 0; j < nk; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
            float* o = O + ((int64_t)b * Lq + t) * C + h * dh;
            for (int i = 0; i < dh; ++i) o[i] = 0;
            float inv = nk > 0 ? 1.0f / sum : 0.f;
            for (int j = 0; j < nk; ++j) {
                const float* vv = V + ((int64_t)b * Lk + j) * C + h * dh; float pj = s[j] * inv;
                for (int i = 0; i < dh; ++i) o[i] += pj * vv[i];
            }
        }
        free(s);
    }
}

/* x += Wo . attn(LN(x) Wq, ctx Wk, ctx Wv); then mask.  rope_mode: -1 none, 0 plain, 1 length-aware */
static void attn_block(const stnref_model* m, const char* p, float* x, int B, int Lq, int C, int H,
                       const float* ctx, int Lk, int Cc, const int* qlen, const int* klen, int rope_mode, int self) {
    int64_t Mq = (int64_t)B * Lq, Mk = (int64_t)B * Lk;
    float* xn = falloc(Mq * C); float* q = falloc(Mq * C); float* k = falloc(Mk * C); float* v = falloc(Mk * C);
    float* o = falloc(Mq * C); float* y = falloc(Mq * C);
    char n[64]; snprintf(n, sizeof n, "%s.ln", p); ln_t(m, n, x, Mq, C, xn);
    if (self) ctx = xn;
    snprintf(n, sizeof n, "%s.q", p); linear_t(m, n, xn, Mq, q);
    snprintf(n, sizeof n, "%s.k", p); linear_t(m, n, ctx, Mk, k);
    snprintf(n, sizeof n, "%s.v", p); linear_t(m, n, ctx, Mk, v);
    (void)Cc;
    if (rope_mode >= 0) {
        rope_rows(q, B, Lq, C, H, qlen, rope_mode, m->a.rope_base, m->a.larope_gamma);
        rope_rows(k, B, Lk, C, H, klen, rope_mode, m->a.rope_base, m->a.larope_gamma);
    }
    stnref_attention_core(q, k, v, B, Lq, Lk, C, H, klen, o);
    snprintf(n, sizeof n, "%s.o", p); linear_t(m, n, o, Mq, y);
    for (int64_t i = 0; i < Mq * C; ++i) x[i] += y[i];
    mask_rows(x, B, Lq, C, qlen);
    free(xn); free(q); free(k); free(v); free(o); free(y);
}

/* prefix-mask [B,1,L] -> lengths (count of entries > 0.5) */
static int* mask_to_len(const float* mask, int B, int L) {
    int* len = malloc(sizeof(int) * (size_t)B);
    for (int b = 0; b < B; ++b) { int n = 0; for (int t = 0; t < L; ++t) n += mask[(int64_t)b * L + t] > 0.5f; len[b] = n; }
    return len;
}

static void embed(const stnref_model* m, const char* name, const int64_t* ids, int B, int L, int C, const int* len, float* x) {
    const tensor* e = get(m, "%s", name);
    for (int b = 0; b < B; ++b) for (int t = 0; t < L; ++t) {
        float* y = x + ((int64_t)b * L + t) * C; int64_t id = ids[(int64_t)b * L + t];
        if (t < len[b] && id >= 0 && id < e->rows) memcpy(y, e->w + id * C, sizeof(float) * C);
        else memset(y, 0, sizeof(float) * C);  /* padded position or out-of-vocabulary id -> zero row */
    }
}

/* ------------------------------------------------------------------------------------ */
/* the four stages                                                                        */
/* ------------------------------------------------------------------------------------ */

void stnref_duration(const stnref_model* m, int B, int Lt, const int64_t* text_ids, const float* style_dp,
                     const float* text_mask, float* duration) {
    const stn_arch* a = &m->a; int C = a->dp_dim; int64_t M = (int64_t)B * Lt; char p[64];
    int* len = mask_to_len(text_mask, B, Lt);
    float* x = falloc(M * C);
    embed(m, "dp.emb", text_ids, B, Lt, C, len, x);
    for (int i = 0; i < a->dp_conv_blocks; ++i) { snprintf(p, sizeof p, "dp.conv%d", i); convnext(m, p, x, B, Lt, C, a->dp_hidden, a->dp_kernel, 1, len); }
    attn_block(m, "dp.st", x, B, Lt, C, a->dp_heads, style_dp, a->n_style_dp, a->d_style_dp, len, NULL, -1, 0);
    float* xn = falloc(M * C); ln_t(m, "dp.out_ln", x, M, C, xn);
    float* pooled = falloc((int64_t)B * C); float* h = falloc((int64_t)B * C);
    for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) {
        float s = 0; for (int t = 0; t < len[b]; ++t) s += xn[((int64_t)b * Lt + t) * C + c];
        pooled[b * C + c] = s / (float)(len[b] > 0 ? len[b] : 1);
    }
    linear_t(m, "dp.fc1", pooled, B, h);
    for (int i = 0; i < B * C; ++i) h[i] = gelu(h[i]);
    linear_t(m, "dp.fc2", h, B, duration);
    for (int b = 0; b < B; ++b) { float y = duration[b]; duration[b] = y > 20.f ? y : log1pf(expf(y)); }
    free(x); free(xn); free(pooled); free(h); free(len);
}

void stnref_text_enc(const stnref_model* m, int B, int Lt, const int64_t* text_ids, const float* style_ttl,
                     const float* text_mask, float* text_emb /* [B, Ce, Lt] */) {
    const stn_arch* a = &m->a; int C = a->te_dim; int64_t M = (int64_t)B * Lt; char p[64], q[64];
    int* len = mask_to_len(text_mask, B, Lt);
    float* x = falloc(M * C);
    embed(m, "te.emb", text_ids, B, Lt, C, len, x);
    for (int i = 0; i < a->te_conv_blocks; ++i) { snprintf(p, sizeof p, "te.conv%d", i); convnext(m, p, x, B, Lt, C, a->te_hidden, a->te_kernel, 1, len); }
    for (int i = 0; i < a->te_attn_blocks; ++i) {
        snprintf(p, sizeof p, "te.sa%d", i);
        attn_block(m, p, x, B, Lt, C, a->te_heads, NULL, Lt, C, len, len, 0, 1);
        float* xn = falloc(M * C); float* u = falloc(M * a->te_ffn); float* v = falloc(M * C);
        snprintf(q, sizeof q, "%s.ffn_ln", p); ln_t(m, q, x, M, C, xn);
        snprintf(q, sizeof q, "%s.ffn1", p); linear_t(m, q, xn, M, u);
        for (int64_t j = 0; j < M * a->te_ffn; ++j) u[j] = gelu(u[j]);
        snprintf(q, sizeof q, "%s.ffn2", p); linear_t(m, q, u, M, v);
        for (int64_t j = 0; j < M * C; ++j) x[j] += v[j];
        mask_rows(x, B, Lt, C, len);
        free(xn); free(u); free(v);
    }
    for (int i = 0; i < a->te_style_blocks; ++i) {
        snprintf(p, sizeof p, "te.st%d", i);
        attn_block(m, p, x, B, Lt, C, a->te_heads, style_ttl, a->n_style_ttl, a->d_style_ttl, len, NULL, -1, 0);
    }
    int Ce = a->te_out_dim; float* xn = falloc(M * C); float* y = falloc(M * Ce);
    ln_t(m, "te.out_ln", x, M, C, xn);
    linear_t(m, "te.proj", xn, M, y);
    mask_rows(y, B, Lt, Ce, len);
    for (int b = 0; b < B; ++b) for (int c = 0; c < Ce; ++c) for (int t = 0; t < Lt; ++t)
        text_emb[((int64_t)b * Ce + c) * Lt + t] = y[((int64_t)b * Lt + t) * Ce + c];
    free(x); free(xn); free(y); free(len);
}

void stnref_vector_est(const stnref_model* m, int B, int L, int Lt, const float* noisy_latent, const float* text_emb,
                       const float* style_ttl, const float* text_mask, const float* latent_mask,
                       const float* total_step, const float* current_step, float* denoised /* [B, D, L] */) {
    const stn_arch* a = &m->a; int C = a->ve_dim, D = a->latent_dim * a->chunk_compress_factor, Ce = a->te_out_dim;
    int64_t M = (int64_t)B * L; char p[64];
    int* llen = mask_to_len(latent_mask, B, L); int* tlen = mask_to_len(text_mask, B, Lt);
    /* [B,D,L] -> rows [B*L][D];  [B,Ce,Lt] -> [B*Lt][Ce] */
    float* z = falloc(M * D);
    for (int b = 0; b < B; ++b) for (int d = 0; d < D; ++d) for (int t = 0; t < L; ++t)
        z[((int64_t)b * L + t) * D + d] = noisy_latent[((int64_t)b * D + d) * L + t];
    float* ctx = falloc((int64_t)B * Lt * Ce);
    for (int b = 0; b < B; ++b) for (int c = 0; c < Ce; ++c) for (int t = 0; t < Lt; ++t)
        ctx[((int64_t)b * Lt + t) * Ce + c] = text_emb[((int64_t)b * Ce + c) * Lt + t];
    float* x = falloc(M * C);
    linear_t(m, "ve.in", z, M, x);
    mask_rows(x, B, L, C, llen);
    /* time conditioning: sinusoid(t * time_scale) -> Linear -> SiLU -> Linear */
    int Td = a->ve_time_dim, hd = Td / 2;
    float* te = falloc((int64_t)B * Td); float* t1 = falloc((int64_t)B * C); float* tc = falloc((int64_t)B * C); float* tb = falloc((int64_t)B * C);
    for (int b = 0; b < B; ++b) {
        float t = current_step[b] / total_step[b] * a->time_scale;
        for (int i = 0; i < hd; ++i) {
            float f = expf(-logf(10000.0f) * (float)i / (float)hd);
            te[b * Td + i] = sinf(t * f); te[b * Td + hd + i] = cosf(t * f);
        }
    }
    linear_t(m, "ve.t1", te, B, t1);
    for (int i = 0; i < B * C; ++i) t1[i] = silu(t1[i]);
    linear_t(m, "ve.t2", t1, B, tc);
    for (int blk = 0; blk < a->ve_main_blocks; ++blk) {
        for (int j = 0; j < a->ve_dilated; ++j) { snprintf(p, sizeof p, "ve.m%d.dil%d", blk, j); convnext(m, p, x, B, L, C, a->ve_hidden, a->ve_kernel, 1 << j, llen); }
        snprintf(p, sizeof p, "ve.m%d.time", blk); linear_t(m, p, tc, B, tb);
        for (int b = 0; b < B; ++b) for (int t = 0; t < llen[b]; ++t) for (int c = 0; c < C; ++c) x[((int64_t)b * L + t) * C + c] += tb[b * C + c];
        snprintf(p, sizeof p, "ve.m%d.cn_a", blk); convnext(m, p, x, B, L, C, a->ve_hidden, a->ve_kernel, 1, llen);
        snprintf(p, sizeof p, "ve.m%d.text", blk); attn_block(m, p, x, B, L, C, a->ve_heads, ctx, Lt, Ce, llen, tlen, 1, 0);
        snprintf(p, sizeof p, "ve.m%d.cn_b", blk); convnext(m, p, x, B, L, C, a->ve_hidden, a->ve_kernel, 1, llen);
        snprintf(p, sizeof p, "ve.m%d.style", blk); attn_block(m, p, x, B, L, C, a->ve_heads, style_ttl, a->n_style_ttl, a->d_style_ttl, llen, NULL, -1, 0);
    }
    for (int j = 0; j < a->ve_tail_blocks; ++j) { snprintf(p, sizeof p, "ve.tail%d", j); convnext(m, p, x, B, L, C, a->ve_hidden, a->ve_kernel, 1, llen); }
    float* xn = falloc(M * C); float* v = falloc(M * D);
    ln_t(m, "ve.out_ln", x, M, C, xn);
    linear_t(m, "ve.out", xn, M, v);
    /* Euler step inside the graph (cpp/helper.cpp:643-658 feeds the output straight back) */
    for (int b = 0; b < B; ++b) for (int d = 0; d < D; ++d) for (int t = 0; t < L; ++t) {
        int64_t o = ((int64_t)b * D + d) * L + t;
        float dt = 1.0f / total_step[b];
        denoised[o] = t < llen[b] ? noisy_latent[o] + v[((int64_t)b * L + t) * D + d] * dt : 0.0f;
    }
    free(z); free(ctx); free(x); free(te); free(t1); free(tc); free(tb); free(xn); free(v); free(llen); free(tlen);
}

void stnref_vocoder(const stnref_model* m, int B, int L, const float* latent /* [B,D,L] */, float* wav /* [B, L*cs] */) {
    const stn_arch* a = &m->a; int ld = a->latent_dim, ccf = a->chunk_compress_factor, D = ld * ccf, C = a->vo_dim;
    int T = L * ccf; int64_t M = (int64_t)B * T; char p[64];
    /* un-compress: frame t = l*ccf + j takes channels [j*ld, (j+1)*ld) of compressed frame l */
    float* z = falloc(M * ld);
    for (int b = 0; b < B; ++b) for (int l = 0; l < L; ++l) for (int j = 0; j < ccf; ++j) for (int c = 0; c < ld; ++c)
        z[((int64_t)b * T + l * ccf + j) * ld + c] = latent[((int64_t)b * D + j * ld + c) * L + l];
    /* input conv ld -> C, kernel vo_in_kernel, 'same' zero padding */
    float* x = falloc(M * C);
    const float* w = get(m, "vo.in.w")->w; const float* bi = get(m, "vo.in.b")->w; int k = a->vo_in_kernel, half = (k - 1) / 2;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < M; ++r) {
        int t = (int)(r % T); int64_t base = r - t;
        for (int co = 0; co < C; ++co) {
            float acc = bi[co];
            for (int ci = 0; ci < ld; ++ci) for (int j = 0; j < k; ++j) {
                int tt = t + j - half; if (tt < 0 || tt >= T) continue;
                acc += w[((int64_t)co * ld + ci) * k + j] * z[(base + tt) * ld + ci];
            }
            x[r * C + co] = acc;
        }
    }
    for (int i = 0; i < a->vo_blocks; ++i) { snprintf(p, sizeof p, "vo.blk%d", i); convnext(m, p, x, B, T, C, a->vo_hidden, a->vo_kernel, a->vo_dilations[i], NULL); }
    float* xn = falloc(M * C);
    ln_t(m, "vo.out_ln", x, M, C, xn);
    /* head = transposed conv with kernel = stride = base_chunk_size == per-frame linear C -> bcs */
    linear_t(m, "vo.head", xn, M, wav);
    free(z); free(x); free(xn);
}

/* ------------------------------------------------------------------------------------ */
/* Philox4x32-10 + Box-Muller noise: element (utt, d, t) depends only on (seed, utt, d, t) */
/* ------------------------------------------------------------------------------------ */

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
void stnref_randn(uint64_t seed, int B, int D, int L, const int64_t* utt_ids, float* out /* [B,D,L] */) {
    for (int b = 0; b < B; ++b) for (int d = 0; d < D; ++d) for (int t4 = 0; t4 < (L + 3) / 4; ++t4) {
        uint64_t u = utt_ids ? (uint64_t)utt_ids[b] : (uint64_t)b;
        uint32_t c[4] = {(uint32_t)t4, (uint32_t)d, (uint32_t)u, (uint32_t)(u >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        float n[4];
        for (int h = 0; h < 2; ++h) {
            float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            float rr = sqrtf(-2.0f * logf(u1)), th = 6.28318530717958647692f * u2;
            n[2 * h] = rr * cosf(th); n[2 * h + 1] = rr * sinf(th);
        }
        for (int i = 0; i < 4 && t4 * 4 + i < L; ++i) out[((int64_t)b * D + d) * L + t4 * 4 + i] = n[i];
    }
}
