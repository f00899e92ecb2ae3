// engine_ops.cpp — single-kernel, phase-stamp and timing entry points (stn_op_*): what the op-level tests and the tools under tools/ call.
#include "engine.hpp"
#include "engine_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace stn {
using detail::up;

// =================================================================================================
// op-level test entry points
// =================================================================================================
void Engine::op_gemm(int dtype, int M, int N, int K, const float* A, const float* W, const float* bias, int act, float* out) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    float* dA = up(ar_, s_, A, (size_t)M * K);
    float* dW = up(ar_, s_, W, (size_t)N * K);
    float* dB = bias ? up(ar_, s_, bias, (size_t)N) : nullptr;
    float* dO = f32_alloc((size_t)M * N);
    const void* pa = dA;
    const void* pw = dW;
    if (is_half(dtype)) {
        void* a16 = ar_.alloc((size_t)M * K * 2);
        void* w16 = ar_.alloc((size_t)N * K * 2);
        launch_cast(s_, dtype, dA, (int64_t)M * K, a16);
        launch_cast(s_, dtype, dW, (int64_t)N * K, w16);
        pa = a16; pw = w16;
    }
    Epilogue e; e.mode = EPI_STORE; e.act = act; e.out_dtype = F32; e.out = dO; e.ldo = N; e.bias = dB;
    const int sk = gemm_splitk_factor(dtype, M, N, K, e);  // the same decision the model path takes (Engine::gemm)
    if (sk > 1) launch_gemm_splitk(s_, dtype, pa, K, pw, K, M, N, K, e, sk, f32_alloc((int64_t)sk * M * N));
    else launch_gemm(s_, dtype, pa, K, pw, K, M, N, K, e);
    STN_HIP(hipMemcpyAsync(out, dO, (size_t)M * N * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::op_dwconv_ln(int dtype, int B, int L, int C, int k, int dil, const float* x, const float* w, const float* bias,
                          const float* g, const float* b, float* y, const int* seqlen) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    const size_t n = (size_t)B * L * C;
    const int* dlen = seqlen ? up(ar_, s_, seqlen, (size_t)B) : nullptr;
    std::vector<float> wt((size_t)C * k);
    for (int c = 0; c < C; ++c) for (int j = 0; j < k; ++j) wt[(size_t)j * C + c] = w[(size_t)c * k + j];
    float* dx = up(ar_, s_, x, n);
    float* dw = up(ar_, s_, wt.data(), wt.size());
    float* db = up(ar_, s_, bias, (size_t)C);
    float* dg = up(ar_, s_, g, (size_t)C);
    float* dbt = up(ar_, s_, b, (size_t)C);
    void* dy = ar_.alloc(n * 4);
    float* dy32 = f32_alloc(n);
    launch_dwconv_ln(s_, dtype, dx, B, L, C, dw, db, k, dil, dg, dbt, 1e-6f, dy, dlen);
    if (is_half(dtype)) launch_half_to_f32(s_, dtype, dy, (int64_t)n, dy32);
    STN_HIP(hipMemcpyAsync(y, is_half(dtype) ? dy32 : static_cast<float*>(dy), n * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::op_attention(int dtype, int B, int Lq, int Lk, int H, int dh, const float* q, const float* k, const float* v,
                          const int* qlen, const int* klen, int rope_mode, float* o) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    const int C = H * dh;
    const size_t nq = (size_t)B * Lq * C, nk = (size_t)B * Lk * C;
    float* dq = up(ar_, s_, q, nq);
    float* dk = up(ar_, s_, k, nk);
    float* dv = up(ar_, s_, v, nk);
    int* dql = qlen ? up(ar_, s_, qlen, (size_t)B) : nullptr;
    int* dkl = klen ? up(ar_, s_, klen, (size_t)B) : nullptr;
    const void *pq = dq, *pk = dk, *pv = dv;
    if (is_half(dtype)) {
        void* a = ar_.alloc(nq * 2); void* b = ar_.alloc(nk * 2); void* c = ar_.alloc(nk * 2);
        launch_cast(s_, dtype, dq, (int64_t)nq, a); launch_cast(s_, dtype, dk, (int64_t)nk, b); launch_cast(s_, dtype, dv, (int64_t)nk, c);
        pq = a; pk = b; pv = c;
    }
    void* dO = ar_.alloc(nq * 4);
    float* dO32 = f32_alloc(nq);
    const float rbase = a_.rope_base > 0 ? a_.rope_base : 10000.f, rgam = a_.larope_gamma > 0 ? a_.larope_gamma : 10.f;
    const bool prerot = rope_mode >= 0 && (rope_mode & 0x100) != 0;  // test hook: rotate the keys in a separate pass first
    if (prerot) rope_mode &= 0xFF;
    if (prerot) launch_rope_rows(s_, dtype, const_cast<void*>(pk), C, B, Lk, dkl, 1, 0, H, dh, rope_mode, rbase, rgam);
    launch_attention(s_, dtype, pq, C, pk, pv, C, dO, C, B, Lq, Lk, H, dh, dql, dkl, rope_mode, rbase, rgam, prerot);
    if (is_half(dtype)) launch_half_to_f32(s_, dtype, dO, (int64_t)nq, dO32);
    STN_HIP(hipMemcpyAsync(o, is_half(dtype) ? dO32 : static_cast<float*>(dO), nq * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

double Engine::op_gemm_bench(int dtype, int M, int N, int K, int mode, int iters) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    const size_t esz = is_half(dtype) ? 2 : 4;
    float* tmp = f32_alloc((size_t)std::max((size_t)M * K, (size_t)N * K));
    void* A = ar_.alloc((size_t)M * K * esz);
    void* Wt = ar_.alloc((size_t)N * K * esz);
    launch_randn_masked(s_, 11, nullptr, 1, 1, (int)std::min<size_t>((size_t)M * K, 1u << 30), nullptr, tmp);
    launch_cast(s_, dtype, tmp, (int64_t)M * K, A);
    launch_randn_masked(s_, 12, nullptr, 1, 1, (int)std::min<size_t>((size_t)N * K, 1u << 30), nullptr, tmp);
    launch_scale(s_, tmp, (int)std::min<size_t>((size_t)N * K, 1u << 30), 1.0f / std::sqrt((float)K));
    launch_cast(s_, dtype, tmp, (int64_t)N * K, Wt);
    float* bias = f32_alloc(N);
    float* gamma = f32_alloc(N);
    launch_fill(s_, bias, N, 0.01f);
    launch_fill(s_, gamma, N, 0.2f);
    float* resid = f32_alloc((size_t)M * N);
    void* out = ar_.alloc((size_t)M * N * 4);
    STN_HIP(hipMemsetAsync(resid, 0, (size_t)M * N * 4, s_));
    Epilogue e;
    e.bias = bias;
    if (mode == 1) { e.mode = EPI_RESID; e.resid = resid; e.ldo = N; e.gamma = gamma; }
    else { e.mode = EPI_STORE; e.act = mode == 2 ? ACT_NONE : ACT_GELU; e.out_dtype = mode == 3 ? F32 : dtype; e.out = out; e.ldo = N; }
    for (int i = 0; i < 3; ++i) launch_gemm(s_, dtype, A, K, Wt, K, M, N, K, e);
    hipEvent_t a, b;
    STN_HIP(hipEventCreate(&a));
    STN_HIP(hipEventCreate(&b));
    STN_HIP(hipEventRecord(a, s_));
    for (int i = 0; i < iters; ++i) launch_gemm(s_, dtype, A, K, Wt, K, M, N, K, e);
    STN_HIP(hipEventRecord(b, s_));
    STN_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    STN_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return (double)ms / iters;
}

void Engine::op_gemm_phases(int dtype, int M, int N, int K, int mode, double* out6) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    const size_t esz = is_half(dtype) ? 2 : 4;
    float* tmp = f32_alloc((size_t)std::max((size_t)M * K, (size_t)N * K));
    void* A = ar_.alloc((size_t)M * K * esz);
    void* Wt = ar_.alloc((size_t)N * K * esz);
    launch_randn_masked(s_, 11, nullptr, 1, 1, (int)std::min<size_t>((size_t)M * K, 1u << 30), nullptr, tmp);
    launch_cast(s_, dtype, tmp, (int64_t)M * K, A);
    launch_randn_masked(s_, 12, nullptr, 1, 1, (int)std::min<size_t>((size_t)N * K, 1u << 30), nullptr, tmp);
    launch_scale(s_, tmp, (int)std::min<size_t>((size_t)N * K, 1u << 30), 1.0f / std::sqrt((float)K));
    launch_cast(s_, dtype, tmp, (int64_t)N * K, Wt);
    float* bias = f32_alloc(N);
    float* gamma = f32_alloc(N);
    launch_fill(s_, bias, N, 0.01f);
    launch_fill(s_, gamma, N, 0.2f);
    float* resid = f32_alloc((size_t)M * N);
    void* out = ar_.alloc((size_t)M * N * 4);
    STN_HIP(hipMemsetAsync(resid, 0, (size_t)M * N * 4, s_));
    const size_t max_wg = 1 << 16;
    unsigned long long* ts = static_cast<unsigned long long*>(ar_.alloc(max_wg * 4 * 8));
    Epilogue e;
    e.bias = bias;
    if (mode == 1) { e.mode = EPI_RESID; e.resid = resid; e.ldo = N; e.gamma = gamma; }
    else { e.mode = EPI_STORE; e.act = mode == 2 ? ACT_NONE : ACT_GELU; e.out_dtype = mode == 3 ? F32 : dtype; e.out = out; e.ldo = N; }
    for (int i = 0; i < 3; ++i) launch_gemm(s_, dtype, A, K, Wt, K, M, N, K, e);
    STN_HIP(hipMemsetAsync(ts, 0, max_wg * 4 * 8, s_));
    e.ts = ts;
    launch_gemm(s_, dtype, A, K, Wt, K, M, N, K, e);
    std::vector<unsigned long long> h(max_wg * 4);
    STN_HIP(hipMemcpyAsync(h.data(), ts, max_wg * 4 * 8, hipMemcpyDeviceToHost, s_));
    sync();
    double p0 = 0, p1 = 0, p2 = 0;
    unsigned long long tmin = ~0ull, tmax_in = 0, tend = 0;
    size_t n = 0;
    for (size_t w = 0; w < max_wg; ++w) {
        const unsigned long long* t = &h[w * 4];
        if (t[3] == 0) continue;
        ++n;
        p0 += (double)(t[1] - t[0]); p1 += (double)(t[2] - t[1]); p2 += (double)(t[3] - t[2]);
        tmin = std::min(tmin, t[0]); tmax_in = std::max(tmax_in, t[0]); tend = std::max(tend, t[3]);
    }
    if (n == 0) throw std::runtime_error("op_gemm_phases: this shape does not run on the tiled kernel");
    out6[0] = p0 / n; out6[1] = p1 / n; out6[2] = p2 / n;
    out6[3] = (double)(tend - tmin); out6[4] = (double)(tmax_in - tmin); out6[5] = (double)n;
}

void Engine::op_ffn(int M, int C, int I, const float* xn, const float* W1, const float* b1, const float* W2, const float* b2, const float* gamma,
                    const float* rowvec, const int* row_b, int nseq, float* x, int mode) {
    STN_HIP(hipSetDevice(device_));
    const bool fused = mode != 0;
    if (!is_half(dt_)) throw std::invalid_argument("op_ffn: 16-bit engines only");
    if (fused && !ffn_fused_supported(dt_, C, I)) throw std::invalid_argument("op_ffn: shape not supported by the fused kernel");
    if (mode == 2 && ffn_split_factor(dt_, C, I) < 2) throw std::invalid_argument("op_ffn: shape not supported by the hidden-split kernel");
    ar_.reset();
    float* d_xn = up(ar_, s_, xn, (size_t)M * C);
    float* d_w1 = up(ar_, s_, W1, (size_t)I * C);
    float* d_w2 = up(ar_, s_, W2, (size_t)C * I);
    float* d_b1 = up(ar_, s_, b1, (size_t)I);
    float* d_b2 = b2 ? up(ar_, s_, b2, (size_t)C) : nullptr;
    float* d_g = gamma ? up(ar_, s_, gamma, (size_t)C) : nullptr;
    float* d_x = up(ar_, s_, x, (size_t)M * C);
    float* d_rv = rowvec ? up(ar_, s_, rowvec, (size_t)nseq * C) : nullptr;
    int* d_rb = (rowvec && row_b) ? up(ar_, s_, row_b, (size_t)M) : nullptr;
    void* xn16 = act_alloc((int64_t)M * C);
    void* w1_16 = act_alloc((int64_t)I * C);
    void* w2_16 = act_alloc((int64_t)I * C);
    launch_cast(s_, dt_, d_xn, (int64_t)M * C, xn16);
    launch_cast(s_, dt_, d_w1, (int64_t)I * C, w1_16);
    launch_cast(s_, dt_, d_w2, (int64_t)I * C, w2_16);
    if (fused) {
        void* tmp = act_alloc((int64_t)2 * I * C);
        void* wseq = act_alloc((int64_t)2 * I * C);
        const int S = mode == 2 ? ffn_split_choose(dt_, C, I, M) : 1;
        launch_ffn_pack(s_, w1_16, w2_16, C, I, tmp, wseq, S);
        FfnArgs fa;
        fa.xn = xn16; fa.ldx = C; fa.wseq = wseq; fa.b1 = d_b1; fa.b2 = d_b2; fa.gamma = d_g; fa.x = d_x; fa.ldo = C;
        fa.M = M; fa.I = I; fa.rowvec = d_rv; fa.rv_ld = C; fa.row_b = d_rb; fa.L = M;
        if (mode == 2) {
            fa.split = S; fa.part_stride = ffn_split_rows(M) * C; fa.part = act_alloc(fa.part_stride * S);
            launch_ffn_fused(s_, dt_, C, fa);
            // the pending update, folded by the LayerNorm form of the fold (its normalised output is not part of this op)
            float* ones = f32_alloc(C);
            launch_fill(s_, ones, C, 1.f);
            float* zeros = f32_alloc(C);
            launch_fill(s_, zeros, C, 0.f);
            FoldArgs fo; fo.part = fa.part; fo.S = S; fo.part_stride = fa.part_stride; fo.b2 = d_b2 ? d_b2 : zeros; fo.gamma = d_g ? d_g : ones;
            fo.rowvec = d_rv; fo.rv_ld = C; fo.row_b = d_rb;
            void* y = act_alloc((int64_t)M * C);
            launch_fold_ln(s_, dt_, d_x, M, C, fo, ones, ones, a_.ln_eps, y);
        } else {
            launch_ffn_fused(s_, dt_, C, fa);
        }
    } else {
        void* u = act_alloc((int64_t)M * I);
        Epilogue e1; e1.mode = EPI_STORE; e1.act = ACT_GELU; e1.out_dtype = dt_; e1.out = u; e1.ldo = I; e1.bias = d_b1;
        launch_gemm(s_, dt_, xn16, C, w1_16, C, M, I, C, e1);
        Epilogue e2; e2.mode = EPI_RESID; e2.resid = d_x; e2.ldo = C; e2.gamma = d_g; e2.bias = d_b2; e2.rowvec = d_rv; e2.rv_ld = C; e2.row_b = d_rb; e2.L = M;
        launch_gemm(s_, dt_, u, I, w2_16, I, M, C, I, e2);
    }
    STN_HIP(hipGetLastError());
    STN_HIP(hipMemcpyAsync(x, d_x, sizeof(float) * (size_t)M * C, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::op_ffn_bench(int M, int C, int I, int mode, int iters, double* out5) {
    STN_HIP(hipSetDevice(device_));
    const bool fused = mode != 0;
    const int S = mode == 2 ? ffn_split_choose(dt_, C, I, M) : 1;
    if (mode == 2 && S < 2) throw std::invalid_argument("op_ffn_bench: shape not supported by the hidden-split kernel");
    if (!is_half(dt_)) throw std::invalid_argument("op_ffn_bench: 16-bit engines only");
    if (fused && !ffn_fused_supported(dt_, C, I)) throw std::invalid_argument("op_ffn_bench: shape not supported by the fused kernel");
    ar_.reset();
    for (int i = 0; i < 5; ++i) out5[i] = 0.0;
    // random operands (the clock a chip holds on zeros is not the clock it holds on data)
    float* rnd = f32_alloc((int64_t)M * C);
    launch_randn_masked(s_, 11, nullptr, 1, M, C, nullptr, rnd);
    float* wr = f32_alloc((int64_t)I * C);
    launch_randn_masked(s_, 12, nullptr, 1, I, C, nullptr, wr);
    launch_scale(s_, wr, I * C, 0.05f);
    void* xn16 = act_alloc((int64_t)M * C);
    void* w1_16 = act_alloc((int64_t)I * C);
    void* w2_16 = act_alloc((int64_t)I * C);
    launch_cast(s_, dt_, rnd, (int64_t)M * C, xn16);
    launch_cast(s_, dt_, wr, (int64_t)I * C, w1_16);
    launch_cast(s_, dt_, wr, (int64_t)I * C, w2_16);
    float* d_b1 = f32_alloc(I);
    float* d_b2 = f32_alloc(C);
    float* d_g = f32_alloc(C);
    launch_fill(s_, d_b1, I, 0.01f); launch_fill(s_, d_b2, C, 0.01f); launch_fill(s_, d_g, C, 0.1f);
    float* d_x = f32_alloc((int64_t)M * C);
    STN_HIP(hipMemsetAsync(d_x, 0, sizeof(float) * (size_t)M * C, s_));
    void* wseq = nullptr;
    if (fused) {
        void* tmp = act_alloc((int64_t)2 * I * C);
        wseq = act_alloc((int64_t)2 * I * C);
        launch_ffn_pack(s_, w1_16, w2_16, C, I, tmp, wseq, S);
    }
    void* u = fused ? nullptr : act_alloc((int64_t)M * I);
    const int64_t pstride = ffn_split_rows(M) * C;
    void* part = mode == 2 ? act_alloc(pstride * S) : nullptr;
    const int nslab = (M + 127) / 128;
    const int nwg = mode == 2 ? (nslab + 7) / 8 * 8 * S : nslab;
    unsigned long long* ts = static_cast<unsigned long long*>(ar_.alloc(sizeof(unsigned long long) * 4 * (size_t)nwg));
    auto run = [&](unsigned long long* stamps) {
        if (fused) {
            FfnArgs fa;
            fa.xn = xn16; fa.ldx = C; fa.wseq = wseq; fa.b1 = d_b1; fa.b2 = d_b2; fa.gamma = d_g; fa.x = d_x; fa.ldo = C;
            fa.M = M; fa.I = I; fa.L = M; fa.ts = stamps;
            if (mode == 2) { fa.split = S; fa.part = part; fa.part_stride = pstride; }
            launch_ffn_fused(s_, dt_, C, fa);
        } else {
            Epilogue e1; e1.mode = EPI_STORE; e1.act = ACT_GELU; e1.out_dtype = dt_; e1.out = u; e1.ldo = I; e1.bias = d_b1;
            if (nt_hints_ && (double)M * I * 2.0 > 128e6) e1.nt = 1;
            launch_gemm(s_, dt_, xn16, C, w1_16, C, M, I, C, e1);
            Epilogue e2; e2.mode = EPI_RESID; e2.resid = d_x; e2.ldo = C; e2.gamma = d_g; e2.bias = d_b2; e2.L = M;
            launch_gemm(s_, dt_, u, I, w2_16, I, M, C, I, e2);
        }
    };
    for (int i = 0; i < 3; ++i) run(nullptr);
    hipEvent_t a, b;
    STN_HIP(hipEventCreate(&a)); STN_HIP(hipEventCreate(&b));
    STN_HIP(hipEventRecord(a, s_));
    for (int i = 0; i < iters; ++i) run(nullptr);
    STN_HIP(hipEventRecord(b, s_));
    STN_HIP(hipEventSynchronize(b));
    float ms = 0.f;
    STN_HIP(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    out5[0] = ms / iters;
    if (fused) {
        STN_HIP(hipMemsetAsync(ts, 0, sizeof(unsigned long long) * 4 * (size_t)nwg, s_));
        run(ts);
        std::vector<unsigned long long> h((size_t)4 * nwg);
        STN_HIP(hipMemcpyAsync(h.data(), ts, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, s_));
        sync();
        double s1 = 0, s2 = 0, s3 = 0;
        int live = 0;  // (workgroups of a hidden-split grid beyond the last slab exit at once and leave no stamps)
        for (int w = 0; w < nwg; ++w) {
            if (!h[4 * w + 3]) continue;
            ++live;
            s1 += (double)(h[4 * w + 1] - h[4 * w]); s2 += (double)(h[4 * w + 2] - h[4 * w + 1]); s3 += (double)(h[4 * w + 3] - h[4 * w + 2]);
        }
        if (live) { out5[1] = s1 / live; out5[2] = s2 / live; out5[3] = s3 / live; }
        out5[4] = live;
    }
    STN_HIP(hipGetLastError());
    sync();
}

void Engine::op_fold_dwconv_ln(int B, int C, int k, int dil, int S, const int* seqlen, const float* x, const float* part, const float* b2,
                               const float* gamma, const float* rowvec, const float* w, const float* bias, const float* g, const float* b,
                               float* x_out, float* y) {
    STN_HIP(hipSetDevice(device_));
    if (!is_half(dt_)) throw std::invalid_argument("op_fold_dwconv_ln: 16-bit engines only");
    ar_.reset();
    std::vector<int> off(B + 1, 0);
    int L = 0;
    for (int i = 0; i < B; ++i) { off[i + 1] = off[i] + seqlen[i]; L = std::max(L, seqlen[i]); }
    const int64_t M = off[B];
    const size_t n = (size_t)M * C;
    std::vector<float> wt((size_t)C * k);
    for (int c = 0; c < C; ++c) for (int j = 0; j < k; ++j) wt[(size_t)j * C + c] = w[(size_t)c * k + j];
    const int* dlen = up(ar_, s_, seqlen, (size_t)B);
    const int* doff = up(ar_, s_, off.data(), (size_t)B + 1);
    float* dx = up(ar_, s_, x, n);
    float* dp32 = up(ar_, s_, part, n * S);
    float* db2 = b2 ? up(ar_, s_, b2, (size_t)C) : nullptr;
    float* dgm = gamma ? up(ar_, s_, gamma, (size_t)C) : nullptr;
    float* drv = rowvec ? up(ar_, s_, rowvec, (size_t)B * C) : nullptr;
    float* dw = up(ar_, s_, wt.data(), wt.size());
    float* db = up(ar_, s_, bias, (size_t)C);
    float* dg = up(ar_, s_, g, (size_t)C);
    float* dbt = up(ar_, s_, b, (size_t)C);
    void* dp16 = act_alloc((int64_t)n * S);
    launch_cast(s_, dt_, dp32, (int64_t)n * S, dp16);
    float* dxo = f32_alloc((int64_t)n);
    void* dy = act_alloc((int64_t)n);
    float* dy32 = f32_alloc((int64_t)n);
    float* ones = f32_alloc(C);
    launch_fill(s_, ones, C, 1.f);
    float* zeros = f32_alloc(C);
    launch_fill(s_, zeros, C, 0.f);
    FoldArgs fo; fo.part = dp16; fo.S = S; fo.part_stride = (int64_t)n; fo.b2 = db2 ? db2 : zeros; fo.gamma = dgm ? dgm : ones; fo.rowvec = drv; fo.rv_ld = C;
    launch_fold_dwconv_ln(s_, dt_, dx, dxo, B, L, C, fo, dw, db, k, dil, dg, dbt, 1e-6f, dy, dlen, doff);
    launch_half_to_f32(s_, dt_, dy, (int64_t)n, dy32);
    STN_HIP(hipGetLastError());
    STN_HIP(hipMemcpyAsync(x_out, dxo, n * 4, hipMemcpyDeviceToHost, s_));
    STN_HIP(hipMemcpyAsync(y, dy32, n * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

void Engine::op_block_bench(int B, int L, int C, int I, int k, int dil, int mode, int iters, double* out2) {
    STN_HIP(hipSetDevice(device_));
    if (!is_half(dt_)) throw std::invalid_argument("op_block_bench: 16-bit engines only");
    const int S = ffn_split_choose(dt_, C, I, (int64_t)B * L);
    if (mode == 2 && S < 2) throw std::invalid_argument("op_block_bench: shape not supported by the hidden-split kernel");
    ar_.reset();
    for (int i = 0; i < 6; ++i) out2[i] = 0.0;
    const int64_t M = (int64_t)B * L;
    std::vector<int> len(B, L), off(B + 1);
    for (int i = 0; i <= B; ++i) off[i] = i * L;
    const int* dlen = up(ar_, s_, len.data(), (size_t)B);
    const int* doff = up(ar_, s_, off.data(), (size_t)B + 1);
    float* xa = f32_alloc(M * C);
    float* xb = f32_alloc(M * C);
    launch_randn_masked(s_, 11, nullptr, 1, (int)M, C, nullptr, xa);
    float* wr = f32_alloc((int64_t)I * C);
    launch_randn_masked(s_, 12, nullptr, 1, I, C, nullptr, wr);
    launch_scale(s_, wr, I * C, 0.05f);
    void* w1_16 = act_alloc((int64_t)I * C);
    void* w2_16 = act_alloc((int64_t)I * C);
    launch_cast(s_, dt_, wr, (int64_t)I * C, w1_16);
    launch_cast(s_, dt_, wr, (int64_t)I * C, w2_16);
    float* d_b1 = f32_alloc(I);
    float* d_b2 = f32_alloc(C);
    float* d_g = f32_alloc(C);
    float* d_one = f32_alloc(C);
    float* dwt = f32_alloc((int64_t)k * C);
    launch_fill(s_, d_b1, I, 0.01f); launch_fill(s_, d_b2, C, 0.01f); launch_fill(s_, d_g, C, 0.01f); launch_fill(s_, d_one, C, 1.f);
    launch_fill(s_, dwt, k * C, 1.f / k);
    void* xn = act_alloc(M * C);
    void* u = mode == 2 ? nullptr : act_alloc(M * I);
    void* wseq = nullptr;
    const int64_t pstride = ffn_split_rows(M) * C;
    void* part = nullptr;
    if (mode == 2) {
        void* tmp = act_alloc((int64_t)2 * I * C);
        wseq = act_alloc((int64_t)2 * I * C);
        launch_ffn_pack(s_, w1_16, w2_16, C, I, tmp, wseq, S);
        part = act_alloc(pstride * S);
        STN_HIP(hipMemsetAsync(part, 0, (size_t)pstride * S * 2, s_));
    }
    FoldArgs fo; fo.part = part; fo.S = S; fo.part_stride = pstride; fo.b2 = d_b2; fo.gamma = d_g;
    auto conv = [&]() {
        if (mode == 2) { launch_fold_dwconv_ln(s_, dt_, xa, xb, B, L, C, fo, dwt, d_b2, k, dil, d_one, d_b2, 1e-6f, xn, dlen, doff); std::swap(xa, xb); }
        else launch_dwconv_ln(s_, dt_, xa, B, L, C, dwt, d_b2, k, dil, d_one, d_b2, 1e-6f, xn, dlen, doff);
    };
    auto block = [&]() {
        conv();
        if (mode == 2) {
            FfnArgs fa; fa.xn = xn; fa.ldx = C; fa.wseq = wseq; fa.b1 = d_b1; fa.M = (int)M; fa.I = I; fa.split = S; fa.part = part; fa.part_stride = pstride;
            launch_ffn_fused(s_, dt_, C, fa);
        } else {
            Epilogue e1; e1.mode = EPI_STORE; e1.act = ACT_GELU; e1.out_dtype = dt_; e1.out = u; e1.ldo = I; e1.bias = d_b1;
            launch_gemm(s_, dt_, xn, C, w1_16, C, (int)M, I, C, e1);
            Epilogue e2; e2.mode = EPI_RESID; e2.resid = xa; e2.ldo = C; e2.gamma = d_g; e2.bias = d_b2; e2.L = (int)M;
            launch_gemm(s_, dt_, u, I, w2_16, I, (int)M, C, I, e2);
        }
    };
    hipEvent_t a, b;
    STN_HIP(hipEventCreate(&a)); STN_HIP(hipEventCreate(&b));
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 3; ++i) { if (which) conv(); else block(); }
        STN_HIP(hipEventRecord(a, s_));
        for (int i = 0; i < iters; ++i) { if (which) conv(); else block(); }
        STN_HIP(hipEventRecord(b, s_));
        STN_HIP(hipEventSynchronize(b));
        float ms = 0.f;
        STN_HIP(hipEventElapsedTime(&ms, a, b));
        out2[which] = ms / iters;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (mode == 2) {  // phase stamps of one fold_dwconv_ln launch
        const int nwg = B * ((L + 7) / 8);  // (runs of 8 frames when there are few sequences, of 32 otherwise: sized for the shorter)
        unsigned long long* ts = static_cast<unsigned long long*>(ar_.alloc(sizeof(unsigned long long) * 4 * (size_t)nwg));
        STN_HIP(hipMemsetAsync(ts, 0, sizeof(unsigned long long) * 4 * (size_t)nwg, s_));
        fo.ts = ts;
        conv();
        fo.ts = nullptr;
        std::vector<unsigned long long> h((size_t)4 * nwg);
        STN_HIP(hipMemcpyAsync(h.data(), ts, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, s_));
        sync();
        double p1 = 0, p2 = 0, p3 = 0; int live = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int w = 0; w < nwg; ++w) {
            if (!h[4 * w + 3]) continue;
            ++live;
            p1 += (double)(h[4 * w + 1] - h[4 * w]); p2 += (double)(h[4 * w + 2] - h[4 * w + 1]); p3 += (double)(h[4 * w + 3] - h[4 * w + 2]);
            tmin = std::min(tmin, h[4 * w]); tmax = std::max(tmax, h[4 * w + 3]);
        }
        if (live) { out2[2] = p1 / live; out2[3] = p2 / live; out2[4] = p3 / live; out2[5] = (double)(tmax - tmin); }
    }
    STN_HIP(hipGetLastError());
    sync();
}

void Engine::op_randn(uint64_t seed, int B, int D, int L, const int64_t* utt_ids, const int* len, float* out) {
    STN_HIP(hipSetDevice(device_));
    ar_.reset();
    int64_t* du = utt_ids ? up(ar_, s_, utt_ids, (size_t)B) : nullptr;
    int* dl = len ? up(ar_, s_, len, (size_t)B) : nullptr;
    float* d = f32_alloc((size_t)B * D * L);
    launch_randn_masked(s_, seed, du, B, D, L, dl, d);
    STN_HIP(hipMemcpyAsync(out, d, (size_t)B * D * L * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

}  // namespace stn
