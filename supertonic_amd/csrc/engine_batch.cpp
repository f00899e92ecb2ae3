// engine_batch.cpp — the resident-batch pipeline (stn_batch_*): upload, the one duration read, the captured latent pipeline and its
// graph cache, the fetches (see engine.hpp).
#include "engine.hpp"
#include "engine_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace stn {
using detail::up;

// =================================================================================================
// resident batch
// =================================================================================================

// Grow-only device buffer of the resident batch.  A reallocation retires the old block (freed at the next upload, once the
// stream has drained) and bumps the generation that is part of the graph key: a captured graph holds raw pointers.
template <typename T>
void Engine::ensure(T*& p, size_t& cap, size_t need) {
    if (p && need <= cap) return;
    const size_t n = std::max<size_t>(need + need / 4, 64);  // headroom: ragged request streams settle after a few uploads
    T* q = nullptr;
    STN_HIP(hipMalloc(reinterpret_cast<void**>(&q), n * sizeof(T)));
    if (p) {
        batch_owned_.erase(std::find(batch_owned_.begin(), batch_owned_.end(), static_cast<void*>(p)));
        batch_retired_.push_back(p);
    }
    batch_owned_.push_back(q);
    p = q;
    cap = n;
    ++bt_.gen;
}

void Engine::batch_upload(int B, int Lt, const int64_t* ids, const float* text_mask, const float* style_ttl,
                          const float* style_dp, const float* duration_override, const int64_t* utt_ids) {
    STN_HIP(hipSetDevice(device_));
    if (!loaded_) throw std::runtime_error("no model loaded");
    if (B <= 0 || Lt <= 0) throw std::runtime_error("empty batch");
    sync();
    for (void* p : batch_retired_) (void)hipFree(p);
    batch_retired_.clear();
    Batch& b = bt_;
    const int Lt_in = Lt;  // the caller's row stride of ids / text_mask
    if (shape_buckets_) Lt = (int)bucket_up(Lt, 32);  // token rows padded with id 0 / mask 0: the lengths come from the mask, not from Lt
    b.B = B; b.Lt = Lt; b.L = 0; b.noise_L = 0; b.total_step = 0;
    b.have_override = false; b.have_noise = false;
    b.h_dur.clear(); b.h_llen.clear();
    const size_t n_ttl = (size_t)B * a_.n_style_ttl * a_.d_style_ttl, n_dp = (size_t)B * a_.n_style_dp * a_.d_style_dp;
    ensure(b.ids, b.ids_cap, (size_t)B * Lt);
    ensure(b.tlen, b.tlen_cap, (size_t)B);
    ensure(b.style_ttl, b.ttl_cap, n_ttl);
    ensure(b.style_dp, b.dp_cap, n_dp);
    ensure(b.dur, b.dur_cap, (size_t)B);
    ensure(b.llen, b.llen_cap, (size_t)B);
    ensure(b.utt_ids, b.utt_cap, (size_t)B);
    if (Lt != Lt_in) {
        STN_HIP(hipMemsetAsync(b.ids, 0, sizeof(int64_t) * B * Lt, s_));
        STN_HIP(hipMemcpy2DAsync(b.ids, sizeof(int64_t) * Lt, ids, sizeof(int64_t) * Lt_in, sizeof(int64_t) * Lt_in, (size_t)B, hipMemcpyHostToDevice, s_));
    } else {
        STN_HIP(hipMemcpyAsync(b.ids, ids, sizeof(int64_t) * B * Lt, hipMemcpyHostToDevice, s_));
    }
    STN_HIP(hipMemcpyAsync(b.style_ttl, style_ttl, sizeof(float) * n_ttl, hipMemcpyHostToDevice, s_));
    STN_HIP(hipMemcpyAsync(b.style_dp, style_dp, sizeof(float) * n_dp, hipMemcpyHostToDevice, s_));
    std::vector<int64_t> uid(B);
    for (int i = 0; i < B; ++i) uid[i] = utt_ids ? utt_ids[i] : i;
    STN_HIP(hipMemcpyAsync(b.utt_ids, uid.data(), sizeof(int64_t) * B, hipMemcpyHostToDevice, s_));
    ar_.reset();
    float* d_mask = up(ar_, s_, text_mask, (size_t)B * Lt_in);
    launch_mask_to_len(s_, d_mask, B, Lt_in, b.tlen);
    // packed text rows: first row of each utterance (fixed for the life of this upload) and their total
    ensure(b.toff, b.toff_cap, (size_t)B + 1);
    b.trows = 0;
    for (int i = 0; i < B; ++i) {
        int n = 0;
        for (int t = 0; t < Lt_in; ++t) n += text_mask[(size_t)i * Lt_in + t] > 0.5f ? 1 : 0;  // as mask_to_len_kernel counts
        b.trows += n;
    }
    if (shape_buckets_) b.trows = (int)bucket_up(b.trows, 64);  // dead text rows behind the last utterance (toff[B] keeps the exact sum)
    if (B <= 1024) launch_row_map(s_, b.tlen, B, b.toff, nullptr);
    b.have_override = duration_override != nullptr;
    if (duration_override) b.h_dur.assign(duration_override, duration_override + B);
    sync();
}

void Engine::batch_set_noise(const float* noise, int L) {
    STN_HIP(hipSetDevice(device_));
    Batch& b = bt_;
    if (b.B == 0) throw std::runtime_error("batch_set_noise: no batch uploaded");
    if (L < 1) throw std::runtime_error("batch_set_noise: L must be >= 1");
    const int D = a_.latent_dim * a_.chunk_compress_factor;
    sync();
    ensure(b.noise, b.noise_cap, (size_t)b.B * D * L);
    STN_HIP(hipMemcpyAsync(b.noise, noise, sizeof(float) * b.B * D * L, hipMemcpyHostToDevice, s_));
    b.have_noise = true;
    b.noise_L = L;  // checked against the durations in batch_run
    sync();
}

// Latent geometry exactly as TextToSpeech::sampleNoisyLatent (/root/reference/cpp/helper.cpp:424-440,457,764-768):
// float32 products, truncation to integers.
static void latent_geometry(const stn_arch& a, const std::vector<float>& dur, int& L, std::vector<int>& llen) {
    const int cs = a.base_chunk_size * a.chunk_compress_factor;
    float mx = dur[0];
    for (float d : dur) mx = std::max(mx, d);
    const float wav_len_max = mx * (float)a.sample_rate;
    L = (int)((wav_len_max + (float)cs - 1.0f) / (float)cs);
    llen.resize(dur.size());
    for (size_t i = 0; i < dur.size(); ++i) {
        const int64_t wl = (int64_t)(dur[i] * (float)a.sample_rate);
        llen[i] = (int)((wl + cs - 1) / cs);
    }
}

void Engine::batch_run(int total_step, float speed, uint64_t noise_seed) {
    STN_HIP(hipSetDevice(device_));
    Batch& b = bt_;
    if (b.B == 0) throw std::runtime_error("batch_run: no batch uploaded");
    if (total_step < 1) throw std::runtime_error("total_step must be >= 1");
    if (!(speed > 0.f)) throw std::runtime_error("speed must be > 0");
    const stn_arch& a = a_;
    const int B = b.B, Lt = b.Lt, D = a.latent_dim * a.chunk_compress_factor;
    b.total_step = total_step; b.speed = speed; b.noise_seed = noise_seed;
    ar_.reset();
    // 1. duration predictor (always executed; its output may be overridden for shape control)
    Ragged trg;
    const bool tpk = packed_text_ok(B) && b.trows > 0;
    if (tpk) { trg.off = b.toff; trg.rows = b.trows; }
    std::vector<float> dur(B);
    // the text rows: persistent, grow-only buffers (a reallocation bumps b.gen, which every graph key holds)
    const size_t text_bytes = (size_t)(tpk ? (int64_t)b.trows : (int64_t)B * Lt) * a.te_out_dim * (is_half(dt_) ? 2 : 4);
    ensure(b.text_side, b.text_side_cap, text_bytes);
    ensure(b.text_rows, b.text_cap, text_bytes);
    {
        // on the side streams, each with its own workspace (see dp_s_): swapped in for the duration of a text stage
        struct Side {
            Engine& e; hipStream_t& s; Arena& ar; bool on;
            Side(Engine& e_, hipStream_t& s_, Arena& ar_) : e(e_), s(s_), ar(ar_), on(s_ != nullptr) { if (on) { std::swap(e.s_, s); e.ar_.swap(ar); e.ar_.reset(); } }
            ~Side() { if (on) { std::swap(e.s_, s); e.ar_.swap(ar); } }
        };
        {   // 2. text encoder -> context rows (act dtype), once the previous run has taken its copy of them
            Side side(*this, te_s_, te_ar_);
            if (side.on && copied_valid_) STN_HIP(hipStreamWaitEvent(s_, ev_copied_, 0));
            text_enc_dev(B, Lt, b.ids, b.style_ttl, b.tlen, nullptr, b.text_side, tpk ? &trg : nullptr);
            STN_HIP(hipEventRecord(ev_te_, s_));
        }
        {   // 1. duration predictor (always executed; its output may be overridden for shape control), beside the encoder
            Side side(*this, dp_s_, dp_ar_);
            duration_dev(B, Lt, b.ids, b.style_dp, b.tlen, b.dur, tpk ? &trg : nullptr);
            if (!b.have_override || dur_read_always_) {
                STN_HIP(hipMemcpyAsync(dur.data(), b.dur, sizeof(float) * B, hipMemcpyDeviceToHost, s_));
                STN_HIP(hipEventRecord(ev_dp_, s_));
            }
        }
        if (!b.have_override || dur_read_always_) STN_HIP(hipEventSynchronize(ev_dp_));  // the one host round trip (the predictor only): L = f(max duration) sizes every later buffer
        if (b.have_override) dur = b.h_dur;  // known on the host: no device->host read, no sync (unless the measurement switch asks for the read: the
                                             // critical path of a predicted-duration run on the controlled shapes of a forced one)
    }
    // Hand-over of this run's text rows to the main pipeline (everything captured below reads b.text_rows).  It happens where the
    // rows are first needed — in front of the text K/V GEMM of the first text cross-attention — so the noise, the style K/V, the time
    // conditioning and the first ConvNeXt blocks of the estimator run beside the encoder; a captured pipeline is therefore TWO
    // graphs with this hand-over between them.
    auto take_text_rows = [this, &b, text_bytes]() {
        if (te_s_) STN_HIP(hipStreamWaitEvent(s_, ev_te_, 0));
        STN_HIP(hipMemcpyAsync(b.text_rows, b.text_side, text_bytes, hipMemcpyDeviceToDevice, s_));
        STN_HIP(hipEventRecord(ev_copied_, s_));
        copied_valid_ = true;
    };
    for (float& d : dur) d /= speed;  // cpp/helper.cpp:529-531
    int L = 0;
    latent_geometry(a, dur, L, b.h_llen);
    if (L < 1) throw std::runtime_error("predicted duration too short: zero latent frames");
    if (shape_buckets_ && !b.have_noise) L = (int)bucket_up(L, 16);  // (injected noise has the caller's exact [B, D, L] layout)
    if (b.have_noise && b.noise_L != L)
        throw std::runtime_error("injected noise has L=" + std::to_string(b.noise_L) + " but the durations imply L=" + std::to_string(L));
    b.L = L;
    reported_dur_ = dur;  // durations after /speed: what the reference returns (cpp/helper.cpp:680)
    const size_t nx = (size_t)B * D * L, nw = (size_t)B * L * a.base_chunk_size * a.chunk_compress_factor;
    ensure(b.xt[0], b.xt_cap[0], nx);
    ensure(b.xt[1], b.xt_cap[1], nx);
    ensure(b.wav, b.wav_cap, nw);
    // per-call data of the captured region lives in pinned host memory (the graph's memcpy nodes re-read it at replay).
    // Sized for 1024 utterances up front so that it is not reallocated under cached graphs; should it ever have to grow, its
    // address is part of the graph key and the graphs that point at the old block are dropped before it is freed.
    if ((size_t)B > pin_llen_cap_) {
        if (pin_llen_) { sync(); drop_graphs(); (void)hipHostFree(pin_llen_); pin_llen_ = nullptr; }
        const size_t cap = std::max<size_t>((size_t)B, 1024);
        STN_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin_llen_), sizeof(int) * cap, hipHostMallocDefault));
        pin_llen_cap_ = cap;
        pin_valid_ = false;
    }
    if (!pin_seed_) {
        STN_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin_seed_), sizeof(unsigned long long), hipHostMallocDefault));
        STN_HIP(hipMalloc(reinterpret_cast<void**>(&seed_dev_), sizeof(unsigned long long)));
        pin_valid_ = false;
    }
    // a previous run's copy nodes may still be reading the staging: rewrite it only when the content changes, and then
    // only after the stream has drained (steady-state replays of an unchanged batch never wait here)
    if (!pin_valid_ || !std::equal(b.h_llen.begin(), b.h_llen.end(), pin_llen_) || *pin_seed_ != (unsigned long long)noise_seed) {
        sync();
        std::copy(b.h_llen.begin(), b.h_llen.end(), pin_llen_);
        *pin_seed_ = (unsigned long long)noise_seed;
        pin_valid_ = true;
    }

    ensure_time_cond(total_step, B);  // eager, on s_, ahead of whatever is replayed or captured below (same stream: ordered)
    GraphKey key;
    key.B = B; key.Lt = Lt; key.L = L; key.steps = total_step; key.noise = b.have_noise; key.ragged = vo_ragged_; key.xattn = fused_xattn_;
    key.ffn = fused_ffn_; key.gen = b.gen; key.wgen = wgen_; key.pin = pin_llen_;
    key.rows = 0;
    if (packed_rows_ok(B)) for (int v : b.h_llen) key.rows += v;
    if (shape_buckets_) key.rows = (int)bucket_up(key.rows, 64);
    last_ve_rows_ = key.rows ? key.rows : (int64_t)B * L;
    key.vrows = trimmed_rows(B, L, nullptr);
    if (shape_buckets_) key.vrows = (int)bucket_up(key.vrows, 64 * a.chunk_compress_factor);
    key.trows = tpk ? b.trows : 0;
    last_vo_rows_ = (int64_t)B * L * a.chunk_compress_factor;
    if (vo_ragged_ && packed_ve_ && is_half(dt_)) {
        last_vo_rows_ = 0;
        for (int v : b.h_llen) last_vo_rows_ += (int64_t)v * a.chunk_compress_factor;
        if (shape_buckets_) last_vo_rows_ = bucket_up(last_vo_rows_, 64 * a.chunk_compress_factor);
        key.vrows = (int)last_vo_rows_;  // (the length-aware vocoder's packed rows are baked into the pipeline like the others)
    }
    else if (key.vrows) last_vo_rows_ = key.vrows;
    key.p0 = b.xt[0]; key.p1 = b.wav; key.s = s_;
    // event timing forces eager launches: hipEventRecord captured into a graph returns garbage spans on ROCm 7.2 (measured)
    const bool graphable = graph_on_ && !prof_on_;
    if (graphable) {
        for (auto& g : graphs_)
            if (g.key == key) {
                g.last_use = ++graph_clock_;
                STN_HIP(hipGraphLaunch(g.exec, s_));
                take_text_rows();
                STN_HIP(hipGraphLaunch(g.exec2, s_));
                ++graph_replays_;
                return;
            }
    }
    const auto warm = std::find(warm_keys_.begin(), warm_keys_.end(), key);
    if (graphable && warm != warm_keys_.end()) {  // second sighting of this shape: the arena is warm, allocation order is fixed
        const Arena::Mark cap0 = ar_.mark();
        const size_t cap_before = ar_.capacity();
        STN_HIP(hipStreamBeginCapture(s_, hipStreamCaptureModeThreadLocal));
        bool ok = true;
        std::string why;
        hipGraph_t g = nullptr, g2 = nullptr;
        hipError_t ec1 = hipErrorUnknown;
        bool split = false;
        // at the hand-over the first graph ends and the second begins (the hand-over itself is issued between their launches)
        auto split_capture = [&]() {
            ec1 = hipStreamEndCapture(s_, &g);
            split = true;
            STN_HIP(hipStreamBeginCapture(s_, hipStreamCaptureModeThreadLocal));
        };
        try { enqueue_after_duration(total_step, split_capture); } catch (const std::exception& e) { ok = false; why = e.what(); }
        const hipError_t ec = hipStreamEndCapture(s_, &g2);
        text_gate_ = nullptr;
        ar_.release(cap0);
        hipGraphExec_t ex = nullptr, ex2 = nullptr;
        if (ok && split && ec1 == hipSuccess && ec == hipSuccess && g && g2 && ar_.capacity() == cap_before &&
            hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && hipGraphInstantiate(&ex2, g2, nullptr, nullptr, 0) == hipSuccess) {
            if (graphs_.size() >= kGraphCache) {  // evict the least recently used entry
                auto lru = std::min_element(graphs_.begin(), graphs_.end(), [](const GraphEntry& x, const GraphEntry& y) { return x.last_use < y.last_use; });
                sync();  // its last replay may still be running
                (void)hipGraphExecDestroy(lru->exec);
                (void)hipGraphDestroy(lru->graph);
                if (lru->exec2) (void)hipGraphExecDestroy(lru->exec2);
                if (lru->graph2) (void)hipGraphDestroy(lru->graph2);
                graphs_.erase(lru);
            }
            GraphEntry e;
            e.key = key; e.graph = g; e.exec = ex; e.graph2 = g2; e.exec2 = ex2; e.last_use = ++graph_clock_;
            graphs_.push_back(e);
            warm_keys_.erase(warm);
            STN_HIP(hipGraphLaunch(ex, s_));
            take_text_rows();
            STN_HIP(hipGraphLaunch(ex2, s_));
            ++graph_replays_;
            return;
        }
        if (ex) (void)hipGraphExecDestroy(ex);
        if (ex2) (void)hipGraphExecDestroy(ex2);
        if (g2) (void)hipGraphDestroy(g2);
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        if (!ok) throw std::runtime_error("graph capture failed: " + why);
        // fall through to an eager run
    }
    if (warm == warm_keys_.end()) {
        if (warm_keys_.size() >= kWarmKeys) warm_keys_.erase(warm_keys_.begin());
        warm_keys_.push_back(key);
    }
    enqueue_after_duration(total_step, take_text_rows);
}

// Everything after the duration read: lengths to the device, text encoder, initial latent, Euler loop, vocoder.
void Engine::ensure_time_cond(int total_step, int B) {
    TimeCond& t = tcond_;
    if (t.steps == total_step && t.B == B && t.wgen == wgen_ && t.buf) return;
    const stn_arch& a = a_;
    const size_t n_cnt = (size_t)2 * total_step * B + (size_t)B, n_tb = (size_t)total_step * B * a.ve_main_blocks * a.ve_dim;
    const size_t need = (n_cnt + 3) / 4 * 4 + n_tb;  // (tb 16-byte aligned behind the counters)
    if (need > t.cap) {
        sync();
        drop_graphs();  // captured pipelines read the old buffer
        if (t.buf) (void)hipFree(t.buf);
        t.buf = nullptr; t.cap = 0;
        STN_HIP(hipMalloc(reinterpret_cast<void**>(&t.buf), need * sizeof(float)));
        t.cap = need;
    }
    t.tot = t.buf; t.cur = t.tot + (size_t)total_step * B; t.dt = t.cur + (size_t)total_step * B; t.tb = t.buf + (n_cnt + 3) / 4 * 4;
    t.steps = 0;  // (invalid until the launches below are enqueued)
    const Arena::Mark mk = ar_.mark();
    launch_step_counters(s_, t.tot, t.cur, t.dt, B, total_step);
    (void)ve_time_cond_dev(total_step * B, t.tot, t.cur, t.tb);
    ar_.release(mk);
    t.steps = total_step; t.B = B; t.wgen = wgen_;
}

void Engine::enqueue_after_duration(int total_step, const std::function<void()>& take_text_rows) {
    Batch& b = bt_;
    const stn_arch& a = a_;
    const int B = b.B, Lt = b.Lt, L = b.L, D = a.latent_dim * a.chunk_compress_factor;
    const size_t nx = (size_t)B * D * L;
    STN_HIP(hipMemcpyAsync(b.llen, pin_llen_, sizeof(int) * B, hipMemcpyHostToDevice, s_));
    STN_HIP(hipMemcpyAsync(seed_dev_, pin_seed_, sizeof(unsigned long long), hipMemcpyHostToDevice, s_));
    // 2. (the text encoder ran on the side stream: batch_run) its rows
    Ragged trg;
    const bool tpk = packed_text_ok(B) && b.trows > 0;
    if (tpk) { trg.off = b.toff; trg.rows = b.trows; }
    const Ragged* trgp = tpk ? &trg : nullptr;
    void* text_rows = b.text_rows;
    // 3. initial latent
    if (b.have_noise) {
        STN_HIP(hipMemcpyAsync(b.xt[0], b.noise, nx * 4, hipMemcpyDeviceToDevice, s_));
        launch_mask_ncl(s_, b.xt[0], B, D, L, b.llen);
    } else {
        launch_randn_masked(s_, 0, b.utt_ids, B, D, L, b.llen, b.xt[0], seed_dev_);
    }
    // 4. Euler loop: step-invariant K/V once, the time conditioning of every step in one pass, then total_step passes
    VeCtx c = ve_prepare_dev(B, Lt, text_rows, b.style_ttl, b.tlen, trgp, /*defer_text=*/true);
    // armed here, fired by the first text cross-attention of the first Euler step (ve_step_dev): hand-over of the rows, then the text K/V
    text_gate_ = [this, &c, &b, &take_text_rows, B, Lt, text_rows, trgp]() {
        take_text_rows();
        ve_text_kv_dev(c, B, Lt, text_rows, b.tlen, trgp);
    };
    struct Disarm { std::function<void()>& g; ~Disarm() { g = nullptr; } } disarm{text_gate_};  // it refers to this frame: never outlives it
    if (a.ve_main_blocks == 0) { auto fire = std::move(text_gate_); text_gate_ = nullptr; fire(); }  // (no cross-attention would ever fire it)
    // (step counters and time conditioning: computed once per (total_step, B, weights) by ensure_time_cond, ahead of the captured pipeline)
    if (tcond_.steps != total_step || tcond_.B != B || tcond_.wgen != wgen_) throw std::runtime_error("time conditioning not prepared for this run");
    const float* tot_all = tcond_.tot;
    const float* cur_all = tcond_.cur;
    const float* dt_all = tcond_.dt;
    const float* tb_all = tcond_.tb;
    const size_t tb_stride = (size_t)B * a.ve_main_blocks * a.ve_dim;
    Ragged rg;
    const Ragged* rgp = nullptr;
    if (packed_rows_ok(B)) {  // the estimator works on the frames the utterances own and nothing else
        rg.rows = 0;
        for (int v : b.h_llen) rg.rows += v;
        if (shape_buckets_) rg.rows = (int)bucket_up(rg.rows, 64);  // dead rows behind the last utterance (as in the graph key)
        int* off = static_cast<int*>(ar_.alloc(sizeof(int) * (size_t)(B + 1)));
        int* row_b = static_cast<int*>(ar_.alloc(sizeof(int) * (size_t)std::max(rg.rows, 1)));
        launch_row_map(s_, b.llen, B, off, row_b, rg.rows);
        rg.off = off; rg.row_b = row_b;
        // one 1024-thread workgroup of fold_dwconv_ln fits a CU: 271 runs of <= 32 frames (this bench's lengths) are two rounds, 256 runs of <= 40 one
        rg.fold_run = fold_run_frames(b.h_llen.data(), B, n_cu_);
        if (fused_xattn_ && is_half(dt_) && B >= 2) {  // the head-split cross-attention's pairing, once per synthesis (the lengths are the run's)
            int* pairs = static_cast<int*>(ar_.alloc(sizeof(int) * (size_t)(B + 2)));
            launch_xattn_hs_pairs(s_, b.llen, B, pairs);
            rg.hs_pairs = pairs;
        }
        rgp = &rg;
    }
    int cur = 0;
    // the latent as rows (the input projection's operand) lives across the steps: every step's Euler update writes it beside the [B][D][L] layout, so only
    // the first step converts (ncl_to_rows: 11 us of strided reads per step otherwise)
    const int Dz = (a.latent_dim * a.chunk_compress_factor + 63) / 64 * 64;
    const int64_t Mz = rgp ? (int64_t)rg.rows : (int64_t)B * L;
    void* z_rows = act_alloc(Mz * Dz);
    for (int st = 0; st < total_step; ++st) {
        ve_step_dev(B, L, c, b.xt[cur], b.tlen, b.llen, tot_all + (size_t)st * B, cur_all + (size_t)st * B, b.xt[cur ^ 1],
                    tb_all + (size_t)st * tb_stride, rgp, dt_all, z_rows, st > 0, st + 1 < total_step);
        cur ^= 1;
    }
    final_xt_ = cur;
    // 5. vocoder
    const int* vlen = nullptr;
    if (vo_ragged_) {
        int* v = static_cast<int*>(ar_.alloc(sizeof(int) * B));
        launch_scale_len(s_, b.llen, B, a.chunk_compress_factor, v);
        vlen = v;
    }
    int vrows = 0;
    const int* valid = nullptr;
    if (vo_ragged_ && packed_ve_) {
        for (int v : b.h_llen) vrows += v * a.chunk_compress_factor;
        if (shape_buckets_) vrows = (int)bucket_up(vrows, 64 * a.chunk_compress_factor);
    } else if (int tr = trimmed_rows(B, L, nullptr)) {
        if (shape_buckets_) tr = (int)bucket_up(tr, 64 * a.chunk_compress_factor);
        // reference (dense) semantics at the cost of the frames that are not position-independent
        int* n_dev = static_cast<int*>(ar_.alloc(sizeof(int) * B));
        int* v_dev = static_cast<int*>(ar_.alloc(sizeof(int) * B));
        launch_trim_len(s_, b.llen, B, a.chunk_compress_factor, L * a.chunk_compress_factor, vo_rf_, n_dev, v_dev);
        vlen = n_dev; valid = v_dev; vrows = tr;
    }
    vocoder_dev(B, L, b.xt[cur], b.wav, vlen, vrows, valid);
    STN_HIP(hipGetLastError());  // a kernel launch that was rejected (bad configuration) must not pass silently
}

void Engine::batch_fetch(float* wav, size_t wav_capacity, float* duration) {
    Batch& b = bt_;
    const size_t nw = (size_t)b.B * b.L * a_.base_chunk_size * a_.chunk_compress_factor;
    if (wav) {
        if (wav_capacity < nw) throw std::runtime_error("wav buffer too small: need " + std::to_string(nw) + " floats");
        STN_HIP(hipMemcpyAsync(wav, b.wav, nw * 4, hipMemcpyDeviceToHost, s_));
    }
    sync();
    if (duration) std::copy(reported_dur_.begin(), reported_dur_.end(), duration);
}
void Engine::batch_fetch_pcm16(int16_t* pcm, size_t capacity, float* duration) {
    STN_HIP(hipSetDevice(device_));
    Batch& b = bt_;
    const size_t nw = (size_t)b.B * b.L * a_.base_chunk_size * a_.chunk_compress_factor;
    if (!b.wav || b.L == 0) throw std::runtime_error("no finished batch");
    if (capacity < nw) throw std::runtime_error("pcm buffer too small: need " + std::to_string(nw) + " samples");
    ensure(b.pcm, b.pcm_cap, nw);
    launch_f32_to_pcm16(s_, b.wav, (int64_t)b.B, (int)(nw / (size_t)b.B), b.pcm, (int64_t)(nw / (size_t)b.B));
    STN_HIP(hipMemcpyAsync(pcm, b.pcm, nw * 2, hipMemcpyDeviceToHost, s_));
    sync();
    if (duration) std::copy(reported_dur_.begin(), reported_dur_.end(), duration);
}
void Engine::batch_fetch_pcm16_begin(int slot) {
    STN_HIP(hipSetDevice(device_));
    if (slot < 0 || slot > 1) throw std::invalid_argument("fetch slot must be 0 or 1");
    Batch& b = bt_;
    if (!b.wav || b.L == 0) throw std::runtime_error("no finished batch");
    const size_t nw = (size_t)b.B * b.L * a_.base_chunk_size * a_.chunk_compress_factor;
    FetchSlot& f = fetch_[slot];
    if (!copy_s_) STN_HIP(hipStreamCreateWithFlags(&copy_s_, hipStreamNonBlocking));
    if (!f.ready) { STN_HIP(hipEventCreateWithFlags(&f.ready, hipEventDisableTiming)); STN_HIP(hipEventCreateWithFlags(&f.done, hipEventDisableTiming)); }
    if (f.busy) STN_HIP(hipEventSynchronize(f.done));  // the slot's previous copy (two batches ago) must be out before it is refilled
    if (nw > f.cap) {
        if (f.dev) (void)hipFree(f.dev);
        if (f.pin) (void)hipHostFree(f.pin);
        f.dev = nullptr; f.pin = nullptr;
        const size_t cap = nw + nw / 4;
        STN_HIP(hipMalloc(reinterpret_cast<void**>(&f.dev), cap * sizeof(int16_t)));
        STN_HIP(hipHostMalloc(reinterpret_cast<void**>(&f.pin), cap * sizeof(int16_t), hipHostMallocDefault));
        f.cap = cap;
    }
    launch_f32_to_pcm16(s_, b.wav, (int64_t)b.B, (int)(nw / (size_t)b.B), f.dev, (int64_t)(nw / (size_t)b.B));
    STN_HIP(hipEventRecord(f.ready, s_));
    STN_HIP(hipStreamWaitEvent(copy_s_, f.ready, 0));
    STN_HIP(hipMemcpyAsync(f.pin, f.dev, nw * sizeof(int16_t), hipMemcpyDeviceToHost, copy_s_));
    STN_HIP(hipEventRecord(f.done, copy_s_));
    f.n = nw;
    f.dur = reported_dur_;
    f.busy = true;
}
void Engine::batch_fetch_pcm16_end(int slot, const int16_t** pcm, size_t* n, float* duration) {
    if (slot < 0 || slot > 1) throw std::invalid_argument("fetch slot must be 0 or 1");
    FetchSlot& f = fetch_[slot];
    if (!f.busy) throw std::runtime_error("no fetch in flight on this slot");
    STN_HIP(hipEventSynchronize(f.done));
    if (pcm) *pcm = f.pin;
    if (n) *n = f.n;
    if (duration) std::copy(f.dur.begin(), f.dur.end(), duration);
}
void Engine::batch_copy_wav_device(float* dst, int64_t dst_stride) {
    Batch& b = bt_;
    const size_t W = (size_t)b.L * a_.base_chunk_size * a_.chunk_compress_factor;
    if (!b.wav || b.L == 0) throw std::runtime_error("no finished batch");
    if ((size_t)dst_stride < W) throw std::invalid_argument("dst_stride smaller than the waveform length");
    STN_HIP(hipMemcpy2DAsync(dst, (size_t)dst_stride * 4, b.wav, W * 4, W * 4, (size_t)b.B, hipMemcpyDeviceToDevice, s_));
}
void Engine::batch_copy_pcm16_device(int16_t* dst, int64_t dst_stride) {
    STN_HIP(hipSetDevice(device_));
    Batch& b = bt_;
    const size_t W = (size_t)b.L * a_.base_chunk_size * a_.chunk_compress_factor;
    if (!b.wav || b.L == 0) throw std::runtime_error("no finished batch");
    if ((size_t)dst_stride < W) throw std::invalid_argument("dst_stride smaller than the waveform length");
    launch_f32_to_pcm16(s_, b.wav, (int64_t)b.B, (int)W, dst, dst_stride);
}
void Engine::batch_fetch_latent(float* latent) {
    Batch& b = bt_;
    const size_t nx = (size_t)b.B * a_.latent_dim * a_.chunk_compress_factor * b.L;
    STN_HIP(hipMemcpyAsync(latent, b.xt[final_xt_], nx * 4, hipMemcpyDeviceToHost, s_));
    sync();
}

}  // namespace stn
