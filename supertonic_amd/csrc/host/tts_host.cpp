#include "tts_host.hpp"

#include <cmath>
#include <fstream>
#include <random>
#include <stdexcept>

#include "json_min.hpp"

namespace stn {
namespace host {

namespace {
void check(stn_handle* h, int rc) {
    if (rc != STN_OK) throw std::runtime_error(std::string("engine: ") + stn_last_error(h));
}
}  // namespace

TextToSpeech::TextToSpeech(stn_handle* engine, UnicodeProcessor tp, const Config& cfgs, uint64_t noise_seed)
    : h_(engine), text_processor_(std::move(tp)), cfgs_(cfgs), noise_seed_(noise_seed) {}
TextToSpeech::TextToSpeech(stn_group* group, UnicodeProcessor tp, const Config& cfgs, uint64_t noise_seed)
    : h_(stn_group_handle(group, 0)), grp_(group), text_processor_(std::move(tp)), cfgs_(cfgs), noise_seed_(noise_seed) {}
TextToSpeech::~TextToSpeech() {
    if (grp_) stn_group_destroy(grp_);  // (owns the handles)
    else stn_destroy(h_);
}

TextToSpeech::SynthesisResult TextToSpeech::infer(const std::vector<std::string>& text_list,
                                                  const std::vector<std::string>& lang_list, const Style& style,
                                                  int total_step, float speed) {
    const int bsz = (int)text_list.size();
    if (bsz != style.getTtlShape()[0]) throw std::runtime_error("Number of texts must match number of style vectors");
    const TokenBatch tb = text_processor_(text_list, lang_list);
    const std::vector<float> mask = tb.mask();
    if (grp_) {
        // several devices: deal, synthesize, gather the 16-bit PCM into the first device (RCCL) and fetch it in caller order.  The
        // float waveform handed back is the centre-ish representative of each sample's quantisation cell, (pcm +- 0.5) / 32767, whose
        // re-quantisation by writeWavFile (clamp, * 32767, truncation: cpp/helper.cpp:986-987) gives exactly the gathered PCM.
        uint64_t seed = noise_seed_;
        if (seed == 0) { std::random_device rd; seed = ((uint64_t)rd() << 32) | rd(); }
        else seed += calls_;
        ++calls_;
        int64_t W = 0;
        if (stn_group_synthesize(grp_, bsz, tb.Lt, tb.ids.data(), mask.data(), style.getTtlData().data(), style.getDpData().data(), total_step, speed,
                                 nullptr, seed, &W) != STN_OK)
            throw std::runtime_error(std::string("engine group: ") + stn_group_last_error(grp_));
        std::vector<int16_t> pcm((size_t)bsz * (size_t)W);
        SynthesisResult r;
        r.duration.resize(bsz);
        if (stn_group_fetch_pcm16(grp_, pcm.data(), pcm.size(), r.duration.data()) != STN_OK)
            throw std::runtime_error(std::string("engine group: ") + stn_group_last_error(grp_));
        r.wav.resize(pcm.size());
        for (size_t i = 0; i < pcm.size(); ++i) r.wav[i] = pcm[i] == 0 ? 0.f : ((float)pcm[i] + (pcm[i] > 0 ? 0.5f : -0.5f)) / 32767.0f;
        return r;
    }
    check(h_, stn_batch_upload(h_, bsz, tb.Lt, tb.ids.data(), mask.data(), style.getTtlData().data(),
                               style.getDpData().data(), nullptr, nullptr));
    uint64_t seed = noise_seed_;
    if (seed == 0) { std::random_device rd; seed = ((uint64_t)rd() << 32) | rd(); }  // unseeded, like cpp/helper.cpp:442-444
    else seed += calls_;
    ++calls_;
    check(h_, stn_batch_run(h_, total_step, speed, seed));
    int B = 0, L = 0;
    int64_t W = 0;
    check(h_, stn_batch_dims(h_, &B, &L, &W));
    SynthesisResult r;
    r.wav.resize((size_t)B * W);
    r.duration.resize(B);
    check(h_, stn_batch_fetch(h_, r.wav.data(), r.wav.size(), r.duration.data()));
    return r;
}

TextToSpeech::SynthesisResult TextToSpeech::call(const std::string& text, const std::string& lang, const Style& style,
                                                 int total_step, float speed, float silence_duration) {
    if (style.getTtlShape()[0] != 1) throw std::runtime_error("Single speaker text to speech only supports single style");
    const std::vector<std::string> chunks = chunk_text(text, lang == "ko" ? 120 : 300);
    if (chunks.size() == 1) return infer({chunks[0]}, {lang}, style, total_step, speed);
    // The reference synthesizes the chunks one after another (cpp/helper.cpp:697-719: one _infer, i.e. four Run calls per
    // step, per chunk).  Here they form ONE batch with the speaker's style replicated; the length-aware vocoder mode makes
    // every chunk's wave over its own frames what the batch-of-one run gives, so the joined result keeps the reference's
    // semantics (untrimmed chunk waves of L_i * chunk_size samples, zeros in between) at one pipeline pass.
    const int n = (int)chunks.size();
    const std::vector<int64_t>& ts = style.getTtlShape();
    const std::vector<int64_t>& ds = style.getDpShape();
    std::vector<float> ttl, dp;
    ttl.reserve(style.getTtlData().size() * n);
    dp.reserve(style.getDpData().size() * n);
    for (int i = 0; i < n; ++i) {
        ttl.insert(ttl.end(), style.getTtlData().begin(), style.getTtlData().end());
        dp.insert(dp.end(), style.getDpData().begin(), style.getDpData().end());
    }
    const Style rep(std::move(ttl), {n, ts[1], ts[2]}, std::move(dp), {n, ds[1], ds[2]});
    struct ModeGuard {  // (every rank's engine when the chunks are dealt over a group)
        std::vector<stn_handle*> hs;
        ModeGuard(stn_handle* h0, stn_group* grp) {
            if (grp) for (int r = 0; r < stn_group_size(grp); ++r) hs.push_back(stn_group_handle(grp, r));
            else hs.push_back(h0);
            for (stn_handle* h : hs) { check(h, stn_set_vocoder_mode(h, 1)); check(h, stn_set_shape_buckets(h, 1)); }  // chunk batches of any
                                                                                                                      // length share captured graphs
        }
        ~ModeGuard() { for (stn_handle* h : hs) { (void)stn_set_vocoder_mode(h, 0); (void)stn_set_shape_buckets(h, 0); } }
    } guard(h_, grp_);
    const SynthesisResult r = infer(chunks, std::vector<std::string>((size_t)n, lang), rep, total_step, speed);
    const size_t W = r.wav.size() / (size_t)n;
    const int chunk_size = cfgs_.ae.base_chunk_size * cfgs_.ttl.chunk_compress_factor;
    const size_t n_sil = (size_t)(int)(silence_duration * (float)cfgs_.ae.sample_rate);
    SynthesisResult out;
    float dur_cat = 0.f;
    for (int i = 0; i < n; ++i) {
        const LatentGeometry g = latent_geometry({r.duration[(size_t)i]}, cfgs_.ae.sample_rate, cfgs_.ae.base_chunk_size,
                                                 cfgs_.ttl.chunk_compress_factor, cfgs_.ttl.latent_dim);
        const size_t n_i = (size_t)g.L * (size_t)chunk_size;  // the wav length the chunk's own run would return
        if (i > 0) {  // untrimmed chunk waves joined by zeros (cpp/helper.cpp:706-715)
            out.wav.insert(out.wav.end(), n_sil, 0.0f);
            dur_cat += r.duration[(size_t)i] + silence_duration;
        } else {
            dur_cat = r.duration[0];
        }
        out.wav.insert(out.wav.end(), r.wav.begin() + (size_t)i * W, r.wav.begin() + (size_t)i * W + std::min(n_i, W));
    }
    out.duration = {dur_cat};
    return out;
}

TextToSpeech::SynthesisResult TextToSpeech::batch(const std::vector<std::string>& text_list,
                                                  const std::vector<std::string>& lang_list, const Style& style,
                                                  int total_step, float speed) {
    return infer(text_list, lang_list, style, total_step, speed);
}

std::unique_ptr<TextToSpeech> loadTextToSpeech(const std::string& onnx_dir, bool use_gpu, const EngineOptions& opts) {
    if (!use_gpu) throw std::runtime_error("CPU mode is not supported: this engine runs on MI355X only");
    const char* dt_name = opts.dtype == STN_DTYPE_BF16 ? "bf16" : opts.dtype == STN_DTYPE_F16 ? "fp16" : "fp32";
    stn_handle* h = nullptr;
    stn_group* grp = nullptr;
    if (opts.gpus > 1 || !opts.devices.empty()) {
        std::vector<int> dev = opts.devices;
        if (dev.empty()) for (int i = 0; i < opts.gpus; ++i) dev.push_back(opts.device + i);
        if (stn_group_create((int)dev.size(), dev.data(), opts.dtype, &grp) != STN_OK) throw std::runtime_error(std::string("engine group: ") + stn_group_last_error(nullptr));
        h = stn_group_handle(grp, 0);
        std::cout << "Using " << dev.size() << " x MI355X (" << dt_name << ", utterances dealt by length, PCM gathered into device " << dev[0]
                  << (stn_group_uses_rccl(grp) ? " over RCCL" : " by device copies: ranks share a GPU") << ") for inference" << std::endl;
    } else {
        stn_config cfg{opts.device, opts.dtype};
        if (stn_create(&cfg, &h) != STN_OK) throw std::runtime_error(std::string("engine: ") + stn_last_error(nullptr));
        std::cout << "Using MI355X (HIP device " << opts.device << ", " << dt_name << ") for inference" << std::endl;
    }
    auto load_dir = [&]() { return grp ? stn_group_load_dir(grp, onnx_dir.c_str()) : stn_load_dir(h, onnx_dir.c_str()); };
    auto load_err = [&]() { return std::string(grp ? stn_group_last_error(grp) : stn_last_error(h)); };
    try {
        const int rc = load_dir();
        bool synthetic = false;
        Config cfgs;
        UnicodeProcessor tp;
        if (rc == STN_OK) {
            cfgs = loadCfgs(onnx_dir);
            tp = UnicodeProcessor::from_file(onnx_dir + "/unicode_indexer.json");
        } else if (opts.allow_synthetic) {
            std::cout << "  model assets unavailable (" << load_err() << ")\n"
                      << "  -> synthetic weights from the default architecture descriptor (seed " << opts.weight_seed << ")" << std::endl;
            stn_arch a;
            stn_arch_default(&a);
            if (grp) { if (stn_group_load_synthetic(grp, &a, opts.weight_seed) != STN_OK) throw std::runtime_error(load_err()); }
            else check(h, stn_load_synthetic(h, &a, opts.weight_seed));
            cfgs.ae.sample_rate = a.sample_rate; cfgs.ae.base_chunk_size = a.base_chunk_size;
            cfgs.ttl.chunk_compress_factor = a.chunk_compress_factor; cfgs.ttl.latent_dim = a.latent_dim;
            std::vector<int64_t> idx(65536);  // synthetic indexer of the same shape as unicode_indexer.json
            for (int cp = 0; cp < 65536; ++cp) idx[cp] = cp < 384 ? cp : 384 + (cp % 128);
            tp = UnicodeProcessor(std::move(idx));
            synthetic = true;
        } else {
            throw std::runtime_error(load_err());
        }
        auto tts = grp ? std::make_unique<TextToSpeech>(grp, std::move(tp), cfgs, opts.noise_seed)
                       : std::make_unique<TextToSpeech>(h, std::move(tp), cfgs, opts.noise_seed);
        if (synthetic) tts->markSynthetic();
        return tts;
    } catch (...) {
        if (grp) stn_group_destroy(grp); else stn_destroy(h);
        throw;
    }
}

}  // namespace host
}  // namespace stn
