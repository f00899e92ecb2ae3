// example_native.cpp — command-line driver with the reference's flags (/root/reference/cpp/example_onnx.cpp:35-50):
//   --onnx-dir --total-step --speed --n-test --voice-style --text --lang --save-dir --batch
// plus engine flags: --device N, --gpus N (deal the batch over N devices: include/stn_group.h), --devices a,b,.. (explicit ordinals),
// --dtype {fp32,bf16,fp16}, --seed S (0 = unseeded noise, like the reference).
// Voice styles: paths to voice-style JSON files; when the model assets are absent (synthetic weights) a
// non-existing path is taken as a voice NAME and mapped to a deterministic synthetic style.
#include <sys/stat.h>

#include <cstdlib>
#include <fstream>
#include <iostream>

#include "../host/tts_host.hpp"

using namespace stn::host;

namespace {
std::vector<std::string> split(const std::string& s, char delim) {
    std::vector<std::string> out;
    size_t a = 0, p;
    while ((p = s.find(delim, a)) != std::string::npos) { out.push_back(s.substr(a, p - a)); a = p + 1; }
    out.push_back(s.substr(a));
    return out;
}
bool exists(const std::string& p) { std::ifstream f(p); return f.is_open(); }
void make_dirs(const std::string& path) {
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') ::mkdir(path.substr(0, i).c_str(), 0755);
}
}  // namespace

int main(int argc, char* argv[]) {
    std::cout << "=== TTS Inference on MI355X (native HIP engine) ===\n\n";
    std::string onnx_dir = "../assets/onnx", save_dir = "results";
    int total_step = 5, n_test = 4;
    float speed = 1.05f;
    std::vector<std::string> voice_style = {"../assets/voice_styles/M1.json"};
    std::vector<std::string> text = {"This morning, I took a walk in the park, and the sound of the birds and the breeze was so "
                                     "pleasant that I stopped for a long time just to listen."};
    std::vector<std::string> lang = {"en"};
    bool batch = false;
    EngineOptions opts;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        const bool more = i + 1 < argc;
        if (a == "--onnx-dir" && more) onnx_dir = argv[++i];
        else if (a == "--total-step" && more) total_step = std::atoi(argv[++i]);
        else if (a == "--speed" && more) speed = (float)std::atof(argv[++i]);
        else if (a == "--n-test" && more) n_test = std::atoi(argv[++i]);
        else if (a == "--voice-style" && more) voice_style = split(argv[++i], ',');
        else if (a == "--text" && more) text = split(argv[++i], '|');
        else if (a == "--lang" && more) lang = split(argv[++i], ',');
        else if (a == "--save-dir" && more) save_dir = argv[++i];
        else if (a == "--batch") batch = true;
        else if (a == "--device" && more) opts.device = std::atoi(argv[++i]);
        else if (a == "--gpus" && more) opts.gpus = std::atoi(argv[++i]);  // > 1: the batch is dealt over devices device .. device + N - 1
        else if (a == "--devices" && more) { for (const std::string& d : split(argv[++i], ',')) opts.devices.push_back(std::atoi(d.c_str())); }
        else if (a == "--dtype" && more) { const std::string d = argv[++i]; opts.dtype = d == "fp32" ? STN_DTYPE_F32 : d == "fp16" ? STN_DTYPE_F16 : STN_DTYPE_BF16; }
        else if (a == "--seed" && more) opts.noise_seed = std::strtoull(argv[++i], nullptr, 10);
        else if (a == "--synthetic") opts.allow_synthetic = true;  // no model assets: run the default architecture on synthetic weights
    }
    if (voice_style.size() != text.size()) {
        std::cerr << "Error: Number of voice styles (" << voice_style.size() << ") must match number of texts (" << text.size() << ")\n";
        return 1;
    }
    if (lang.size() != text.size()) {
        std::cerr << "Error: Number of languages (" << lang.size() << ") must match number of texts (" << text.size() << ")\n";
        return 1;
    }
    const int bsz = (int)voice_style.size();
    try {
        auto tts = loadTextToSpeech(onnx_dir, true, opts);
        std::cout << std::endl;
        stn_arch arch;
        stn_get_arch(tts->engine(), &arch);
        // voice styles are files (loadVoiceStyle throws on a missing one, cpp/helper.cpp:835); only an engine on synthetic
        // weights, given NO existing file at all, maps the names to deterministic synthetic styles
        bool any_file = false;
        for (auto& p : voice_style) any_file = any_file || exists(p);
        const bool by_name = tts->synthetic() && !any_file;
        const Style style = by_name ? syntheticVoiceStyle(voice_style, arch) : loadVoiceStyle(voice_style, true);
        if (by_name) std::cout << "Voice style files not found -> synthetic styles keyed by name" << std::endl;
        make_dirs(save_dir);
        for (int n = 0; n < n_test; ++n) {
            std::cout << "\n[" << (n + 1) << "/" << n_test << "] Starting synthesis...\n";
            auto result = timer("Generating speech from text", [&]() {
                return batch ? tts->batch(text, lang, style, total_step, speed) : tts->call(text[0], lang[0], style, total_step, speed);
            });
            const int sr = tts->getSampleRate();
            const size_t per = result.wav.size() / (size_t)bsz;
            for (int b = 0; b < bsz; ++b) {
                const std::string fname = sanitizeFilename(text[b], 20) + "_" + std::to_string(n + 1) + ".wav";
                size_t wav_len = (size_t)(int)((float)sr * result.duration[b]);
                if (wav_len > per) wav_len = per;
                std::vector<float> out(result.wav.begin() + (long)(b * per), result.wav.begin() + (long)(b * per + wav_len));
                writeWavFile(save_dir + "/" + fname, out, sr);
                std::cout << "Saved: " << save_dir << "/" << fname << "\n";
            }
        }
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        return 2;
    }
    std::cout << "\n=== Synthesis completed successfully! ===\n";
    return 0;
}
