// host_fuzz.cpp — driver of the sanitizer build of the host-side parsers (make host-asan): every file of every directory named on the
// command line goes through the entry points that read caller-supplied bytes — stn_onnx_summary (protobuf reader), stn_bind_graphs /
// stn_bound_tensor (graph walk over tts.json + the four graphs), stn_load_voice_style (JSON), the text frontend (stn_text_preprocess,
// stn_text_to_ids, stn_chunk_text, stn_sanitize_filename over the lines of *.txt) — and must come back with a result or an STN_ERR_*
// code.  A sanitizer report aborts the process (-fno-sanitize-recover): exit code 0 means every input was handled.
// Stands in for the robustness the reference delegates to ONNX Runtime and nlohmann/json (/root/reference/cpp/helper.cpp:784-823,
// 829-897, 1054-1064).  CPU only; tests/test_host_asan_cpu.py builds the corpus.
#include <dirent.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../include/stn.h"
#include "../include/stn_host.h"

static std::vector<std::string> list_dir(const std::string& d) {
    std::vector<std::string> out;
    if (DIR* dir = opendir(d.c_str())) {
        while (dirent* e = readdir(dir))
            if (e->d_name[0] != '.') out.push_back(e->d_name);
        closedir(dir);
    }
    return out;
}
static bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

int main(int argc, char** argv) {
    long calls = 0, errors = 0;
    std::vector<char> buf(1 << 20);
    std::vector<float> fbuf(1 << 18);
    for (int a = 1; a < argc; ++a) {
        const std::string dir = argv[a];
        int64_t r = stn_bind_graphs(dir.c_str(), buf.data(), buf.size());
        ++calls; errors += r < 0;
        for (const char* name : {"vo.head.w", "no.such.tensor"}) {
            r = stn_bound_tensor(dir.c_str(), name, fbuf.data(), fbuf.size());
            ++calls; errors += r < 0;
        }
        for (const std::string& f : list_dir(dir)) {
            const std::string path = dir + "/" + f;
            if (ends_with(f, ".onnx")) {
                r = stn_onnx_summary(path.c_str(), buf.data(), buf.size());
                ++calls; errors += r < 0;
            } else if (ends_with(f, ".json")) {
                const char* paths[2] = {path.c_str(), path.c_str()};
                int64_t dims[6] = {0, 0, 0, 0, 0, 0};
                int rc = stn_load_voice_style(paths, 2, nullptr, 0, nullptr, 0, dims);
                ++calls; errors += rc < 0;
                if (rc >= 0 && dims[0] > 0 && dims[1] > 0 && dims[2] > 0 && dims[4] > 0 && dims[5] > 0 && dims[0] * dims[1] * dims[2] < (1 << 18) && dims[3] * dims[4] * dims[5] < (1 << 18)) {
                    std::vector<float> t((size_t)(dims[0] * dims[1] * dims[2])), d((size_t)(dims[3] * dims[4] * dims[5]));
                    rc = stn_load_voice_style(paths, 2, t.data(), t.size(), d.data(), d.size(), dims);
                    ++calls; errors += rc < 0;
                }
            } else if (ends_with(f, ".txt")) {
                std::ifstream in(path, std::ios::binary);
                std::string line;
                std::vector<int64_t> indexer(65536);
                for (int cp = 0; cp < 65536; ++cp) indexer[cp] = cp < 384 ? cp : (cp % 7 == 0 ? -1 : 384 + cp % 128);
                while (std::getline(in, line)) {
                    for (const char* lang : {"en", "ko", "xx"}) {
                        r = stn_text_preprocess(line.c_str(), lang, buf.data(), buf.size());
                        ++calls; errors += r < 0;
                        const char* texts[1] = {line.c_str()};
                        const char* langs[1] = {lang};
                        std::vector<int64_t> ids(4096);
                        int32_t len = 0;
                        int Lt = 0;
                        int rc = stn_text_to_ids(indexer.data(), indexer.size(), texts, langs, 1, ids.data(), 4096, &len, &Lt);
                        ++calls; errors += rc < 0;
                    }
                    int nch = 0;
                    for (int max_len : {0, 1, 7, 300}) {
                        r = stn_chunk_text(line.c_str(), max_len, buf.data(), buf.size(), &nch);
                        ++calls; errors += r < 0;
                    }
                    for (int max_len : {0, 3, 20}) {
                        r = stn_sanitize_filename(line.c_str(), max_len, buf.data(), buf.size());
                        ++calls; errors += r < 0;
                    }
                }
            }
        }
    }
    std::printf("{\"dirs\": %d, \"calls\": %ld, \"errors_returned\": %ld}\n", argc - 1, calls, errors);
    return 0;
}
