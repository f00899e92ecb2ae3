// text_frontend.cpp — see text_frontend.hpp.  Every function states the reference behaviour it reproduces.
#include "text_frontend.hpp"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include "json_min.hpp"

namespace stn {
namespace host {

const char* const kLanguages[5] = {"en", "ko", "es", "pt", "fr"};  // cpp/helper.cpp:15

bool language_supported(const std::string& lang) {
    for (const char* l : kLanguages)
        if (lang == l) return true;
    return false;
}

namespace {

inline bool c_space(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }  // isspace, "C" locale

std::string trimmed(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && c_space((unsigned char)s[a])) ++a;
    while (b > a && c_space((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

// one left-to-right, non-overlapping substitution pass (what the reference's find/replace loop and its
// single regex_replace calls amount to)
void substitute(std::string& s, const char* from, const char* to) {
    const size_t nf = std::strlen(from), nt = std::strlen(to);
    if (!nf) return;
    std::string out;
    out.reserve(s.size());
    size_t i = 0;
    while (i < s.size()) {
        if (s.compare(i, nf, from) == 0) { out.append(to, nt); i += nf; }
        else out.push_back(s[i++]);
    }
    s.swap(out);
}

struct Sub { const char* from; const char* to; };

// symbol map, applied in this order before anything else (cpp/helper.cpp:69-95)
const Sub kSymbols[] = {
    {"\xE2\x80\x93", "-"}, {"\xE2\x80\x91", "-"}, {"\xE2\x80\x94", "-"}, {"_", " "},
    {"\xE2\x80\x9C", "\""}, {"\xE2\x80\x9D", "\""}, {"\xE2\x80\x98", "'"}, {"\xE2\x80\x99", "'"},
    {"\xC2\xB4", "'"}, {"`", "'"}, {"[", " "}, {"]", " "}, {"|", " "}, {"/", " "}, {"#", " "},
    {"\xE2\x86\x92", " "}, {"\xE2\x86\x90", " "},
};
// dropped outright (cpp/helper.cpp:105-111): heart, star, white heart, copyright, backslash
const char* const kDropped[] = {"\xE2\x99\xA5", "\xE2\x98\x86", "\xE2\x99\xA1", "\xC2\xA9", "\\"};
const Sub kExpressions[] = {{"@", " at "}, {"e.g.,", "for example, "}, {"i.e.,", "that is, "}};  // :114-126
const Sub kPunctSpacing[] = {{" ,", ","}, {" .", "."}, {" !", "!"}, {" ?", "?"}, {" ;", ";"}, {" :", ":"}, {" '", "'"}};
// multi-byte sentence enders the reference compares against the LAST THREE BYTES (cpp/helper.cpp:166-176);
// U+00BB is two bytes long and therefore never matches there — reproduced on purpose.
const char* const kEnders3[] = {"\xE2\x80\xA6", "\xE3\x80\x82", "\xE3\x80\x8D", "\xE3\x80\x8F", "\xE3\x80\x91",
                                "\xE3\x80\x89", "\xE3\x80\x8B", "\xE2\x80\xBA", "\xC2\xBB",
                                "\xE2\x80\x9C", "\xE2\x80\x9D", "\xE2\x80\x98", "\xE2\x80\x99"};

void collapse_pairs(std::string& s, const char* pair, const char* single) {  // while (find(pair)) replace first
    for (size_t p; (p = s.find(pair)) != std::string::npos;) s.replace(p, 2, single);
}

}  // namespace

std::string preprocess_text(const std::string& text, const std::string& lang) {
    std::string s = text;
    for (const Sub& r : kSymbols) substitute(s, r.from, r.to);
    {  // 4-byte sequences F0 9F xx xx (emoji planes) vanish (cpp/helper.cpp:99-102)
        std::string out;
        out.reserve(s.size());
        for (size_t i = 0; i < s.size();) {
            const unsigned char c0 = s[i];
            if (c0 == 0xF0 && i + 3 < s.size() && (unsigned char)s[i + 1] == 0x9F &&
                ((unsigned char)s[i + 2] & 0xC0) == 0x80 && ((unsigned char)s[i + 3] & 0xC0) == 0x80) { i += 4; continue; }
            out.push_back(s[i++]);
        }
        s.swap(out);
    }
    for (const char* d : kDropped) substitute(s, d, "");
    for (const Sub& r : kExpressions) substitute(s, r.from, r.to);
    for (const Sub& r : kPunctSpacing) substitute(s, r.from, r.to);
    collapse_pairs(s, "\"\"", "\"");
    collapse_pairs(s, "''", "'");
    collapse_pairs(s, "``", "`");
    {  // whitespace runs -> one space, then trim (cpp/helper.cpp:152-153)
        std::string out;
        out.reserve(s.size());
        bool in_ws = false;
        for (unsigned char c : s) {
            if (c_space(c)) { if (!in_ws) out.push_back(' '); in_ws = true; }
            else { out.push_back((char)c); in_ws = false; }
        }
        s = trimmed(out);
    }
    if (!s.empty()) {  // terminal punctuation (cpp/helper.cpp:156-182); an empty text stays empty
        bool ended = std::strchr(".!?;:,'\")]}>", s.back()) != nullptr;
        if (!ended && s.size() >= 3) {
            const std::string tail = s.substr(s.size() - 3);
            for (const char* e : kEnders3)
                if (tail == e) { ended = true; break; }
        }
        if (!ended) s.push_back('.');
    }
    if (!language_supported(lang)) throw std::runtime_error("Invalid language: " + lang + ". Available: en, ko, es, pt, fr");
    return "<" + lang + ">" + s + "</" + lang + ">";
}

namespace {
// precomposed Latin letters of es/pt/fr -> base letter + combining mark (cpp/helper.cpp:214-269)
bool latin_decompose(uint32_t cp, uint16_t& base, uint16_t& mark) {
    struct Row { uint16_t mark; const uint16_t* cps; const char* letters; };
    static const uint16_t acute[] = {0xC1, 0xC9, 0xCD, 0xD3, 0xDA, 0xE1, 0xE9, 0xED, 0xF3, 0xFA};
    static const uint16_t grave[] = {0xC0, 0xC8, 0xCC, 0xD2, 0xD9, 0xE0, 0xE8, 0xEC, 0xF2, 0xF9};
    static const uint16_t circ[] = {0xC2, 0xCA, 0xCE, 0xD4, 0xDB, 0xE2, 0xEA, 0xEE, 0xF4, 0xFB};
    static const uint16_t diaer[] = {0xC4, 0xCB, 0xCF, 0xD6, 0xDC, 0xE4, 0xEB, 0xEF, 0xF6, 0xFC};
    static const uint16_t tilde[] = {0xC3, 0xD1, 0xD5, 0xE3, 0xF1, 0xF5};
    static const uint16_t cedil[] = {0xC7, 0xE7};
    static const Row rows[] = {{0x0301, acute, "AEIOUaeiou"}, {0x0300, grave, "AEIOUaeiou"}, {0x0302, circ, "AEIOUaeiou"},
                               {0x0308, diaer, "AEIOUaeiou"}, {0x0303, tilde, "ANOano"},     {0x0327, cedil, "Cc"}};
    if (cp < 0xC0 || cp > 0xFC) return false;
    for (const Row& r : rows)
        for (size_t i = 0; r.letters[i]; ++i)
            if (r.cps[i] == cp) { base = (uint16_t)(unsigned char)r.letters[i]; mark = r.mark; return true; }
    return false;
}
}  // namespace

std::vector<uint16_t> text_to_unicode_values(const std::string& text) {
    std::vector<uint16_t> out;
    out.reserve(text.size());
    const size_t n = text.size();
    size_t i = 0;
    while (i < n) {
        const unsigned char c = text[i];
        uint32_t cp;
        auto cont = [&](size_t k) { return (uint32_t)((unsigned char)text[i + k] & 0x3F); };
        if (c < 0x80) { cp = c; i += 1; }
        else if ((c & 0xE0) == 0xC0 && i + 1 < n) { cp = ((c & 0x1Fu) << 6) | cont(1); i += 2; }
        else if ((c & 0xF0) == 0xE0 && i + 2 < n) { cp = ((c & 0x0Fu) << 12) | (cont(1) << 6) | cont(2); i += 3; }
        else if ((c & 0xF8) == 0xF0 && i + 3 < n) { cp = ((c & 0x07u) << 18) | (cont(1) << 12) | (cont(2) << 6) | cont(3); i += 4; }
        else { i += 1; continue; }  // stray byte: skipped (cpp/helper.cpp:336-340)
        if (cp >= 0xAC00 && cp < 0xAC00 + 11172) {  // Hangul syllable -> L V (T) jamo (UAX #15 arithmetic)
            const uint32_t s = cp - 0xAC00;
            out.push_back((uint16_t)(0x1100 + s / 588));
            out.push_back((uint16_t)(0x1161 + (s % 588) / 28));
            if (s % 28) out.push_back((uint16_t)(0x11A7 + s % 28));
            continue;
        }
        uint16_t base, mark;
        if (latin_decompose(cp, base, mark)) { out.push_back(base); out.push_back(mark); continue; }
        out.push_back((uint16_t)(cp & 0xFFFF));  // beyond the BMP: truncated, as the reference does
    }
    return out;
}

std::vector<float> TokenBatch::mask() const {
    std::vector<float> m((size_t)B * Lt, 0.f);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < lengths[b] && t < Lt; ++t) m[(size_t)b * Lt + t] = 1.f;
    return m;
}

UnicodeProcessor UnicodeProcessor::from_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("Failed to open file: " + path);  // cpp/helper.cpp:1057
    std::string txt((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const json::Value v = json::parse(txt);
    if (!v.is_array()) throw std::runtime_error("unicode_indexer.json: expected a flat array of integers");
    std::vector<int64_t> idx;
    idx.reserve(v.arr.size());
    for (const json::Value& e : v.arr) idx.push_back((int64_t)e.num);
    return UnicodeProcessor(std::move(idx));
}

TokenBatch UnicodeProcessor::operator()(const std::vector<std::string>& texts, const std::vector<std::string>& langs) const {
    if (texts.size() != langs.size()) throw std::runtime_error("Number of texts must match number of languages");
    if (texts.empty()) throw std::runtime_error("empty text list");
    TokenBatch tb;
    tb.B = (int)texts.size();
    std::vector<std::vector<uint16_t>> units(tb.B);
    for (int b = 0; b < tb.B; ++b) {
        units[b] = text_to_unicode_values(preprocess_text(texts[b], langs[b]));
        tb.lengths.push_back((int32_t)units[b].size());  // code units, not bytes (cpp/helper.cpp:366-376)
        tb.Lt = std::max(tb.Lt, (int)units[b].size());
    }
    tb.ids.assign((size_t)tb.B * tb.Lt, 0);
    for (int b = 0; b < tb.B; ++b)
        for (size_t j = 0; j < units[b].size(); ++j)
            if (units[b][j] < indexer_.size()) tb.ids[(size_t)b * tb.Lt + j] = indexer_[units[b][j]];  // else stays 0 (:383-385)
    return tb;
}

std::vector<float> length_to_mask(const std::vector<int64_t>& lengths, int64_t max_len) {
    if (max_len < 0) max_len = lengths.empty() ? 0 : *std::max_element(lengths.begin(), lengths.end());
    std::vector<float> m(lengths.size() * (size_t)max_len, 0.f);
    for (size_t b = 0; b < lengths.size(); ++b)
        for (int64_t t = 0; t < max_len && t < lengths[b]; ++t) m[b * (size_t)max_len + t] = 1.f;
    return m;
}

LatentGeometry latent_geometry(const std::vector<float>& duration, int sample_rate, int base_chunk_size,
                               int chunk_compress_factor, int latent_dim) {
    if (duration.empty()) throw std::runtime_error("latent_geometry: empty duration list");
    LatentGeometry g;
    const int cs = base_chunk_size * chunk_compress_factor;
    const float wav_len_max = *std::max_element(duration.begin(), duration.end()) * (float)sample_rate;
    g.L = (int)((wav_len_max + (float)cs - 1.0f) / (float)cs);
    g.D = latent_dim * chunk_compress_factor;
    for (float d : duration) {
        const int64_t wl = (int64_t)(d * (float)sample_rate);
        g.lengths.push_back((int32_t)((wl + cs - 1) / cs));
    }
    return g;
}

// ---- chunker (cpp/helper.cpp:1117-1186) ---------------------------------------------------------------
// Paragraphs: split where "\n \s* \n+" matches; sentences: split where "[.!?] \s+" matches, every piece
// KEEPS its delimiter run; pieces are greedily re-joined with one extra space while the BYTE length stays
// <= max_len.  No abbreviation guard (unlike the Python host): "Dr. Smith" splits after "Dr.".
std::vector<std::string> chunk_text(const std::string& text, int max_len) {
    std::vector<std::string> paragraphs;
    {
        std::string cur;
        const size_t n = text.size();
        size_t i = 0;
        while (i < n) {
            if (text[i] == '\n') {
                size_t j = i + 1;
                while (j < n && c_space((unsigned char)text[j])) ++j;
                size_t k = j;  // greedy \s* backtracks to the last '\n' of the run
                while (k > i + 1 && text[k - 1] != '\n') --k;
                if (k > i + 1) { paragraphs.push_back(cur); cur.clear(); i = k; continue; }
            }
            cur.push_back(text[i++]);
        }
        paragraphs.push_back(cur);
    }
    std::vector<std::string> chunks;
    for (const std::string& raw : paragraphs) {
        const std::string para = trimmed(raw);
        if (para.empty()) continue;
        std::vector<std::string> pieces;
        const size_t n = para.size();
        size_t start = 0, i = 0;
        while (i < n) {
            const char c = para[i];
            if ((c == '.' || c == '!' || c == '?') && i + 1 < n && c_space((unsigned char)para[i + 1])) {
                size_t j = i + 1;
                while (j < n && c_space((unsigned char)para[j])) ++j;
                if (i > start) pieces.push_back(para.substr(start, j - start));  // token + its delimiter run
                start = i = j;
                continue;
            }
            ++i;
        }
        if (start < n) pieces.push_back(para.substr(start));
        std::string cur;
        for (const std::string& p : pieces) {
            if ((int)(cur.size() + p.size() + 1) <= max_len) {
                if (!cur.empty()) cur.push_back(' ');
                cur += p;
            } else {
                if (!cur.empty()) chunks.push_back(trimmed(cur));
                cur = p;
            }
        }
        if (!cur.empty()) chunks.push_back(trimmed(cur));
    }
    if (chunks.empty()) chunks.push_back(trimmed(text));  // cpp/helper.cpp:1181-1183
    return chunks;
}

// cpp/helper.cpp:1070-1111: ASCII alphanumerics and '_' kept, any multi-byte UTF-8 sequence kept whole,
// everything else -> '_'; max_len counts characters, not bytes.
std::string sanitize_filename(const std::string& text, int max_len) {
    std::string out;
    const size_t n = text.size();
    size_t i = 0;
    int count = 0;
    while (i < n && count < max_len) {
        const unsigned char c = text[i];
        size_t len = 0;
        if ((c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_') len = 1;
        else if ((c & 0xE0) == 0xC0 && i + 1 < n) len = 2;
        else if ((c & 0xF0) == 0xE0 && i + 2 < n) len = 3;
        else if ((c & 0xF8) == 0xF0 && i + 3 < n) len = 4;
        if (len) { out.append(text, i, len); i += len; }
        else { out.push_back('_'); i += 1; }
        ++count;
    }
    return out;
}

std::vector<unsigned char> wav_bytes(const float* audio, size_t n, int sample_rate) {
    std::vector<unsigned char> w(44 + n * 2);
    auto put32 = [&](size_t off, int32_t v) { std::memcpy(&w[off], &v, 4); };
    auto put16 = [&](size_t off, int16_t v) { std::memcpy(&w[off], &v, 2); };
    const int32_t data_size = (int32_t)(n * 2);
    std::memcpy(&w[0], "RIFF", 4); put32(4, 36 + data_size); std::memcpy(&w[8], "WAVEfmt ", 8);
    put32(16, 16); put16(20, 1); put16(22, 1); put32(24, sample_rate); put32(28, sample_rate * 2); put16(32, 2); put16(34, 16);
    std::memcpy(&w[36], "data", 4); put32(40, data_size);
    for (size_t i = 0; i < n; ++i) {
        const float c = std::max(-1.0f, std::min(1.0f, audio[i]));
        put16(44 + 2 * i, (int16_t)(c * 32767));  // truncation toward zero (cpp/helper.cpp:986-987)
    }
    return w;
}

void write_wav_file(const std::string& filename, const std::vector<float>& audio, int sample_rate) {
    std::ofstream f(filename, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("Failed to open file for writing: " + filename);  // cpp/helper.cpp:950
    const std::vector<unsigned char> w = wav_bytes(audio.data(), audio.size(), sample_rate);
    f.write(reinterpret_cast<const char*>(w.data()), (std::streamsize)w.size());  // one write, not one per sample
}

}  // namespace host
}  // namespace stn
