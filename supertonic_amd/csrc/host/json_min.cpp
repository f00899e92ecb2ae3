#include "json_min.hpp"

#include <cmath>
#include <cstdlib>
#include <stdexcept>

namespace stn {
namespace json {

bool Value::has(const std::string& key) const {
    for (auto& kv : obj) if (kv.first == key) return true;
    return false;
}
const Value& Value::at(const std::string& key) const {
    if (type != Object) throw std::runtime_error("JSON: not an object while looking up \"" + key + "\"");
    for (auto& kv : obj) if (kv.first == key) return kv.second;
    throw std::runtime_error("JSON: missing key \"" + key + "\"");
}
const Value& Value::at(size_t i) const {
    if (type != Array || i >= arr.size()) throw std::runtime_error("JSON: array index out of range");
    return arr[i];
}
int Value::as_int() const {
    if (type != Number) throw std::runtime_error("JSON: expected a number");
    return (int)num;
}
void Value::flatten_numbers(std::vector<float>& out) const {
    if (type == Number) { out.push_back((float)num); return; }
    if (type != Array) throw std::runtime_error("JSON: expected nested numeric arrays");
    for (const Value& v : arr) v.flatten_numbers(out);
}

namespace {
struct Parser {
    const std::string& s;
    size_t i = 0;
    explicit Parser(const std::string& t) : s(t) {}
    [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("JSON parse error at byte ") + std::to_string(i) + ": " + what); }
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; }
    bool eat(char c) { ws(); if (i < s.size() && s[i] == c) { ++i; return true; } return false; }
    void expect(char c) { if (!eat(c)) fail("unexpected character"); }

    static void utf8(std::string& o, unsigned cp) {
        if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    unsigned hex4() {
        if (i + 4 > s.size()) fail("truncated \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; ++k) {
            const char c = s[i++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else fail("bad hex digit");
        }
        return v;
    }
    std::string string() {
        expect('"');
        std::string o;
        while (true) {
            if (i >= s.size()) fail("unterminated string");
            const char c = s[i++];
            if (c == '"') return o;
            if (c != '\\') { o.push_back(c); continue; }
            if (i >= s.size()) fail("unterminated escape");
            const char e = s[i++];
            switch (e) {
                case '"': o.push_back('"'); break;   case '\\': o.push_back('\\'); break; case '/': o.push_back('/'); break;
                case 'b': o.push_back('\b'); break;  case 'f': o.push_back('\f'); break;  case 'n': o.push_back('\n'); break;
                case 'r': o.push_back('\r'); break;  case 't': o.push_back('\t'); break;
                case 'u': {
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && i + 1 < s.size() && s[i] == '\\' && s[i + 1] == 'u') {
                        i += 2;
                        const unsigned lo = hex4();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    utf8(o, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
    }
    Value value(int depth) {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (i >= s.size()) fail("unexpected end of input");
        Value v;
        const char c = s[i];
        if (c == '{') {
            ++i; v.type = Value::Object;
            if (eat('}')) return v;
            do { ws(); std::string k = string(); expect(':'); v.obj.emplace_back(std::move(k), value(depth + 1)); } while (eat(','));
            expect('}');
        } else if (c == '[') {
            ++i; v.type = Value::Array;
            if (eat(']')) return v;
            do { v.arr.push_back(value(depth + 1)); } while (eat(','));
            expect(']');
        } else if (c == '"') {
            v.type = Value::String; v.str = string();
        } else if (s.compare(i, 4, "true") == 0) { i += 4; v.type = Value::Bool; v.boolean = true; }
        else if (s.compare(i, 5, "false") == 0) { i += 5; v.type = Value::Bool; }
        else if (s.compare(i, 4, "null") == 0) { i += 4; }
        else {
            const char* b = s.c_str() + i; char* e = nullptr;
            v.num = std::strtod(b, &e);
            if (e == b) fail("expected a value");
            i += (size_t)(e - b); v.type = Value::Number;
        }
        return v;
    }
};
}  // namespace

Value parse(const std::string& text) {
    Parser p(text);
    Value v = p.value(0);
    p.ws();
    if (p.i != text.size()) p.fail("trailing characters");
    return v;
}

}  // namespace json
}  // namespace stn
