// json_min.hpp — a small recursive-descent JSON reader (the reference links nlohmann/json, which this image
// does not have).  Enough for tts.json, unicode_indexer.json and the voice-style files.
#pragma once
#include <string>
#include <utility>
#include <vector>

namespace stn {
namespace json {

struct Value {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    double num = 0;
    bool boolean = false;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    bool is_array() const { return type == Array; }
    bool is_object() const { return type == Object; }
    bool has(const std::string& key) const;
    const Value& at(const std::string& key) const;  // throws std::runtime_error naming the missing key
    const Value& at(size_t i) const;
    int as_int() const;
    // depth-first flatten of nested numeric arrays (voice-style "data": [[[...]]])
    void flatten_numbers(std::vector<float>& out) const;
};

Value parse(const std::string& text);  // throws std::runtime_error with a byte offset on malformed input

}  // namespace json
}  // namespace stn
