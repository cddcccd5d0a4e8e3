// json.h -- minimal JSON DOM (config.json, safetensors headers). Header-only, no dependencies.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace q3 {

struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;  // insertion order kept

    // A JSON number as an integer of the asked range; a value no such integer holds (1e30, -inf, a hostile header) is an error of
    // the file, not a conversion the language leaves undefined (tests/test_sanitizers.py feeds such files).
    static int64_t to_int(double v, double lo, double hi, const std::string& what) {
        if (!(v >= lo && v <= hi)) throw Error(6, "JSON number out of range for " + what);
        return int64_t(v);
    }
    const Json* get(const std::string& k) const {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
    bool has(const std::string& k) const {
        const Json* j = get(k);
        return j && j->kind != Null;
    }
    int64_t i64(const std::string& k, int64_t def) const {
        const Json* j = get(k);
        return (j && j->kind == Num) ? to_int(j->num, -9.2e18, 9.2e18, "'" + k + "'") : def;
    }
    double f64(const std::string& k, double def) const {
        const Json* j = get(k);
        return (j && j->kind == Num) ? j->num : def;
    }
    std::string s(const std::string& k, const std::string& def) const {
        const Json* j = get(k);
        return (j && j->kind == Str) ? j->str : def;
    }
    std::vector<int> ints(const std::string& k, std::vector<int> def) const {
        const Json* j = get(k);
        if (!j || j->kind != Arr) return def;
        std::vector<int> out;
        for (auto& e : j->arr) {
            if (e.kind != Num) throw Error(6, "JSON array '" + k + "' holds something that is not a number");
            out.push_back(int(to_int(e.num, -2147483648.0, 2147483647.0, "'" + k + "'")));
        }
        return out;
    }
};

class JsonParser {
  public:
    JsonParser(const char* p, size_t n) : p_(p), e_(p + n) {}
    Json parse() {
        depth_ = 0;
        Json j = value();
        ws();
        return j;
    }

  private:
    const char* p_;
    const char* e_;
    int depth_ = 0;  // open containers: a file of ten thousand '[' must not be able to overflow the stack
    struct Nest {
        int& d;
        explicit Nest(int& depth) : d(depth) { ++d; }
        ~Nest() { --d; }
    };
    void ws() {
        while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
    }
    [[noreturn]] void fail(const char* what) { throw Error(6, std::string("JSON parse error: ") + what); }
    Json value() {
        ws();
        if (p_ >= e_) fail("unexpected end");
        char c = *p_;
        Json j;
        Nest nest(depth_);
        if (depth_ > 128) fail("nested too deeply");
        if (c == '{') {
            j.kind = Json::Obj;
            ++p_;
            ws();
            if (p_ < e_ && *p_ == '}') { ++p_; return j; }
            for (;;) {
                ws();
                std::string k = string();
                ws();
                if (p_ >= e_ || *p_ != ':') fail("expected ':'");
                ++p_;
                Json v = value();
                j.obj.emplace_back(std::move(k), std::move(v));
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            j.kind = Json::Arr;
            ++p_;
            ws();
            if (p_ < e_ && *p_ == ']') { ++p_; return j; }
            for (;;) {
                j.arr.push_back(value());
                ws();
                if (p_ < e_ && *p_ == ',') { ++p_; continue; }
                if (p_ < e_ && *p_ == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            j.kind = Json::Str;
            j.str = string();
        } else if (c == 't' && e_ - p_ >= 4 && !strncmp(p_, "true", 4)) {
            j.kind = Json::Bool; j.b = true; p_ += 4;
        } else if (c == 'f' && e_ - p_ >= 5 && !strncmp(p_, "false", 5)) {
            j.kind = Json::Bool; j.b = false; p_ += 5;
        } else if (c == 'n' && e_ - p_ >= 4 && !strncmp(p_, "null", 4)) {
            p_ += 4;
        } else {
            char* end = nullptr;
            std::string tmp(p_, size_t(std::min<ptrdiff_t>(e_ - p_, 64)));
            j.num = strtod(tmp.c_str(), &end);
            if (end == tmp.c_str()) fail("bad number");
            p_ += (end - tmp.c_str());
            j.kind = Json::Num;
        }
        return j;
    }
    std::string string() {
        if (p_ >= e_ || *p_ != '"') fail("expected string");
        ++p_;
        std::string out;
        while (p_ < e_ && *p_ != '"') {
            if (*p_ == '\\' && p_ + 1 < e_) {
                ++p_;
                switch (*p_) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {  // keep BMP code points as UTF-8
                        if (e_ - p_ < 5) fail("bad \\u escape");
                        unsigned cp = unsigned(strtoul(std::string(p_ + 1, 4).c_str(), nullptr, 16));
                        p_ += 4;
                        if (cp >= 0xD800 && cp < 0xDC00 && e_ - p_ >= 7 && p_[1] == '\\' && p_[2] == 'u') {  // surrogate pair
                            const unsigned lo = unsigned(strtoul(std::string(p_ + 3, 4).c_str(), nullptr, 16));
                            if (lo >= 0xDC00 && lo < 0xE000) {
                                cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                                p_ += 6;
                            }
                        }
                        if (cp < 0x80) out += char(cp);
                        else if (cp < 0x800) { out += char(0xC0 | (cp >> 6)); out += char(0x80 | (cp & 0x3F)); }
                        else if (cp < 0x10000) { out += char(0xE0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
                        else { out += char(0xF0 | (cp >> 18)); out += char(0x80 | ((cp >> 12) & 0x3F)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *p_;
                }
                ++p_;
            } else {
                out += *p_++;
            }
        }
        if (p_ >= e_) fail("unterminated string");
        ++p_;
        return out;
    }
};

}  // namespace q3
