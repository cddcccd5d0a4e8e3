// host_san.cc -- TEST INFRASTRUCTURE (tests/test_sanitizers.py): the product's host-only sources -- the BPE tokeniser
// (csrc/tokenizer.cc), the JSON reader (csrc/json.h) and the safetensors directory reader (csrc/safetensors.h) -- compiled with
// g++ -fsanitize=address,undefined and driven from the command line. The GPU build cannot run under a sanitizer on this pool
// (no XNACK / GPU ASan), so the code that parses files a user hands to q3tts_model_load is checked here, on the CPU: every input,
// well-formed or mangled, must end in a result or in a q3::Error -- never in a sanitizer report.
//   host_san tok <tokenizer.json> <cases.bin>          cases.bin: u32 count, then (u32 length, bytes) per text; prints one line of ids per text
//   host_san jsonfuzz <file.json> <seed> <n>           n seeded manglings of the file through JsonParser; prints "ok A rejected B"
//   host_san stfuzz <file.safetensors> <tmpdir> <seed> <n>   the same for the header of one safetensors file through SafetensorsDir
//   host_san tokfuzz <tokenizer.json> <tmpdir> <seed> <n>    the same for a tokenizer.json through BpeTokenizer::load_json_file + encode
//   host_san cfgfuzz <config.json> <speech_tokenizer/config.json> <seed> <n>   manglings through ModelConfig::parse + validate + use
//   host_san textfuzz <tokenizer.json> <seed> <n>            n byte strings that are not (or are unusual) UTF-8 through encode
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "config.h"
#include "json.h"
#include "safetensors.h"
#include "tokenizer.h"

namespace {

std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", path.c_str());
        std::exit(2);
    }
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

struct Rng {  // splitmix64: the manglings are the same on every machine
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    size_t below(size_t n) { return n ? size_t(next() % n) : 0; }
};

// truncate, overwrite bytes, insert structural characters, duplicate a slice: the ways a damaged or hostile file differs from a good one
std::string mangle(const std::string& good, Rng& r) {
    std::string s = good;
    static const char structural[] = "{}[]\",:\\-0e.tfn\x00\xff";
    int edits = 1 + int(r.below(4));
    for (int e = 0; e < edits && !s.empty(); ++e) {
        switch (r.below(6)) {
            case 0: s.resize(r.below(s.size())); break;
            case 5: {  // a number becomes one no integer type holds
                static const char* extreme[] = {"-1", "1e308", "18446744073709551616", "9223372036854775808", "1e999", "-1e999",
                                                "4294967296", "-9223372036854775809", "0.5", "1e19"};
                size_t a = r.below(s.size());
                while (a < s.size() && !(s[a] >= '0' && s[a] <= '9')) ++a;
                size_t b = a;
                while (b < s.size() && s[b] >= '0' && s[b] <= '9') ++b;
                if (a < b) s.replace(a, b - a, extreme[r.below(sizeof(extreme) / sizeof(extreme[0]))]);
                break;
            }
            case 1: s[r.below(s.size())] = char(r.next()); break;
            case 2: s.insert(r.below(s.size() + 1), 1, structural[r.below(sizeof(structural) - 1)]); break;
            case 3: {
                size_t a = r.below(s.size()), n = r.below(std::min<size_t>(64, s.size() - a) + 1);
                s.insert(r.below(s.size() + 1), s.substr(a, n));
                break;
            }
            default: s.erase(r.below(s.size()), 1 + r.below(8)); break;
        }
    }
    return s;
}

// everything a loader would do with a parsed document: walk it, read numbers and strings by key with defaults
size_t walk(const q3::Json& j) {
    size_t n = 1;
    for (auto& kv : j.obj) n += kv.first.size() + walk(kv.second);
    for (auto& v : j.arr) n += walk(v);
    n += j.str.size();
    (void)j.i64("hidden_size", 7);
    (void)j.f64("rms_norm_eps", 1e-6);
    (void)j.s("tts_model_type", "x");
    (void)j.ints("upsample_rates", {1, 2});
    return n;
}

int cmd_tok(int argc, char** argv) {
    if (argc < 4) return 2;
    q3::BpeTokenizer t;
    t.load_json_file(argv[2]);
    std::string cases = slurp(argv[3]);
    size_t off = 0;
    auto u32 = [&]() {
        uint32_t v = 0;
        if (off + 4 > cases.size()) std::exit(2);
        std::memcpy(&v, cases.data() + off, 4);
        off += 4;
        return v;
    };
    uint32_t n = u32();
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t len = u32();
        if (off + len > cases.size()) return 2;
        std::string text = cases.substr(off, len);
        off += len;
        try {
            auto ids = t.encode(text);
            for (size_t k = 0; k < ids.size(); ++k) std::printf(k ? " %d" : "%d", ids[k]);
            std::printf("\n");
        } catch (const q3::Error& e) {
            std::printf("error %d\n", e.status);
        }
    }
    return 0;
}

int cmd_jsonfuzz(int argc, char** argv) {
    if (argc < 5) return 2;
    std::string good = slurp(argv[2]);
    Rng r{uint64_t(std::strtoull(argv[3], nullptr, 10))};
    int n = std::atoi(argv[4]), ok = 0, rejected = 0;
    {
        q3::JsonParser p(good.data(), good.size());
        if (walk(p.parse()) == 0) return 3;  // the unmangled file parses
    }
    for (const char* open : {"[", "{\"a\":"}) {  // nesting no stack holds
        std::string deep;
        for (int i = 0; i < 200000; ++i) deep += open;
        try {
            q3::JsonParser p(deep.data(), deep.size());
            (void)p.parse();
            return 3;
        } catch (const q3::Error&) {
        }
    }
    for (int i = 0; i < n; ++i) {
        std::string s = mangle(good, r);
        // an exact-size heap copy: one byte read past the end is a report, not a lucky hit in std::string's slack
        std::vector<char> buf(s.begin(), s.end());
        try {
            q3::JsonParser p(buf.data(), buf.size());
            (void)walk(p.parse());
            ++ok;
        } catch (const q3::Error&) {
            ++rejected;
        }
    }
    std::printf("ok %d rejected %d\n", ok, rejected);
    return 0;
}

// a mangled tokenizer.json: load it and, when it still loads, encode with it
int cmd_tokfuzz(int argc, char** argv) {
    if (argc < 6) return 2;
    std::string good = slurp(argv[2]);
    std::string path = std::string(argv[3]) + "/tokenizer.json";
    Rng r{uint64_t(std::strtoull(argv[4], nullptr, 10))};
    int n = std::atoi(argv[5]), ok = 0, rejected = 0;
    const std::string text = "Hello, wor\xc5\x82" "d! 123 <|im_start|>assistant\n  \xe4\xbd\xa0\xe5\xa5\xbd" " e\xcc\x81" "\t\r\n";
    for (int i = 0; i < n; ++i) {
        std::string s = mangle(good, r);
        {
            std::ofstream f(path, std::ios::binary | std::ios::trunc);
            f.write(s.data(), std::streamsize(s.size()));
        }
        try {
            q3::BpeTokenizer t;
            t.load_json_file(path);
            (void)t.encode(text);
            ++ok;
        } catch (const q3::Error&) {
            ++rejected;
        }
    }
    std::printf("ok %d rejected %d\n", ok, rejected);
    return 0;
}

// text that is not UTF-8, or is unusual UTF-8, through a good tokenizer
int cmd_textfuzz(int argc, char** argv) {
    if (argc < 5) return 2;
    q3::BpeTokenizer t;
    t.load_json_file(argv[2]);
    Rng r{uint64_t(std::strtoull(argv[3], nullptr, 10))};
    int n = std::atoi(argv[4]), ok = 0, rejected = 0;
    for (int i = 0; i < n; ++i) {
        std::string s;
        size_t len = r.below(48);
        for (size_t k = 0; k < len; ++k) {
            switch (r.below(4)) {
                case 0: s += char(r.next()); break;                                            // any byte: broken sequences, overlongs, lone continuation bytes
                case 1: s += char(0x20 + r.below(0x5f)); break;
                case 2: s += "\xf4\x90\x80\x80"; break;                                        // beyond U+10FFFF
                default: { static const char* bits[] = {"\xed\xa0\x80", "\xef\xbf\xbf", "\xcc\x81", "<|im_", "\xe1\x84\x80\xe1\x85\xa1\xe1\x86\xa8", "\r\n\r\n", "   "};
                           s += bits[r.below(7)]; }
            }
        }
        try {
            (void)t.encode(s);
            ++ok;
        } catch (const q3::Error&) {
            ++rejected;
        }
    }
    std::printf("ok %d rejected %d\n", ok, rejected);
    return 0;
}

// config.json / speech_tokenizer/config.json through ModelConfig: parse, validate, then use it the way the loader does
int cmd_cfgfuzz(int argc, char** argv) {
    if (argc < 6) return 2;
    std::string main_good = slurp(argv[2]), st_good = slurp(argv[3]);
    Rng r{uint64_t(std::strtoull(argv[4], nullptr, 10))};
    int n = std::atoi(argv[5]), ok = 0, rejected = 0;
    for (int i = -1; i < n; ++i) {
        std::string a = (i < 0 || i % 2) ? main_good : mangle(main_good, r);
        std::string b = (i < 0 || !(i % 2)) ? st_good : mangle(st_good, r);
        try {
            q3::ModelConfig c;
            c.parse(q3::JsonParser(a.data(), a.size()).parse());
            c.parse_speech_tokenizer(q3::JsonParser(b.data(), b.size()).parse());
            c.validate();
            // the loader's own arithmetic on the validated numbers: indices, products, the float division of the encoder rate
            int64_t sum = 0;
            if (c.has_talker)
                for (int l = 0; l < c.talker.num_hidden_layers; ++l) sum += c.talker.inter(l);
            if (c.has_codec) sum += c.codec.total_upsample();
            if (c.has_codec_encoder) sum += c.codec_enc.hop() + c.codec_enc.downsample_stride();
            if (sum < 0) return 3;
            ++ok;
        } catch (const q3::Error&) {
            if (i < 0) return 3;  // the unmangled pair is a valid configuration
            ++rejected;
        }
    }
    std::printf("ok %d rejected %d\n", ok - 1, rejected);
    return 0;
}

int cmd_stfuzz(int argc, char** argv) {
    if (argc < 6) return 2;
    std::string good = slurp(argv[2]);
    std::string dir = argv[3];
    Rng r{uint64_t(std::strtoull(argv[4], nullptr, 10))};
    int n = std::atoi(argv[5]), ok = 0, rejected = 0;
    if (good.size() < 8) return 2;
    uint64_t hlen = 0;
    std::memcpy(&hlen, good.data(), 8);
    if (8 + hlen > good.size()) return 2;
    std::string header = good.substr(8, size_t(hlen)), body = good.substr(8 + size_t(hlen));
    std::string path = dir + "/model.safetensors";
    for (int i = -1; i < n; ++i) {
        std::string h = i < 0 ? header : mangle(header, r);
        uint64_t len = h.size();
        // every fourth case lies about the header length or cuts the tensor data short instead
        std::string b = body;
        if (i >= 0 && i % 4 == 1) {
            static const uint64_t edge[] = {~0ull, ~0ull - 7, ~0ull - 8, 1ull << 63, 0};
            len = r.below(3) == 0 ? edge[r.below(5)] : (r.below(2) ? h.size() + b.size() + r.below(3) : r.next() >> r.below(64));
        }
        if (i >= 0 && i % 4 == 3) b.resize(r.below(b.size() + 1));
        {
            std::ofstream f(path, std::ios::binary | std::ios::trunc);
            f.write(reinterpret_cast<const char*>(&len), 8);
            f.write(h.data(), std::streamsize(h.size()));
            f.write(b.data(), std::streamsize(b.size()));
        }
        try {
            q3::SafetensorsDir d;
            d.open_dir(dir);
            // touch the first and the last byte of every tensor the header promised: a view outside the mapping faults here
            size_t sum = 0;
            for (auto& kv : d.all()) {
                const q3::TensorView& v = kv.second;
                size_t bytes = size_t(v.numel()) * q3::dtype_size(v.dtype);
                if (bytes) sum += size_t(reinterpret_cast<const unsigned char*>(v.data)[0]) + reinterpret_cast<const unsigned char*>(v.data)[bytes - 1];
            }
            (void)sum;
            if (i < 0 && d.all().empty()) return 3;  // the unmangled file opens
            ++ok;
        } catch (const q3::Error&) {
            if (i < 0) return 3;
            ++rejected;
        }
    }
    std::printf("ok %d rejected %d\n", ok - 1, rejected);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::string mode = argv[1];
    try {
        if (mode == "tok") return cmd_tok(argc, argv);
        if (mode == "jsonfuzz") return cmd_jsonfuzz(argc, argv);
        if (mode == "stfuzz") return cmd_stfuzz(argc, argv);
        if (mode == "tokfuzz") return cmd_tokfuzz(argc, argv);
        if (mode == "cfgfuzz") return cmd_cfgfuzz(argc, argv);
        if (mode == "textfuzz") return cmd_textfuzz(argc, argv);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "uncaught: %s\n", e.what());
        return 4;
    }
    return 2;
}
