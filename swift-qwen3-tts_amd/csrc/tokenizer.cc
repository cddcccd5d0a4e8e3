// tokenizer.cc -- see tokenizer.h.
#include "tokenizer.h"

#include <algorithm>
#include <climits>

#include "common.h"
#include "config.h"
#include "json.h"
#include "unicode_tables.h"

namespace q3 {

// ------------------------------------------------------------------------------------------------
// UTF-8
// ------------------------------------------------------------------------------------------------
std::u32string utf8_to_u32(const std::string& s) {
    std::u32string out;
    out.reserve(s.size());
    size_t i = 0;
    while (i < s.size()) {
        const unsigned char c = (unsigned char)s[i];
        uint32_t cp = 0xFFFD;
        size_t n = 1;
        if (c < 0x80) cp = c;
        else if ((c >> 5) == 6 && i + 1 < s.size()) { cp = ((c & 0x1F) << 6) | (s[i + 1] & 0x3F); n = 2; }
        else if ((c >> 4) == 14 && i + 2 < s.size()) { cp = ((c & 0x0F) << 12) | ((s[i + 1] & 0x3F) << 6) | (s[i + 2] & 0x3F); n = 3; }
        else if ((c >> 3) == 30 && i + 3 < s.size()) {
            cp = ((c & 0x07) << 18) | ((s[i + 1] & 0x3F) << 12) | ((s[i + 2] & 0x3F) << 6) | (s[i + 3] & 0x3F);
            n = 4;
        }
        out.push_back(char32_t(cp));
        i += n;
    }
    return out;
}

static void append_utf8(std::string& out, uint32_t cp) {
    if (cp < 0x80) out += char(cp);
    else if (cp < 0x800) { out += char(0xC0 | (cp >> 6)); out += char(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { out += char(0xE0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
    else { out += char(0xF0 | (cp >> 18)); out += char(0x80 | ((cp >> 12) & 0x3F)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
}
std::string u32_to_utf8(const std::u32string& s) {
    std::string out;
    for (char32_t c : s) append_utf8(out, uint32_t(c));
    return out;
}

// ------------------------------------------------------------------------------------------------
// Unicode properties
// ------------------------------------------------------------------------------------------------
namespace {

template <size_t N>
bool in_ranges(const uni::Range (&r)[N], uint32_t cp) {
    size_t lo = 0, hi = N;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp < r[mid].lo) hi = mid;
        else if (cp > r[mid].hi) lo = mid + 1;
        else return true;
    }
    return false;
}
bool is_letter(uint32_t cp) { return in_ranges(uni::kLetter, cp); }
bool is_number(uint32_t cp) { return in_ranges(uni::kNumber, cp); }
bool is_space(uint32_t cp) { return in_ranges(uni::kWhiteSpace, cp); }
bool is_newline(uint32_t cp) { return cp == '\r' || cp == '\n'; }
bool is_other(uint32_t cp) { return !is_space(cp) && !is_letter(cp) && !is_number(cp); }  // [^\s\p{L}\p{N}]

int ccc_of(uint32_t cp) {
    size_t lo = 0, hi = sizeof(uni::kCcc) / sizeof(uni::kCcc[0]);
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (uni::kCcc[mid].cp < cp) lo = mid + 1;
        else hi = mid;
    }
    const size_t n = sizeof(uni::kCcc) / sizeof(uni::kCcc[0]);
    return (lo < n && uni::kCcc[lo].cp == cp) ? uni::kCcc[lo].ccc : 0;
}
const uni::DecompIndex* decomp_of(uint32_t cp) {
    size_t lo = 0, hi = sizeof(uni::kDecompIndex) / sizeof(uni::kDecompIndex[0]);
    const size_t n = hi;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (uni::kDecompIndex[mid].cp < cp) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && uni::kDecompIndex[lo].cp == cp) ? &uni::kDecompIndex[lo] : nullptr;
}
uint32_t compose_pair(uint32_t a, uint32_t b) {
    // Hangul (algorithmic)
    if (a >= 0x1100 && a < 0x1113 && b >= 0x1161 && b < 0x1176) return 0xAC00 + ((a - 0x1100) * 21 + (b - 0x1161)) * 28;
    if (a >= 0xAC00 && a <= 0xD7A3 && (a - 0xAC00) % 28 == 0 && b > 0x11A7 && b < 0x11C3) return a + (b - 0x11A7);
    size_t lo = 0, hi = sizeof(uni::kComp) / sizeof(uni::kComp[0]);
    const size_t n = hi;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        const auto& c = uni::kComp[mid];
        if (c.a < a || (c.a == a && c.b < b)) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && uni::kComp[lo].a == a && uni::kComp[lo].b == b) ? uni::kComp[lo].c : 0;
}

}  // namespace

// Unicode Standard Annex #15: canonical decomposition, canonical ordering, canonical composition.
std::u32string nfc_normalize(const std::u32string& s) {
    bool ascii = true;
    for (char32_t c : s)
        if (c >= 0xC0) { ascii = false; break; }
    if (ascii) return s;  // Latin-1 below U+00C0 is NFC-inert
    std::u32string d;
    d.reserve(s.size() + 8);
    for (char32_t c : s) {
        const uint32_t cp = uint32_t(c);
        if (cp >= 0xAC00 && cp <= 0xD7A3) {  // Hangul syllable
            const uint32_t si = cp - 0xAC00;
            d.push_back(char32_t(0x1100 + si / 588));
            d.push_back(char32_t(0x1161 + (si % 588) / 28));
            if (si % 28) d.push_back(char32_t(0x11A7 + si % 28));
        } else if (const uni::DecompIndex* di = decomp_of(cp)) {
            for (uint32_t k = 0; k < di->len; ++k) d.push_back(char32_t(uni::kDecompData[di->off + k]));
        } else {
            d.push_back(c);
        }
    }
    // canonical ordering: stable sort of runs of non-starters by combining class
    for (size_t i = 0; i < d.size();) {
        if (ccc_of(d[i]) == 0) { ++i; continue; }
        size_t j = i;
        while (j < d.size() && ccc_of(d[j]) != 0) ++j;
        std::stable_sort(d.begin() + long(i), d.begin() + long(j), [](char32_t a, char32_t b) { return ccc_of(a) < ccc_of(b); });
        i = j;
    }
    // canonical composition: a character combines with the last starter unless a character in between has combining
    // class 0 or a class >= its own (after canonical ordering the largest class in between is the previous character's)
    std::u32string out;
    out.reserve(d.size());
    size_t starter = SIZE_MAX;
    int last_cc = 0;
    for (char32_t c : d) {
        const int cc = ccc_of(c);
        if (starter != SIZE_MAX) {
            const bool has_between = out.size() - 1 > starter;
            const bool blocked = has_between && (last_cc == 0 || last_cc >= cc);
            if (!blocked) {
                const uint32_t comp = compose_pair(uint32_t(out[starter]), uint32_t(c));
                if (comp) {
                    out[starter] = char32_t(comp);
                    continue;
                }
            }
        }
        if (cc == 0) starter = out.size();
        out.push_back(c);
        last_cc = cc;
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// (?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+
// hand-written matcher: leftmost match, alternatives tried in order, greedy quantifiers with backtracking
// ------------------------------------------------------------------------------------------------
std::vector<std::pair<size_t, size_t>> qwen2_pretokenize(const std::u32string& s) {
    std::vector<std::pair<size_t, size_t>> out;
    const size_t n = s.size();
    auto lower = [](uint32_t c) -> uint32_t {
        if (c >= 'A' && c <= 'Z') return c + 32;
        if (c == 0x17F) return 's';   // LATIN SMALL LETTER LONG S folds to s
        if (c == 0x212A) return 'k';  // KELVIN SIGN folds to k (no contraction uses it; kept for completeness)
        return c;
    };
    size_t i = 0;
    while (i < n) {
        const uint32_t c = s[i];
        size_t end = 0;
        // 1. contractions
        if (c == '\'' && i + 1 < n) {
            const uint32_t a = lower(s[i + 1]);
            const uint32_t b = i + 2 < n ? lower(s[i + 2]) : 0;
            if (a == 's' || a == 't') end = i + 2;
            else if (a == 'r' && b == 'e') end = i + 3;
            else if (a == 'v' && b == 'e') end = i + 3;
            else if (a == 'm') end = i + 2;
            else if (a == 'l' && b == 'l') end = i + 3;
            else if (a == 'd') end = i + 2;
        }
        // 2. [^\r\n\p{L}\p{N}]?\p{L}+
        if (!end) {
            size_t j = i;
            if (!is_newline(c) && !is_letter(c) && !is_number(c) && i + 1 < n && is_letter(s[i + 1])) j = i + 1;
            if (is_letter(s[j])) {
                while (j < n && is_letter(s[j])) ++j;
                end = j;
            }
        }
        // 3. \p{N}
        if (!end && is_number(c)) end = i + 1;
        // 4.  ?[^\s\p{L}\p{N}]+[\r\n]*
        if (!end) {
            size_t j = i;
            if (c == ' ' && i + 1 < n && is_other(s[i + 1])) j = i + 1;
            if (is_other(s[j])) {
                while (j < n && is_other(s[j])) ++j;
                while (j < n && is_newline(s[j])) ++j;
                end = j;
            }
        }
        if (!end && is_space(c)) {
            size_t j = i;
            while (j < n && is_space(s[j])) ++j;  // maximal whitespace run [i, j)
            // 5. \s*[\r\n]+ : up to and including the last newline of the run
            size_t k = j;
            while (k > i && !is_newline(s[k - 1])) --k;
            if (k > i) end = k;
            // 6. \s+(?!\S) : the whole run at the end of the text, otherwise all but its last character
            else if (j == n) end = j;
            else if (j - i >= 2) end = j - 1;
            // 7. \s+
            else end = j;
        }
        if (!end) end = i + 1;  // unreachable: every character is a letter, a number, whitespace or "other"
        out.emplace_back(i, end);
        i = end;
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// BPE
// ------------------------------------------------------------------------------------------------
void BpeTokenizer::init_byte_map() {
    // GPT-2 bytes_to_unicode: printable Latin-1 bytes map to themselves, the rest to U+0100...
    int nxt = 0;
    for (int b = 0; b < 256; ++b) {
        const bool keep = (b >= '!' && b <= '~') || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE && b <= 0xFF);
        std::string u;
        append_utf8(u, keep ? uint32_t(b) : uint32_t(256 + nxt++));
        byte_char_[b] = u;
    }
}

// a vocabulary entry's id: a non-negative integer that fits the engine's int32 token ids, or the file is damaged
static int32_t vocab_id(const Json& v) {
    Q3_CHECK(v.kind == Json::Num, 6, "tokenizer: a vocabulary id is not a number");
    return int32_t(Json::to_int(v.num, 0, 2147483647.0, "a vocabulary id"));
}

void BpeTokenizer::load(const std::string& dir) {
    if (file_exists(dir + "/tokenizer.json")) {
        load_json_file(dir + "/tokenizer.json");
        return;
    }
    Q3_CHECK(file_exists(dir + "/vocab.json") && file_exists(dir + "/merges.txt"), 1,
             "Model not initialized: Tokenizer not loaded (no tokenizer.json, vocab.json or merges.txt in " + dir + ")");
    init_byte_map();
    {
        const std::string txt = read_file(dir + "/vocab.json");
        Json j = JsonParser(txt.data(), txt.size()).parse();
        for (auto& kv : j.obj) vocab_[kv.first] = vocab_id(kv.second);
    }
    {
        const std::string txt = read_file(dir + "/merges.txt");
        size_t pos = 0;
        int rank = 0;
        while (pos < txt.size()) {
            size_t eol = txt.find('\n', pos);
            if (eol == std::string::npos) eol = txt.size();
            std::string line = txt.substr(pos, eol - pos);
            pos = eol + 1;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty() || line.rfind("#version", 0) == 0) continue;
            const size_t sp = line.find(' ');
            if (sp == std::string::npos) continue;
            merge_rank_[line.substr(0, sp) + '\x01' + line.substr(sp + 1)] = rank++;
        }
    }
    if (file_exists(dir + "/tokenizer_config.json")) {  // added_tokens_decoder: {"151643": {"content": "<|endoftext|>", ...}}
        const std::string txt = read_file(dir + "/tokenizer_config.json");
        Json j = JsonParser(txt.data(), txt.size()).parse();
        if (const Json* d = j.get("added_tokens_decoder"); d && d->kind == Json::Obj)
            for (auto& kv : d->obj) added_.push_back(Added{kv.second.s("content", ""), int32_t(std::atoi(kv.first.c_str())), false});
    }
}

void BpeTokenizer::load_json_file(const std::string& path) {
    init_byte_map();
    const std::string txt = read_file(path);
    Json j = JsonParser(txt.data(), txt.size()).parse();
    const Json* model = j.get("model");
    Q3_CHECK(model && model->s("type", "BPE") == "BPE", 6, "tokenizer.json: only BPE models are supported");
    const Json* vocab = model->get("vocab");
    const Json* merges = model->get("merges");
    Q3_CHECK(vocab && vocab->kind == Json::Obj && merges && merges->kind == Json::Arr, 6, "tokenizer.json: vocab / merges missing");
    vocab_.reserve(vocab->obj.size() * 2);
    for (auto& kv : vocab->obj) vocab_[kv.first] = vocab_id(kv.second);
    int rank = 0;
    for (auto& m : merges->arr) {
        if (m.kind == Json::Str) {  // "left right"
            const size_t sp = m.str.find(' ');
            Q3_CHECK(sp != std::string::npos, 6, "tokenizer.json: malformed merge");
            merge_rank_[m.str.substr(0, sp) + '\x01' + m.str.substr(sp + 1)] = rank++;
        } else {  // ["left", "right"]
            Q3_CHECK(m.kind == Json::Arr && m.arr.size() == 2, 6, "tokenizer.json: malformed merge");
            merge_rank_[m.arr[0].str + '\x01' + m.arr[1].str] = rank++;
        }
    }
    if (const Json* ig = model->get("ignore_merges"); ig && ig->kind == Json::Bool) ignore_merges_ = ig->b;
    nfc_ = false;
    if (const Json* nz = j.get("normalizer"); nz && nz->kind == Json::Obj) {
        const std::string t = nz->s("type", "");
        Q3_CHECK(t == "NFC", 6, "tokenizer.json: unsupported normalizer '" + t + "' (Qwen2 uses NFC)");
        nfc_ = true;
    }
    if (const Json* pt = j.get("pre_tokenizer"); pt && pt->kind == Json::Obj) {  // must be the Qwen2 split + byte level
        bool ok = false;
        if (const Json* seq = pt->get("pretokenizers"); seq && seq->kind == Json::Arr)
            for (auto& p : seq->arr)
                if (p.s("type", "") == "Split")
                    if (const Json* pat = p.get("pattern"))
                        ok = pat->s("Regex", "").find("\\p{L}+|\\p{N}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+") != std::string::npos;
        Q3_CHECK(ok, 6, "tokenizer.json: the pre-tokenizer is not the Qwen2 split pattern");
    }
    if (const Json* at = j.get("added_tokens"); at && at->kind == Json::Arr)
        for (auto& a : at->arr) {
            const int64_t id = a.i64("id", -1);
            Added ad{a.s("content", ""), id > 2147483647 ? -1 : int32_t(id), false};
            if (const Json* nm = a.get("normalized"); nm && nm->kind == Json::Bool) ad.normalized = nm->b;
            if (!ad.content.empty() && ad.id >= 0) added_.push_back(ad);
        }
}

// one pre-token: byte-level symbols, then merges by rank
void BpeTokenizer::encode_piece(const std::u32string& piece, std::vector<int32_t>& out) const {
    const std::string bytes = u32_to_utf8(piece);
    std::vector<std::string> sym;
    sym.reserve(bytes.size());
    std::string whole;
    for (unsigned char b : bytes) {
        sym.push_back(byte_char_[b]);
        whole += byte_char_[b];
    }
    if (ignore_merges_) {
        auto it = vocab_.find(whole);
        if (it != vocab_.end()) {
            out.push_back(it->second);
            return;
        }
    }
    while (sym.size() > 1) {
        int best = INT_MAX;
        size_t at = 0;
        for (size_t i = 0; i + 1 < sym.size(); ++i) {
            auto it = merge_rank_.find(sym[i] + '\x01' + sym[i + 1]);
            if (it != merge_rank_.end() && it->second < best) {
                best = it->second;
                at = i;
            }
        }
        if (best == INT_MAX) break;
        // merge every non-overlapping occurrence of that pair, left to right (as the word-level merge does)
        const std::string l = sym[at], r = sym[at + 1];
        std::vector<std::string> nx;
        nx.reserve(sym.size());
        for (size_t i = 0; i < sym.size();) {
            if (i + 1 < sym.size() && sym[i] == l && sym[i + 1] == r) {
                nx.push_back(l + r);
                i += 2;
            } else {
                nx.push_back(sym[i]);
                ++i;
            }
        }
        sym.swap(nx);
    }
    for (auto& t : sym) {
        auto it = vocab_.find(t);
        Q3_CHECK(it != vocab_.end(), 3, "Invalid input: tokenizer vocabulary has no entry for a byte-level symbol");
        out.push_back(it->second);
    }
}

void BpeTokenizer::encode_text(const std::string& utf8, std::vector<int32_t>& out) const {
    if (utf8.empty()) return;
    std::u32string s = utf8_to_u32(utf8);
    if (nfc_) s = nfc_normalize(s);
    for (auto& span : qwen2_pretokenize(s)) encode_piece(s.substr(span.first, span.second - span.first), out);
}

std::vector<int32_t> BpeTokenizer::encode(const std::string& text) const {
    std::vector<int32_t> out;
    // added tokens first: leftmost match, longest among those starting at the same position
    size_t seg = 0, i = 0;
    while (i < text.size()) {
        const Added* hit = nullptr;
        for (auto& a : added_)
            if (!a.normalized && text.compare(i, a.content.size(), a.content) == 0 && (!hit || a.content.size() > hit->content.size())) hit = &a;
        if (hit) {
            encode_text(text.substr(seg, i - seg), out);
            out.push_back(hit->id);
            i += hit->content.size();
            seg = i;
        } else {
            ++i;
        }
    }
    encode_text(text.substr(seg), out);
    return out;
}

}  // namespace q3
