// tokenizer.h -- byte-level BPE text tokeniser (the Qwen2 tokenizer that Qwen3-TTS checkpoints ship as tokenizer.json).
//
// The reference tokenises through swift-transformers (`AutoTokenizer.from(modelFolder:)`, /root/reference/Sources/
// Qwen3TTS/Models/Qwen3.swift:1458; encode calls :274-275, :364-365, :448-457, :822; dependency pinned in
// Package.resolved:58-64), which is not vendored: this is a restatement of the published tokenizer.json semantics
// (Hugging Face `tokenizers`): added (special) tokens split out first, NFC normalisation, the Qwen2 pre-tokeniser regex
// applied as an isolating split, GPT-2 byte-to-unicode mapping, rank-ordered BPE merges. Host code only (SURVEY.md
// row f2); parity is pinned against the `tokenizers` wheel on a synthetic tokenizer.json (tests/test_tokenizer.py).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace q3 {

class BpeTokenizer {
  public:
    // model_dir holds tokenizer.json (fast format), or vocab.json + merges.txt (slow format, Qwen2 defaults assumed)
    void load(const std::string& model_dir);
    void load_json_file(const std::string& path);
    std::vector<int32_t> encode(const std::string& utf8) const;
    size_t vocab_size() const { return vocab_.size() + added_.size(); }

  private:
    std::unordered_map<std::string, int32_t> vocab_;
    std::unordered_map<std::string, int32_t> merge_rank_;  // "left\x01right" -> rank
    struct Added {
        std::string content;
        int32_t id;
        bool normalized;
    };
    std::vector<Added> added_;
    bool nfc_ = true;
    bool ignore_merges_ = false;
    std::string byte_char_[256];  // GPT-2 bytes_to_unicode, UTF-8 encoded
    void init_byte_map();
    void encode_piece(const std::u32string& piece, std::vector<int32_t>& out) const;
    void encode_text(const std::string& utf8, std::vector<int32_t>& out) const;
};

// exposed for tests
std::u32string utf8_to_u32(const std::string& s);
std::string u32_to_utf8(const std::u32string& s);
std::u32string nfc_normalize(const std::u32string& s);
// the Qwen2 pre-tokeniser pattern, as (start, end) code-point spans covering the whole input
std::vector<std::pair<size_t, size_t>> qwen2_pretokenize(const std::u32string& s);

}  // namespace q3
