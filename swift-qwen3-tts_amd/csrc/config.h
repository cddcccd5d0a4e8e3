// config.h -- config.json mirrors with the reference's per-field defaults
// (/root/reference/Sources/Qwen3TTS/Models/Config.swift). The defaults are load-bearing
// (e.g. codec_eos_token_id ?? 2150, Config.swift:309), so every one is restated here.
#pragma once
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "json.h"

namespace q3 {

struct CodePredictorConfig {  // Config.swift:145-159
    int vocab_size = 2048, hidden_size = 1024, intermediate_size = 3072, num_hidden_layers = 5;
    int num_attention_heads = 16, num_key_value_heads = 8, head_dim = 128, num_code_groups = 16;
    float rms_norm_eps = 1e-6f, rope_theta = 1e6f;
    void parse(const Json& j) {
        vocab_size = int(j.i64("vocab_size", vocab_size));
        hidden_size = int(j.i64("hidden_size", hidden_size));
        intermediate_size = int(j.i64("intermediate_size", intermediate_size));
        num_hidden_layers = int(j.i64("num_hidden_layers", num_hidden_layers));
        num_attention_heads = int(j.i64("num_attention_heads", num_attention_heads));
        num_key_value_heads = int(j.i64("num_key_value_heads", num_key_value_heads));
        head_dim = int(j.i64("head_dim", head_dim));
        num_code_groups = int(j.i64("num_code_groups", num_code_groups));
        rms_norm_eps = float(j.f64("rms_norm_eps", rms_norm_eps));
        rope_theta = float(j.f64("rope_theta", rope_theta));
    }
};

struct TalkerConfig {  // Config.swift:289-333
    int vocab_size = 3072, text_vocab_size = 151936, hidden_size = 2048, text_hidden_size = 2048;
    int intermediate_size = 6144, num_hidden_layers = 28, num_attention_heads = 16;
    int num_key_value_heads = 8, head_dim = 128, num_code_groups = 16;
    std::vector<int> per_layer_intermediate_sizes;  // empty = uniform
    float rms_norm_eps = 1e-6f, rope_theta = 1e6f;
    int codec_eos_token_id = 2150, codec_think_id = 2154, codec_nothink_id = 2155;
    int codec_think_bos_id = 2156, codec_think_eos_id = 2157, codec_pad_id = 2148, codec_bos_id = 2149;
    std::map<std::string, int> codec_language_id{{"chinese", 2055}, {"english", 2050}, {"german", 2053},
                                                  {"italian", 2070}, {"portuguese", 2071}, {"spanish", 2054},
                                                  {"japanese", 2058}, {"korean", 2064}, {"french", 2061},
                                                  {"russian", 2069}};
    bool has_spk_id = false;
    std::map<std::string, int> spk_id;
    std::map<std::string, std::string> spk_dialect;  // only entries that name a dialect
    bool has_code_predictor = false;
    CodePredictorConfig cp;

    int inter(int layer) const {
        return per_layer_intermediate_sizes.empty() ? intermediate_size : per_layer_intermediate_sizes[size_t(layer)];
    }
    void parse(const Json& j) {
        vocab_size = int(j.i64("vocab_size", vocab_size));
        text_vocab_size = int(j.i64("text_vocab_size", text_vocab_size));
        hidden_size = int(j.i64("hidden_size", hidden_size));
        text_hidden_size = int(j.i64("text_hidden_size", text_hidden_size));
        intermediate_size = int(j.i64("intermediate_size", intermediate_size));
        per_layer_intermediate_sizes = j.ints("per_layer_intermediate_sizes", {});
        num_hidden_layers = int(j.i64("num_hidden_layers", num_hidden_layers));
        num_attention_heads = int(j.i64("num_attention_heads", num_attention_heads));
        num_key_value_heads = int(j.i64("num_key_value_heads", num_key_value_heads));
        head_dim = int(j.i64("head_dim", head_dim));
        num_code_groups = int(j.i64("num_code_groups", num_code_groups));
        rms_norm_eps = float(j.f64("rms_norm_eps", rms_norm_eps));
        rope_theta = float(j.f64("rope_theta", rope_theta));
        codec_eos_token_id = int(j.i64("codec_eos_token_id", codec_eos_token_id));
        codec_think_id = int(j.i64("codec_think_id", codec_think_id));
        codec_nothink_id = int(j.i64("codec_nothink_id", codec_nothink_id));
        codec_think_bos_id = int(j.i64("codec_think_bos_id", codec_think_bos_id));
        codec_think_eos_id = int(j.i64("codec_think_eos_id", codec_think_eos_id));
        codec_pad_id = int(j.i64("codec_pad_id", codec_pad_id));
        codec_bos_id = int(j.i64("codec_bos_id", codec_bos_id));
        if (const Json* l = j.get("codec_language_id"); l && l->kind == Json::Obj) {
            codec_language_id.clear();
            for (auto& kv : l->obj) codec_language_id[kv.first] = int(kv.second.num);
        }
        if (const Json* s = j.get("spk_id"); s && s->kind == Json::Obj) {
            has_spk_id = true;
            for (auto& kv : s->obj) spk_id[kv.first] = int(kv.second.num);
        }
        if (const Json* s = j.get("spk_is_dialect"); s && s->kind == Json::Obj) {
            for (auto& kv : s->obj)  // DialectValue: false or a dialect name (Config.swift:17-53)
                if (kv.second.kind == Json::Str) spk_dialect[kv.first] = kv.second.str;
        }
        if (const Json* c = j.get("code_predictor_config"); c && c->kind == Json::Obj) {
            has_code_predictor = true;
            cp.parse(*c);
        }
    }
};

struct CodecDecoderConfig {  // Config.swift:385-415
    int latent_dim = 1024, codebook_dim = 512, codebook_size = 2048, decoder_dim = 1536;
    int hidden_size = 512, intermediate_size = 1024, num_hidden_layers = 8, num_attention_heads = 16;
    int num_key_value_heads = 16, head_dim = 64;
    float rms_norm_eps = 1e-5f;
    int num_quantizers = 16, num_semantic_quantizers = 1, semantic_codebook_size = 4096;
    std::vector<int> upsample_rates{8, 5, 4, 3}, upsampling_ratios{2, 2};
    float layer_scale_initial_scale = 0.01f;
    int total_upsample() const {
        int t = 1;
        for (int r : upsample_rates) t *= r;
        for (int r : upsampling_ratios) t *= r;
        return t;
    }
    void parse(const Json& j) {
        latent_dim = int(j.i64("latent_dim", latent_dim));
        codebook_dim = int(j.i64("codebook_dim", codebook_dim));
        codebook_size = int(j.i64("codebook_size", codebook_size));
        decoder_dim = int(j.i64("decoder_dim", decoder_dim));
        hidden_size = int(j.i64("hidden_size", hidden_size));
        intermediate_size = int(j.i64("intermediate_size", intermediate_size));
        num_hidden_layers = int(j.i64("num_hidden_layers", num_hidden_layers));
        num_attention_heads = int(j.i64("num_attention_heads", num_attention_heads));
        num_key_value_heads = int(j.i64("num_key_value_heads", num_key_value_heads));
        head_dim = int(j.i64("head_dim", head_dim));
        rms_norm_eps = float(j.f64("rms_norm_eps", rms_norm_eps));
        num_quantizers = int(j.i64("num_quantizers", num_quantizers));
        num_semantic_quantizers = int(j.i64("num_semantic_quantizers", num_semantic_quantizers));
        semantic_codebook_size = int(j.i64("semantic_codebook_size", semantic_codebook_size));
        upsample_rates = j.ints("upsample_rates", upsample_rates);
        upsampling_ratios = j.ints("upsampling_ratios", upsampling_ratios);
        layer_scale_initial_scale = float(j.f64("layer_scale_initial_scale", layer_scale_initial_scale));
    }
};

struct ModelConfig {  // Config.swift:635-657, 584-594
    std::string tts_model_type = "voice_design", tts_model_size = "1b7";
    int tts_pad_token_id = 151671, tts_bos_token_id = 151672, tts_eos_token_id = 151673;
    int sample_rate = 24000;
    bool has_talker = false;
    TalkerConfig talker;
    bool has_quantization = false;
    int quant_group_size = 64, quant_bits = 4;
    bool has_speaker_encoder = false;
    // speech_tokenizer/config.json
    bool has_codec = false;
    int decode_upsample_rate = 1920;
    bool has_codec_encoder = false;
    CodecDecoderConfig codec;

    void parse(const Json& j) {
        tts_model_type = j.s("tts_model_type", tts_model_type);
        tts_model_size = j.s("tts_model_size", tts_model_size);
        tts_pad_token_id = int(j.i64("tts_pad_token_id", tts_pad_token_id));
        tts_bos_token_id = int(j.i64("tts_bos_token_id", tts_bos_token_id));
        tts_eos_token_id = int(j.i64("tts_eos_token_id", tts_eos_token_id));
        sample_rate = int(j.i64("sample_rate", sample_rate));
        if (const Json* t = j.get("talker_config"); t && t->kind == Json::Obj) {
            has_talker = true;
            talker.parse(*t);
        }
        if (const Json* q = j.get("quantization"); q && q->kind == Json::Obj) {
            has_quantization = true;
            quant_group_size = int(q->i64("group_size", 64));
            quant_bits = int(q->i64("bits", 4));
        }
        has_speaker_encoder = j.has("speaker_encoder_config");
    }
    void parse_speech_tokenizer(const Json& j) {
        decode_upsample_rate = int(j.i64("decode_upsample_rate", decode_upsample_rate));
        has_codec_encoder = j.has("encoder_config");
        if (const Json* d = j.get("decoder_config"); d && d->kind == Json::Obj) {
            has_codec = true;
            codec.parse(*d);
        }
    }
};

inline std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    Q3_CHECK(bool(f), 6, "cannot read " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

inline bool file_exists(const std::string& path) {
    std::ifstream f(path);
    return bool(f);
}

}  // namespace q3
