// config.h -- config.json mirrors with the reference's per-field defaults
// (/root/reference/Sources/Qwen3TTS/Models/Config.swift). The defaults are load-bearing
// (e.g. codec_eos_token_id ?? 2150, Config.swift:309), so every one is restated here.
#pragma once
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <utility>
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
    static int id_of(const Json& v, const char* what) {
        Q3_CHECK(v.kind == Json::Num, 6, std::string("config.json: an entry of ") + what + " is not a number");
        return int(Json::to_int(v.num, -2147483648.0, 2147483647.0, what));
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
            for (auto& kv : l->obj) codec_language_id[kv.first] = id_of(kv.second, "codec_language_id");
        }
        if (const Json* s = j.get("spk_id"); s && s->kind == Json::Obj) {
            has_spk_id = true;
            for (auto& kv : s->obj) spk_id[kv.first] = id_of(kv.second, "spk_id");
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

struct CodecEncoderConfig {  // Qwen3TTSTokenizerEncoderConfig, Config.swift:476-504
    float frame_rate = 12.5f;
    int audio_channels = 1, codebook_dim = 256, codebook_size = 2048, compress = 2, dilation_growth_rate = 2;
    int head_dim = 64, hidden_size = 512, intermediate_size = 2048, kernel_size = 7, last_kernel_size = 3;
    float layer_scale_initial_scale = 0.01f;
    int max_position_embeddings = 8000, num_attention_heads = 8, num_filters = 64, num_hidden_layers = 8;
    int num_key_value_heads = 8, num_quantizers = 32, num_residual_layers = 1, residual_kernel_size = 3;
    float rope_theta = 10000.0f;
    int sampling_rate = 24000, sliding_window = 250;
    std::vector<int> upsampling_ratios{8, 6, 5, 4};
    bool use_causal_conv = true, use_conv_shortcut = false;
    static bool flag(const Json& j, const char* k, bool def) {
        const Json* v = j.get(k);
        return (v && v->kind == Json::Bool) ? v->b : def;
    }
    void parse(const Json& j) {
        frame_rate = float(j.f64("frame_rate", frame_rate));
        audio_channels = int(j.i64("audio_channels", audio_channels));
        codebook_dim = int(j.i64("codebook_dim", codebook_dim));
        codebook_size = int(j.i64("codebook_size", codebook_size));
        compress = int(j.i64("compress", compress));
        dilation_growth_rate = int(j.i64("dilation_growth_rate", dilation_growth_rate));
        head_dim = int(j.i64("head_dim", head_dim));
        hidden_size = int(j.i64("hidden_size", hidden_size));
        intermediate_size = int(j.i64("intermediate_size", intermediate_size));
        kernel_size = int(j.i64("kernel_size", kernel_size));
        last_kernel_size = int(j.i64("last_kernel_size", last_kernel_size));
        layer_scale_initial_scale = float(j.f64("layer_scale_initial_scale", layer_scale_initial_scale));
        max_position_embeddings = int(j.i64("max_position_embeddings", max_position_embeddings));
        num_attention_heads = int(j.i64("num_attention_heads", num_attention_heads));
        num_filters = int(j.i64("num_filters", num_filters));
        num_hidden_layers = int(j.i64("num_hidden_layers", num_hidden_layers));
        num_key_value_heads = int(j.i64("num_key_value_heads", num_key_value_heads));
        num_quantizers = int(j.i64("num_quantizers", num_quantizers));
        num_residual_layers = int(j.i64("num_residual_layers", num_residual_layers));
        residual_kernel_size = int(j.i64("residual_kernel_size", residual_kernel_size));
        rope_theta = float(j.f64("rope_theta", rope_theta));
        sampling_rate = int(j.i64("sampling_rate", sampling_rate));
        sliding_window = int(j.i64("sliding_window", sliding_window));
        upsampling_ratios = j.ints("upsampling_ratios", upsampling_ratios);
        use_causal_conv = flag(j, "use_causal_conv", use_causal_conv);
        use_conv_shortcut = flag(j, "use_conv_shortcut", use_conv_shortcut);
    }
    int hop() const {  // samples per encoder-transformer position
        int t = 1;
        for (int r : upsampling_ratios) t *= r;
        return t;
    }
    int downsample_stride() const {  // SpeechTokenizerEncoder.swift:1008-1009, Float arithmetic
        const float enc_rate = float(sampling_rate) / float(hop());
        return int(enc_rate / frame_rate);
    }
};

struct SpeakerEncoderConfig {  // Qwen3TTSSpeakerEncoderConfig, Config.swift:80-91
    int mel_dim = 128, enc_dim = 1024;
    std::vector<int> enc_channels{512, 512, 512, 512, 1536}, enc_kernel_sizes{5, 3, 3, 3, 1}, enc_dilations{1, 2, 3, 4, 1};
    int enc_attention_channels = 128, enc_res2net_scale = 8, enc_se_channels = 128, sample_rate = 24000;
    void parse(const Json& j) {
        mel_dim = int(j.i64("mel_dim", mel_dim));
        enc_dim = int(j.i64("enc_dim", enc_dim));
        enc_channels = j.ints("enc_channels", enc_channels);
        enc_kernel_sizes = j.ints("enc_kernel_sizes", enc_kernel_sizes);
        enc_dilations = j.ints("enc_dilations", enc_dilations);
        enc_attention_channels = int(j.i64("enc_attention_channels", enc_attention_channels));
        enc_res2net_scale = int(j.i64("enc_res2net_scale", enc_res2net_scale));
        enc_se_channels = int(j.i64("enc_se_channels", enc_se_channels));
        sample_rate = int(j.i64("sample_rate", sample_rate));
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
    SpeakerEncoderConfig speaker;
    CodecEncoderConfig codec_enc;
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
        if (const Json* sp = j.get("speaker_encoder_config"); sp && sp->kind == Json::Obj) {  // Qwen3.swift:55-57
            has_speaker_encoder = true;
            speaker.parse(*sp);
        }
    }
    // What the engine relies on without looking again: sizes that are sizes, one intermediate size per layer, ids inside the tables
    // they index (an id beyond the codec vocabulary would be a row gather outside the embedding table ON THE GPU). Every field of
    // a config.json arrives through an `int(...)` of whatever number the file held; this is where a damaged or hostile file stops.
    // The tensors' shapes are checked against these numbers afterwards (model.cc), so nothing here needs to know the kernels' tiles.
    static void in_range(int64_t v, int64_t lo, int64_t hi, const std::string& what) {
        Q3_CHECK(v >= lo && v <= hi, 6, "config.json: " + what + " = " + std::to_string(v) + " is outside [" + std::to_string(lo) + ", " + std::to_string(hi) + "]");
    }
    static void finite_pos(float v, const std::string& what) {
        Q3_CHECK(v > 0.0f && v < 3.0e38f, 6, "config.json: " + what + " must be a positive finite number");
    }
    void validate() const {
        if (has_talker) {
            const TalkerConfig& t = talker;
            in_range(t.hidden_size, 16, 1 << 16, "talker hidden_size");
            in_range(t.text_hidden_size, 16, 1 << 16, "talker text_hidden_size");
            in_range(t.intermediate_size, 16, 1 << 20, "talker intermediate_size");
            in_range(t.num_hidden_layers, 1, 1024, "talker num_hidden_layers");
            in_range(t.num_attention_heads, 1, 1024, "talker num_attention_heads");
            in_range(t.num_key_value_heads, 1, t.num_attention_heads, "talker num_key_value_heads");
            Q3_CHECK(t.num_attention_heads % t.num_key_value_heads == 0, 6, "config.json: talker attention heads are not a multiple of the kv heads");
            in_range(t.vocab_size, 16, 1 << 24, "talker vocab_size");
            in_range(t.text_vocab_size, 1, 1 << 24, "talker text_vocab_size");
            in_range(t.num_code_groups, 2, 64, "talker num_code_groups");
            Q3_CHECK(t.per_layer_intermediate_sizes.empty() || int64_t(t.per_layer_intermediate_sizes.size()) == t.num_hidden_layers, 6,
                     "config.json: per_layer_intermediate_sizes must name every layer");
            for (int v : t.per_layer_intermediate_sizes) in_range(v, 16, 1 << 20, "a per-layer intermediate size");
            finite_pos(t.rms_norm_eps, "talker rms_norm_eps");
            finite_pos(t.rope_theta, "talker rope_theta");
            for (auto [v, name] : {std::pair<int, const char*>{t.codec_eos_token_id, "codec_eos_token_id"}, {t.codec_think_id, "codec_think_id"},
                                   {t.codec_nothink_id, "codec_nothink_id"}, {t.codec_think_bos_id, "codec_think_bos_id"},
                                   {t.codec_think_eos_id, "codec_think_eos_id"}, {t.codec_pad_id, "codec_pad_id"}, {t.codec_bos_id, "codec_bos_id"}})
                in_range(v, 0, t.vocab_size - 1, name);
            for (auto& kv : t.codec_language_id) in_range(kv.second, 0, t.vocab_size - 1, "codec_language_id." + kv.first);
            for (auto& kv : t.spk_id) in_range(kv.second, 0, t.vocab_size - 1, "spk_id." + kv.first);
            for (auto [v, name] : {std::pair<int, const char*>{tts_pad_token_id, "tts_pad_token_id"}, {tts_bos_token_id, "tts_bos_token_id"},
                                   {tts_eos_token_id, "tts_eos_token_id"}})
                in_range(v, 0, t.text_vocab_size - 1, name);
            if (t.has_code_predictor) {
                const CodePredictorConfig& c = t.cp;
                in_range(c.hidden_size, 16, 1 << 16, "code predictor hidden_size");
                in_range(c.intermediate_size, 16, 1 << 20, "code predictor intermediate_size");
                in_range(c.num_hidden_layers, 1, 1024, "code predictor num_hidden_layers");
                in_range(c.num_attention_heads, 1, 1024, "code predictor num_attention_heads");
                in_range(c.num_key_value_heads, 1, c.num_attention_heads, "code predictor num_key_value_heads");
                Q3_CHECK(c.num_attention_heads % c.num_key_value_heads == 0, 6, "config.json: code predictor attention heads are not a multiple of the kv heads");
                in_range(c.vocab_size, 16, 1 << 24, "code predictor vocab_size");
                finite_pos(c.rms_norm_eps, "code predictor rms_norm_eps");
                finite_pos(c.rope_theta, "code predictor rope_theta");
            }
        }
        in_range(sample_rate, 1, 1 << 20, "sample_rate");
        if (has_quantization) {
            in_range(quant_bits, 1, 8, "quantization.bits");
            in_range(quant_group_size, 1, 1 << 16, "quantization.group_size");
        }
        if (has_speaker_encoder) {
            const SpeakerEncoderConfig& s = speaker;
            in_range(s.mel_dim, 1, 4096, "speaker encoder mel_dim");
            in_range(s.enc_dim, 1, 1 << 16, "speaker encoder enc_dim");
            in_range(s.enc_attention_channels, 1, 1 << 16, "speaker encoder enc_attention_channels");
            in_range(s.enc_res2net_scale, 1, 64, "speaker encoder enc_res2net_scale");
            in_range(s.enc_se_channels, 1, 1 << 16, "speaker encoder enc_se_channels");
            in_range(s.sample_rate, 1, 1 << 20, "speaker encoder sample_rate");
            in_range(int64_t(s.enc_channels.size()), 1, 16, "speaker encoder enc_channels (count)");
            for (int v : s.enc_channels) in_range(v, 1, 1 << 16, "a speaker encoder channel count");
            for (int v : s.enc_kernel_sizes) in_range(v, 1, 64, "a speaker encoder kernel size");
            for (int v : s.enc_dilations) in_range(v, 1, 64, "a speaker encoder dilation");
        }
        if (has_codec) {
            const CodecDecoderConfig& d = codec;
            in_range(decode_upsample_rate, 1, 1 << 20, "decode_upsample_rate");
            in_range(d.latent_dim, 1, 1 << 16, "codec latent_dim");
            in_range(d.codebook_dim, 2, 1 << 16, "codec codebook_dim");
            in_range(d.codebook_size, 1, 1 << 24, "codec codebook_size");
            in_range(d.semantic_codebook_size, 1, 1 << 24, "codec semantic_codebook_size");
            in_range(d.decoder_dim, 1, 1 << 16, "codec decoder_dim");
            in_range(d.hidden_size, 1, 1 << 16, "codec hidden_size");
            in_range(d.intermediate_size, 1, 1 << 20, "codec intermediate_size");
            in_range(d.num_hidden_layers, 0, 1024, "codec num_hidden_layers");
            in_range(d.num_attention_heads, 1, 1024, "codec num_attention_heads");
            in_range(d.num_quantizers, 1, 256, "codec num_quantizers");
            in_range(d.num_semantic_quantizers, 0, d.num_quantizers, "codec num_semantic_quantizers");
            finite_pos(d.rms_norm_eps, "codec rms_norm_eps");
            in_range(int64_t(d.upsample_rates.size()), 1, 16, "codec upsample_rates (count)");
            in_range(int64_t(d.upsampling_ratios.size()), 0, 16, "codec upsampling_ratios (count)");
            int64_t total = 1;
            for (int v : d.upsample_rates) { in_range(v, 1, 64, "a codec upsample rate"); total *= v; }
            for (int v : d.upsampling_ratios) { in_range(v, 1, 64, "a codec upsampling ratio"); total *= v; }
            in_range(total, 1, 1 << 20, "the codec's total upsampling");
        }
        if (has_codec_encoder) {
            const CodecEncoderConfig& e = codec_enc;
            Q3_CHECK(e.frame_rate >= 0.01f && e.frame_rate <= 1.0e6f, 6, "config.json: codec encoder frame_rate is outside [0.01, 1e6]");
            finite_pos(e.rope_theta, "codec encoder rope_theta");
            in_range(e.codebook_dim, 1, 1 << 16, "codec encoder codebook_dim");
            in_range(e.codebook_size, 1, 1 << 24, "codec encoder codebook_size");
            in_range(e.hidden_size, 1, 1 << 16, "codec encoder hidden_size");
            in_range(e.intermediate_size, 1, 1 << 20, "codec encoder intermediate_size");
            in_range(e.kernel_size, 1, 64, "codec encoder kernel_size");
            in_range(e.last_kernel_size, 1, 64, "codec encoder last_kernel_size");
            in_range(e.residual_kernel_size, 1, 64, "codec encoder residual_kernel_size");
            in_range(e.compress, 1, 64, "codec encoder compress");
            in_range(e.dilation_growth_rate, 1, 64, "codec encoder dilation_growth_rate");
            in_range(e.num_filters, 1, 1 << 16, "codec encoder num_filters");
            in_range(e.num_hidden_layers, 0, 1024, "codec encoder num_hidden_layers");
            in_range(e.num_attention_heads, 1, 1024, "codec encoder num_attention_heads");
            in_range(e.num_key_value_heads, 1, 1024, "codec encoder num_key_value_heads");
            in_range(e.num_quantizers, 1, 256, "codec encoder num_quantizers");
            in_range(e.num_residual_layers, 0, 64, "codec encoder num_residual_layers");
            in_range(e.sampling_rate, 1, 1 << 20, "codec encoder sampling_rate");
            in_range(e.max_position_embeddings, 1, 1 << 24, "codec encoder max_position_embeddings");
            in_range(int64_t(e.upsampling_ratios.size()), 1, 16, "codec encoder upsampling_ratios (count)");
            int64_t hop = 1;
            for (int v : e.upsampling_ratios) { in_range(v, 1, 64, "a codec encoder ratio"); hop *= v; }
            in_range(hop, 1, 1 << 20, "the codec encoder's hop");
            in_range(e.downsample_stride(), 1, 1 << 16, "the codec encoder's downsample stride");
        }
    }
    void parse_speech_tokenizer(const Json& j) {
        decode_upsample_rate = int(j.i64("decode_upsample_rate", decode_upsample_rate));
        if (const Json* e = j.get("encoder_config"); e && e->kind == Json::Obj) {  // SpeechTokenizer.swift:808-812
            has_codec_encoder = true;
            codec_enc.parse(*e);
        }
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
