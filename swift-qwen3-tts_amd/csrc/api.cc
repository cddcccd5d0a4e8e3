// api.cc -- extern "C" surface declared in include/q3tts.h. No exception crosses the boundary:
// every entry point returns a q3tts_status and records the message for q3tts_last_error().
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "codec.h"
#include "comm.h"
#include "engine.h"
#include "tokenizer.h"

using q3::EngineGroup;

struct q3tts_model {
    std::unique_ptr<EngineGroup> eng;
    // "Calls on one handle are serialised by the caller" (q3tts.h; the reference's model object is not re-entrant either). The
    // contract is checked, not assumed: a call that finds the handle inside another thread's call fails with INVALID_INPUT instead
    // of racing on the engine's state. Nested calls from the SAME thread (an event callback asking for q3tts_model_info) pass.
    std::atomic<std::thread::id> owner{std::thread::id()};
    std::atomic<int> depth{0};
};
struct q3tts_tokenizer {
    q3::BpeTokenizer tok;
};

namespace {
thread_local std::string g_load_error;
thread_local const q3tts_model* g_refused = nullptr;  // this thread's last call on that handle found it in use (HandleUse)
const char* const kBusyMsg =
    "Invalid input: the model handle is inside another thread's call (calls on one handle must be serialised by the caller)";

struct HandleUse {  // the handle's one-caller-at-a-time contract (struct q3tts_model)
    q3tts_model* m;
    bool mine = false;
    explicit HandleUse(q3tts_model* model) : m(model) {
        if (!m) return;
        const std::thread::id me = std::this_thread::get_id();
        std::thread::id none;
        if (m->owner.load(std::memory_order_acquire) == me || m->owner.compare_exchange_strong(none, me, std::memory_order_acq_rel)) {
            m->depth.fetch_add(1, std::memory_order_relaxed);
            mine = true;
        }
    }
    ~HandleUse() {
        if (m && mine && m->depth.fetch_sub(1, std::memory_order_relaxed) == 1) m->owner.store(std::thread::id(), std::memory_order_release);
    }
};

template <class F>
q3tts_status guarded(q3tts_model* m, F&& f) {
    HandleUse use(m);
    if (m && !use.mine) {
        g_refused = m;  // (not written to the engine's last_error: that string belongs to the call that is running)
        return Q3TTS_ERR_INVALID_INPUT;
    }
    if (m && g_refused == m) g_refused = nullptr;
    try {
        // every call on a handle runs with the handle's GPU current, whatever the calling thread had selected (a process may hold one
        // handle per GPU; allocations and per-device kernel attributes inside the call belong to THIS device)
        if (m && m->eng) Q3_HIP(hipSetDevice(m->eng->model().device));
        f();
        return Q3TTS_OK;
    } catch (const q3::Error& e) {
        if (m && m->eng) m->eng->last_error = e.what();
        else g_load_error = e.what();
        return static_cast<q3tts_status>(e.status);
    } catch (const std::exception& e) {
        if (m && m->eng) m->eng->last_error = e.what();
        else g_load_error = e.what();
        return Q3TTS_ERR_DEVICE;
    }
}
}  // namespace

extern "C" {

void q3tts_default_load_opts(q3tts_load_opts* o) {
    std::memset(o, 0, sizeof(*o));
    o->device = 0;
    o->max_batch = 1;
    o->max_frames = 2048;
    o->max_prompt = 512;
    o->use_graph = 1;
    o->weights_from_broadcast = 0;
    o->n_streams = 0;
    o->codec_overlap_cus = 0;
    o->codec_fp32 = 0;
}

void q3tts_default_sampling(q3tts_sampling* s) {  // Qwen3.swift:1296-1299
    s->temperature = 0.9f;
    s->top_k = 50;
    s->top_p = 1.0f;
    s->repetition_penalty = 1.05f;
    s->seed = 0;
    s->force_frames = 0;
    s->audio_chunk_frames = 0;
    s->audio_window_frames = 0;
    s->audio_lookahead_frames = 4;
    s->row_base = 0;
}

q3tts_status q3tts_model_load(const char* model_dir, const q3tts_load_opts* opts, q3tts_model** out) {
    if (out) *out = nullptr;
    return guarded(nullptr, [&] {
        Q3_CHECK(model_dir && out, 3, "Invalid input: null argument");
        q3::debug_env_reload();  // the launchers' diagnostic switches are read here, once per load (kernels.h DebugEnv)
        q3tts_load_opts o;
        if (opts) o = *opts;
        else q3tts_default_load_opts(&o);
        Q3_CHECK(o.max_batch >= 1 && o.max_batch <= 64, 3, "Invalid input: max_batch must be in 1..64");
        Q3_CHECK(o.max_frames >= 1 && o.max_prompt >= 16, 3, "Invalid input: max_frames / max_prompt too small");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        Q3_CHECK(e == hipSuccess && ndev > 0, 7, "no HIP device available: this engine has no CPU fallback");
        Q3_CHECK(o.device >= 0 && o.device < ndev, 3, "Invalid input: device ordinal out of range");
        q3::LoadOptions lo;
        lo.device = o.device;
        lo.max_pos_talker = o.max_prompt + o.max_frames + 8;
        lo.skip_tensor_data = o.weights_from_broadcast != 0;
        auto model = q3::load_model(model_dir, lo);
        auto h = std::make_unique<q3tts_model>();
        h->eng = std::make_unique<EngineGroup>(std::move(model), o);
        *out = h.release();
    });
}

void q3tts_model_free(q3tts_model* m) {
    if (g_refused == m) g_refused = nullptr;  // (a later handle may be allocated at the same address)
    delete m;
}

const char* q3tts_last_error(const q3tts_model* m) {
    if (m && g_refused == m) return kBusyMsg;
    if (m && m->eng) return m->eng->last_error.c_str();
    return g_load_error.c_str();
}

q3tts_status q3tts_model_arena(q3tts_model* m, void** device_ptr, size_t* bytes) {
    return guarded(m, [&] {
        Q3_CHECK(m && device_ptr && bytes, 3, "Invalid input: null argument");
        *device_ptr = m->eng->model().arena;
        *bytes = m->eng->model().arena_bytes;
    });
}

q3tts_status q3tts_comm_get_unique_id(q3tts_comm_id* out) {
    return guarded(nullptr, [&] {
        Q3_CHECK(out, 3, "Invalid input: null argument");
        q3::comm_unique_id(out);
    });
}

q3tts_status q3tts_model_broadcast(q3tts_model* m, const q3tts_comm_id* id, int32_t rank, int32_t world, int32_t root) {
    return guarded(m, [&] {
        Q3_CHECK(m && id, 3, "Invalid input: null argument");
        q3::Model& md = m->eng->model();
        q3::comm_broadcast_arena(md.device, md.arena, md.arena_bytes, *id, rank, world, root);
    });
}

q3tts_status q3tts_model_arena_checksum(q3tts_model* m, uint64_t* out) {
    return guarded(m, [&] {
        Q3_CHECK(m && out, 3, "Invalid input: null argument");
        q3::Model& md = m->eng->model();
        *out = q3::arena_checksum(md.device, md.arena, md.arena_bytes);
    });
}

q3tts_status q3tts_model_get_info(const q3tts_model* m, q3tts_model_info* out) {
    return guarded(const_cast<q3tts_model*>(m), [&] {
        Q3_CHECK(m && out, 3, "Invalid input: null argument");
        std::memset(out, 0, sizeof(*out));
        const q3::Model& md = m->eng->model();
        const q3::ModelConfig& c = md.cfg;
        std::strncpy(out->tts_model_type, c.tts_model_type.c_str(), sizeof(out->tts_model_type) - 1);
        out->sample_rate = c.sample_rate;
        // supportsVoiceCloning: base model with a codec encoder (Qwen3.swift:1210-1214); hasVoiceCloning: speaker encoder (:61-63)
        out->supports_voice_cloning = (c.tts_model_type == "base" && md.has_codec && md.has_codec_encoder) ? 1 : 0;
        out->has_voice_cloning = md.has_speaker_encoder ? 1 : 0;
        out->speaker_embedding_dim = md.has_speaker_encoder ? md.speaker.enc_dim : 0;
        out->hidden_size = c.talker.hidden_size;
        out->num_layers = c.talker.num_hidden_layers;
        out->vocab_size = c.talker.vocab_size;
        out->text_vocab_size = c.talker.text_vocab_size;
        out->num_code_groups = c.talker.num_code_groups;
        out->cp_hidden_size = c.talker.cp.hidden_size;
        out->cp_num_layers = c.talker.cp.num_hidden_layers;
        out->cp_vocab_size = c.talker.cp.vocab_size;
        out->codec_eos_token_id = c.talker.codec_eos_token_id;
        out->samples_per_frame = c.has_codec ? c.codec.total_upsample() : c.decode_upsample_rate;
        out->max_batch = m->eng->opts().max_batch;
        out->weight_bytes = md.step_weight_bytes;
    });
}

int32_t q3tts_model_num_speakers(const q3tts_model* m) { return m ? int32_t(m->eng->speakers.size()) : 0; }
const char* q3tts_model_speaker_name(const q3tts_model* m, int32_t i) {
    if (!m || i < 0 || size_t(i) >= m->eng->speakers.size()) return nullptr;
    return m->eng->speakers[size_t(i)].c_str();
}

q3tts_status q3tts_generate(q3tts_model* m, const q3tts_request* reqs, int32_t n_reqs, const q3tts_sampling* sampling,
                            q3tts_event_cb cb, void* user, q3tts_result* results) {
    return guarded(m, [&] {
        Q3_CHECK(m && reqs && results, 3, "Invalid input: null argument");
        q3tts_sampling sp;
        if (sampling) sp = *sampling;
        else q3tts_default_sampling(&sp);
        std::memset(results, 0, sizeof(q3tts_result) * size_t(n_reqs > 0 ? n_reqs : 0));
        m->eng->generate(reqs, n_reqs, sp, cb, user, results, nullptr);
        for (int i = 0; i < n_reqs; ++i)
            if (results[i].status == Q3TTS_ERR_GENERATION_FAILED)
                m->eng->last_error = "Generation failed: No tokens generated";  // Qwen3.swift:940
    });
}

struct q3tts_job {
    int slot = -1;
    int n = 0;
};

q3tts_status q3tts_generate_begin(q3tts_model* m, const q3tts_request* reqs, int32_t n_reqs, const q3tts_sampling* sampling,
                                  q3tts_event_cb cb, void* user, int32_t more_follows, q3tts_job** job) {
    return guarded(m, [&] {
        Q3_CHECK(m && reqs && job, 3, "Invalid input: null argument");
        *job = nullptr;
        q3tts_sampling sp;
        if (sampling) sp = *sampling;
        else q3tts_default_sampling(&sp);
        auto j = std::make_unique<q3tts_job>();
        j->slot = m->eng->begin(reqs, n_reqs, sp, cb, user, more_follows != 0);
        j->n = n_reqs;
        *job = j.release();
    });
}

q3tts_status q3tts_generate_end(q3tts_model* m, q3tts_job* job, q3tts_result* results) {
    return guarded(m, [&] {
        Q3_CHECK(m && job && results, 3, "Invalid input: null argument");
        std::unique_ptr<q3tts_job> j(job);  // the job is released whatever happens
        std::memset(results, 0, sizeof(q3tts_result) * size_t(j->n > 0 ? j->n : 0));
        m->eng->end(j->slot, results);
        for (int i = 0; i < j->n; ++i)
            if (results[i].status == Q3TTS_ERR_GENERATION_FAILED)
                m->eng->last_error = "Generation failed: No tokens generated";  // Qwen3.swift:940
    });
}

void q3tts_pcm_to_int16(const float* pcm, int64_t n_samples, int16_t* out) {
    for (int64_t i = 0; i < n_samples; ++i) {
        const float clamped = std::fmax(-1.0f, std::fmin(1.0f, pcm[i]));  // main.swift:159
        out[i] = static_cast<int16_t>(clamped * 32767.0f);                // Int16(Float): toward zero (:160)
    }
}

q3tts_status q3tts_write_wav(const char* path, const float* pcm, int64_t n_samples, int32_t sample_rate) {
    return guarded(nullptr, [&] {
        Q3_CHECK(path && (pcm || n_samples == 0) && n_samples >= 0 && sample_rate > 0, 3, "Invalid input: null argument");
        Q3_CHECK(n_samples <= (int64_t(0xffffffffu) - 36) / 2, 3, "Invalid input: too many samples for a RIFF file");
        std::vector<uint8_t> d;
        d.reserve(size_t(44 + n_samples * 2));
        auto u32 = [&](uint32_t v) { for (int i = 0; i < 4; ++i) d.push_back(uint8_t(v >> (8 * i))); };
        auto u16 = [&](uint16_t v) { d.push_back(uint8_t(v)); d.push_back(uint8_t(v >> 8)); };
        auto tag = [&](const char* t) { d.insert(d.end(), t, t + 4); };
        tag("RIFF"); u32(uint32_t(36 + n_samples * 2)); tag("WAVE");          // main.swift:138-142
        tag("fmt "); u32(16); u16(1); u16(1); u32(uint32_t(sample_rate));      // :145-149
        u32(uint32_t(sample_rate) * 2); u16(2); u16(16);                       // :150-152
        tag("data"); u32(uint32_t(n_samples * 2));                             // :155-156
        std::vector<int16_t> s16(static_cast<size_t>(n_samples));
        q3tts_pcm_to_int16(pcm, n_samples, s16.data());
        for (int16_t v : s16) u16(uint16_t(v));
        FILE* f = std::fopen(path, "wb");
        Q3_CHECK(f != nullptr, 3, std::string("Invalid input: cannot open ") + path);
        const size_t w = std::fwrite(d.data(), 1, d.size(), f);
        const int rc = std::fclose(f);
        Q3_CHECK(w == d.size() && rc == 0, 3, std::string("Invalid input: short write to ") + path);
    });
}

void q3tts_result_free(q3tts_result* results, int32_t n) {
    if (!results) return;
    for (int i = 0; i < n; ++i) {
        std::free(results[i].pcm);
        std::free(results[i].codes);
        results[i].pcm = nullptr;
        results[i].codes = nullptr;
    }
}

q3tts_status q3tts_codec_decode(q3tts_model* m, const int32_t* codes, const int32_t* n_frames, int32_t batch,
                                int32_t max_frames, float* pcm, int64_t* audio_lengths) {
    return guarded(m, [&] {
        Q3_CHECK(m && codes && n_frames && pcm && audio_lengths, 3, "Invalid input: null argument");
        m->eng->lane0().codec_decode(codes, n_frames, batch, max_frames, pcm, audio_lengths);
        m->eng->timing.codec_ms = m->eng->lane0().timing.codec_ms;
    });
}

q3tts_status q3tts_codec_encode(q3tts_model* m, const float* audio, int64_t n_samples, int32_t* codes, int32_t cap_frames,
                                int32_t* n_frames) {
    return guarded(m, [&] {
        Q3_CHECK(m && audio && codes && n_frames, 3, "Invalid input: null argument");
        *n_frames = m->eng->lane0().codec_encode(audio, n_samples, codes, cap_frames);
        m->eng->timing = m->eng->lane0().timing;
    });
}

int32_t q3tts_codec_encoded_frames(const q3tts_model* m, int64_t n_samples) {
    if (!m || n_samples <= 0 || n_samples > (int64_t(1) << 24)) return 0;
    return m->eng->lane0().encoded_frames(n_samples);
}

q3tts_status q3tts_speaker_embedding(q3tts_model* m, const float* audio, int64_t n_samples, int32_t sample_rate, float* out,
                                     int32_t cap) {
    return guarded(m, [&] {
        Q3_CHECK(m && audio && out, 3, "Invalid input: null argument");
        Q3_CHECK(sample_rate == 24000, 3,
                 "Invalid input: Only 24kHz audio is supported for speaker embedding extraction");  // Qwen3.swift:223-225
        m->eng->lane0().speaker_embedding(audio, n_samples, out, cap);
        m->eng->timing = m->eng->lane0().timing;
    });
}

q3tts_status q3tts_debug_frontend_stage(q3tts_model* m, const float* audio, int64_t n_samples, const char* stage, float* out,
                                        int64_t cap_floats, int32_t* T, int32_t* C) {
    return guarded(m, [&] {
        Q3_CHECK(m && audio && stage && out && T && C, 3, "Invalid input: null argument");
        m->eng->lane0().debug_frontend_stage(audio, n_samples, stage, out, cap_floats, T, C);
    });
}

q3tts_status q3tts_tokenizer_load(const char* path, q3tts_tokenizer** out) {
    return guarded(nullptr, [&] {
        Q3_CHECK(path && out, 3, "Invalid input: null argument");
        auto t = std::make_unique<q3tts_tokenizer>();
        const std::string p = path;
        if (p.size() > 5 && p.compare(p.size() - 5, 5, ".json") == 0) t->tok.load_json_file(p);
        else t->tok.load(p);
        *out = t.release();
    });
}
void q3tts_tokenizer_free(q3tts_tokenizer* t) { delete t; }
q3tts_status q3tts_tokenizer_encode(const q3tts_tokenizer* t, const char* utf8, int32_t* ids, int32_t cap, int32_t* n) {
    return guarded(nullptr, [&] {
        Q3_CHECK(t && utf8 && n, 3, "Invalid input: null argument");
        const std::vector<int32_t> v = t->tok.encode(utf8);
        *n = int32_t(v.size());
        Q3_CHECK(ids == nullptr || cap >= *n, 3, "Invalid input: output buffer too small for the token ids");
        if (ids) std::memcpy(ids, v.data(), v.size() * 4);
    });
}

q3tts_status q3tts_last_timing(const q3tts_model* m, q3tts_timing* out) {
    if (!m || !out) return Q3TTS_ERR_INVALID_INPUT;
    *out = m->eng->timing;
    return Q3TTS_OK;
}

q3tts_status q3tts_debug_prepare_inputs(q3tts_model* m, const q3tts_request* req, uint16_t* input_embeds,
                                        int32_t cap_prompt, int32_t* n_prompt, uint16_t* trailing, int32_t cap_trailing,
                                        int32_t* n_trailing, uint16_t* tts_pad) {
    return guarded(m, [&] {
        Q3_CHECK(m && req && input_embeds && n_prompt && trailing && n_trailing && tts_pad, 3, "Invalid input: null argument");
        m->eng->lane0().debug_prepare_inputs(*req, input_embeds, cap_prompt, n_prompt, trailing, cap_trailing, n_trailing, tts_pad);
    });
}

q3tts_status q3tts_debug_generate_forced(q3tts_model* m, const q3tts_request* reqs, int32_t n_reqs,
                                         const q3tts_sampling* sampling, const int32_t* forced_codes, int32_t n_frames,
                                         uint16_t* talker_logits, uint16_t* cp_logits, int32_t* sampled) {
    return guarded(m, [&] {
        Q3_CHECK(m && reqs && n_frames > 0, 3, "Invalid input: null argument");
        q3tts_sampling sp;
        if (sampling) sp = *sampling;
        else q3tts_default_sampling(&sp);
        q3::DebugOpts d;
        d.forced_codes = forced_codes;
        d.frames = n_frames;
        d.talker_logits = talker_logits;
        d.cp_logits = cp_logits;
        d.sampled = sampled;
        std::vector<q3tts_result> res((size_t)(n_reqs));
        m->eng->generate(reqs, n_reqs, sp, nullptr, nullptr, res.data(), &d);
        q3tts_result_free(res.data(), n_reqs);
    });
}

q3tts_status q3tts_debug_sample(q3tts_model* m, const uint16_t* logits, int32_t rows, int32_t V,
                                const q3tts_sampling* sampling, const uint8_t* seen, int32_t suppress_lo,
                                int32_t suppress_hi, int32_t eos_id, uint32_t row0, uint32_t draw, int32_t* tokens) {
    return guarded(m, [&] {
        Q3_CHECK(m && logits && sampling && tokens, 3, "Invalid input: null argument");
        m->eng->lane0().debug_sample(logits, rows, V, *sampling, seen, suppress_lo, suppress_hi, eos_id, row0, draw, tokens);
    });
}

q3tts_status q3tts_debug_linear(q3tts_model* m, const uint16_t* x, const uint16_t* W, const uint16_t* bias, int32_t M,
                                int32_t K, int32_t N, uint16_t* y) {
    return guarded(m, [&] {
        Q3_CHECK(m && x && W && y, 3, "Invalid input: null argument");
        m->eng->lane0().debug_linear(x, W, bias, M, K, N, y);
    });
}

q3tts_status q3tts_codec_decode_streamed(q3tts_model* m, const int32_t* codes, const int32_t* n_frames, int32_t batch, int32_t max_frames,
                                         int32_t chunk_frames, int32_t window, int32_t lookahead, float* pcm) {
    return guarded(m, [&] {
        Q3_CHECK(m && codes && n_frames && pcm, 3, "Invalid input: null argument");
        m->eng->lane0().codec_decode_streamed(codes, n_frames, batch, max_frames, chunk_frames, window, lookahead, pcm);
    });
}

void q3tts_debug_set_codec_scratch(uint64_t bytes) { q3::CodecRunner::set_scratch_budget(size_t(bytes)); }

void q3tts_debug_reload_env(void) { q3::debug_env_reload(); }

q3tts_status q3tts_debug_codec_stage(q3tts_model* m, const int32_t* codes, int32_t n_frames, const char* stage,
                                     float* out, int64_t cap_floats, int32_t* T, int32_t* C) {
    return guarded(m, [&] {
        Q3_CHECK(m && codes && stage && out && T && C && n_frames > 0, 3, "Invalid input: null argument");
        m->eng->lane0().debug_codec_stage(codes, n_frames, stage, out, cap_floats, T, C);
    });
}

}  // extern "C"
