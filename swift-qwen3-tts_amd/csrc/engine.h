// engine.h -- host-side generation driver on top of the HIP kernels.
//
// Mirrors the reference's generation driver (/root/reference/Sources/Qwen3TTS/Models/Qwen3.swift:
// prompt assembly :259-409, AR loop :847-936 / :640-729, routing :1291-1373, decode + trim
// :943-961) with batching added: rows are independent sequences with their own KV pages,
// repetition sets and RNG streams.
#pragma once
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/q3tts.h"
#include "kernels.h"
#include "model.h"

namespace q3 {

struct DebugOpts {
    const int32_t* forced_codes = nullptr;  // host [n][frames][16]
    int frames = 0;
    uint16_t* talker_logits = nullptr;  // host [n][frames][V]
    uint16_t* cp_logits = nullptr;      // host [n][frames][groups-1][Vcp]
    int32_t* sampled = nullptr;         // host [n][frames][16]
};

struct ResolvedRequest {
    std::vector<int32_t> text_ids, instruct_ids;
    int speaker_token = -1;  // row of the codec embedding table, -1: none
    int language_id = -1;    // -1: none ("auto" without dialect)
    int max_frames = 0;
    int target_token_count = 0;
    // voice clone (generateVoiceClone, Qwen3.swift:1009-1203)
    bool clone = false;
    std::vector<int32_t> ref_text_ids;
    const float* ref_audio = nullptr;  // caller memory, valid during the call
    int64_t n_ref_samples = 0;
    int ref_T = 0;        // reference frames (filled by prepare_clone_rows)
    int extra_base = 0;   // first row of this request in extra_: speaker x-vector, then ref_T embedding sums
    int ref_off = 0;      // offset of this request's [16][ref_T] codes in ref_codes_dev_
};

class CodecRunner;
class VoiceFrontEnd;

class Engine {
  public:
    // One lane: a slice of the batch with its own stream, workspace, KV pool and frame graph.
    Engine(Model* model, const q3tts_load_opts& opts);
    ~Engine();

    Model& model() { return *m_; }
    const q3tts_load_opts& opts() const { return opts_; }
    std::string last_error;
    q3tts_timing timing{};

    void generate(const q3tts_request* reqs, int n, const q3tts_sampling& sp, q3tts_event_cb cb, void* user,
                  q3tts_result* results, const DebugOpts* dbg);
    // generate() in two halves (q3tts_generate_begin / _end): begin returns once the AR loop has finished and the codec
    // decode of its codes is queued on the codec stream; end waits for the PCM and fills the results. A second begin()
    // may run between the two: its AR loop (a latency-bound chain that leaves the matrix cores idle) then overlaps the
    // first job's decode (matrix-core bound). Jobs may end in any order; at most kJobSlots are outstanding.
    static constexpr int kJobSlots = 2;
    int begin(const q3tts_request* reqs, int n, const q3tts_sampling& sp, q3tts_event_cb cb, void* user, const DebugOpts* dbg,
              bool overlapped);  // overlapped: another batch's AR loop is expected to run beside this one's decode
    void end(int job, q3tts_result* results);
    void debug_prepare_inputs(const q3tts_request& req, uint16_t* input_embeds, int cap_prompt, int* n_prompt,
                              uint16_t* trailing, int cap_trailing, int* n_trailing, uint16_t* tts_pad);
    void debug_sample(const uint16_t* logits, int rows, int V, const q3tts_sampling& sp, const uint8_t* seen,
                      int suppress_lo, int suppress_hi, int eos_id, uint32_t row0, uint32_t draw, int32_t* tokens);
    void debug_linear(const uint16_t* x, const uint16_t* W, const uint16_t* bias, int M, int K, int N, uint16_t* y);
    void codec_decode(const int32_t* codes, const int32_t* n_frames, int batch, int max_frames, float* pcm,
                      int64_t* audio_lengths);
    void codec_decode_streamed(const int32_t* codes, const int32_t* n_frames, int batch, int max_frames, int chunk_frames, int window,
                               int lookahead, float* pcm);
    void debug_codec_stage(const int32_t* codes, int n_frames, const char* stage, float* out, int64_t cap, int* T, int* C);
    // voice-clone front end (SpeechTokenizer.swift:841-846; Qwen3.swift:222-249); host buffers in and out
    int codec_encode(const float* audio, int64_t n_samples, int32_t* codes, int cap_frames);
    int encoded_frames(int64_t n_samples) const;
    void speaker_embedding(const float* audio, int64_t n_samples, float* out, int cap);
    void debug_frontend_stage(const float* audio, int64_t n_samples, const char* stage, float* out, int64_t cap, int* T, int* C);

    std::vector<std::string> speakers;  // sorted (Qwen3.swift:965-971)

  private:
    Model* m_;
  public:
    uint32_t row_offset = 0;       // global index of this lane's first row (RNG stream id)
    std::mutex* cb_mutex = nullptr;  // serialises event callbacks across lanes
    int request_base = 0;          // added to request_index in events
    int max_inflight_frames = 16;  // frame steps queued but not finished (two bursts of half this)
  private:
    q3tts_load_opts opts_;
    hipStream_t st_ = nullptr;
    hipStream_t st_codec_ = nullptr;       // codec decode that nothing overlaps (lower priority than st_)
    hipStream_t st_codec_part_ = nullptr;  // codec decode beside the next batch's AR loop: confined to half of the CUs
    hipStream_t codec_stream(bool overlapped);
    hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t burst_ev_[2] = {nullptr, nullptr};
    hipEvent_t ev_fe_[2] = {nullptr, nullptr};
    hipEvent_t fe_uploaded_ = nullptr;
    int Bm_ = 0, Mp_ = 0;  // max batch, padded to 16
    int Pcap_ = 0, Tcap_ = 0, Fcap_ = 0, max_pages_ = 0, n_pages_ = 0;

    // device workspace (one allocation)
    uint8_t* ws_ = nullptr;
    size_t ws_bytes_ = 0;
    struct Stream {  // activation buffers of one decoder stack
        uint16_t *h, *xn, *qkv, *ao, *act, *logits;  // h, xn, ao, act: fragment-major; qkv, logits: row-major
        float *ss_a, *ss_b;                     // per-tile sums of squares of h ([H/16][Mp])
        int ld_qkv, ld_act, ld_logits;
    } tk_{}, cp_{};
    uint16_t* cp_x_ = nullptr;   // fragment-major [Mp][H] code-predictor input before the projection
    uint16_t* cp_x2_ = nullptr;
    // small_to_mtp_projection applied to every row of the code predictor's embedding tables at load (by the decode GEMM
    // itself, so rows are bit-identical to projecting at run time); kept in the Model, see Model::cp_pe
    bool cp_tables_ = false;  // the samplers hand projected rows (Model::cp_pe) straight to the next pass
    void build_cp_proj_tables();  // staged embedding of code 0 (second position of predictor step 0)
    float* cp_ss2_ = nullptr;
    uint16_t *kpool_ = nullptr, *vpool_ = nullptr, *cp_kpool_ = nullptr, *cp_vpool_ = nullptr;
    size_t kv_layer_stride_ = 0, cp_kv_layer_stride_ = 0;
    int32_t *block_table_ = nullptr, *cp_block_table_ = nullptr;
    int32_t *kv_len_ = nullptr, *cp_len_ = nullptr, *n_frames_ = nullptr, *max_frames_ = nullptr;
    int32_t *trailing_idx_ = nullptr, *n_trailing_ = nullptr, *n_prompt_ = nullptr, *cur_codes_ = nullptr, *codes_ = nullptr;
    uint8_t *active_ = nullptr, *finished_ = nullptr, *seen_ = nullptr;
    uint16_t *prompt_ = nullptr, *trailing_ = nullptr, *tts_pad_ = nullptr;
    SamplingParams* sp_dev_ = nullptr;
    // prompt-assembly scratch
    int32_t* ids_dev_ = nullptr;
    uint16_t *proj_in_ = nullptr, *proj_mid_ = nullptr, *proj_out_ = nullptr;
    int proj_cap_ = 0;
    int32_t *compose_a_ = nullptr, *compose_b_ = nullptr;
    // debug buffers (allocated on demand)
    int32_t *forced_dev_ = nullptr, *sampled_dev_ = nullptr;
    uint16_t *tl_dump_ = nullptr, *cl_dump_ = nullptr;

    // One slot per outstanding job: everything the second half (codec decode -> results) needs after the next begin()
    // has started to overwrite the engine's per-call state.
    struct Job {
        bool busy = false;
        uint64_t seq = 0;                 // begin order
        int n = 0, Fdec = 0, up = 0;
        std::vector<int> frames, ref_T, target_tokens, n_prompt;
        std::vector<std::vector<int32_t>> ref_code0;  // first code row of each reference (valid-length count)
        std::vector<int32_t> codes_host;  // [n][Fcap][16]
        int32_t* dec_codes = nullptr;     // device [n][Fdec][16]: what the decoder reads (reference ++ generated for clone rows)
        size_t dec_codes_cap = 0;
        float* pcm_host = nullptr;        // pinned [n][Fdec * up]
        size_t pcm_host_cap = 0;
        hipEvent_t ev_codec[2] = {nullptr, nullptr};
        int32_t* nf_host = nullptr;  // pinned [max_batch]: rows whose waveform came out non-finite (CodecRunner::decode)
        int32_t* nf_chunk_host = nullptr;  // pinned [chunks][n]: the same flags behind every chunk of a streamed job
        size_t nf_chunk_cap = 0;
        std::vector<int> held_from;  // streamed job: first chunk of row b that is held back for the fp32 re-decode (-1: none)
        hipEvent_t ev_begin = nullptr, ev_first_audio = nullptr;  // request in / first streamed chunk on the host
        std::vector<hipEvent_t> chunk_done;  // chunked decode (audio_chunk_frames > 0): one per chunk, behind its copy
        int n_chunks = 0, chunk_frames = 0;
        bool streamed = false;   // audio_window_frames > 0: chunks were decoded (and partly delivered) inside the frame loop
        int chunks_fired = 0;    // AUDIO_CHUNK events already delivered for chunks [0, chunks_fired)
        double t_first_audio = 0;
        q3tts_timing timing{};
        double t_start = 0, t_done = 0;  // begin() entered / PCM on the host (stage_rows)
        q3tts_event_cb cb = nullptr;
        void* user = nullptr;
        int request_base = 0;
        bool decoded = false;
        // results of the rows: cut computed when the codes are known; PCM and codes copied out of the job's buffers once
        // the decode has finished -- by the staging thread for a pipelined job (so that end() hands over pointers while
        // the next batch's frame loop keeps the device busy), inside end() otherwise
        std::vector<int64_t> row_cut, row_ns;
        std::vector<float*> st_pcm;
        std::vector<int32_t*> st_codes;
        int stage = 0;  // 0: not staged, 1: queued for the staging thread, 2: staged, 3: staging failed (stage_err)
        std::string stage_err;
    } jobs_[kJobSlots];
    void compute_cuts(Job& J);
    // AUDIO_CHUNK events of chunks [J.chunks_fired, upto); rows are clipped to known[b] frames (their final length when known)
    void fire_chunks(Job& J, int upto, const std::vector<int>* known, bool wait);
    void stage_rows(Job& J);   // waits for the decode, then copies; throws
    // rows whose waveform left the fp16 range of the default codec kernels are decoded again on the fp32 matrix cores
    // (the reference's range) before end() hands them out; returns the rows that are non-finite even then
    std::vector<int> redo_rows_fp32(Job& J);
    void staging_loop();
    std::thread stager_;
    std::mutex stage_mu_;
    std::condition_variable stage_cv_;
    bool stage_stop_ = false;
    uint64_t job_seq_ = 0;

    unsigned long long* stamps_ = nullptr;  // Q3TTS_FRAME_STAMPS=1: [0] frame steps, [k] ticks of phase k, [63] last stamp
    void stamp(int k) {
        if (stamps_) other([&] { launch_stamp(stamps_, stamps_ + 63, k, st_); });
    }
    // ---- next-launch weight touch (kernels/prefetch.h) -----------------------------------------------------------------
    // A frame step is enqueued twice: a recording pass in which every launch site only notes what it would launch (the weight
    // stream it reads, whether its kernel carries the touch code and how many lines it can touch), then the real pass, in which
    // launch i receives the descriptor plan_touches() chose for it. plan_mode_ 0: launch sites launch directly (prefill, load).
    struct PlanItem {
        const uint8_t* w = nullptr;  // weight stream this launch reads (nullptr: none worth touching ahead)
        uint32_t span = 0, nspan = 0;
        uint32_t cap_lines = 0;      // lines per XCD this launch can touch (0: its kernel has no touch code)
    };
    int plan_mode_ = 0;  // 0 off, 1 recording, 2 replaying
    std::vector<PlanItem> plan_;
    std::vector<PfArgs> pf_;
    size_t plan_pos_ = 0;
    std::map<int, std::vector<PfArgs>> pf_plans_;  // keyed by batch size
    void plan_touches();
    void enqueue_frame_body(int B, const DebugOpts* dbg);
    void gemm(GemmArgs a);                     // launch_gemm_skinny through the plan
    bool gemm_with_norm_rows(GemmArgs a, const NormRowsArgs& n);
    void attn(AttnArgs a);                     // launch_attn_decode through the plan
    template <class F>
    void other(F&& f) {                        // any launch that neither streams weights worth touching nor touches
        if (plan_mode_ == 1) { plan_.emplace_back(); return; }
        if (plan_mode_ == 2) ++plan_pos_;
        ++launches_;
        f();
    }
    int launches_ = 0;                         // launches enqueued since enqueue_frame last reset it
    std::map<int, int> frame_launches_;        // launches of one frame step, keyed by batch size (q3tts_timing)
    std::map<int, hipGraphExec_t> graphs_;  // keyed by batch size
    std::unique_ptr<CodecRunner> codec_;
    std::unique_ptr<VoiceFrontEnd> fe_;
    // voice-clone scratch (grown on demand)
    float* ref_audio_dev_ = nullptr;
    size_t ref_audio_cap_ = 0;
    int32_t* ref_codes_dev_ = nullptr;
    size_t ref_codes_cap_ = 0;
    uint16_t* extra_ = nullptr;  // [rows][H] bf16: speaker x-vectors and reference-frame embedding sums
    size_t extra_cap_ = 0;
    float* spk_f32_ = nullptr;
    // Clone rows are independent and their front-end kernels are small: a few of them run side by side, each on its
    // own stream with its own scratch.
    struct FeLane {
        hipStream_t st = nullptr;
        hipEvent_t done = nullptr;
        std::unique_ptr<VoiceFrontEnd> fe;
        float* spk = nullptr;
    };
    std::vector<FeLane> fe_lanes_;
    const float* upload_audio(const float* audio, int64_t n);
    void prepare_clone_rows(std::vector<ResolvedRequest>& reqs);

    void alloc_workspace();
    ResolvedRequest resolve(const q3tts_request& r, const q3tts_sampling& sp) const;
    // builds prompt_/trailing_/tts_pad_ for rows [0,n); fills host-side lengths
    void assemble_prompts(const std::vector<ResolvedRequest>& reqs, std::vector<int>& n_prompt, std::vector<int>& n_trailing);
    void project_rows(const std::vector<int32_t>& ids, int rows);  // ids -> proj_out_[rows][H]
    void enqueue_layers(const StackW& s, Stream& w, int B, uint16_t* kpool, uint16_t* vpool, size_t layer_stride,
                        const int32_t* block_table, int max_pages, const int32_t* kv_len, const uint8_t* active,
                        int ss_count_in, int fixed_len, int chunk, const int32_t* chunk_n_prompt, int chunk_r_base);
    void enqueue_talker_step(int B, bool with_head);
    void enqueue_cp_pass(int B, bool from_talker, int head, int cp_pos, bool projected = false);  // head: lm_head index or -1; cp_pos: tokens already cached
    void enqueue_frame(int B, const DebugOpts* dbg);
    hipGraphExec_t frame_graph(int B);
    GemmArgs gemm_args(const LinearW& L, const uint16_t* x, int M) const;
};


// The object behind q3tts_model: the model plus `n_lanes` engines. q3tts_generate splits its rows
// contiguously over the lanes; each lane is driven by its own host thread on its own HIP stream, so
// the latency-bound frame steps of independent rows overlap on the GPU (the frame step is a chain
// of ~800 short dependent kernels; one chain cannot fill 256 CUs, several chains can).
class EngineGroup {
  public:
    EngineGroup(std::unique_ptr<Model> model, const q3tts_load_opts& opts);
    Model& model() { return *model_; }
    Engine& lane0() { return *lanes_[0]; }
    int n_lanes() const { return int(lanes_.size()); }
    const q3tts_load_opts& opts() const { return opts_; }
    void generate(const q3tts_request* reqs, int n, const q3tts_sampling& sp, q3tts_event_cb cb, void* user,
                  q3tts_result* results, const DebugOpts* dbg);
    // Two-deep pipeline (Engine::begin / end). With more than one lane a job runs to completion inside begin.
    int begin(const q3tts_request* reqs, int n, const q3tts_sampling& sp, q3tts_event_cb cb, void* user, bool more_follows);
    void end(int job, q3tts_result* results);
    std::string last_error;
    q3tts_timing timing{};
    std::vector<std::string> speakers;

  private:
    struct Parked {  // lanes > 1: finished results waiting for end()
        bool busy = false;
        std::vector<q3tts_result> results;
        q3tts_timing timing{};
    } parked_[Engine::kJobSlots];
    std::unique_ptr<Model> model_;
    q3tts_load_opts opts_;
    std::vector<std::unique_ptr<Engine>> lanes_;
    std::mutex cb_mutex_;
};

}  // namespace q3
