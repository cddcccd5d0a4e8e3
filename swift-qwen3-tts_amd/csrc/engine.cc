// engine.cc -- generation driver. See engine.h.
#include "engine.h"

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <thread>

#include "codec.h"
#include "frontend.h"

namespace q3 {

namespace {
std::string lower(const std::string& s) {
    std::string o = s;
    for (auto& c : o) c = char(std::tolower((unsigned char)c));
    return o;
}
const char* const kCodecRangeMsg =
    "Audio decoding failed: the codec decoder produced a non-finite waveform -- also on the fp32 matrix-core convolutions (the "
    "reference's range), which a row is re-decoded on when it leaves the fp16 range of the default two-plane kernels";
double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct Bump {
    uint8_t* base;
    size_t off = 0;
    template <class T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};
}  // namespace

Engine::Engine(Model* model, const q3tts_load_opts& opts) : m_(model), opts_(opts) {
    Q3_HIP(hipSetDevice(m_->device));
    {
        // The AR loop is a chain of short dependent launches (latency), the codec decode a few hundred large ones
        // (matrix cores): the decode of a finished batch runs on its own stream so that the NEXT batch's AR loop can
        // overlap it (begin / end). Stream priorities alone do not help: a decode kernel's workgroups fill every CU and the
        // AR chain's workgroups then queue behind them (prefill 23 -> 165 ms, nothing gained). Confined to half of the CUs
        // (mask bits interleave over the XCDs) the decode takes 1.7x as long but leaves the chain room: 879 -> 826 ms
        // per pipelined step at 1.7B / batch 32 (96 CUs: 841, 160: 841, 192: 856). A decode that nothing overlaps
        // (generate(), codec_decode) uses the unmasked stream. q3tts_load_opts.codec_overlap_cus overrides the CU count.
        int least = 0, greatest = 0;
        Q3_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        Q3_HIP(hipStreamCreateWithPriority(&st_, hipStreamNonBlocking, greatest));
        Q3_HIP(hipStreamCreateWithPriority(&st_codec_, hipStreamNonBlocking, least));
        hipDeviceProp_t prop{};
        Q3_HIP(hipGetDeviceProperties(&prop, m_->device));
        int cus = prop.multiProcessorCount / 2;
        if (opts.codec_overlap_cus > 0) cus = opts.codec_overlap_cus;   // q3tts_load_opts: the caller's own tuning
        else if (opts.codec_overlap_cus < 0) cus = 0;
        cus = std::min(cus, prop.multiProcessorCount) / 8 * 8;
        if (cus > 0 && cus < prop.multiProcessorCount) {
            std::vector<uint32_t> mask(size_t(ceil_div(prop.multiProcessorCount, 32)), 0u);
            // mask bit i = CU (i / 8) of XCD (i % 8), CUs of an XCD numbered round-robin over its shader engines: the first
            // half of the bits is half of every shader engine of every XCD. (Every other CU instead -- whole shader
            // engines -- left the AR chain's workgroups queueing on the busy engines: 1008 against 807 ms per step.)
            for (int i = 0; i < cus; ++i) mask[size_t(i / 32)] |= 1u << (i % 32);
            Q3_HIP(hipExtStreamCreateWithCUMask(&st_codec_part_, uint32_t(mask.size()), mask.data()));
        }
    }
    for (auto& J : jobs_) {
        for (auto& e : J.ev_codec) Q3_HIP(hipEventCreate(&e));
        Q3_HIP(hipEventCreate(&J.ev_begin));
        Q3_HIP(hipEventCreate(&J.ev_first_audio));
        Q3_HIP(hipHostMalloc(reinterpret_cast<void**>(&J.nf_host), size_t(std::max(opts.max_batch, 1)) * 4, hipHostMallocDefault));
        std::memset(J.nf_host, 0, size_t(std::max(opts.max_batch, 1)) * 4);
    }
    for (auto& e : ev_) Q3_HIP(hipEventCreate(&e));
    for (auto& e : burst_ev_) Q3_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : ev_fe_) Q3_HIP(hipEventCreate(&e));
    Q3_HIP(hipEventCreateWithFlags(&fe_uploaded_, hipEventDisableTiming));
    Bm_ = opts.max_batch;
    // activation rows per launch (batch rows x positions): 256 rows run the code predictor's two-position step 0 as one pass up
    // to batch 128; a prefill chunk is up to 16 positions of every row, so the buffers hold 16 rows per batch row (the GEMMs
    // take them as row blocks of <= 64 on grid.y and stream the weights once per chunk: 32 x 48 prompt positions at 1.7B
    // 14.1 -> 11.0 ms against 8-position chunks; the chunk boundaries do not change a bit, tests/test_scheduling.py)
    Mp_ = std::getenv("Q3TTS_ROWS_64") ? 64 : std::max(256, 16 * int(align_up(size_t(opts.max_batch), 16)));
    Pcap_ = opts.max_prompt;
    Tcap_ = opts.max_prompt;
    Fcap_ = opts.max_frames;
    max_pages_ = ceil_div(Pcap_ + Fcap_ + 1, kPageTokens);
    n_pages_ = Bm_ * max_pages_;
    for (auto& kv : m_->cfg.talker.spk_id) speakers.push_back(kv.first);
    std::sort(speakers.begin(), speakers.end());
    alloc_workspace();
    if (std::getenv("Q3TTS_FRAME_STAMPS")) {
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&stamps_), 64 * 8));
        Q3_HIP(hipMemset(stamps_, 0, 64 * 8));
    }
    if (m_->has_codec) codec_ = std::make_unique<CodecRunner>(*m_, st_codec_, opts.codec_fp32 != 0);
    if (m_->has_codec_encoder || m_->has_speaker_encoder) fe_ = std::make_unique<VoiceFrontEnd>(*m_, st_);
}

Engine::~Engine() {
    if (stager_.joinable()) {
        {
            std::lock_guard<std::mutex> lk(stage_mu_);
            stage_stop_ = true;
        }
        stage_cv_.notify_all();
        stager_.join();
    }
    for (auto& J : jobs_) {  // rows staged for a job that was never ended
        for (float* p : J.st_pcm) std::free(p);
        for (int32_t* p : J.st_codes) std::free(p);
    }
    for (auto& g : graphs_) (void)hipGraphExecDestroy(g.second);
    codec_.reset();
    fe_.reset();
    for (auto& L : fe_lanes_) {
        L.fe.reset();
        if (L.spk) (void)hipFree(L.spk);
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.st) (void)hipStreamDestroy(L.st);
    }
    for (void* p : {(void*)ref_audio_dev_, (void*)ref_codes_dev_, (void*)extra_, (void*)spk_f32_})
        if (p) (void)hipFree(p);
    for (auto& J : jobs_) {
        for (auto& e : J.chunk_done)
            if (e) (void)hipEventDestroy(e);
        if (J.dec_codes) (void)hipFree(J.dec_codes);
        if (J.pcm_host) (void)hipHostFree(J.pcm_host);
        for (auto& e : J.ev_codec)
            if (e) (void)hipEventDestroy(e);
        if (J.nf_host) (void)hipHostFree(J.nf_host);
        if (J.nf_chunk_host) (void)hipHostFree(J.nf_chunk_host);
        if (J.ev_begin) (void)hipEventDestroy(J.ev_begin);
        if (J.ev_first_audio) (void)hipEventDestroy(J.ev_first_audio);
    }
    if (ws_) (void)hipFree(ws_);
    for (void* p : {(void*)forced_dev_, (void*)sampled_dev_, (void*)tl_dump_, (void*)cl_dump_})
        if (p) (void)hipFree(p);
    for (auto& e : ev_)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : burst_ev_)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ev_fe_)
        if (e) (void)hipEventDestroy(e);
    if (fe_uploaded_) (void)hipEventDestroy(fe_uploaded_);
    if (stamps_) {  // phase times of the frame step, accumulated over every frame step this engine ran
        unsigned long long h[64];
        if (hipMemcpy(h, stamps_, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[0] > 0) {
            static const char* nm[] = {"", "talker layers", "codec_head + norm + sampler (+ proj)", "predictor pair pass",
                                       "pass-0 head + sampler", "predictor passes 1..14 (layers)", "their heads + samplers", "frame_end"};
            std::fprintf(stderr, "[q3tts frame stamps] %llu frame steps\n", h[0]);
            for (int k = 1; k < 8; ++k) std::fprintf(stderr, "  %-40s %8.1f us per step\n", nm[k], double(h[k]) / 100.0 / double(h[0]));
        }
        (void)hipFree(stamps_);
    }
    if (st_codec_part_) (void)hipStreamDestroy(st_codec_part_);
    if (st_codec_) (void)hipStreamDestroy(st_codec_);
    if (st_) (void)hipStreamDestroy(st_);
}

// CodePredictor.swift:327-330 projects the embedding of every sampled code (H wide) down to the predictor's width before
// each of passes 1..14. The projection of a table row does not depend on anything else, so it is taken once per row at
// load -- by the very GEMM kernel the frame step would have launched, Mp_ codes at a time, so the rows (and their per-tile sums
// of squares for the next norm prologue) are bit-identical to projecting at run time -- and the frame step loses 14 launches.
void Engine::build_cp_proj_tables() {
    const TalkerConfig& t = m_->cfg.talker;
    const int H = t.hidden_size, CH = m_->cp.hidden, MBL = Mp_ / 16, Vc = t.cp.vocab_size, nss = CH / 16;
    const int ntab = t.num_code_groups - 2;  // embeddings 0..13 feed passes 1..14; the last code feeds nothing
    if (ntab <= 0) return;
    std::lock_guard<std::mutex> lock(m_->lazy_mutex);  // lanes share the model; the first one to generate builds
    if (m_->cp_pe.empty()) {
        for (int i = 0; i < ntab; ++i) {
            uint16_t* pe = nullptr;
            float* pss = nullptr;
            Q3_HIP(hipMalloc(reinterpret_cast<void**>(&pe), size_t(Vc) * CH * 2));
            m_->lazy_allocs.push_back(pe);
            Q3_HIP(hipMalloc(reinterpret_cast<void**>(&pss), size_t(Vc) * nss * 4));
            m_->lazy_allocs.push_back(pss);
            for (int c0 = 0; c0 < Vc; c0 += Mp_) {
                const int rows = std::min(Mp_, Vc - c0);
                launch_tile_rows(m_->cp_emb[size_t(i)] + size_t(c0) * H, H, cp_x_, MBL, rows, H, st_);
                GemmArgs p = gemm_args(m_->cp_proj, cp_x_, rows);
                p.epi = 3; p.y = cp_.h; p.yMB = MBL; p.resid = 0; p.ss_out = cp_.ss_a;
                launch_gemm_skinny(p, st_);
                launch_untile_rows(cp_.h, MBL, pe + size_t(c0) * CH, CH, rows, CH, st_);
                launch_ss_to_table(cp_.ss_a, Mp_, pss + size_t(c0) * nss, nss, rows, st_);
            }
            m_->cp_pe.push_back(pe);
            m_->cp_pss.push_back(pss);
        }
        Q3_HIP(hipStreamSynchronize(st_));
    }
    cp_tables_ = true;
}

void Engine::alloc_workspace() {
    const TalkerConfig& t = m_->cfg.talker;
    const int H = t.hidden_size, CH = t.cp.hidden_size, TH = t.text_hidden_size;
    const int qd = t.num_attention_heads * kHeadDim, kd = t.num_key_value_heads * kHeadDim;
    const int cqd = t.cp.num_attention_heads * kHeadDim, ckd = t.cp.num_key_value_heads * kHeadDim;
    const int L = t.num_hidden_layers, CL = t.cp.num_hidden_layers;
    kv_layer_stride_ = size_t(n_pages_) * t.num_key_value_heads * kPageTokens * kHeadDim;
    cp_kv_layer_stride_ = size_t(Bm_) * t.cp.num_key_value_heads * kPageTokens * kHeadDim;
    proj_cap_ = Bm_ * (2 * Pcap_ + 8);  // text + instruct or reference-text ids + the three tts tokens per row
    for (int pass = 0; pass < 2; ++pass) {
        Bump b{pass ? ws_ : nullptr};
        Bump& x = b;
        auto stream = [&](Stream& s, int hid, int q, int k, int inter_p, int vocab) {
            s.ld_qkv = q + 2 * k;
            s.ld_act = inter_p;
            s.ld_logits = vocab;
            s.h = x.take<uint16_t>(size_t(Mp_) * hid);
            s.xn = x.take<uint16_t>(size_t(Mp_) * hid);
            s.ss_a = x.take<float>(size_t(hid / 16) * Mp_);
            s.ss_b = x.take<float>(size_t(hid / 16) * Mp_);
            s.qkv = x.take<uint16_t>(size_t(Mp_) * s.ld_qkv);
            s.ao = x.take<uint16_t>(size_t(Mp_) * q);
            s.act = x.take<uint16_t>(size_t(Mp_) * inter_p);
            s.logits = b.take<uint16_t>(size_t(Mp_) * vocab);
        };
        stream(tk_, H, qd, kd, m_->talker.max_inter_p, t.vocab_size);
        stream(cp_, CH, cqd, ckd, m_->cp.max_inter_p, t.cp.vocab_size);
        cp_x_ = x.take<uint16_t>(size_t(Mp_) * H);
        cp_x2_ = x.take<uint16_t>(size_t(Mp_) * H);
        cp_ss2_ = x.take<float>(size_t(Mp_));
        kpool_ = b.take<uint16_t>(kv_layer_stride_ * L);
        vpool_ = b.take<uint16_t>(kv_layer_stride_ * L);
        cp_kpool_ = b.take<uint16_t>(cp_kv_layer_stride_ * CL);
        cp_vpool_ = b.take<uint16_t>(cp_kv_layer_stride_ * CL);
        block_table_ = b.take<int32_t>(size_t(Bm_) * max_pages_);
        cp_block_table_ = b.take<int32_t>(size_t(Bm_));
        kv_len_ = b.take<int32_t>(size_t(Bm_));
        cp_len_ = b.take<int32_t>(size_t(Bm_));
        n_frames_ = b.take<int32_t>(size_t(Bm_));
        max_frames_ = b.take<int32_t>(size_t(Bm_));
        trailing_idx_ = b.take<int32_t>(size_t(Bm_));
        n_trailing_ = b.take<int32_t>(size_t(Bm_));
        n_prompt_ = b.take<int32_t>(size_t(Bm_));
        cur_codes_ = b.take<int32_t>(size_t(Bm_) * 16);
        codes_ = b.take<int32_t>(size_t(Bm_) * Fcap_ * 16);
        active_ = b.take<uint8_t>(size_t(Bm_));
        finished_ = b.take<uint8_t>(size_t(Bm_));
        seen_ = b.take<uint8_t>(size_t(Bm_) * t.vocab_size);
        prompt_ = b.take<uint16_t>(size_t(Bm_) * Pcap_ * H);
        trailing_ = b.take<uint16_t>(size_t(Bm_) * Tcap_ * H);
        tts_pad_ = b.take<uint16_t>(size_t(H));
        sp_dev_ = b.take<SamplingParams>(1);
        ids_dev_ = b.take<int32_t>(size_t(proj_cap_));
        proj_in_ = b.take<uint16_t>(size_t(64) * TH);
        proj_mid_ = b.take<uint16_t>(size_t(64) * TH);
        proj_out_ = b.take<uint16_t>(size_t(proj_cap_ + 64) * H);
        compose_a_ = b.take<int32_t>(size_t(3) * Bm_ * (Pcap_ + Tcap_));
        compose_b_ = nullptr;
        if (!pass) {
            ws_bytes_ = align_up(b.off, 256);
            Q3_HIP(hipMalloc(reinterpret_cast<void**>(&ws_), ws_bytes_));
            Q3_HIP(hipMemset(ws_, 0, ws_bytes_));
        }
    }
    std::vector<int32_t> cbt((size_t)(Bm_));  // code-predictor cache: one private page per row
    for (int i = 0; i < Bm_; ++i) cbt[size_t(i)] = i;
    Q3_HIP(hipMemcpy(cp_block_table_, cbt.data(), cbt.size() * 4, hipMemcpyHostToDevice));
}

GemmArgs Engine::gemm_args(const LinearW& L, const uint16_t* x, int M) const {
    GemmArgs a{};
    a.W = L.w;
    a.Wsb = L.sb;
    a.x = x;
    a.xMB = Mp_ / 16;
    a.M = M;
    a.Mpad = int(align_up(size_t(M), 16));
    a.N = L.Np;
    a.K = L.Kp;
    a.bias = L.bias;
    a.ss_ld = Mp_;
    return a;
}

// One pre-norm decoder layer = 5 launches (Talker.swift:451-469):
//   qkv GEMM [RMSNorm prologue] -> attention -> o_proj GEMM [residual + sum(h^2) epilogue]
//   -> gate/up GEMM [RMSNorm prologue, SwiGLU epilogue] -> down GEMM [residual + sum(h^2) epilogue]
// w.h is the fragment-major residual stream; ss_a holds the per-tile sums of squares of the rows
// entering a layer (`ss_count_in` partials for the first layer), ss_b those after o_proj.
void Engine::enqueue_layers(const StackW& s, Stream& w, int B, uint16_t* kpool, uint16_t* vpool, size_t layer_stride,
                            const int32_t* block_table, int max_pages, const int32_t* kv_len, const uint8_t* active,
                            int ss_count_in, int fixed_len, int chunk, const int32_t* chunk_n_prompt, int chunk_r_base) {
    const int H = s.hidden, MBL = Mp_ / 16, tiles = H / 16;
    const int M = B * (chunk > 1 ? chunk : 1);  // GEMM rows: chunk element p of batch row b is row p * B + b
    Q3_CHECK(M <= Mp_, 7, "internal error: chunk does not fit the activation buffers");
    // The RMSNorm prologue re-normalises all of x in every workgroup: VALU work on the critical path that grows with K and
    // with the row blocks per workgroup (0.8-1.3 us). Against a separate row-norm launch (4.5 us) it wins at both widths
    // since the norm-prologue kernels request x before the weight tiles, take at most two row blocks per workgroup and
    // the 6144-wide gate/up runs as one round of workgroups (gemm_decode.hip); wider stacks keep the row-norm kernel.
    // Above 64 rows per launch (prefill chunks) the prologue is repeated by (column tiles x row-block groups) workgroups --
    // 2048 of them for a 6144-wide gate/up at 256 rows, ~3 us of SIMD time each -- so the rows are normalised once by
    // the row kernel instead, from the SAME per-tile partials in the same order (NormRowsArgs::ss_in): bit-identical.
    const bool row_norm = M > 64;
    const bool prologue_qkv = H <= 2048 && !row_norm, prologue_mlp = H <= 2048 && !row_norm;
    // The talker's layer weights and KV cache are read once per frame step out of gigabytes; the code predictor's
    // 0.22 GB are read fifteen times per step. Non-temporal loads on the former leave the Infinity Cache (256 MB) to the
    // latter: measured 3.53 -> 3.42 ms per 1.7B frame step with both (either one alone: < 1 %; on the predictor's
    // weights as well: 3.62 ms). Q3TTS_NT=0 turns the hint off (diagnostics).
    const bool nt_off = debug_env().nt_off;
    const bool is_talker = &s == &m_->talker;
    const int ntw = is_talker && !nt_off ? 1 : 0;
    const int ntkv = ntw;
    auto norm_into_xn = [&](const uint16_t* nw, const float* ss, int ss_count) {
        NormRowsArgs n{};
        n.h = w.h; n.hMB = MBL; n.w = nw; n.eps = s.eps; n.out = w.xn; n.outMB = MBL; n.M = M; n.H = H;
        if (row_norm && H <= 2048) { n.ss_in = ss; n.ss_count = ss_count; n.ss_ld = Mp_; }
        other([&] { launch_norm_rows(n, st_); });
    };
    for (size_t l = 0; l < s.layers.size(); ++l) {
        const LayerW& L = s.layers[l];
        if (!prologue_qkv) norm_into_xn(L.ln1, w.ss_a, (l == 0) ? ss_count_in : tiles);
        GemmArgs q = gemm_args(L.qkv, prologue_qkv ? w.h : w.xn, M);
        q.epi = 0; q.y = w.qkv; q.ldy = w.ld_qkv; q.nt_weights = ntw;
        if (prologue_qkv) {
            q.norm_w = L.ln1; q.ss_in = w.ss_a; q.ss_count = (l == 0) ? ss_count_in : tiles; q.norm_dim = H; q.norm_eps = s.eps;
        }
        AttnArgs at{};
        at.qkv = w.qkv; at.ld = w.ld_qkv; at.qn_w = L.qn; at.kn_w = L.kn; at.eps = s.eps;
        at.rope_cos = s.rope_cos; at.rope_sin = s.rope_sin;
        at.kpool = kpool + l * layer_stride; at.vpool = vpool + l * layer_stride;
        at.block_table = block_table; at.max_pages = max_pages; at.kv_len = kv_len; at.active = active;
        at.out = w.ao; at.outMB = MBL; at.n_heads = s.n_heads; at.n_kv = s.n_kv; at.B = B;
        at.fixed_len = fixed_len; at.identity_pages = fixed_len >= 0 ? 1 : 0;
        at.chunk = chunk; at.chunk_n_prompt = chunk_n_prompt; at.chunk_r_base = chunk_r_base;
        at.scale = powf(float(kHeadDim), -0.5f);  // Talker.swift:179
        at.nt_kv = ntkv;
        gemm(q);
        attn(at);
        GemmArgs o = gemm_args(L.o, w.ao, M);
        o.epi = 3; o.y = w.h; o.yMB = MBL; o.resid = 1; o.ss_out = w.ss_b; o.nt_weights = ntw;
        gemm(o);
        if (!prologue_mlp) norm_into_xn(L.ln2, w.ss_b, tiles);
        GemmArgs g = gemm_args(L.gateup, prologue_mlp ? w.h : w.xn, M);
        g.epi = 2; g.y = w.act; g.yMB = MBL; g.nt_weights = ntw;
        if (prologue_mlp) {
            g.norm_w = L.ln2; g.ss_in = w.ss_b; g.ss_count = tiles; g.norm_dim = H; g.norm_eps = s.eps;
        }
        gemm(g);
        GemmArgs d = gemm_args(L.down, w.act, M);
        d.epi = 3; d.y = w.h; d.yMB = MBL; d.resid = 1; d.ss_out = w.ss_a; d.nt_weights = ntw;
        gemm(d);
    }
}

void Engine::enqueue_talker_step(int B, bool with_head) {
    (void)with_head;
    enqueue_layers(m_->talker, tk_, B, kpool_, vpool_, kv_layer_stride_, block_table_, max_pages_, kv_len_, active_, 1, -1, 1, nullptr, 0);
}

// One code-predictor pass (CodePredictor.swift:320-339 without the head). `from_talker`: the input is
// the talker's final-normed hidden state (step 0, first position); otherwise it is the embedding the
// previous sampler gathered (fragment-major in cp_x_ when a projection follows, else straight in cp_.h).
void Engine::enqueue_cp_pass(int B, bool from_talker, int head, int cp_pos, bool projected) {
    const TalkerConfig& t = m_->cfg.talker;
    const int H = t.hidden_size, CH = m_->cp.hidden, MBL = Mp_ / 16;
    int ss_count = 1;
    // The talker's final norm (Talker.swift:573) is always a row kernel, so that every way of scheduling step 0 (two
    // passes, one two-position pass) rounds it identically.
    auto talker_norm_into = [&](uint16_t* dst, float* ss_out) {
        NormRowsArgs n{};
        n.h = tk_.h; n.hMB = MBL; n.w = m_->talker.final_norm; n.eps = m_->talker.eps;
        n.out = dst; n.outMB = MBL; n.ss_out = ss_out; n.M = B; n.H = H;
        other([&] { launch_norm_rows(n, st_); });
    };
    if (m_->has_cp_proj && projected) {  // the sampler gathered an already projected row and its sums (build_cp_proj_tables)
        ss_count = CH / 16;
    } else if (m_->has_cp_proj) {  // small_to_mtp_projection (biased), CodePredictor.swift:327-330
        if (from_talker) talker_norm_into(cp_x2_, nullptr);  // cp_x_ already holds embed(code0) for the second position
        GemmArgs p = gemm_args(m_->cp_proj, from_talker ? cp_x2_ : cp_x_, B);
        p.epi = 3; p.y = cp_.h; p.yMB = MBL; p.resid = 0; p.ss_out = cp_.ss_a;
        gemm(p);
        ss_count = CH / 16;
    } else if (from_talker) {
        talker_norm_into(cp_.h, cp_.ss_a);
    }  // else: the sampler wrote cp_.h and cp_.ss_a[0] itself
    // every row's predictor cache holds cp_pos tokens at this point (cp_len_ advances in lock-step, finished rows included)
    enqueue_layers(m_->cp, cp_, B, cp_kpool_, cp_vpool_, cp_kv_layer_stride_, cp_block_table_, 1, cp_len_, nullptr, ss_count, cp_pos, 1, nullptr, 0);
}

void Engine::enqueue_frame_body(int B, const DebugOpts* dbg) {
    const TalkerConfig& t = m_->cfg.talker;
    const int H = t.hidden_size, V = t.vocab_size, Vc = t.cp.vocab_size, CH = t.cp.hidden_size;
    const int groups = t.num_code_groups, MBL = Mp_ / 16;
    stamp(0);
    enqueue_talker_step(B, true);
    stamp(1);
    // where the samplers put the next code-predictor input
    uint16_t* next_x = m_->has_cp_proj ? cp_x_ : cp_.h;
    float* next_ss = m_->has_cp_proj ? nullptr : cp_.ss_a;
    // Predictor step 0 takes two positions, [talker hidden, embed(code0)] (Qwen3.swift:884-887). When 2 * B rows fit the
    // activation buffers they go through the stack together (rows 0..B-1 and B..2B-1, chunk attention), which saves a
    // whole pass of launches per frame; otherwise they are two passes.
    const bool pair = 2 * B <= Mp_;
    {   // final norm (prologue) + codec_head (Talker.swift:573, 644)
        GemmArgs hd = gemm_args(m_->codec_head, tk_.h, B);
        hd.epi = 0; hd.y = tk_.logits; hd.ldy = tk_.ld_logits;
        hd.norm_w = m_->talker.final_norm; hd.ss_in = tk_.ss_a; hd.ss_count = H / 16; hd.norm_dim = H; hd.norm_eps = m_->talker.eps;
        bool rode = false;
        if (pair) {
            // the predictor's first position -- the talker's final-normed hidden state (Talker.swift:573), materialised next to
            // the embedding -- reads what this GEMM reads: B extra workgroups of its launch instead of a launch of its own
            NormRowsArgs n{};
            n.h = tk_.h; n.hMB = MBL; n.w = m_->talker.final_norm; n.eps = m_->talker.eps;
            n.out = next_x; n.outMB = MBL; n.ss_out = next_ss; n.M = B; n.H = H;
            rode = gemm_with_norm_rows(hd, n);
            if (!rode) other([&] { launch_norm_rows(n, st_); });
        }
        if (!rode) gemm(hd);
    }
    SamplerArgs sa{};
    sa.logits = tk_.logits; sa.ldl = tk_.ld_logits; sa.V = V; sa.sp = sp_dev_; sa.is_talker = 1;
    sa.suppress_lo = V - 1024; sa.suppress_hi = V; sa.eos_id = t.codec_eos_token_id;  // Qwen3.swift:829-835
    sa.seen = seen_; sa.cb = 0; sa.n_frames = n_frames_; sa.max_frames = max_frames_;
    sa.finished = finished_; sa.active = active_; sa.kv_len = kv_len_; sa.advance = 1; sa.advance_gate = active_;
    sa.cur_codes = cur_codes_; sa.codes = codes_; sa.Fmax = Fcap_;
    sa.forced = dbg ? forced_dev_ : nullptr; sa.forced_frames = dbg ? dbg->frames : 0;
    sa.sampled = dbg ? sampled_dev_ : nullptr;
    sa.emb = m_->codec_emb; sa.emb_ld = H; sa.next_MB = MBL; sa.H = H; sa.B = B;
    if (pair) {  // embed(code0) goes straight to the second row block of the pass input
        sa.next_x = next_x; sa.next_row0 = B; sa.next_ss = next_ss ? next_ss + B : nullptr;
    } else {
        // Without a projection a pass takes its input in cp_.h, which the first position (the talker hidden state) still
        // needs: stage the embedding in cp_x2_
        sa.next_x = m_->has_cp_proj ? cp_x_ : cp_x2_; sa.next_ss = m_->has_cp_proj ? nullptr : cp_ss2_;
    }
    sa.logits_dump = (dbg && dbg->talker_logits) ? tl_dump_ : nullptr; sa.dump_ld = V; sa.dump_off = 0;
    if (pair) {
        other([&] { launch_sampler(sa, st_); });
        int ss_count = 1;
        if (m_->has_cp_proj) {  // small_to_mtp_projection over both positions (CodePredictor.swift:327-330)
            GemmArgs p = gemm_args(m_->cp_proj, cp_x_, 2 * B);
            p.epi = 3; p.y = cp_.h; p.yMB = MBL; p.resid = 0; p.ss_out = cp_.ss_a;
            gemm(p);
            ss_count = CH / 16;
        }
        stamp(2);
        enqueue_layers(m_->cp, cp_, B, cp_kpool_, cp_vpool_, cp_kv_layer_stride_, cp_block_table_, 1, cp_len_, nullptr, ss_count, 0, 2,
                       nullptr, 0);
        stamp(3);  // (no launch for cp_len_: the predictor's attention takes its cache length from the pass index, fixed_len)
    } else {
        other([&] { launch_sampler(sa, st_); });
        // code predictor, step 0 = [hidden, embed(code0)] run as two positions
        enqueue_cp_pass(B, true, -1, 0);
        if (!m_->has_cp_proj) {  // second position: move the staged embedding (and its sum of squares) into place
            other([&] { launch_copy_rows(cp_x2_, 0, cp_.h, 0, 1, Mp_ * H, st_); });
            other([&] { launch_copy_rows(reinterpret_cast<const uint16_t*>(cp_ss2_), 0, reinterpret_cast<uint16_t*>(cp_.ss_a), 0, 1, Mp_ * 2, st_); });
        }
    }
    FrameEndArgs fe{};
    fe.cur_codes = cur_codes_; fe.codec_emb = m_->codec_emb; fe.cp_emb = m_->cp_emb_dev;
    fe.trailing = trailing_; fe.n_trailing = n_trailing_; fe.trailing_idx = trailing_idx_; fe.Tmax = Tcap_;
    fe.tts_pad = tts_pad_; fe.h = tk_.h; fe.hMB = MBL; fe.ss_out = tk_.ss_a; fe.H = H; fe.B = B; fe.groups = groups;
    fe.n_frames = n_frames_; fe.max_frames = max_frames_; fe.finished = finished_; fe.active = active_; fe.cp_len = cp_len_;
    bool fe_done = false;
    for (int i = 0; i < groups - 1; ++i) {
        const bool second_of_pair = pair && i == 0;  // its stack forward already ran above; rows B..2B-1 hold it
        const int Mh = second_of_pair ? 2 * B : B;
        if (!second_of_pair) {
            enqueue_cp_pass(B, false, i, i + 1, cp_tables_ && i >= 1);
            stamp(5);
        }
        {
            GemmArgs lh = gemm_args(m_->lm_head[size_t(i)], cp_.h, Mh);
            lh.epi = 0; lh.y = cp_.logits; lh.ldy = cp_.ld_logits;
            lh.norm_w = m_->cp.final_norm; lh.ss_in = cp_.ss_a; lh.ss_count = CH / 16; lh.norm_dim = CH; lh.norm_eps = m_->cp.eps;
            gemm(lh);
        }
        SamplerArgs sc{};
        sc.logits = cp_.logits + (second_of_pair ? size_t(B) * cp_.ld_logits : 0); sc.ldl = cp_.ld_logits; sc.V = Vc; sc.sp = sp_dev_;
        sc.is_talker = 0;
        sc.eos_id = -1; sc.cb = i + 1; sc.n_frames = n_frames_; sc.max_frames = max_frames_;
        sc.finished = finished_; sc.active = active_; sc.kv_len = cp_len_; sc.advance = 1; sc.advance_gate = nullptr;
        sc.cur_codes = cur_codes_; sc.codes = codes_; sc.Fmax = Fcap_;
        sc.forced = dbg ? forced_dev_ : nullptr; sc.forced_frames = dbg ? dbg->frames : 0;
        sc.sampled = dbg ? sampled_dev_ : nullptr;
        if (i + 1 < groups - 1) {  // embedding of this code feeds the next pass (Qwen3.swift:889-892)
            sc.emb = m_->cp_emb[size_t(i)]; sc.emb_ld = H; sc.next_x = next_x; sc.next_MB = MBL; sc.next_ss = next_ss;
        }
        sc.H = H; sc.B = B;
        if (cp_tables_ && i + 1 < groups - 1) {  // projected row + its per-tile sums straight into the next pass's input
            sc.emb = m_->cp_pe[size_t(i)]; sc.emb_ld = CH; sc.H = CH; sc.next_x = cp_.h;
            sc.emb_ss = m_->cp_pss[size_t(i)]; sc.nss = CH / 16; sc.next_ss = cp_.ss_a; sc.next_ss_ld = Mp_;
        }
        sc.logits_dump = (dbg && dbg->cp_logits) ? cl_dump_ : nullptr; sc.dump_ld = (groups - 1) * Vc; sc.dump_off = i * Vc;
        if (i == groups - 2 && Vc <= 2048) {  // the frame's last draw carries its row's end-of-frame job (Qwen3.swift:919-935; row_jobs.h)
            other([&] { launch_sampler_with_frame_end(sc, fe, st_); });
            fe_done = true;
        } else {
            other([&] { launch_sampler(sc, st_); });
        }
        stamp(second_of_pair ? 4 : 6);
    }
    if (!fe_done) other([&] { launch_frame_end(fe, st_); });
    stamp(7);
}


// ---- the plan: which later launch's weight lines each launch of the frame step touches (kernels/prefetch.h) ------------
namespace {
PfArgs empty_touch(const void* valid) {
    PfArgs p{};
    p.base = static_cast<const uint8_t*>(valid);
    p.span = 128; p.lines = 1; p.inv_lines = 1.0f;
    return p;
}
}  // namespace

void Engine::gemm(GemmArgs a) {
    if (plan_mode_ == 1) {
        PlanItem it{};
        const SkinnyGeom g = skinny_geometry(a);
        if (!g.tall) {
            it.w = reinterpret_cast<const uint8_t*>(a.W);
            it.span = uint32_t(g.span_bytes(a));
            it.nspan = uint32_t(g.gx);
            if (g.touches) it.cap_lines = uint32_t(kPfTouches) * uint32_t((g.gx * g.split + 7) / 8) * uint32_t(g.threads());
        }
        plan_.push_back(it);
        return;
    }
    a.pf = plan_mode_ == 2 ? pf_[plan_pos_++] : empty_touch(a.W);
    ++launches_;
    launch_gemm_skinny(a, st_);
}

bool Engine::gemm_with_norm_rows(GemmArgs a, const NormRowsArgs& n) {
    if (plan_mode_ == 1) {
        // recorded as the plain launch; whether the riders fit is decided by the launcher's own test in both passes
        const bool rides = gemm_norm_rows_rides(a, n);
        if (rides) gemm(a);
        return rides;
    }
    a.pf = plan_mode_ == 2 ? pf_[plan_pos_] : empty_touch(a.W);
    const bool rode = launch_gemm_skinny_with_norm_rows(a, n, st_);
    if (rode && plan_mode_ == 2) ++plan_pos_;
    if (rode) ++launches_;
    return rode;
}

void Engine::attn(AttnArgs a) {
    if (plan_mode_ == 1) {
        PlanItem it{};
        if (Q3_PF_MODE != 0 && a.chunk <= 1) it.cap_lines = uint32_t(kPfTouches) * uint32_t((a.n_kv * a.B + 7) / 8) * uint32_t(attn_decode_threads(a));
        plan_.push_back(it);
        return;
    }
    a.pf = plan_mode_ == 2 ? pf_[plan_pos_++] : empty_touch(a.qkv);
    ++launches_;
    launch_attn_decode(a, st_);
}

// Greedy, in launch order, over the frame step taken as a cycle (the graph is replayed back to back: the last launches of a
// step touch the first weights of the next): launch i looks up to `ahead` launches forward for the nearest weight stream
// that is not yet fully touched and takes as many of its lines as (a) its own threads can touch and (b) the per-XCD budget of
// lines sitting touched-but-unread in an L2 allows. A stream's lines are counted per XCD: the spans x' = c (mod 8) belong to
// XCD c, floor(nspan / 8) of them for every c.
void Engine::plan_touches() {
    const DebugEnv& env = debug_env();
    const size_t n = plan_.size();
    pf_.assign(n, PfArgs{});
    const int64_t budget = int64_t(env.pf_budget_kb) * 1024 / 128;
    const int ahead = std::max(1, env.pf_ahead);
    std::vector<int64_t> touched(3 * n + size_t(ahead) + 1, 0);
    int64_t resident = 0;
    for (size_t v = 0; v < 2 * n; ++v) {  // two rounds: the first only sets up what the end of a step leaves for the next
        const PlanItem& me = plan_[v % n];
        resident -= touched[v];
        PfArgs p = empty_touch(ws_);
        if (env.prefetch && me.cap_lines > 0) {
            int skip = env.pf_skip;
            for (size_t j = v + 1; j <= v + size_t(ahead); ++j) {
                const PlanItem& t = plan_[j % n];
                if (!t.w || t.nspan < 8 || t.span % 128 != 0) continue;
                if (skip-- > 0) continue;
                const int64_t lines = t.span / 128, total = int64_t(t.nspan / 8) * lines;
                if (total >= (int64_t(1) << 22)) continue;  // pf_issue's exact-division range
                const int64_t rem = total - touched[j];
                if (rem <= 0) continue;
                const int64_t amt = std::min<int64_t>({rem, int64_t(me.cap_lines), budget - resident});
                if (amt <= 0) break;
                p.base = t.w;
                p.span = t.span;
                p.lines = uint32_t(lines);
                p.inv_lines = 1.0f / float(lines);
                p.u0 = uint32_t(touched[j]);
                p.u1 = uint32_t(touched[j] + amt);
                touched[j] += amt;
                resident += amt;
                break;
            }
        }
        if (v >= n) pf_[v - n] = p;
    }
}

void Engine::enqueue_frame(int B, const DebugOpts* dbg) {
    struct Count {  // launches of this frame step, for q3tts_timing (the same count whether captured or launched eagerly)
        Engine* e;
        int B;
        ~Count() { e->frame_launches_[B] = e->launches_; }
    } count{this, B};
    launches_ = 0;
    if (Q3_PF_MODE == 0) {  // the shipped build: no kernel carries the touch code, nothing to plan
        enqueue_frame_body(B, dbg);
        return;
    }
    auto it = pf_plans_.find(B);
    if (it == pf_plans_.end()) {
        plan_mode_ = 1;
        plan_.clear();
        try {
            enqueue_frame_body(B, dbg);
        } catch (...) {
            plan_mode_ = 0;
            throw;
        }
        plan_mode_ = 0;
        plan_touches();
        it = pf_plans_.emplace(B, pf_).first;
    }
    pf_ = it->second;
    plan_mode_ = 2;
    plan_pos_ = 0;
    try {
        enqueue_frame_body(B, dbg);
    } catch (...) {
        plan_mode_ = 0;
        throw;
    }
    plan_mode_ = 0;
    Q3_CHECK(plan_pos_ == pf_.size(), 7, "internal error: the frame step's launches do not match their plan");
}

hipGraphExec_t Engine::frame_graph(int B) {
    auto it = graphs_.find(B);
    if (it != graphs_.end()) return it->second;
    hipGraph_t g = nullptr;
    Q3_HIP(hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal));
    try {
        enqueue_frame(B, nullptr);
    } catch (...) {
        (void)hipStreamEndCapture(st_, &g);
        if (g) (void)hipGraphDestroy(g);
        throw;
    }
    Q3_HIP(hipStreamEndCapture(st_, &g));
    hipGraphExec_t ge = nullptr;
    Q3_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    Q3_HIP(hipGraphDestroy(g));
    graphs_[B] = ge;
    return ge;
}

// ------------------------------------------------------------------------------------------------
// request resolution: routing and validation of generate() (Qwen3.swift:1291-1373, 803-811, 303-319)
// ------------------------------------------------------------------------------------------------
ResolvedRequest Engine::resolve(const q3tts_request& r, const q3tts_sampling& sp) const {
    const ModelConfig& cfg = m_->cfg;
    const TalkerConfig& t = cfg.talker;
    ResolvedRequest o;
    Q3_CHECK(r.text_ids && r.n_text_ids >= 4, 3, "Invalid input: text_ids must hold the chat-template tokens");
    Q3_CHECK(r.route >= 0 && r.route <= 2, 3, "Invalid input: unknown q3tts_request.route");
    o.text_ids.assign(r.text_ids, r.text_ids + r.n_text_ids);
    const bool have_instruct = r.instruct_ids && r.n_instruct_ids > 0;
    const std::string type = cfg.tts_model_type;
    auto speaker_list = [&]() {
        std::string s;
        for (size_t i = 0; i < speakers.size(); ++i) s += (i ? ", " : "") + speakers[i];
        return s;
    };
    bool use_speaker = false, use_instruct = false;
    if (r.ref_audio != nullptr) {  // generateVoiceClone (Qwen3.swift:1009-1046): no routing by model type, no speaker/instruct
        Q3_CHECK(m_->has_codec, 1, "Model not initialized: Speech tokenizer not loaded");  // :1029-1031
        Q3_CHECK(m_->has_codec_encoder, 1,
                 "Model not initialized: Voice cloning (ICL mode) requires the speech tokenizer encoder. Make sure to load a model "
                 "with encoder weights.");  // :1033-1038
        Q3_CHECK(r.n_ref_samples > 0, 3, "Invalid input: reference audio is empty");
        {   // a NaN sample would spread through the encoders into every logit of the row
            bool finite = true;
            for (int64_t i = 0; i < int64_t(r.n_ref_samples); ++i) finite = finite && (std::fabs(r.ref_audio[i]) <= 3.0e38f);
            Q3_CHECK(finite, 3, "Invalid input: reference audio holds non-finite samples");
        }
        Q3_CHECK(r.ref_text_ids && r.n_ref_text_ids >= 5, 3, "Invalid input: ref_text_ids must hold the chat-template tokens");
        Q3_CHECK(r.n_text_ids >= 8, 3, "Invalid input: text_ids must hold the chat-template tokens");
        Q3_CHECK(m_->codec_enc.bins <= t.vocab_size && m_->codec_enc.bins <= t.cp.vocab_size, 3,
                 "Invalid input: encoder codebook larger than the codec embedding tables");
        o.clone = true;
        o.ref_audio = r.ref_audio;
        o.n_ref_samples = r.n_ref_samples;
        o.ref_text_ids.assign(r.ref_text_ids, r.ref_text_ids + r.n_ref_text_ids);
        const std::string lang = lower(r.language ? r.language : "auto");
        if (lang != "auto") {  // :515-519 (no dialect override on this path)
            auto it = t.codec_language_id.find(lang);
            if (it != t.codec_language_id.end()) o.language_id = it->second;
        }
        o.target_token_count = r.target_token_count;
        const int mt = r.max_tokens > 0 ? r.max_tokens : 2048;
        o.max_frames = sp.force_frames > 0 ? sp.force_frames : int(std::min<int64_t>(mt, std::max<int64_t>(75, int64_t(r.target_token_count) * 6)));  // :1051-1052
        for (int id : o.text_ids) Q3_CHECK(id >= 0 && id < t.text_vocab_size, 3, "Invalid input: text token id out of range");
        for (int id : o.ref_text_ids) Q3_CHECK(id >= 0 && id < t.text_vocab_size, 3, "Invalid input: reference text token id out of range");
        return o;
    }
    if (r.route == 1) {         // generateVoiceDesign(text:language:instruct:...) on any checkpoint (Qwen3.swift:587-620)
        use_instruct = true;
    } else if (r.route == 2) {  // generateCustomVoice(text:speaker:language:instruct:...) on any checkpoint (:783-811)
        Q3_CHECK(r.speaker != nullptr, 3, "Invalid input: generateCustomVoice requires 'speaker'");
        use_speaker = true;
        use_instruct = true;
    } else if (type == "custom_voice" || type == "base") {
        const char* nm = type == "custom_voice" ? "CustomVoice" : "Base";
        Q3_CHECK(r.speaker != nullptr, 3,
                 std::string("Invalid input: ") + nm + " model requires 'speaker' (e.g., 'Vivian', 'Ryan'). Available speakers: " + speaker_list());
        use_speaker = true;
        use_instruct = (type == "custom_voice");  // Base ignores instruct (Qwen3.swift:1352)
    } else {  // voice_design and unknown types (Qwen3.swift:1360-1371)
        if (type == "voice_design")
            Q3_CHECK(have_instruct, 3,
                     "Invalid input: VoiceDesign model requires 'instruct' to describe the voice (e.g., 'A cheerful young female voice with high pitch')");
        use_instruct = true;
    }
    if (use_speaker) {  // Qwen3.swift:803-811
        Q3_CHECK(t.has_spk_id, 3, "Invalid input: This model does not support CustomVoice. No speakers defined.");
        auto it = t.spk_id.find(lower(r.speaker));
        Q3_CHECK(it != t.spk_id.end(), 3,
                 std::string("Invalid input: Speaker '") + r.speaker + "' not found. Available speakers: " + speaker_list());
        o.speaker_token = it->second;
    }
    if (use_instruct && have_instruct) o.instruct_ids.assign(r.instruct_ids, r.instruct_ids + r.n_instruct_ids);
    const std::string lang = lower(r.language ? r.language : "auto");
    if (lang != "auto") {  // Qwen3.swift:304-308 (unknown names silently mean "no language")
        auto it = t.codec_language_id.find(lang);
        if (it != t.codec_language_id.end()) o.language_id = it->second;
    }
    if ((lang == "chinese" || lang == "auto") && use_speaker) {  // dialect override, Qwen3.swift:311-319
        auto d = t.spk_dialect.find(lower(r.speaker));
        if (d != t.spk_dialect.end()) {
            auto it = t.codec_language_id.find(d->second);
            if (it != t.codec_language_id.end()) o.language_id = it->second;
        }
    }
    o.target_token_count = r.target_token_count;
    const int mt = r.max_tokens > 0 ? r.max_tokens : 2048;
    o.max_frames = sp.force_frames > 0 ? sp.force_frames : int(std::min<int64_t>(mt, std::max<int64_t>(75, int64_t(r.target_token_count) * 6)));  // :822-823
    for (int id : o.text_ids) Q3_CHECK(id >= 0 && id < t.text_vocab_size, 3, "Invalid input: text token id out of range");
    for (int id : o.instruct_ids) Q3_CHECK(id >= 0 && id < t.text_vocab_size, 3, "Invalid input: instruct token id out of range");
    return o;
}

// ids -> text_projection(embedText(ids)) (Talker.swift:627-633, 475-487), 64 rows per GEMM pass
void Engine::project_rows(const std::vector<int32_t>& ids, int rows) {
    const TalkerConfig& t = m_->cfg.talker;
    const int TH = t.text_hidden_size, H = t.hidden_size;
    Q3_CHECK(rows <= proj_cap_, 3, "Invalid input: prompt too long for the configured max_prompt");
    Q3_HIP(hipMemcpyAsync(ids_dev_, ids.data(), size_t(rows) * 4, hipMemcpyHostToDevice, st_));
    for (int r0 = 0; r0 < rows; r0 += 64) {
        const int n = std::min(64, rows - r0);
        launch_gather_rows(m_->text_emb, TH, ids_dev_ + r0, m_->token_map, n, TH, proj_in_, TH, 4, st_);
        GemmArgs f1 = gemm_args(m_->fc1, proj_in_, n);
        f1.xMB = 4; f1.epi = 0; f1.y = proj_mid_; f1.y_tiled = 1; f1.yMB = 4; f1.act_silu = 1;
        launch_gemm_skinny(f1, st_);
        GemmArgs f2 = gemm_args(m_->fc2, proj_mid_, n);
        f2.xMB = 4; f2.epi = 0; f2.y = proj_out_ + size_t(r0) * H; f2.ldy = H;
        launch_gemm_skinny(f2, st_);
    }
}

void Engine::assemble_prompts(const std::vector<ResolvedRequest>& reqs, std::vector<int>& n_prompt, std::vector<int>& n_trailing) {
    const ModelConfig& cfg = m_->cfg;
    const TalkerConfig& t = cfg.talker;
    const int H = t.hidden_size;
    const int n = int(reqs.size());
    std::vector<int32_t> ids;
    std::vector<int> text_off((size_t)(n)), instr_off((size_t)(n)), tts_off((size_t)(n));
    for (int b = 0; b < n; ++b) {
        const auto& r = reqs[size_t(b)];
        text_off[size_t(b)] = int(ids.size());
        ids.insert(ids.end(), r.text_ids.begin(), r.text_ids.end());
        instr_off[size_t(b)] = int(ids.size());  // doubles as the reference-text offset of voice-clone rows
        ids.insert(ids.end(), r.instruct_ids.begin(), r.instruct_ids.end());
        ids.insert(ids.end(), r.ref_text_ids.begin(), r.ref_text_ids.end());
        tts_off[size_t(b)] = int(ids.size());
        ids.push_back(cfg.tts_bos_token_id);  // Qwen3.swift:282-292
        ids.push_back(cfg.tts_eos_token_id);
        ids.push_back(cfg.tts_pad_token_id);
    }
    project_rows(ids, int(ids.size()));
    std::vector<int32_t> pa, pb, pd, ta, tb, td;  // (proj row, codec id or -1, destination row)
    n_prompt.assign(size_t(n), 0);
    n_trailing.assign(size_t(n), 0);
    for (int b = 0; b < n; ++b) {
        const auto& r = reqs[size_t(b)];
        const int bos = tts_off[size_t(b)], eos = bos + 1, pad = bos + 2;
        std::vector<int> cp_ids;  // codec prefix, Qwen3.swift:322-359 / 527-561
        if (r.language_id < 0) cp_ids = {t.codec_nothink_id, t.codec_think_bos_id, t.codec_think_eos_id};
        else cp_ids = {t.codec_think_id, t.codec_think_bos_id, r.language_id, t.codec_think_eos_id};
        if (r.speaker_token >= 0) cp_ids.push_back(r.speaker_token);
        if (r.clone && m_->has_speaker_encoder) cp_ids.push_back(-2 - r.extra_base);  // x-vector row (:553-558)
        cp_ids.push_back(t.codec_pad_id);
        cp_ids.push_back(t.codec_bos_id);
        for (int id : cp_ids) Q3_CHECK(id < t.vocab_size && id != -1, 3, "Invalid input: codec prefix id out of range");
        const int nc = int(cp_ids.size());
        int p = 0;
        auto push = [&](int a, int c) {
            Q3_CHECK(p < Pcap_, 3, "Invalid input: prompt longer than max_prompt");
            pa.push_back(a);
            pb.push_back(c);
            pd.push_back(b * Pcap_ + p);
            ++p;
        };
        if (r.clone) {  // prepareICLGenerationInputs (Qwen3.swift:418-582)
            const int tl = int(r.text_ids.size()), rl = int(r.ref_text_ids.size()), ro = instr_off[size_t(b)];
            for (int i = 0; i < 3; ++i) push(text_off[size_t(b)] + i, -1);                     // role, :564-566
            for (int i = 0; i < nc - 1; ++i) push(i < nc - 2 ? pad : bos, cp_ids[size_t(i)]);  // :569-573
            for (int i = 3; i < rl - 2; ++i) push(ro + i, t.codec_pad_id);                     // reference text, :451, :505-506
            for (int i = 3; i < tl - 5; ++i) push(text_off[size_t(b)] + i, t.codec_pad_id);    // target text, :457
            push(eos, t.codec_pad_id);                                                         // :476
            push(pad, t.codec_bos_id);                                                         // :494-496, :509-510
            for (int f = 0; f < r.ref_T; ++f) push(pad, -2 - (r.extra_base + 1 + f));
            n_prompt[size_t(b)] = p;
            ta.push_back(pad);  // trailing text is just tts_pad (:579)
            tb.push_back(-1);
            td.push_back(b * Tcap_);
            n_trailing[size_t(b)] = 1;
            continue;
        }
        for (int i = 0; i < int(r.instruct_ids.size()); ++i) push(instr_off[size_t(b)] + i, -1);  // :383-384
        for (int i = 0; i < 3; ++i) push(text_off[size_t(b)] + i, -1);                            // role, :371
        for (int i = 0; i < nc - 1; ++i) push(i < nc - 2 ? pad : bos, cp_ids[size_t(i)]);         // :375-379
        push(text_off[size_t(b)] + 3, cp_ids[size_t(nc - 1)]);                                    // :390
        Q3_CHECK(p <= Pcap_, 3, "Invalid input: prompt longer than max_prompt");
        n_prompt[size_t(b)] = p;
        const int tl = int(r.text_ids.size());
        int q = 0;
        auto pusht = [&](int a) {
            ta.push_back(a);
            tb.push_back(-1);
            td.push_back(b * Tcap_ + q);
            ++q;
        };
        if (tl - 5 > 4)  // Qwen3.swift:394-406
            for (int i = 4; i < tl - 5; ++i) pusht(text_off[size_t(b)] + i);
        pusht(eos);
        Q3_CHECK(q <= Tcap_, 3, "Invalid input: text longer than max_prompt");
        n_trailing[size_t(b)] = q;
    }
    auto run = [&](std::vector<int32_t>& a, std::vector<int32_t>& bb, std::vector<int32_t>& d, uint16_t* dst) {
        const size_t k = a.size();
        int32_t* da = compose_a_;
        int32_t* db = compose_a_ + k;
        int32_t* dd = compose_a_ + 2 * k;
        Q3_HIP(hipMemcpyAsync(da, a.data(), k * 4, hipMemcpyHostToDevice, st_));
        Q3_HIP(hipMemcpyAsync(db, bb.data(), k * 4, hipMemcpyHostToDevice, st_));
        Q3_HIP(hipMemcpyAsync(dd, d.data(), k * 4, hipMemcpyHostToDevice, st_));
        launch_compose_rows(proj_out_, H, m_->codec_emb, H, extra_, H, da, db, dd, dst, H, int(k), H, st_);
        Q3_HIP(hipStreamSynchronize(st_));  // host vectors are reused by the next call
    };
    run(pa, pb, pd, prompt_);
    run(ta, tb, td, trailing_);
    launch_copy_rows(proj_out_ + size_t(tts_off[0] + 2) * H, H, tts_pad_, H, 1, H, st_);
}

const float* Engine::upload_audio(const float* audio, int64_t n) {
    if (size_t(n) > ref_audio_cap_) {
        Q3_HIP(hipStreamSynchronize(st_));
        if (ref_audio_dev_) Q3_HIP(hipFree(ref_audio_dev_));
        ref_audio_dev_ = nullptr;
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&ref_audio_dev_), size_t(n) * 4));
        ref_audio_cap_ = size_t(n);
    }
    Q3_HIP(hipMemcpyAsync(ref_audio_dev_, audio, size_t(n) * 4, hipMemcpyHostToDevice, st_));
    return ref_audio_dev_;
}

// Steps 1, 5 and 8 of prepareICLGenerationInputs (Qwen3.swift:436-444, 479-491, 521-525) for every voice-clone row:
// reference codes, the per-frame sums of their 16 embeddings and the speaker x-vector, all left on the device.
void Engine::prepare_clone_rows(std::vector<ResolvedRequest>& reqs) {
    const int H = m_->cfg.talker.hidden_size;
    Q3_CHECK(fe_ != nullptr, 1, "Model not initialized: Speech tokenizer encoder not available");
    size_t total_codes = 0, total_rows = 0;
    for (auto& r : reqs) {
        if (!r.clone) continue;
        r.ref_T = fe_->encoded_frames(r.n_ref_samples);
        r.ref_off = int(total_codes);
        r.extra_base = int(total_rows);
        total_codes += size_t(16) * r.ref_T;
        total_rows += size_t(1) + r.ref_T;
    }
    if (total_codes > ref_codes_cap_) {
        if (ref_codes_dev_) Q3_HIP(hipFree(ref_codes_dev_));
        ref_codes_dev_ = nullptr;
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&ref_codes_dev_), total_codes * 4));
        ref_codes_cap_ = total_codes;
    }
    if (total_rows > extra_cap_) {
        if (extra_) Q3_HIP(hipFree(extra_));
        extra_ = nullptr;
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&extra_), total_rows * H * 2));
        extra_cap_ = total_rows;
    }
    int n_clone = 0;
    for (auto& r : reqs) n_clone += r.clone ? 1 : 0;
    const int K = std::min(4, n_clone);
    while (int(fe_lanes_.size()) < K) {
        FeLane L;
        Q3_HIP(hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking));
        Q3_HIP(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        L.fe = std::make_unique<VoiceFrontEnd>(*m_, L.st);
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&L.spk), size_t(H) * 4));
        fe_lanes_.push_back(std::move(L));
    }
    // every clip zero-padded to the longest one: the (causal) codec encoder then runs ONCE over all of them
    int64_t S_max = 0;
    for (auto& r : reqs)
        if (r.clone) S_max = std::max<int64_t>(S_max, r.n_ref_samples);
    if (size_t(S_max) * n_clone > ref_audio_cap_) {
        Q3_HIP(hipStreamSynchronize(st_));
        if (ref_audio_dev_) Q3_HIP(hipFree(ref_audio_dev_));
        ref_audio_dev_ = nullptr;
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&ref_audio_dev_), size_t(S_max) * n_clone * 4));
        ref_audio_cap_ = size_t(S_max) * n_clone;
    }
    Q3_HIP(hipMemsetAsync(ref_audio_dev_, 0, size_t(S_max) * n_clone * 4, st_));
    std::vector<int64_t> valid, offs;  // samples per clip, offset of its codes
    int i = 0;
    for (auto& r : reqs) {
        if (!r.clone) continue;
        Q3_HIP(hipMemcpyAsync(ref_audio_dev_ + size_t(i) * S_max, r.ref_audio, size_t(r.n_ref_samples) * 4, hipMemcpyHostToDevice, st_));
        valid.push_back(r.n_ref_samples);
        offs.push_back(r.ref_off);
        ++i;
    }
    Q3_HIP(hipEventRecord(fe_uploaded_, st_));
    // speaker x-vectors: not causal (reflect padding, statistics over the whole clip), so one clip at a time, a few side by side
    if (m_->has_speaker_encoder) {
        i = 0;
        for (auto& r : reqs) {
            if (!r.clone) continue;
            FeLane& L = fe_lanes_[size_t(i % K)];
            if (i < K) Q3_HIP(hipStreamWaitEvent(L.st, fe_uploaded_, 0));
            // The x-vector is fp32; it enters the prompt in the talker's storage dtype like every other row
            // (the reference's MLX concat would instead promote the prompt to fp32).
            L.fe->speaker_embedding(ref_audio_dev_ + size_t(i) * S_max, r.n_ref_samples, L.spk);
            launch_f32_to_bf16(L.spk, extra_ + size_t(r.extra_base) * H, H, L.st);
            ++i;
        }
    }
    const size_t per_clip = size_t(S_max) * 64 * 4 * 3 + (size_t(1) << 20);  // rough scratch per clip (bytes)
    const int rows_per_pass = int(std::max<size_t>(1, std::min<size_t>(VoiceFrontEnd::kMaxClips, (size_t(16) << 30) / per_clip)));
    for (int lo = 0; lo < n_clone; lo += rows_per_pass) {
        const int nb = std::min(rows_per_pass, n_clone - lo);
        fe_->encode_batch(ref_audio_dev_ + size_t(lo) * S_max, nb, S_max, valid.data() + lo, offs.data() + lo, ref_codes_dev_);
    }
    for (auto& r : reqs) {
        if (!r.clone) continue;
        launch_ref_embed_rows(ref_codes_dev_ + r.ref_off, r.ref_T, 16, m_->codec_emb, m_->cp_emb_dev, H,
                              extra_ + size_t(r.extra_base + 1) * H, H, st_);
    }
    if (m_->has_speaker_encoder)
        for (int k = 0; k < K; ++k) {  // the prompt assembly on st_ follows every lane
            Q3_HIP(hipEventRecord(fe_lanes_[size_t(k)].done, fe_lanes_[size_t(k)].st));
            Q3_HIP(hipStreamWaitEvent(st_, fe_lanes_[size_t(k)].done, 0));
        }
}

int Engine::encoded_frames(int64_t n_samples) const {
    return (fe_ && m_->has_codec_encoder) ? fe_->encoded_frames(n_samples) : 0;
}

int Engine::codec_encode(const float* audio, int64_t n_samples, int32_t* codes, int cap_frames) {
    Q3_CHECK(fe_ && m_->has_codec_encoder, 1, "Model not initialized: Speech tokenizer encoder not available");
    Q3_CHECK(audio && n_samples > 0, 3, "Invalid input: empty audio");
    const int T = fe_->encoded_frames(n_samples);
    Q3_CHECK(T <= cap_frames, 3, "Invalid input: output buffer too small for the encoded frames");
    if (size_t(16) * T > ref_codes_cap_) {
        if (ref_codes_dev_) Q3_HIP(hipFree(ref_codes_dev_));
        ref_codes_dev_ = nullptr;
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&ref_codes_dev_), size_t(16) * T * 4));
        ref_codes_cap_ = size_t(16) * T;
    }
    const float* a = upload_audio(audio, n_samples);
    Q3_HIP(hipEventRecord(ev_fe_[0], st_));
    fe_->encode(a, n_samples, ref_codes_dev_);
    Q3_HIP(hipEventRecord(ev_fe_[1], st_));
    Q3_HIP(hipStreamSynchronize(st_));
    Q3_HIP(hipMemcpy(codes, ref_codes_dev_, size_t(16) * T * 4, hipMemcpyDeviceToHost));
    float ms = 0;
    Q3_HIP(hipEventElapsedTime(&ms, ev_fe_[0], ev_fe_[1]));
    timing = q3tts_timing{};
    timing.frontend_ms = ms;
    return T;
}

void Engine::speaker_embedding(const float* audio, int64_t n_samples, float* out, int cap) {
    Q3_CHECK(fe_ && m_->has_speaker_encoder, 1, "Model not initialized: Speaker encoder not available for this model");
    Q3_CHECK(audio && n_samples > 0, 3, "Invalid input: empty audio");
    const int D = m_->speaker.enc_dim;
    Q3_CHECK(cap >= D, 3, "Invalid input: output buffer too small for the speaker embedding");
    if (!spk_f32_) Q3_HIP(hipMalloc(reinterpret_cast<void**>(&spk_f32_), size_t(m_->cfg.talker.hidden_size) * 4));
    const float* a = upload_audio(audio, n_samples);
    Q3_HIP(hipEventRecord(ev_fe_[0], st_));
    fe_->speaker_embedding(a, n_samples, spk_f32_);
    Q3_HIP(hipEventRecord(ev_fe_[1], st_));
    Q3_HIP(hipStreamSynchronize(st_));
    Q3_HIP(hipMemcpy(out, spk_f32_, size_t(D) * 4, hipMemcpyDeviceToHost));
    float ms = 0;
    Q3_HIP(hipEventElapsedTime(&ms, ev_fe_[0], ev_fe_[1]));
    timing = q3tts_timing{};
    timing.frontend_ms = ms;
}

void Engine::debug_frontend_stage(const float* audio, int64_t n_samples, const char* stage, float* out, int64_t cap, int* T, int* C) {
    Q3_CHECK(fe_ != nullptr, 1, "Model not initialized: Speech tokenizer encoder not available");
    Q3_CHECK(audio && n_samples > 0 && stage, 3, "Invalid input: empty audio or stage name");
    const float* a = upload_audio(audio, n_samples);
    StageCapture cap_s;
    cap_s.name = stage;
    static const char* kSpeaker[] = {"mel", "h0", "h1", "h2", "h3", "mfa", "pooled"};
    bool is_spk = false;
    for (const char* s : kSpeaker) is_spk = is_spk || cap_s.name == s;
    if (is_spk) {
        Q3_CHECK(m_->has_speaker_encoder, 1, "Model not initialized: Speaker encoder not available for this model");
        if (!spk_f32_) Q3_HIP(hipMalloc(reinterpret_cast<void**>(&spk_f32_), size_t(m_->cfg.talker.hidden_size) * 4));
        fe_->speaker_embedding(a, n_samples, spk_f32_, &cap_s);
    } else {
        Q3_CHECK(m_->has_codec_encoder, 1, "Model not initialized: Speech tokenizer encoder not available");
        const int Tq = fe_->encoded_frames(n_samples);
        if (size_t(16) * Tq > ref_codes_cap_) {
            if (ref_codes_dev_) Q3_HIP(hipFree(ref_codes_dev_));
            ref_codes_dev_ = nullptr;
            Q3_HIP(hipMalloc(reinterpret_cast<void**>(&ref_codes_dev_), size_t(16) * Tq * 4));
            ref_codes_cap_ = size_t(16) * Tq;
        }
        fe_->encode(a, n_samples, ref_codes_dev_, &cap_s);
    }
    Q3_HIP(hipStreamSynchronize(st_));
    Q3_CHECK(!cap_s.data.empty(), 3, std::string("Invalid input: unknown front-end stage '") + stage + "'");
    Q3_CHECK(int64_t(cap_s.data.size()) <= cap, 3, "Invalid input: output buffer too small");
    std::memcpy(out, cap_s.data.data(), cap_s.data.size() * 4);
    *T = cap_s.T;
    *C = cap_s.C;
}

void Engine::debug_prepare_inputs(const q3tts_request& req, uint16_t* input_embeds, int cap_prompt, int* n_prompt,
                                  uint16_t* trailing, int cap_trailing, int* n_trailing, uint16_t* tts_pad) {
    q3tts_sampling sp{};
    q3tts_default_sampling(&sp);
    std::vector<ResolvedRequest> rr{resolve(req, sp)};
    if (rr[0].clone) prepare_clone_rows(rr);
    std::vector<int> np, nt;
    assemble_prompts(rr, np, nt);
    const int H = m_->cfg.talker.hidden_size;
    Q3_CHECK(np[0] <= cap_prompt && nt[0] <= cap_trailing, 3, "Invalid input: output buffers too small");
    Q3_HIP(hipStreamSynchronize(st_));
    Q3_HIP(hipMemcpy(input_embeds, prompt_, size_t(np[0]) * H * 2, hipMemcpyDeviceToHost));
    Q3_HIP(hipMemcpy(trailing, trailing_, size_t(nt[0]) * H * 2, hipMemcpyDeviceToHost));
    Q3_HIP(hipMemcpy(tts_pad, tts_pad_, size_t(H) * 2, hipMemcpyDeviceToHost));
    *n_prompt = np[0];
    *n_trailing = nt[0];
}

// ------------------------------------------------------------------------------------------------
// generate
// ------------------------------------------------------------------------------------------------
void Engine::generate(const q3tts_request* reqs, int n, const q3tts_sampling& sp, q3tts_event_cb cb, void* user,
                      q3tts_result* results, const DebugOpts* dbg) {
    end(begin(reqs, n, sp, cb, user, dbg, false), results);
}

// The codec runner's scratch is shared by both codec streams: before it moves to the other one, the one it ran on drains.
hipStream_t Engine::codec_stream(bool overlapped) {
    hipStream_t want = overlapped && st_codec_part_ ? st_codec_part_ : st_codec_;
    if (want != codec_->stream()) {
        Q3_HIP(hipStreamSynchronize(codec_->stream()));
        codec_->set_stream(want);
    }
    return want;
}

int Engine::begin(const q3tts_request* reqs, int n, const q3tts_sampling& sp, q3tts_event_cb cb, void* user, const DebugOpts* dbg,
                  bool overlapped) {
    const TalkerConfig& t = m_->cfg.talker;
    const int H = t.hidden_size, V = t.vocab_size, Vc = t.cp.vocab_size, groups = t.num_code_groups;
    Q3_HIP(hipSetDevice(m_->device));  // lanes run on their own host threads
    Q3_CHECK(n >= 1 && n <= Bm_, 3, "Invalid input: batch size must be between 1 and max_batch");
    Q3_CHECK(groups == 16, 3, "Invalid input: num_code_groups must be 16");
    Q3_CHECK(sp.audio_chunk_frames >= 0, 3, "Invalid input: audio_chunk_frames must not be negative");
    if (sp.audio_chunk_frames > 0 && sp.audio_window_frames > 0 && m_->has_codec)  // before any GPU work (the stream would refuse it later)
        Q3_CHECK(sp.audio_chunk_frames >= codec_->hist_frames(), 3,
                 "Invalid input: audio_chunk_frames of a streamed decode must be at least " + std::to_string(codec_->hist_frames()));
    int slot = -1;  // any free slot: jobs may be ended in any order
    for (int i = 0; i < kJobSlots; ++i)
        if (!jobs_[i].busy && slot < 0) slot = i;
    Q3_CHECK(slot >= 0, 3, "Invalid input: two jobs are already outstanding (q3tts_generate_end must be called first)");
    const double t_start = now_s();
    Q3_HIP(hipEventRecord(jobs_[slot].ev_begin, st_));
    std::vector<ResolvedRequest> rr;
    for (int i = 0; i < n; ++i) rr.push_back(resolve(reqs[i], sp));
    if (dbg)
        for (auto& r : rr) r.max_frames = dbg->frames;
    for (auto& r : rr) Q3_CHECK(r.max_frames <= Fcap_, 3, "Invalid input: max_tokens exceeds the configured max_frames");
    if (m_->has_codec == false)
        throw Error(1, "Model not initialized: Speech tokenizer not loaded");  // Qwen3.swift:799-801

    bool any_clone = false;
    for (auto& r : rr) any_clone = any_clone || r.clone;
    Q3_HIP(hipEventRecord(ev_fe_[0], st_));
    if (any_clone) prepare_clone_rows(rr);  // codec encoder + speaker encoder, once per request (Qwen3.swift:443, :524)
    Q3_HIP(hipEventRecord(ev_fe_[1], st_));

    std::vector<int> np, nt;
    assemble_prompts(rr, np, nt);
    int Pmax = 0;
    for (int p : np) Pmax = std::max(Pmax, p);

    // ---- per-row state ----
    std::vector<int32_t> bt((size_t)(n) * max_pages_, 0), zeros((size_t)(n), 0), maxf((size_t)(n)), ntr((size_t)(n)), npr((size_t)(n));
    int next_page = 0;
    for (int b = 0; b < n; ++b) {
        const int need = ceil_div(np[size_t(b)] + rr[size_t(b)].max_frames + 1, kPageTokens);
        Q3_CHECK(need <= max_pages_ && next_page + need <= n_pages_, 3, "Invalid input: KV pool exhausted");
        for (int i = 0; i < need; ++i) bt[size_t(b) * max_pages_ + i] = next_page++;
        maxf[size_t(b)] = rr[size_t(b)].max_frames;
        ntr[size_t(b)] = nt[size_t(b)];
        npr[size_t(b)] = np[size_t(b)];
    }
    Q3_CHECK(Pmax + *std::max_element(maxf.begin(), maxf.end()) + 1 <= m_->talker.max_pos, 3,
             "Invalid input: sequence longer than the RoPE table");
    Q3_HIP(hipMemcpyAsync(block_table_, bt.data(), bt.size() * 4, hipMemcpyHostToDevice, st_));
    Q3_HIP(hipMemcpyAsync(max_frames_, maxf.data(), size_t(n) * 4, hipMemcpyHostToDevice, st_));
    Q3_HIP(hipMemcpyAsync(n_trailing_, ntr.data(), size_t(n) * 4, hipMemcpyHostToDevice, st_));
    Q3_HIP(hipMemcpyAsync(n_prompt_, npr.data(), size_t(n) * 4, hipMemcpyHostToDevice, st_));
    for (int32_t* p : {kv_len_, cp_len_, n_frames_, trailing_idx_}) Q3_HIP(hipMemsetAsync(p, 0, size_t(n) * 4, st_));
    Q3_HIP(hipMemsetAsync(active_, 0, size_t(n), st_));
    Q3_HIP(hipMemsetAsync(finished_, 0, size_t(n), st_));
    Q3_HIP(hipMemsetAsync(seen_, 0, size_t(n) * V, st_));
    Q3_HIP(hipMemsetAsync(codes_, 0, size_t(n) * Fcap_ * 16 * 4, st_));
    SamplingParams sph{sp.temperature, sp.top_k, sp.top_p, sp.repetition_penalty, sp.seed, row_offset + sp.row_base, sp.force_frames > 0 ? 1 : 0};
    Q3_HIP(hipMemcpyAsync(sp_dev_, &sph, sizeof(sph), hipMemcpyHostToDevice, st_));
    int frames_cap = 0;
    for (int f : maxf) frames_cap = std::max(frames_cap, f);
    if (dbg) {
        for (void* p : {(void*)forced_dev_, (void*)sampled_dev_, (void*)tl_dump_, (void*)cl_dump_})
            if (p) (void)hipFree(p);
        forced_dev_ = sampled_dev_ = nullptr;
        tl_dump_ = cl_dump_ = nullptr;
        const size_t nf = size_t(n) * dbg->frames;
        if (dbg->forced_codes) {
            // teacher-forced codes are fed back as rows of the codec / predictor embedding tables
            for (size_t i = 0; i < nf * 16; ++i) {
                const int32_t c = dbg->forced_codes[i];
                Q3_CHECK(c >= 0 && c < ((i & 15) == 0 ? V : Vc), 3, "Invalid input: forced code outside its vocabulary");
            }
            Q3_HIP(hipMalloc(reinterpret_cast<void**>(&forced_dev_), nf * 16 * 4));
            Q3_HIP(hipMemcpy(forced_dev_, dbg->forced_codes, nf * 16 * 4, hipMemcpyHostToDevice));
        }
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&sampled_dev_), nf * 16 * 4));
        Q3_HIP(hipMemset(sampled_dev_, 0xff, nf * 16 * 4));
        if (dbg->talker_logits) Q3_HIP(hipMalloc(reinterpret_cast<void**>(&tl_dump_), nf * V * 2));
        if (dbg->cp_logits) Q3_HIP(hipMalloc(reinterpret_cast<void**>(&cl_dump_), nf * (groups - 1) * Vc * 2));
    }

    // ---- prefill: positions 0 .. Pmax-2 of the right-aligned prompts, then load the last one ----
    Q3_HIP(hipEventRecord(ev_[0], st_));
    PrefillLoadArgs pl{};
    pl.prompt = prompt_; pl.n_prompt = n_prompt_; pl.Pmax = Pcap_; pl.H = H; pl.B = n; pl.h = tk_.h; pl.hMB = Mp_ / 16;
    pl.ss_out = tk_.ss_a; pl.active = active_;
    // prompt_ rows are laid out with stride Pcap_; right alignment is relative to the longest prompt.
    // Positions 0 .. Pmax-2 go through the decode kernels C at a time (C * n <= Mp_ activation rows: the GEMMs stream
    // the weights once per chunk instead of once per position); rows whose prompt is shorter start inside a chunk.
    {
        const int P1 = Pmax - 1;  // positions before the one the first frame step consumes
        int C = std::max(1, std::min(16, Mp_ / n));
        const int S = (P1 + C - 1) / C;
        for (int s = 0; s < S; ++s) {
            // element p of row b in chunk s is prompt position r = s*C - S*C + (n_prompt[b] - 1) + p
            const int r_base = s * C - S * C - 1;
            if (C == 1) {
                pl.step = s + (Pcap_ - Pmax);
                launch_prefill_load(pl, st_);
                enqueue_talker_step(n, false);
                launch_advance_len(kv_len_, active_, n, st_);
            } else {
                PrefillLoadArgs pc = pl;
                pc.step = r_base;
                launch_prefill_chunk_load(pc, C, st_);
                enqueue_layers(m_->talker, tk_, n, kpool_, vpool_, kv_layer_stride_, block_table_, max_pages_, kv_len_, nullptr, 1, -1,
                               C, n_prompt_, r_base);
                launch_advance_len_chunk(kv_len_, n_prompt_, r_base, C, n, st_);
            }
        }
        pl.step = (Pmax - 1) + (Pcap_ - Pmax);  // the last prompt position: consumed by the first frame step
        launch_prefill_load(pl, st_);
    }
    Q3_HIP(hipEventRecord(ev_[1], st_));

    // ---- streamed decode (row f1): chunks of the waveform leave while the loop below is still producing tokens ----
    Job& J = jobs_[slot];
    const bool streamed = sp.audio_chunk_frames > 0 && sp.audio_window_frames > 0 && !any_clone && !dbg && m_->has_codec;
    hipStream_t sst = nullptr;
    std::vector<int> s_avail((size_t)(n), 0);
    std::vector<uint8_t> s_final((size_t)(n), 0);
    std::memset(J.nf_host, 0, size_t(Bm_) * 4);
    J.streamed = false;
    // the rows' non-finite flags behind every chunk of a decode in pieces (fire_chunks holds a row back from its first flagged chunk)
    auto chunk_flags = [&](int frames_cap) {
        const size_t slots = size_t(ceil_div(frames_cap, sp.audio_chunk_frames) + 1) * size_t(n);
        if (slots > J.nf_chunk_cap) {
            if (J.nf_chunk_host) Q3_HIP(hipHostFree(J.nf_chunk_host));
            J.nf_chunk_host = nullptr;
            J.nf_chunk_cap = 0;
            Q3_HIP(hipHostMalloc(reinterpret_cast<void**>(&J.nf_chunk_host), slots * 4, hipHostMallocDefault));
            J.nf_chunk_cap = slots;
        }
        std::memset(J.nf_chunk_host, 0, slots * 4);
    };
    J.held_from.assign(size_t(n), -1);
    J.chunks_fired = 0;
    J.t_first_audio = 0;
    J.n_chunks = 0;
    if (streamed) {
        J.n = n;
        J.up = codec_->upsample();
        J.Fdec = Fcap_;  // row stride of the job's code and PCM buffers: the final lengths are not known yet
        J.chunk_frames = sp.audio_chunk_frames;
        J.cb = cb;
        J.user = user;
        J.request_base = request_base;
        const size_t need = size_t(n) * Fcap_ * 16, floats = size_t(n) * Fcap_ * J.up;
        if (need > J.dec_codes_cap) {
            if (J.dec_codes) Q3_HIP(hipFree(J.dec_codes));
            J.dec_codes = nullptr;
            Q3_HIP(hipMalloc(reinterpret_cast<void**>(&J.dec_codes), need * 4));
            J.dec_codes_cap = need;
        }
        if (floats > J.pcm_host_cap) {
            if (J.pcm_host) Q3_HIP(hipHostFree(J.pcm_host));
            J.pcm_host = nullptr;
            Q3_HIP(hipHostMalloc(reinterpret_cast<void**>(&J.pcm_host), floats * 4, hipHostMallocDefault));
            J.pcm_host_cap = floats;
        }
        chunk_flags(Fcap_);
        // the decode runs beside this batch's own frame loop: the confined stream, like a decode beside the next batch's
        sst = codec_stream(true);
        CodecRunner::StreamCfg cfg;
        cfg.rows = n; cfg.chunk_frames = sp.audio_chunk_frames; cfg.window = sp.audio_window_frames;
        cfg.lookahead = std::max(0, sp.audio_lookahead_frames); cfg.max_frames = Fcap_;
        Q3_HIP(hipEventRecord(J.ev_codec[0], sst));
        codec_->stream_open(cfg);
        J.streamed = true;
    }
    struct StreamGuard {  // an exception below must not leave the runner's stream open
        CodecRunner* c;
        bool on;
        ~StreamGuard() { if (on && c->streaming()) c->stream_close(); }
    } stream_guard{codec_.get(), streamed};
    // frames [0, upto) of every row exist on the device once the copy below has run: hand them to the decoder, which
    // issues every chunk that s_avail / s_final now allow
    int s_copied = 0;
    auto stream_feed = [&](int upto) {
        if (!streamed) return;
        if (upto > s_copied) {
            Q3_HIP(hipMemcpy2DAsync(J.dec_codes + size_t(s_copied) * 16, size_t(Fcap_) * 64, codes_ + size_t(s_copied) * 16,
                                    size_t(Fcap_) * 64, size_t(upto - s_copied) * 64, size_t(n), hipMemcpyDeviceToDevice, st_));
            s_copied = upto;
            Q3_HIP(hipEventRecord(ev_[3], st_));
            Q3_HIP(hipStreamWaitEvent(sst, ev_[3], 0));
        }
        const int before = J.n_chunks;
        J.n_chunks = codec_->stream_push(J.dec_codes, Fcap_, s_avail.data(), s_final.data(), J.pcm_host, size_t(Fcap_) * J.up, J.chunk_done,
                                         J.nf_chunk_host);
        if (before == 0 && J.n_chunks > 0) Q3_HIP(hipEventRecord(J.ev_first_audio, sst));  // behind chunk 0's copy to the host
    };

    // ---- frame loop ----
    const bool use_graph = opts_.use_graph && !dbg;
    // the tables are cut from the weights, which may arrive after load (weights_from_broadcast): first use, not load
    if (!cp_tables_ && m_->has_cp_proj && !std::getenv("Q3TTS_NO_PROJ_TABLES")) build_cp_proj_tables();
    hipGraphExec_t ge = use_graph ? frame_graph(n) : nullptr;
    std::vector<int32_t> h_nframes((size_t)(n), 0), h_codes;
    std::vector<uint8_t> h_fin((size_t)(n), 0);
    std::vector<int> reported((size_t)(n), 0);
    int launched = 0;
    const bool fixed_len = sp.force_frames > 0 || dbg;
    // Frames are enqueued in bursts with at most two bursts in flight (event ring), so the AQL queue never fills:
    // a host thread blocked on queue back-pressure starves the other lanes' submissions (measured: lanes gave no
    // speed-up until the depth was bounded). Variable-length runs also poll the finished flags once per burst.
    const int burst_frames = std::max(1, max_inflight_frames / 2);
    hipEvent_t ring[2] = {burst_ev_[0], burst_ev_[1]};
    int bursts = 0;
    bool done = false;
    while (!done && launched < frames_cap) {
        if (bursts >= 2) Q3_HIP(hipEventSynchronize(ring[bursts & 1]));  // burst (bursts-2) has drained
        int burst = std::min(burst_frames, frames_cap - launched);
        if (streamed) {
            // a burst ends where the next chunk becomes decodable (its frames + the lookahead), so the chunk is issued behind
            // exactly the frames it needs instead of behind the rest of a full burst (first audio 139 -> 115 ms at 1.7B / batch 32)
            const int need = std::min(frames_cap, (J.n_chunks + 1) * sp.audio_chunk_frames + std::max(0, sp.audio_lookahead_frames));
            if (need > launched) burst = std::min(burst, need - launched);
        }
        for (int i = 0; i < burst; ++i) {
            if (use_graph) Q3_HIP(hipGraphLaunch(ge, st_));
            else enqueue_frame(n, dbg);
        }
        launched += burst;
        if (!(fixed_len && !cb)) {  // per-burst poll of the flags (async copies ordered after the burst)
            Q3_HIP(hipMemcpyAsync(h_nframes.data(), n_frames_, size_t(n) * 4, hipMemcpyDeviceToHost, st_));
            Q3_HIP(hipMemcpyAsync(h_fin.data(), finished_, size_t(n), hipMemcpyDeviceToHost, st_));
        }
        Q3_HIP(hipEventRecord(ring[bursts & 1], st_));
        ++bursts;
        if (fixed_len && !cb) {
            if (streamed) {  // every row has exactly `launched` frames (nothing ends early)
                for (int b = 0; b < n; ++b) s_avail[size_t(b)] = std::min(launched, maxf[size_t(b)]);
                stream_feed(launched);
            }
            continue;
        }
        Q3_HIP(hipEventSynchronize(ring[(bursts - 1) & 1]));
        done = true;
        for (int b = 0; b < n; ++b) done = done && h_fin[size_t(b)];
        if (streamed) {
            for (int b = 0; b < n; ++b) {
                s_avail[size_t(b)] = h_nframes[size_t(b)];
                s_final[size_t(b)] = h_fin[size_t(b)];
            }
            stream_feed(launched);
            fire_chunks(J, J.n_chunks, &s_avail, false);  // what has already landed on the host, without waiting
        }
        if (cb) {  // .token events in generation order (Qwen3+Streaming.swift:24-27)
            for (int b = 0; b < n; ++b) {
                const int nf = h_nframes[size_t(b)];
                if (nf > reported[size_t(b)]) {
                    std::vector<int32_t> tmp((size_t)(nf - reported[size_t(b)]) * 16);
                    Q3_HIP(hipMemcpy(tmp.data(), codes_ + (size_t(b) * Fcap_ + reported[size_t(b)]) * 16, tmp.size() * 4,
                                     hipMemcpyDeviceToHost));
                    std::unique_lock<std::mutex> lk;
                    if (cb_mutex) lk = std::unique_lock<std::mutex>(*cb_mutex);
                    for (int f = 0; f < nf - reported[size_t(b)]; ++f) {
                        q3tts_event ev{};
                        ev.kind = Q3TTS_EVENT_TOKEN;
                        ev.request_index = request_base + b;
                        ev.token = tmp[size_t(f) * 16];
                        cb(user, &ev);
                    }
                    reported[size_t(b)] = nf;
                }
            }
        }
    }
    Q3_HIP(hipEventRecord(ev_[2], st_));
    Q3_HIP(hipMemcpyAsync(h_nframes.data(), n_frames_, size_t(n) * 4, hipMemcpyDeviceToHost, st_));
    Q3_HIP(hipStreamSynchronize(st_));
    if (dbg) {
        const size_t nf = size_t(n) * dbg->frames;
        if (dbg->sampled) Q3_HIP(hipMemcpy(dbg->sampled, sampled_dev_, nf * 16 * 4, hipMemcpyDeviceToHost));
        if (dbg->talker_logits) Q3_HIP(hipMemcpy(dbg->talker_logits, tl_dump_, nf * V * 2, hipMemcpyDeviceToHost));
        if (dbg->cp_logits) Q3_HIP(hipMemcpy(dbg->cp_logits, cl_dump_, nf * (groups - 1) * Vc * 2, hipMemcpyDeviceToHost));
    }

    // ---- hand the codes to the codec decoder (Qwen3.swift:943-961) on its own stream ----
    J.n = n;
    J.up = codec_->upsample();
    J.frames.assign(size_t(n), 0);
    J.ref_T.assign(size_t(n), 0);
    J.target_tokens.assign(size_t(n), 0);
    J.ref_code0.assign(size_t(n), {});
    std::vector<int> dframes((size_t)(n), 0);  // frames the decoder sees per row: [reference ++] generated (:1176-1186)
    int Fdec = 0;
    for (int b = 0; b < n; ++b) {
        const int F = h_nframes[size_t(b)];
        J.frames[size_t(b)] = F;
        J.ref_T[size_t(b)] = rr[size_t(b)].clone ? rr[size_t(b)].ref_T : 0;
        J.target_tokens[size_t(b)] = rr[size_t(b)].target_token_count;
        dframes[size_t(b)] = F > 0 ? F + J.ref_T[size_t(b)] : 0;
        Fdec = std::max(Fdec, dframes[size_t(b)]);
    }
    J.Fdec = streamed ? Fcap_ : Fdec;
    J.codes_host.resize(size_t(n) * Fcap_ * 16);
    Q3_HIP(hipMemcpyAsync(J.codes_host.data(), codes_, J.codes_host.size() * 4, hipMemcpyDeviceToHost, st_));
    if (streamed) {  // the remaining chunks: every row is final now
        for (int b = 0; b < n; ++b) {
            s_avail[size_t(b)] = J.frames[size_t(b)];
            s_final[size_t(b)] = 1;
        }
        stream_feed(launched);
        codec_->stream_close(J.nf_host);
        Q3_HIP(hipStreamSynchronize(st_));
        Q3_HIP(hipEventRecord(J.ev_codec[1], sst));
        J.decoded = Fdec > 0;
    } else if (Fdec > 0) {
        // The decoder reads a copy owned by the job: the next begin() overwrites codes_ while this decode may still run.
        const size_t need = size_t(n) * Fdec * 16;
        if (need > J.dec_codes_cap) {
            if (J.dec_codes) Q3_HIP(hipFree(J.dec_codes));
            J.dec_codes = nullptr;
            Q3_HIP(hipMalloc(reinterpret_cast<void**>(&J.dec_codes), need * 4));
            J.dec_codes_cap = need;
        }
        if (any_clone) {
            for (int b = 0; b < n; ++b) {
                const auto& r = rr[size_t(b)];
                if (J.frames[size_t(b)] == 0) continue;
                launch_build_decode_codes(r.clone ? ref_codes_dev_ + r.ref_off : nullptr, r.clone ? r.ref_T : 0,
                                          codes_ + size_t(b) * Fcap_ * 16, J.frames[size_t(b)], J.dec_codes + size_t(b) * Fdec * 16, st_);
                if (r.clone) {
                    J.ref_code0[size_t(b)].resize(size_t(r.ref_T));
                    Q3_HIP(hipMemcpyAsync(J.ref_code0[size_t(b)].data(), ref_codes_dev_ + r.ref_off, size_t(r.ref_T) * 4,
                                          hipMemcpyDeviceToHost, st_));
                }
            }
        } else {
            Q3_HIP(hipMemcpy2DAsync(J.dec_codes, size_t(Fdec) * 64, codes_, size_t(Fcap_) * 64, size_t(Fdec) * 64, size_t(n),
                                    hipMemcpyDeviceToDevice, st_));
        }
    }
    if (!streamed) {
    Q3_HIP(hipStreamSynchronize(st_));  // everything of this call on st_ is done; only the decode is still to come
    hipStream_t cst = codec_stream(overlapped);
    Q3_HIP(hipEventRecord(J.ev_codec[0], cst));
    J.decoded = false;
    J.n_chunks = 0;
    J.chunk_frames = sp.audio_chunk_frames > 0 ? sp.audio_chunk_frames : 0;
    if (Fdec > 0) {
        const size_t floats = size_t(n) * Fdec * J.up;
        if (floats > J.pcm_host_cap) {
            if (J.pcm_host) Q3_HIP(hipHostFree(J.pcm_host));
            J.pcm_host = nullptr;
            Q3_HIP(hipHostMalloc(reinterpret_cast<void**>(&J.pcm_host), floats * 4, hipHostMallocDefault));
            J.pcm_host_cap = floats;
        }
        if (J.chunk_frames > 0) {
            // pre-transformer once over all frames, then the causal tail chunk by chunk (codec.h decode_chunked)
            chunk_flags(Fdec);
            J.n_chunks = codec_->decode_chunked(J.dec_codes, Fdec, dframes, J.chunk_frames, J.pcm_host, J.chunk_done, J.nf_host, J.nf_chunk_host);
        } else {
            float* pcm_dev = nullptr;
            codec_->decode(J.dec_codes, Fdec, dframes, &pcm_dev, std::string(), nullptr, nullptr, nullptr, J.nf_host);
            Q3_HIP(hipMemcpyAsync(J.pcm_host, pcm_dev, floats * 4, hipMemcpyDeviceToHost, cst));
        }
        J.decoded = true;
    }
    Q3_HIP(hipEventRecord(J.ev_codec[1], cst));
    }
    J.timing = q3tts_timing{};
    float ms = 0;
    Q3_HIP(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
    J.timing.prefill_ms = ms;
    Q3_HIP(hipEventElapsedTime(&ms, ev_[1], ev_[2]));
    J.timing.decode_ms = ms;
    Q3_HIP(hipEventElapsedTime(&ms, ev_fe_[0], ev_fe_[1]));
    J.timing.frontend_ms = ms;
    J.timing.frame_steps = launched;
    {
        auto fl = frame_launches_.find(n);
        J.timing.launches_per_frame_step = fl == frame_launches_.end() ? 0 : fl->second;
    }
    J.timing.rows = n;
    {
        int64_t kvb = 0;
        const int64_t per_tok = int64_t(t.num_hidden_layers) * t.num_key_value_heads * kHeadDim * 2 * 2;
        for (int b = 0; b < n; ++b)
            for (int f = 0; f < J.frames[size_t(b)]; ++f) kvb += int64_t(np[size_t(b)] - 1 + f) * per_tok;
        J.timing.kv_bytes_read = kvb;
    }
    J.t_start = t_start;
    J.t_done = 0;
    J.cb = cb;
    J.user = user;
    J.request_base = request_base;
    J.seq = job_seq_++;
    J.busy = true;
    compute_cuts(J);
    {
        std::lock_guard<std::mutex> lk(stage_mu_);
        J.stage_err.clear();
        J.stage = (overlapped && J.decoded) ? 1 : 0;  // a pipelined job: its rows are copied out while the next batch runs
        if (J.stage == 1) {
            if (!stager_.joinable()) stager_ = std::thread([this] { staging_loop(); });
            stage_cv_.notify_all();
        }
    }
    return slot;
}

// samples [cut, cut + ns) of row b's decoded stream are its audio: audioLengths = count(code0 > 0) * 1920, trimmed when
// 0 < valid < len (SpeechTokenizer.swift:831-833, Qwen3.swift:954-959); clone rows lose the reference's share
// (Qwen3.swift:1195-1199, Float arithmetic)
void Engine::compute_cuts(Job& J) {
    const int n = J.n, up = J.up;
    J.row_cut.assign(size_t(n), 0);
    J.row_ns.assign(size_t(n), 0);
    for (int b = 0; b < n; ++b) {
        const int F = J.frames[size_t(b)];
        if (F == 0 || !J.decoded) continue;
        const int32_t* codes = J.codes_host.data() + size_t(b) * Fcap_ * 16;
        int valid_tok = 0;
        for (int f = 0; f < F; ++f) valid_tok += codes[size_t(f) * 16] > 0 ? 1 : 0;
        for (int32_t c : J.ref_code0[size_t(b)]) valid_tok += c > 0 ? 1 : 0;
        const int ref_T = J.ref_T[size_t(b)], total_f = ref_T + F;
        int64_t ns = int64_t(total_f) * up;
        const int64_t valid = int64_t(valid_tok) * up;
        // (a streamed row has already delivered its chunks frame by frame when the count becomes known: its AUDIO is their
        // concatenation, untrimmed -- include/q3tts.h)
        if (!J.streamed && valid > 0 && valid < ns) ns = valid;
        int64_t cut = 0;
        if (ref_T > 0) {
            cut = int64_t(float(ref_T) / float(std::max(total_f, 1)) * float(ns));
            if (!(cut > 0 && cut < ns)) cut = 0;
        }
        J.row_cut[size_t(b)] = cut;
        J.row_ns[size_t(b)] = ns - cut;
    }
}

// AUDIO_CHUNK events of chunks [J.chunks_fired, upto). known == nullptr: the rows' final cuts (compute_cuts) bound the pieces;
// otherwise row b has known[b] frames so far and nothing is cut in front (a streamed job, still inside its frame loop).
// wait = false delivers only what has already landed on the host.
void Engine::fire_chunks(Job& J, int upto, const std::vector<int>* known, bool wait) {
    const int n = J.n, up = J.up;
    for (int k = J.chunks_fired; k < upto; ++k) {
        if (wait) {
            Q3_HIP(hipEventSynchronize(J.chunk_done[size_t(k)]));
        } else {
            const hipError_t q = hipEventQuery(J.chunk_done[size_t(k)]);
            if (q == hipErrorNotReady) return;
            Q3_HIP(q);
        }
        J.chunks_fired = k + 1;
        if (J.nf_chunk_host)  // a row that has left the fp16 range delivers nothing more until end() has decoded it again
            for (int b = 0; b < n; ++b)
                if (J.held_from[size_t(b)] < 0 && J.nf_chunk_host[size_t(k) * n + b]) J.held_from[size_t(b)] = k;
        if (!J.cb) continue;
        const int64_t c0 = int64_t(k) * J.chunk_frames * up, c1 = std::min<int64_t>(int64_t(J.Fdec), int64_t(k + 1) * J.chunk_frames) * up;
        std::unique_lock<std::mutex> lk;
        if (cb_mutex) lk = std::unique_lock<std::mutex>(*cb_mutex);
        for (int b = 0; b < n; ++b) {
            if (J.held_from[size_t(b)] >= 0) continue;
            const int64_t cut = known ? 0 : J.row_cut[size_t(b)];
            const int64_t len = known ? int64_t((*known)[size_t(b)]) * up : J.row_ns[size_t(b)];
            const int64_t lo = std::max(c0, cut), hi = std::min(c1, cut + len);
            if (hi <= lo) continue;
            q3tts_event ev{};
            ev.kind = Q3TTS_EVENT_AUDIO_CHUNK;
            ev.request_index = J.request_base + b;
            ev.pcm = J.pcm_host + size_t(b) * J.Fdec * up + lo;
            ev.n_samples = hi - lo;
            ev.sample_offset = lo - cut;
            J.cb(J.user, &ev);
        }
    }
}

void Engine::stage_rows(Job& J) {
    Q3_HIP(hipSetDevice(m_->device));
    Q3_HIP(hipEventSynchronize(J.ev_codec[1]));
    J.t_done = now_s();  // the job's own completion, not the moment end() happens to be called (a pipelined job's end()
                         // comes after the NEXT batch's whole frame loop)
    const int n = J.n;
    J.st_pcm.assign(size_t(n), nullptr);
    J.st_codes.assign(size_t(n), nullptr);
    for (int b = 0; b < n; ++b) {
        const int F = J.frames[size_t(b)];
        if (F == 0 || !J.decoded) continue;
        const int64_t ns = J.row_ns[size_t(b)];
        J.st_codes[size_t(b)] = static_cast<int32_t*>(std::malloc(size_t(F) * 16 * 4));
        J.st_pcm[size_t(b)] = static_cast<float*>(std::malloc(std::max<size_t>(size_t(ns) * 4, 4)));
        Q3_CHECK(J.st_codes[size_t(b)] && J.st_pcm[size_t(b)], 5, "out of host memory for the results");
        std::memcpy(J.st_codes[size_t(b)], J.codes_host.data() + size_t(b) * Fcap_ * 16, size_t(F) * 16 * 4);
        std::memcpy(J.st_pcm[size_t(b)], J.pcm_host + size_t(b) * J.Fdec * J.up + J.row_cut[size_t(b)], size_t(ns) * 4);
    }
}

void Engine::staging_loop() {
    std::unique_lock<std::mutex> lk(stage_mu_);
    for (;;) {
        Job* next = nullptr;
        for (auto& J : jobs_)
            if (J.stage == 1 && (!next || J.seq < next->seq)) next = &J;
        if (!next) {
            if (stage_stop_) return;
            stage_cv_.wait(lk);
            continue;
        }
        lk.unlock();
        int state = 2;
        std::string err;
        try {
            stage_rows(*next);
        } catch (const std::exception& e) {
            state = 3;
            err = e.what();
        }
        lk.lock();
        next->stage = state;
        next->stage_err = err;
        stage_cv_.notify_all();
    }
}

// The default codec kernels contract on the fp16 matrix cores (two planes per fp32 operand, codec_conv.hip): an activation
// beyond 65504 turns into inf / NaN, which out_conv flags per row. The reference computes in fp32 and decodes such inputs, so a
// flagged row is decoded AGAIN here on the fp32 matrix cores (the fp32 weights stay resident; 2.4x the time, for those rows
// only) and handed out like any other. A streamed row (audio_window_frames > 0) stopped delivering chunks at its first flagged
// one (fire_chunks): its remaining AUDIO_CHUNK events come from this decode -- the exact one-shot arithmetic -- and its AUDIO
// is what was delivered. Returns the rows that are non-finite even in fp32 (they fail with AUDIO_DECODING_FAILED).
std::vector<int> Engine::redo_rows_fp32(Job& J) {
    std::vector<int> rows, bad;
    if (!J.decoded) return bad;
    for (int b = 0; b < J.n; ++b)
        if (J.nf_host[b] && J.frames[size_t(b)] > 0 && J.st_pcm[size_t(b)]) rows.push_back(b);
    if (rows.empty()) return bad;
    if (codec_->fp32_convs()) return rows;  // already the fp32 kernels: nothing wider to fall back to
    const int R = int(rows.size()), up = J.up;
    std::vector<int> dframes((size_t)(R));
    int Fd = 0;
    for (int i = 0; i < R; ++i) {
        const int b = rows[size_t(i)];
        dframes[size_t(i)] = J.frames[size_t(b)] + J.ref_T[size_t(b)];
        Fd = std::max(Fd, dframes[size_t(i)]);
    }
    hipStream_t cst = codec_stream(false);
    int32_t* dcodes = nullptr;
    int32_t* nf = nullptr;
    float* hpcm = nullptr;
    auto cleanup = [&] {
        if (dcodes) (void)hipFree(dcodes);
        if (nf) (void)hipHostFree(nf);
        if (hpcm) (void)hipHostFree(hpcm);
    };
    try {
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&dcodes), size_t(R) * Fd * 64));
        Q3_HIP(hipHostMalloc(reinterpret_cast<void**>(&nf), size_t(R) * 4, hipHostMallocDefault));
        Q3_HIP(hipHostMalloc(reinterpret_cast<void**>(&hpcm), size_t(R) * Fd * up * 4, hipHostMallocDefault));
        std::memset(nf, 0, size_t(R) * 4);
        // the codes the first decode read: the job's own device copy ([reference ++] generated; row stride J.Fdec frames)
        for (int i = 0; i < R; ++i)
            Q3_HIP(hipMemcpyAsync(dcodes + size_t(i) * Fd * 16, J.dec_codes + size_t(rows[size_t(i)]) * J.Fdec * 16,
                                  size_t(dframes[size_t(i)]) * 64, hipMemcpyDeviceToDevice, cst));
        float* pcm_dev = nullptr;
        codec_->decode(dcodes, Fd, dframes, &pcm_dev, std::string(), nullptr, nullptr, nullptr, nf, true);
        Q3_HIP(hipMemcpyAsync(hpcm, pcm_dev, size_t(R) * Fd * up * 4, hipMemcpyDeviceToHost, cst));
        Q3_HIP(hipStreamSynchronize(cst));
        for (int i = 0; i < R; ++i) {
            const int b = rows[size_t(i)];
            if (nf[i]) {
                bad.push_back(b);
                continue;
            }
            // decoded-stream coordinates: sample p of the decode is sample p - cut of the row's audio; chunk k covers
            // [k * step, (k + 1) * step). Chunks below `held` have been delivered from the first decode and stay as they are.
            const int64_t cut = J.row_cut[size_t(b)], ns = J.row_ns[size_t(b)], step = int64_t(std::max(J.chunk_frames, 1)) * up;
            const int held = J.n_chunks > 0 ? std::max(0, J.held_from[size_t(b)]) : 0;
            const int64_t from = J.n_chunks > 0 ? std::min(ns, std::max<int64_t>(0, int64_t(held) * step - cut)) : 0;
            std::memcpy(J.st_pcm[size_t(b)] + from, hpcm + size_t(i) * Fd * up + cut + from, size_t(ns - from) * 4);
            if (J.cb && J.n_chunks > 0) {  // the pieces fire_chunks held back
                std::unique_lock<std::mutex> lk;
                if (cb_mutex) lk = std::unique_lock<std::mutex>(*cb_mutex);
                for (int k = held; k < J.n_chunks; ++k) {
                    const int64_t lo = std::max(int64_t(k) * step, cut), hi = std::min(int64_t(k + 1) * step, cut + ns);
                    if (hi <= lo) continue;
                    q3tts_event ev{};
                    ev.kind = Q3TTS_EVENT_AUDIO_CHUNK;
                    ev.request_index = J.request_base + b;
                    ev.pcm = J.st_pcm[size_t(b)] + (lo - cut);
                    ev.n_samples = hi - lo;
                    ev.sample_offset = lo - cut;
                    J.cb(J.user, &ev);
                }
            }
        }
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
    return bad;
}


void Engine::end(int job, q3tts_result* results) {
    Q3_CHECK(job >= 0 && job < kJobSlots && jobs_[job].busy, 3, "Invalid input: no such outstanding job");
    Job& J = jobs_[job];
    // Whatever happens below, the slot is released and rows staged for it are freed: a failure inside end() must not
    // leave the handle with a job nobody can end (q3tts_generate_end has already dropped the caller's handle by then).
    struct Release {
        Engine* e;
        Job* j;
        bool handed_over = false;
        ~Release() {
            if (!handed_over) {
                {   // a staging thread may still be copying into the vectors
                    std::unique_lock<std::mutex> lk(e->stage_mu_);
                    e->stage_cv_.wait(lk, [&] { return j->stage != 1; });
                }
                for (float* p : j->st_pcm) std::free(p);
                for (int32_t* p : j->st_codes) std::free(p);
            }
            j->st_pcm.clear();
            j->st_codes.clear();
            j->busy = false;
        }
    } release{this, &J};
    Q3_HIP(hipSetDevice(m_->device));
    const int n = J.n;
    const std::vector<int64_t>& row_ns = J.row_ns;
    if (J.n_chunks > 0) fire_chunks(J, J.n_chunks, nullptr, true);  // AUDIO_CHUNK events not delivered inside the loop
    {
        std::unique_lock<std::mutex> lk(stage_mu_);
        stage_cv_.wait(lk, [&] { return J.stage != 1; });
    }
    if (J.stage == 0) {
        try {
            stage_rows(J);
            J.stage = 2;
        } catch (const std::exception& e) {
            J.stage = 3;
            J.stage_err = e.what();
        }
    }
    if (J.stage == 3) throw Error(5, J.stage_err);
    // rows whose activations left the fp16 range of the default codec kernels: once more on the fp32 matrix cores
    const std::vector<int> still_bad = redo_rows_fp32(J);
    float ms = 0;
    Q3_HIP(hipEventElapsedTime(&ms, J.ev_codec[0], J.ev_codec[1]));
    J.timing.codec_ms = ms;  // on the codec stream: includes whatever the next batch's AR loop took away from it
    if (J.streamed && J.n_chunks > 0) {
        Q3_HIP(hipEventElapsedTime(&ms, J.ev_begin, J.ev_first_audio));
        J.timing.first_audio_ms = ms;
    }
    timing = J.timing;
    const double total = (J.t_done > 0 ? J.t_done : now_s()) - J.t_start;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    for (int b = 0; b < n; ++b) {
        q3tts_result& r = results[b];
        std::memset(&r, 0, sizeof(r));
        const int F = J.frames[size_t(b)];
        r.info.prompt_token_count = J.target_tokens[size_t(b)];  // tokens of `text` (Qwen3+Streaming.swift:106)
        r.info.generation_token_count = F;
        r.info.prefill_time = 0;  // hard-coded in the reference (Qwen3+Streaming.swift:112)
        r.info.generate_time = total;
        r.info.tokens_per_second = total > 0 ? double(F) / total : 0;
        r.info.peak_memory_usage = double(total_b - free_b) / 1e9;
        if (F == 0 || !J.decoded) {  // Qwen3.swift:939-941
            r.status = Q3TTS_ERR_GENERATION_FAILED;
            continue;
        }
        if (std::find(still_bad.begin(), still_bad.end(), b) != still_bad.end()) {  // never hand out a waveform with holes in it
            r.status = Q3TTS_ERR_AUDIO_DECODING_FAILED;
            last_error = kCodecRangeMsg;
            std::free(J.st_pcm[size_t(b)]);
            std::free(J.st_codes[size_t(b)]);
            J.st_pcm[size_t(b)] = nullptr;
            J.st_codes[size_t(b)] = nullptr;
            continue;
        }
        r.n_frames = F;
        r.codes = J.st_codes[size_t(b)];  // ownership passes to the result (q3tts_result_free)
        r.n_samples = row_ns[size_t(b)];
        r.pcm = J.st_pcm[size_t(b)];
        r.status = Q3TTS_OK;
    }
    release.handed_over = true;
    if (J.cb) {
        std::unique_lock<std::mutex> lk;
        if (cb_mutex) lk = std::unique_lock<std::mutex>(*cb_mutex);
        for (int b = 0; b < n; ++b) {
            if (results[b].status != Q3TTS_OK) continue;
            q3tts_event ev{};
            ev.kind = Q3TTS_EVENT_INFO;
            ev.request_index = J.request_base + b;
            ev.info = &results[b].info;
            J.cb(J.user, &ev);
            ev.kind = Q3TTS_EVENT_AUDIO;
            ev.info = nullptr;
            ev.pcm = results[b].pcm;
            ev.n_samples = results[b].n_samples;
            J.cb(J.user, &ev);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// block-level hooks
// ------------------------------------------------------------------------------------------------
void Engine::debug_sample(const uint16_t* logits, int rows, int V, const q3tts_sampling& sp, const uint8_t* seen,
                          int suppress_lo, int suppress_hi, int eos_id, uint32_t row0, uint32_t draw, int32_t* tokens) {
    Q3_CHECK(rows >= 1 && rows <= Bm_ && V <= m_->cfg.talker.vocab_size, 3, "debug_sample: rows/V out of range");
    Q3_CHECK(draw % 16 == 0, 3, "debug_sample: draw must be a multiple of 16 (frame * 16)");
    const int Vt = m_->cfg.talker.vocab_size;
    Q3_HIP(hipMemcpy2D(tk_.logits, size_t(Vt) * 2, logits, size_t(V) * 2, size_t(V) * 2, size_t(rows), hipMemcpyHostToDevice));
    if (seen) Q3_HIP(hipMemcpy(seen_, seen, size_t(rows) * V, hipMemcpyHostToDevice));
    std::vector<int32_t> fr((size_t)(rows), int32_t(draw / 16)), big((size_t)(rows), 1 << 30);
    Q3_HIP(hipMemcpy(n_frames_, fr.data(), size_t(rows) * 4, hipMemcpyHostToDevice));
    Q3_HIP(hipMemcpy(max_frames_, big.data(), size_t(rows) * 4, hipMemcpyHostToDevice));
    Q3_HIP(hipMemset(finished_, 0, size_t(rows)));
    Q3_HIP(hipMemset(active_, 1, size_t(rows)));
    SamplingParams sph{sp.temperature, sp.top_k, sp.top_p, sp.repetition_penalty, sp.seed, row0, sp.force_frames > 0 ? 1 : 0};
    Q3_HIP(hipMemcpy(sp_dev_, &sph, sizeof(sph), hipMemcpyHostToDevice));
    SamplerArgs sa{};
    sa.logits = tk_.logits; sa.ldl = Vt; sa.V = V; sa.sp = sp_dev_;
    sa.is_talker = (eos_id >= 0 || suppress_hi > suppress_lo || seen) ? 1 : 0;
    sa.suppress_lo = suppress_lo; sa.suppress_hi = suppress_hi; sa.eos_id = eos_id;
    sa.seen = seen ? seen_ : nullptr; sa.cb = 0; sa.n_frames = n_frames_; sa.max_frames = max_frames_;
    sa.finished = finished_; sa.active = active_; sa.kv_len = kv_len_; sa.advance = 0;
    sa.cur_codes = cur_codes_; sa.codes = codes_; sa.Fmax = 0; sa.B = rows; sa.H = m_->cfg.talker.hidden_size;
    launch_sampler(sa, st_);
    Q3_HIP(hipStreamSynchronize(st_));
    std::vector<int32_t> cc((size_t)(rows) * 16);
    Q3_HIP(hipMemcpy(cc.data(), cur_codes_, cc.size() * 4, hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; ++r) tokens[r] = cc[size_t(r) * 16];
    Q3_HIP(hipMemset(seen_, 0, size_t(rows) * V));
}

void Engine::debug_linear(const uint16_t* x, const uint16_t* W, const uint16_t* bias, int M, int K, int N, uint16_t* y) {
    Q3_CHECK(M >= 1 && M <= 1024 && K % 8 == 0 && N >= 1, 3, "debug_linear: unsupported shape");  // > 64 rows: the tall form
    const int Kp = int(align_up(size_t(K), 128)), Np = int(align_up(size_t(N), 16)), Mp = int(align_up(size_t(M), 16));
    uint16_t *dW = nullptr, *dWt = nullptr, *dx = nullptr, *dy = nullptr, *db = nullptr;
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&dW), size_t(N) * K * 2));
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&dWt), size_t(Np) * Kp * 2));
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&dx), size_t(Mp) * Kp * 2));
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&dy), size_t(Mp) * Np * 2));
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&db), size_t(Np) * 2));
    Q3_HIP(hipMemset(dWt, 0, size_t(Np) * Kp * 2));
    Q3_HIP(hipMemset(dx, 0, size_t(Mp) * Kp * 2));
    Q3_HIP(hipMemset(db, 0, size_t(Np) * 2));
    Q3_HIP(hipMemcpy(dW, W, size_t(N) * K * 2, hipMemcpyHostToDevice));
    uint16_t* dxl = nullptr;  // row-major staging, converted to the fragment-major operand layout
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&dxl), size_t(Mp) * Kp * 2));
    Q3_HIP(hipMemset(dxl, 0, size_t(Mp) * Kp * 2));
    Q3_HIP(hipMemcpy2D(dxl, size_t(Kp) * 2, x, size_t(K) * 2, size_t(K) * 2, size_t(M), hipMemcpyHostToDevice));
    launch_tile_rows(dxl, Kp, dx, Mp / 16, Mp, Kp, st_);
    if (bias) Q3_HIP(hipMemcpy(db, bias, size_t(N) * 2, hipMemcpyHostToDevice));
    launch_tile_weights(dW, N, K, dWt, Kp / 128, 0, 1, st_);
    LinearW L;
    L.w = dWt; L.bias = bias ? db : nullptr; L.N = N; L.K = K; L.Np = Np; L.Kp = Kp;
    GemmArgs ga{};
    ga.W = dWt; ga.x = dx; ga.xMB = Mp / 16; ga.M = M; ga.Mpad = Mp; ga.N = Np; ga.K = Kp; ga.epi = 0; ga.y = dy; ga.ldy = Np;
    ga.bias = L.bias; ga.ss_ld = Mp;
    launch_gemm_skinny(ga, st_);
    Q3_HIP(hipStreamSynchronize(st_));
    Q3_HIP(hipMemcpy2D(y, size_t(N) * 2, dy, size_t(Np) * 2, size_t(N) * 2, size_t(M), hipMemcpyDeviceToHost));
    for (void* p : {(void*)dW, (void*)dWt, (void*)dx, (void*)dxl, (void*)dy, (void*)db})
        if (p) (void)hipFree(p);
}

// Codes a CALLER hands in (q3tts_codec_decode, q3tts_codec_decode_streamed) index the RVQ tables on the GPU: every code of every frame
// that will be decoded is checked against the tables as loaded. (Codes the engine sampled itself are inside by construction:
// the samplers draw below the vocabulary, the tables have at least that many rows.)
static void check_caller_codes(const CodecW& w, const int32_t* codes, const int32_t* n_frames, int batch, int max_frames) {
    for (int b = 0; b < batch; ++b)
        for (int f = 0; f < n_frames[b]; ++f) {
            const int32_t* c = codes + (size_t(b) * max_frames + f) * 16;
            Q3_CHECK(c[0] >= 0 && c[0] < w.cb_first_rows, 3, "Invalid input: first code outside the semantic codebook");
            for (size_t j = 0; j < w.cb_rest.size(); ++j)
                Q3_CHECK(c[1 + j] >= 0 && c[1 + j] < w.cb_rest_rows, 3, "Invalid input: code outside the acoustic codebook");
        }
}

void Engine::codec_decode(const int32_t* codes, const int32_t* n_frames, int batch, int max_frames, float* pcm,
                          int64_t* audio_lengths) {
    Q3_CHECK(m_->has_codec, 1, "Model not initialized: Speech tokenizer not loaded");
    Q3_CHECK(batch >= 1 && max_frames >= 1, 3, "Invalid input: empty codec batch");
    const int up = codec_->upsample();
    std::vector<int> frames((size_t)(batch));
    int Fmax = 0;
    for (int b = 0; b < batch; ++b) {
        Q3_CHECK(n_frames[b] >= 0 && n_frames[b] <= max_frames, 3, "Invalid input: n_frames out of range");
        frames[size_t(b)] = n_frames[b];
        Fmax = std::max(Fmax, n_frames[b]);
    }
    check_caller_codes(m_->codec, codes, n_frames, batch, max_frames);
    int32_t* dcodes = nullptr;
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&dcodes), size_t(batch) * max_frames * 16 * 4));
    Q3_HIP(hipMemcpy(dcodes, codes, size_t(batch) * max_frames * 16 * 4, hipMemcpyHostToDevice));
    float* pcm_dev = nullptr;
    hipStream_t cst = codec_stream(false);
    int32_t* nf = nullptr;  // pinned: rows whose waveform came out non-finite
    Q3_HIP(hipHostMalloc(reinterpret_cast<void**>(&nf), size_t(batch) * 4, hipHostMallocDefault));
    std::memset(nf, 0, size_t(batch) * 4);
    Q3_HIP(hipEventRecord(ev_[2], cst));
    try {
        if (Fmax > 0) codec_->decode(dcodes, max_frames, frames, &pcm_dev, std::string(), nullptr, nullptr, nullptr, nf);
    } catch (...) {
        (void)hipFree(dcodes);
        (void)hipHostFree(nf);
        throw;
    }
    Q3_HIP(hipEventRecord(ev_[3], cst));
    Q3_HIP(hipStreamSynchronize(cst));
    {
        bool bad = false;
        for (int b = 0; b < batch; ++b) bad = bad || nf[b] != 0;
        if (bad && !codec_->fp32_convs()) {
            // an activation left the fp16 range of the default kernels: the whole call once more on the fp32 matrix cores (the
            // reference's range; redo_rows_fp32 does the same for rows of a generate call)
            std::memset(nf, 0, size_t(batch) * 4);
            try {
                codec_->decode(dcodes, max_frames, frames, &pcm_dev, std::string(), nullptr, nullptr, nullptr, nf, true);
                Q3_HIP(hipEventRecord(ev_[3], cst));
                Q3_HIP(hipStreamSynchronize(cst));
            } catch (...) {
                (void)hipFree(dcodes);
                (void)hipHostFree(nf);
                throw;
            }
            bad = false;
            for (int b = 0; b < batch; ++b) bad = bad || nf[b] != 0;
        }
        (void)hipHostFree(nf);
        if (bad) {
            (void)hipFree(dcodes);
            throw Error(4, kCodecRangeMsg);
        }
    }
    float ms = 0;
    Q3_HIP(hipEventElapsedTime(&ms, ev_[2], ev_[3]));
    timing.codec_ms = ms;
    for (int b = 0; b < batch; ++b) {
        const int F = frames[size_t(b)];
        if (F > 0)
            Q3_HIP(hipMemcpy(pcm + size_t(b) * max_frames * up, pcm_dev + size_t(b) * Fmax * up, size_t(F) * up * 4,
                             hipMemcpyDeviceToHost));
        int valid = 0;
        for (int f = 0; f < F; ++f) valid += codes[(size_t(b) * max_frames + f) * 16] > 0 ? 1 : 0;
        audio_lengths[b] = int64_t(valid) * up;  // SpeechTokenizer.swift:831-833
    }
    (void)hipFree(dcodes);
}

void Engine::codec_decode_streamed(const int32_t* codes, const int32_t* n_frames, int batch, int max_frames, int chunk_frames, int window,
                                   int lookahead, float* pcm) {
    Q3_CHECK(m_->has_codec, 1, "Model not initialized: Speech tokenizer not loaded");
    Q3_CHECK(batch >= 1 && max_frames >= 1 && chunk_frames >= 1, 3, "Invalid input: empty codec batch");
    const int up = codec_->upsample();
    int32_t* dcodes = nullptr;
    float* hpcm = nullptr;
    std::vector<hipEvent_t> done;
    auto cleanup = [&] {
        for (auto e : done) (void)hipEventDestroy(e);
        if (dcodes) (void)hipFree(dcodes);
        if (hpcm) (void)hipHostFree(hpcm);
    };
    try {
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&dcodes), size_t(batch) * max_frames * 16 * 4));
        Q3_HIP(hipMemcpy(dcodes, codes, size_t(batch) * max_frames * 16 * 4, hipMemcpyHostToDevice));
        Q3_HIP(hipHostMalloc(reinterpret_cast<void**>(&hpcm), size_t(batch) * max_frames * up * 4, hipHostMallocDefault));
        std::memset(hpcm, 0, size_t(batch) * max_frames * up * 4);
        std::vector<int> avail((size_t)(batch));
        std::vector<uint8_t> fin((size_t)(batch), 1);
        int Fmax = 0;
        for (int b = 0; b < batch; ++b) {
            Q3_CHECK(n_frames[b] >= 0 && n_frames[b] <= max_frames, 3, "Invalid input: n_frames out of range");
            avail[size_t(b)] = n_frames[b];
            Fmax = std::max(Fmax, n_frames[b]);
        }
        check_caller_codes(m_->codec, codes, n_frames, batch, max_frames);
        hipStream_t cst = codec_stream(false);
        CodecRunner::StreamCfg cfg;
        cfg.rows = batch; cfg.chunk_frames = chunk_frames; cfg.window = window; cfg.lookahead = lookahead; cfg.max_frames = std::max(Fmax, 1);
        codec_->stream_open(cfg);
        try {
            // as a stream would deliver them: frames become available chunk by chunk (window >= 0); all at once otherwise
            if (window >= 0) {
                std::vector<uint8_t> notyet((size_t)(batch), 0);
                for (int have = chunk_frames; have < Fmax + chunk_frames + lookahead; have += chunk_frames) {
                    std::vector<int> a((size_t)(batch));
                    for (int b = 0; b < batch; ++b) {
                        a[size_t(b)] = std::min(avail[size_t(b)], have);
                        notyet[size_t(b)] = a[size_t(b)] == avail[size_t(b)] ? 1 : 0;
                    }
                    codec_->stream_push(dcodes, max_frames, a.data(), notyet.data(), hpcm, size_t(max_frames) * up, done);
                }
            }
            codec_->stream_push(dcodes, max_frames, avail.data(), fin.data(), hpcm, size_t(max_frames) * up, done);
        } catch (...) {
            codec_->stream_close();
            throw;
        }
        codec_->stream_close();
        Q3_HIP(hipStreamSynchronize(cst));
        for (int b = 0; b < batch; ++b)
            std::memcpy(pcm + size_t(b) * max_frames * up, hpcm + size_t(b) * max_frames * up, size_t(n_frames[b]) * up * 4);
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
}

void Engine::debug_codec_stage(const int32_t* codes, int n_frames, const char* stage, float* out, int64_t cap, int* T, int* C) {
    Q3_CHECK(m_->has_codec, 1, "Model not initialized: Speech tokenizer not loaded");
    int32_t* dcodes = nullptr;
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&dcodes), size_t(n_frames) * 16 * 4));
    Q3_HIP(hipMemcpy(dcodes, codes, size_t(n_frames) * 16 * 4, hipMemcpyHostToDevice));
    std::vector<float> so;
    float* pcm_dev = nullptr;
    (void)codec_stream(false);
    try {
        codec_->decode(dcodes, n_frames, {n_frames}, &pcm_dev, stage, &so, T, C);
    } catch (...) {
        (void)hipFree(dcodes);
        throw;
    }
    (void)hipFree(dcodes);
    Q3_CHECK(int64_t(so.size()) <= cap, 3, "debug_codec_stage: output buffer too small");
    std::memcpy(out, so.data(), so.size() * 4);
}


// ------------------------------------------------------------------------------------------------
// EngineGroup
// ------------------------------------------------------------------------------------------------
EngineGroup::EngineGroup(std::unique_ptr<Model> model, const q3tts_load_opts& opts) : model_(std::move(model)), opts_(opts) {
    int lanes = opts.n_streams;
    // Measured: a lane's frame step takes ~4-5 ms whatever its batch size (latency-bound chain),
    // and n concurrent chains overlap by only 1.6x (n=2) / 2.2x (n=4), so splitting a batch into lanes never pays.
    if (lanes <= 0) lanes = 1;
    lanes = std::max(1, std::min(lanes, opts.max_batch));
    q3tts_load_opts lo = opts;
    lo.max_batch = ceil_div(opts.max_batch, lanes);
    for (int i = 0; i < lanes; ++i) {
        lanes_.push_back(std::make_unique<Engine>(model_.get(), lo));
        lanes_.back()->cb_mutex = &cb_mutex_;
        // keep the sum of queued kernel packets of all lanes well under the 16k-entry AQL queue (~650 nodes per frame)
        lanes_.back()->max_inflight_frames = std::max(2, 12000 / 650 / lanes);
    }
    speakers = lanes_[0]->speakers;
}

int EngineGroup::begin(const q3tts_request* reqs, int n, const q3tts_sampling& sp, q3tts_event_cb cb, void* user, bool more_follows) {
    Q3_CHECK(n >= 1 && n <= opts_.max_batch, 3, "Invalid input: batch size must be between 1 and max_batch");
    if (lanes_.size() == 1) {
        Engine& e = *lanes_[0];
        e.row_offset = 0;
        e.request_base = 0;
        return e.begin(reqs, n, sp, cb, user, nullptr, more_follows);
    }
    int slot = -1;
    for (int i = 0; i < Engine::kJobSlots; ++i)
        if (!parked_[i].busy) slot = i;
    Q3_CHECK(slot >= 0, 3, "Invalid input: two jobs are already outstanding (q3tts_generate_end must be called first)");
    parked_[slot].results.assign(size_t(n), q3tts_result{});
    generate(reqs, n, sp, cb, user, parked_[slot].results.data(), nullptr);
    parked_[slot].timing = timing;
    parked_[slot].busy = true;
    return slot;
}

void EngineGroup::end(int job, q3tts_result* results) {
    if (lanes_.size() == 1) {
        lanes_[0]->end(job, results);
        timing = lanes_[0]->timing;
        return;
    }
    Q3_CHECK(job >= 0 && job < Engine::kJobSlots && parked_[job].busy, 3, "Invalid input: no such outstanding job");
    std::copy(parked_[job].results.begin(), parked_[job].results.end(), results);  // buffers change owner
    timing = parked_[job].timing;
    parked_[job].results.clear();
    parked_[job].busy = false;
}

void EngineGroup::generate(const q3tts_request* reqs, int n, const q3tts_sampling& sp, q3tts_event_cb cb, void* user,
                           q3tts_result* results, const DebugOpts* dbg) {
    Q3_CHECK(n >= 1 && n <= opts_.max_batch, 3, "Invalid input: batch size must be between 1 and max_batch");
    const int L = int(lanes_.size());
    // contiguous split: lane i takes rows [lo_i, hi_i)
    std::vector<int> lo((size_t)(L + 1), 0);
    const int per = n / L, rem = n % L;
    for (int i = 0; i < L; ++i) lo[size_t(i) + 1] = lo[size_t(i)] + per + (i < rem ? 1 : 0);
    std::vector<std::string> errs((size_t)(L));
    std::vector<int> codes((size_t)(L), 0);
    auto run = [&](int i) {
        const int a = lo[size_t(i)], b = lo[size_t(i) + 1];
        if (b <= a) return;
        Engine& e = *lanes_[size_t(i)];
        e.row_offset = uint32_t(a);
        e.request_base = a;
        try {
            DebugOpts d;
            const DebugOpts* dp = nullptr;
            if (dbg) {  // slice the per-row debug arrays
                const TalkerConfig& t = model_->cfg.talker;
                d = *dbg;
                const size_t fr = size_t(dbg->frames);
                if (d.forced_codes) d.forced_codes += size_t(a) * fr * 16;
                if (d.sampled) d.sampled += size_t(a) * fr * 16;
                if (d.talker_logits) d.talker_logits += size_t(a) * fr * t.vocab_size;
                if (d.cp_logits) d.cp_logits += size_t(a) * fr * (t.num_code_groups - 1) * t.cp.vocab_size;
                dp = &d;
            }
            e.generate(reqs + a, b - a, sp, cb, user, results + a, dp);
        } catch (const Error& ex) {
            errs[size_t(i)] = ex.what();
            codes[size_t(i)] = ex.status;
        } catch (const std::exception& ex) {
            errs[size_t(i)] = ex.what();
            codes[size_t(i)] = 7;
        }
    };
    if (L == 1 || n == 1) {
        run(0);
        for (int i = 1; i < L; ++i) run(i);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < L; ++i) th.emplace_back(run, i);
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < L; ++i)
        if (codes[size_t(i)]) throw Error(codes[size_t(i)], errs[size_t(i)]);
    // aggregate timing: lanes run concurrently, so spans are maxima and volumes are sums
    timing = q3tts_timing{};
    for (int i = 0; i < L; ++i) {
        if (lo[size_t(i) + 1] <= lo[size_t(i)]) continue;
        const q3tts_timing& t = lanes_[size_t(i)]->timing;
        timing.prefill_ms = std::max(timing.prefill_ms, t.prefill_ms);
        timing.decode_ms = std::max(timing.decode_ms, t.decode_ms);
        timing.codec_ms = std::max(timing.codec_ms, t.codec_ms);
        timing.frontend_ms = std::max(timing.frontend_ms, t.frontend_ms);
        timing.first_audio_ms = std::max(timing.first_audio_ms, t.first_audio_ms);
        timing.frame_steps = std::max(timing.frame_steps, t.frame_steps);
        timing.launches_per_frame_step = std::max(timing.launches_per_frame_step, t.launches_per_frame_step);
        timing.rows += t.rows;
        timing.kv_bytes_read += t.kv_bytes_read;
    }
}

}  // namespace q3
