// codec.cc -- codec decoder pipeline (codes -> PCM). See codec.h.
#include "codec.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "codec_kernels.h"

namespace q3 {

namespace {
size_t kScratchBudget = size_t(24) << 30;  // activation scratch per group of rows (of 288 GB HBM); tests lower it
}
void CodecRunner::set_scratch_budget(size_t bytes) { kScratchBudget = bytes ? bytes : (size_t(24) << 30); }

CodecRunner::CodecRunner(const Model& m, hipStream_t st, bool fp32_convs) : m_(m), st_(st) {
    up_ = m.cfg.codec.total_upsample();
    Q3_CHECK(m.cfg.codec.head_dim == 64, 6, "codec transformer head_dim must be 64");
    const char* e = std::getenv("Q3TTS_CODEC_FP32");  // tests switch per model load without touching the load options
    fp32_mfma_ = fp32_convs || (e && e[0] == '1');
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&nf_dev_), size_t(kMaxRows) * 4));
    Q3_HIP(hipMemset(nf_dev_, 0, size_t(kMaxRows) * 4));
    const char* nf = std::getenv("Q3TTS_CODEC_NO_FUSE");
    no_fuse_ = nf && nf[0] == '1';
    const char* nh = std::getenv("Q3TTS_CODEC_NO_F16");  // float16 checkpoints through the up-cast two-plane path (comparisons)
    no_h1_ = nh && nh[0] == '1';
}

CodecRunner::~CodecRunner() {
    if (buf_) (void)hipFree(buf_);
    if (lens_dev_) (void)hipFree(lens_dev_);
    if (lens_host_) (void)hipHostFree(lens_host_);
    if (nf_dev_) (void)hipFree(nf_dev_);
    if (stream_.arena) (void)hipFree(stream_.arena);
    if (stream_.lens_host) (void)hipHostFree(stream_.lens_host);
    if (stream_.lens_dev) (void)hipFree(stream_.lens_dev);
}

void CodecRunner::ensure(size_t bytes) {
    if (bytes <= buf_bytes_) return;
    if (buf_) Q3_HIP(hipFree(buf_));
    buf_ = nullptr;
    buf_bytes_ = 0;
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&buf_), bytes));
    buf_bytes_ = bytes;
}

// floats per frame of the largest intermediate tensor
size_t CodecRunner::floats_per_frame() const {
    const CodecDecoderConfig& dc = m_.cfg.codec;
    const CodecW& w = m_.codec;
    size_t per_frame = std::max<size_t>(size_t(2) * w.inner, size_t(dc.codebook_dim));
    per_frame = std::max(per_frame, size_t(3) * dc.num_attention_heads * 64);
    per_frame = std::max(per_frame, size_t(2) * dc.intermediate_size);
    int ppf = 1;
    for (int r : dc.upsampling_ratios) {
        ppf *= r;
        per_frame = std::max(per_frame, size_t(ppf) * 4 * dc.latent_dim);
    }
    per_frame = std::max(per_frame, size_t(ppf) * dc.decoder_dim);
    int C = dc.decoder_dim;
    for (int r : dc.upsample_rates) {
        ppf *= r;
        C /= 2;
        per_frame = std::max(per_frame, size_t(ppf) * C);
    }
    return per_frame;
}

// Frames of left context after which the causal tail of the decoder (everything behind the pre-transformer:
// SpeechTokenizer.swift:767-781) no longer sees where its input began: every CausalConv1d looks (K - 1) * dilation
// positions back at its own rate (:298-301), a CausalTransposeConv1d with K = 2 * stride one input position (:346-351).
int CodecRunner::tail_context_frames() const {
    const CodecDecoderConfig& dc = m_.cfg.codec;
    double ctx = 0.0, rate = 1.0;
    for (int r : dc.upsampling_ratios) {
        rate *= r;          // K == stride: no overlap between input positions
        ctx += 6.0 / rate;  // ConvNeXt depthwise k7 (:372-377)
    }
    ctx += 6.0 / rate;      // initConv k7
    for (int r : dc.upsample_rates) {
        ctx += 1.0 / rate;  // transposed conv k = 2r, stride r: one earlier input position
        rate *= r;
        ctx += 6.0 * (1 + 3 + 9) / rate;  // three residual units, k7 with dilation 1, 3, 9
    }
    ctx += 6.0 / rate;      // outConv k7
    return int(ctx) + 2;
}

void CodecRunner::upload_lens(const int32_t* lens, int n) {
    if (lens_cap_ < n) {
        Q3_HIP(hipStreamSynchronize(st_));
        if (lens_dev_) Q3_HIP(hipFree(lens_dev_));
        if (lens_host_) Q3_HIP(hipHostFree(lens_host_));
        lens_dev_ = nullptr;
        lens_host_ = nullptr;
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&lens_dev_), size_t(n) * 4));
        Q3_HIP(hipHostMalloc(reinterpret_cast<void**>(&lens_host_), size_t(n) * 4, hipHostMallocDefault));
        lens_cap_ = n;
    }
    // the pinned staging copy may still feed the previous call's transfer
    Q3_HIP(hipStreamSynchronize(st_));
    std::memcpy(lens_host_, lens, size_t(n) * 4);
    Q3_HIP(hipMemcpyAsync(lens_dev_, lens_host_, size_t(n) * 4, hipMemcpyHostToDevice, st_));
}

// `post`: also (or, with out == nullptr, only) write SnakeBeta_post(result) to out2 for the next conv
void CodecRunner::conv(const Pass& ps, const ConvW& cw, const float* x, int Tmax, int ppf, float* out, const SnakeW* sn,
                       const float* res, int act, const SnakeW* post, float* out2) {
    if (stream_.dry) return;
    ConvGemmArgs a{};
    // streamed decode: Tmax counts the allocation's rows (history margin + chunk); row 0 of the chunk sits behind the margin
    const int64_t m_in = int64_t(ps.hist_frames) * ppf * cw.Cin, m_out = int64_t(ps.hist_frames) * ppf * cw.N;
    x += m_in;
    if (out) out += m_out;
    if (res) res += m_out;
    if (out2) out2 += m_out;
    a.hist = ps.hist_frames * ppf;
    a.x = x; a.ldx = cw.Cin; a.x_bstride = int64_t(Tmax) * cw.Cin;
    a.w = cw.w; a.bias = cw.bias; a.scale = cw.scale;
    if (!fp32_mfma_) { a.wh = cw.wh; a.wsc = cw.wsc; }
    a.res = res; a.ldr = cw.N; a.res_bstride = int64_t(Tmax) * cw.N;
    a.out = out; a.ldo = cw.N; a.out_bstride = int64_t(Tmax) * cw.N;
    a.snake_ea = sn ? sn->ea : nullptr; a.snake_ib = sn ? sn->ib : nullptr;
    if (post) { a.out2 = out2; a.post_ea = post->ea; a.post_ib = post->ib; a.post_C = post->C; }
    a.frames = ps.fr; a.ppf = ppf; a.Tmax = Tmax; a.B = ps.nb;
    a.Cin = cw.Cin; a.N = cw.N; a.K = cw.K; a.dil = cw.dil; a.act = act;
    launch_conv_gemm(a, st_);
}

void CodecRunner::capture(const Pass& ps, const char* name, const float* t, int T, int C) {
    if (!ps.stage_out || *ps.stage != name) return;
    Q3_HIP(hipStreamSynchronize(st_));
    ps.stage_out->resize(size_t(ps.nb) * T * C);
    Q3_HIP(hipMemcpy(ps.stage_out->data(), t, ps.stage_out->size() * 4, hipMemcpyDeviceToHost));
    if (ps.stage_T) *ps.stage_T = T;
    if (ps.stage_C) *ps.stage_C = C;
}

// Steps 1-4 (SpeechTokenizer.swift:757-765): split-RVQ dequantisation, pre_conv, pre_transformer over ALL frames of a row
// (its attention has neither mask nor positions, :512-528). Result: bufs[0] = [nb][Fmax][latent].
void CodecRunner::run_front(const Pass& ps, const int32_t* codes, int code_stride_frames, int Fmax, float* const* bufs) {
    const CodecDecoderConfig& dc = m_.cfg.codec;
    const CodecW& w = m_.codec;
    const int nb = ps.nb;
    const int32_t* fr = ps.fr;
    int T = Fmax, ppf = 1;
    // 1-2. Split-RVQ dequantisation (SpeechTokenizer.swift:214-226)
    launch_rvq_gather(codes, code_stride_frames, w.cb_first, w.cb_rest_dev, int(w.cb_rest.size()), w.inner, fr, Fmax, nb,
                      bufs[0], w.cb_first_rows, w.cb_rest_rows, st_);
    conv(ps, w.rvq_out, bufs[0], T, ppf, bufs[1], nullptr, nullptr, 0);
    capture(ps, "quantizer", bufs[1], T, w.rvq_out.N);
    // 3. pre_conv (:759)
    conv(ps, w.pre_conv, bufs[1], T, ppf, bufs[0], nullptr, nullptr, 0);
    capture(ps, "pre_conv", bufs[0], T, w.pre_conv.N);
    // 4. pre_transformer (:629-643)
    {
        const int hid = dc.hidden_size, heads = dc.num_attention_heads, I = dc.intermediate_size;
        float *x = bufs[1], *t1 = bufs[2], *t2 = bufs[3];
        conv(ps, w.t_in, bufs[0], T, ppf, x, nullptr, nullptr, 0);
        for (auto& L : w.tlayers) {
            launch_rmsnorm_f32(x, L.ln1, dc.rms_norm_eps, hid, fr, ppf, T, nb, t1, st_);
            conv(ps, L.qkv, t1, T, ppf, t2, nullptr, nullptr, 0);
            launch_attn_full_f32(t2, heads, fr, T, nb, t1, st_);
            conv(ps, L.o, t1, T, ppf, x, nullptr, x, 0);  // x = x + layer_scale * o_proj(attn)  (:589-592)
            launch_rmsnorm_f32(x, L.ln2, dc.rms_norm_eps, hid, fr, ppf, T, nb, t1, st_);
            conv(ps, L.gateup, t1, T, ppf, t2, nullptr, nullptr, 0);
            launch_silu_mul_f32(t2, I, fr, ppf, T, nb, t1, st_);
            conv(ps, L.down, t1, T, ppf, x, nullptr, x, 0);  // (:594-598)
        }
        launch_rmsnorm_f32(x, w.t_norm, dc.rms_norm_eps, hid, fr, ppf, T, nb, t1, st_);
        conv(ps, w.t_out, t1, T, ppf, bufs[0], nullptr, nullptr, 0);
    }
    capture(ps, "pre_transformer", bufs[0], T, w.t_out.N);
}

// Steps 5-7 (:767-781): the causal tail. In: bufs[0] = [nb][T][latent] with fr[b] valid frames per row; out: pcm
// [nb][T * upsample] (row stride T * upsample).
void CodecRunner::run_tail(const Pass& ps, int Tframes, float* const* bufs, float* pcm) {
    const CodecDecoderConfig& dc = m_.cfg.codec;
    const CodecW& w = m_.codec;
    const int nb = ps.nb;
    const int32_t* fr = ps.fr;
    int T = Tframes, ppf = 1;
    int cur = 0;
    // 5. upsample stages: transposed conv (k = stride) + ConvNeXt (:767-775)
    for (size_t i = 0; i < w.ups.size(); ++i) {
        const auto& U = w.ups[i];
        const int C = U.tconv.N / U.stride;
        float *h = bufs[cur], *y = bufs[(cur + 1) & 3], *t1 = bufs[(cur + 2) & 3], *t2 = bufs[(cur + 3) & 3];
        conv(ps, U.tconv, h, T, ppf, y, nullptr, nullptr, 0);  // [T][s*C] == [T*s][C]
        T *= U.stride;
        ppf *= U.stride;
        launch_dwconv_ln(y, U.dw_w, U.dw_b, U.ln_w, U.ln_b, 1e-6f, C, fr, ppf, T, nb, t1, st_);
        conv(ps, U.pw1, t1, T, ppf, t2, nullptr, nullptr, 1);
        conv(ps, U.pw2, t2, T, ppf, y, nullptr, y, 0);  // y = y + gamma * (pwconv2(...) + b)  (:396-400)
        cur = (cur + 1) & 3;
        capture(ps, ("upsample" + std::to_string(i)).c_str(), bufs[cur], T, C);
    }
    // 6. MainDecoder (:681-690). Every SnakeBeta sits in front of a conv; it is evaluated in the epilogue of the
    // conv that PRODUCES the tensor (one sinf per element) and the activated copy is what the next conv stages.
    const size_t nblk = w.blocks.size();
    if (w.f16_main && !fp32_mfma_ && !no_h1_) {  // a float16 speech tokenizer: the reference computes this part in float16
        run_main_h1(ps, T, ppf, cur, bufs, pcm);
        return;
    }
    {
        float *y = bufs[(cur + 1) & 3], *ys = bufs[(cur + 2) & 3];
        conv(ps, w.init_conv, bufs[cur], T, ppf, y, nullptr, nullptr, 0, nblk ? &w.blocks[0].snake : nullptr, ys);
        cur = (cur + 1) & 3;  // bufs[cur] = init_conv output, bufs[cur + 1] = snake_0 of it
        capture(ps, "init_conv", bufs[cur], T, w.init_conv.N);
    }
    for (size_t i = 0; i < nblk; ++i) {
        const auto& Bk = w.blocks[i];
        // in: bufs[cur + 1] = snake_i(previous stage). y (raw residual stream), ya = act1(y) / next snake(y), t1 = act2(conv1)
        float *hs = bufs[(cur + 1) & 3], *y = bufs[(cur + 2) & 3], *ya = bufs[(cur + 3) & 3], *t1 = bufs[cur];
        const SnakeW* after = i + 1 < nblk ? &w.blocks[i + 1].snake : nullptr;
        bool fused = !fp32_mfma_ && !no_fuse_ && resunit_supported(Bk.Cout, Bk.res[0].conv1.K, 9);
        for (int j = 0; j < 3; ++j)
            fused = fused && Bk.res[j].conv1.wh && Bk.res[j].conv2.whp &&
                    Bk.res[j].conv1.N == Bk.Cout && Bk.res[j].conv2.K == 1 &&
                    resunit_supported(Bk.Cout, Bk.res[j].conv1.K, Bk.res[j].conv1.dil);
        if (fused) {
            // narrow blocks: each residual unit is one launch, y ping-pongs between two buffers (codec_conv.hip)
            conv(ps, Bk.tconv, hs, T, ppf, y, nullptr, nullptr, 0);  // snake (already applied by the producer) -> transposed conv
            T *= Bk.stride;
            ppf *= Bk.stride;
            float *yin = y, *yout = t1;
            for (int j = 0; j < 3; ++j) {
                ResUnitArgs r{};
                r.y = yin; r.out = yout;
                if (j == 2 && after) { r.out2 = hs; r.post_ea = after->ea; r.post_ib = after->ib; }
                r.b1 = Bk.res[j].conv1.bias; r.b2 = Bk.res[j].conv2.bias;
                r.w1h = Bk.res[j].conv1.wh; r.w2ph = Bk.res[j].conv2.whp; r.wsc1 = Bk.res[j].conv1.wsc; r.wsc2 = Bk.res[j].conv2.wsc;
                r.ea1 = Bk.res[j].act1.ea; r.ib1 = Bk.res[j].act1.ib; r.ea2 = Bk.res[j].act2.ea; r.ib2 = Bk.res[j].act2.ib;
                r.frames = fr; r.ppf = ppf; r.Tmax = T; r.B = nb; r.C = Bk.Cout; r.K = Bk.res[j].conv1.K; r.dil = Bk.res[j].conv1.dil;
                launch_resunit(r, st_);
                std::swap(yin, yout);
            }
            // three units: the result sits in t1 = bufs[cur], its activated copy (if any) in hs = bufs[cur + 1]
            capture(ps, ("block" + std::to_string(i)).c_str(), bufs[cur], T, Bk.Cout);
            continue;
        }
        conv(ps, Bk.tconv, hs, T, ppf, y, nullptr, nullptr, 0, &Bk.res[0].act1, ya);  // snake -> transposed conv (:474-475)
        T *= Bk.stride;
        ppf *= Bk.stride;
        for (int j = 0; j < 3; ++j) {  // DecoderResidualUnit (:430-437): y += conv2(act2(conv1(act1(y))))
            conv(ps, Bk.res[j].conv1, ya, T, ppf, nullptr, nullptr, nullptr, 0, &Bk.res[j].act2, t1);
            const SnakeW* next = j < 2 ? &Bk.res[j + 1].act1 : after;
            conv(ps, Bk.res[j].conv2, t1, T, ppf, y, nullptr, y, 0, next, ya);
        }
        cur = (cur + 2) & 3;  // bufs[cur] = y, bufs[cur + 1] = next block's snake of it
        capture(ps, ("block" + std::to_string(i)).c_str(), bufs[cur], T, Bk.Cout);
    }
    // 7. outSnake -> outConv -> clip (:687-688, :781)
    launch_out_conv(bufs[cur], w.out_C, w.out_snake.ea, w.out_snake.ib, w.out_w, w.out_b, fr, ppf, T, nb,
                    pcm, st_, 0, nf_dev_ + ps.row0);
    Q3_CHECK(T == Tframes * up_, 7, "internal error: codec upsampling mismatch");
    (void)dc;
}

// ---- float16 speech tokenizers ("lite" checkpoints, docs/paper.tex:207): the MainDecoder as the reference computes it ----
void CodecRunner::conv_h1(const Pass& ps, const ConvW& cw, const void* x, bool x_f32, int Tmax, int ppf, uint16_t* out, const uint16_t* res,
                          const SnakeW* post, uint16_t* out2) {
    if (stream_.dry) return;
    Q3_CHECK(cw.w1 != nullptr, 7, "internal error: float16 decoder without its one-plane weights");
    ConvH1Args a{};
    // streamed decode (as conv()): Tmax counts the allocation's rows, row 0 of the chunk sits behind the history margin
    const int64_t m_in = int64_t(ps.hist_frames) * ppf * cw.Cin, m_out = int64_t(ps.hist_frames) * ppf * cw.N;
    x = x_f32 ? static_cast<const void*>(static_cast<const float*>(x) + m_in) : static_cast<const void*>(static_cast<const uint16_t*>(x) + m_in);
    if (out) out += m_out;
    if (res) res += m_out;
    if (out2) out2 += m_out;
    a.hist = ps.hist_frames * ppf;
    a.x = x; a.x_f32 = x_f32 ? 1 : 0; a.ldx = cw.Cin; a.x_bstride = int64_t(Tmax) * cw.Cin;
    a.w1 = cw.w1; a.bias = cw.bias;
    a.res = res; a.ldr = cw.N; a.res_bstride = int64_t(Tmax) * cw.N;
    a.out = out; a.ldo = cw.N; a.out_bstride = int64_t(Tmax) * cw.N;
    if (post) { a.out2 = out2; a.post_ea = post->ea16; a.post_ib = post->ib16; a.post_C = post->C; }
    a.frames = ps.fr; a.ppf = ppf; a.Tmax = Tmax; a.B = ps.nb;
    a.Cin = cw.Cin; a.N = cw.N; a.K = cw.K; a.dil = cw.dil;
    launch_conv_gemm_h1(a, st_);
}

void CodecRunner::capture_h(const Pass& ps, const char* name, const uint16_t* t, int T, int C) {
    if (!ps.stage_out || *ps.stage != name) return;
    Q3_HIP(hipStreamSynchronize(st_));
    std::vector<uint16_t> h(size_t(ps.nb) * T * C);
    Q3_HIP(hipMemcpy(h.data(), t, h.size() * 2, hipMemcpyDeviceToHost));
    ps.stage_out->resize(h.size());
    for (size_t i = 0; i < h.size(); ++i) {
        _Float16 v;
        std::memcpy(&v, &h[i], 2);
        (*ps.stage_out)[i] = float(v);
    }
    if (ps.stage_T) *ps.stage_T = T;
    if (ps.stage_C) *ps.stage_C = C;
}

// initConv -> four DecoderBlocks -> outSnake -> outConv -> clip (SpeechTokenizer.swift:681-690, 781) on float16 tensors. The launch
// structure is the unfused two-plane path's (run_tail): every SnakeBeta is evaluated in the epilogue of the conv that produces
// the tensor; y is the residual stream, ya / t1 / hs the activated copies the next conv reads.
void CodecRunner::run_main_h1(const Pass& ps, int T, int ppf, int cur, float* const* bufs, float* pcm) {
    const CodecW& w = m_.codec;
    const size_t nblk = w.blocks.size();
    auto H = [&](int i) { return reinterpret_cast<uint16_t*>(bufs[i & 3]); };
    {
        uint16_t *y = H(cur + 1), *ys = H(cur + 2);
        conv_h1(ps, w.init_conv, bufs[cur], true, T, ppf, y, nullptr, nblk ? &w.blocks[0].snake : nullptr, ys);
        cur = (cur + 1) & 3;  // H(cur) = initConv output, H(cur + 1) = snake_0 of it
        capture_h(ps, "init_conv", H(cur), T, w.init_conv.N);
    }
    for (size_t i = 0; i < nblk; ++i) {
        const auto& Bk = w.blocks[i];
        uint16_t *hs = H(cur + 1), *y = H(cur + 2), *ya = H(cur + 3), *t1 = H(cur);
        const SnakeW* after = i + 1 < nblk ? &w.blocks[i + 1].snake : nullptr;
        bool fused = !no_fuse_ && resunit_h1_supported(Bk.Cout, Bk.res[0].conv1.K, 9);
        for (int j = 0; j < 3; ++j)
            fused = fused && Bk.res[j].conv1.w1 && Bk.res[j].conv2.w1p && Bk.res[j].conv1.N == Bk.Cout && Bk.res[j].conv2.K == 1 &&
                    resunit_h1_supported(Bk.Cout, Bk.res[j].conv1.K, Bk.res[j].conv1.dil);
        if (fused) {
            // narrow blocks: each residual unit is one launch (resunit_h1_kernel), y ping-pongs between two buffers
            conv_h1(ps, Bk.tconv, hs, false, T, ppf, y, nullptr, nullptr, nullptr);  // snake (applied by the producer) -> transposed conv
            T *= Bk.stride;
            ppf *= Bk.stride;
            uint16_t *yin = y, *yout = t1;
            for (int j = 0; j < 3; ++j) {
                ResUnitH1Args r{};
                r.y = yin; r.out = yout;
                if (j == 2 && after) { r.out2 = hs; r.post_ea = after->ea16; r.post_ib = after->ib16; }
                r.b1 = Bk.res[j].conv1.bias; r.b2 = Bk.res[j].conv2.bias;
                r.w1 = Bk.res[j].conv1.w1; r.w2p = Bk.res[j].conv2.w1p;
                r.ea1 = Bk.res[j].act1.ea16; r.ib1 = Bk.res[j].act1.ib16; r.ea2 = Bk.res[j].act2.ea16; r.ib2 = Bk.res[j].act2.ib16;
                r.frames = ps.fr; r.ppf = ppf; r.Tmax = T; r.B = ps.nb; r.C = Bk.Cout; r.dil = Bk.res[j].conv1.dil;
                launch_resunit_h1(r, st_);
                std::swap(yin, yout);
            }
            // three units: the result sits in t1 = H(cur), its activated copy (if any) in hs = H(cur + 1)
            capture_h(ps, ("block" + std::to_string(i)).c_str(), H(cur), T, Bk.Cout);
            continue;
        }
        conv_h1(ps, Bk.tconv, hs, false, T, ppf, y, nullptr, &Bk.res[0].act1, ya);  // snake -> transposed conv (:474-475)
        T *= Bk.stride;
        ppf *= Bk.stride;
        for (int j = 0; j < 3; ++j) {  // DecoderResidualUnit (:430-437): y += conv2(act2(conv1(act1(y))))
            conv_h1(ps, Bk.res[j].conv1, ya, false, T, ppf, nullptr, nullptr, &Bk.res[j].act2, t1);
            const SnakeW* next = j < 2 ? &Bk.res[j + 1].act1 : after;
            conv_h1(ps, Bk.res[j].conv2, t1, false, T, ppf, y, y, next, ya);  // (the last unit leaves the NEXT block's snake in ya)
        }
        cur = (cur + 2) & 3;  // H(cur) = y, H(cur + 1) = the next block's snake of it
        capture_h(ps, ("block" + std::to_string(i)).c_str(), H(cur), T, Bk.Cout);
    }
    launch_out_conv_h1(H(cur), w.out_C, w.out_snake.ea16, w.out_snake.ib16, w.out_w, w.out_b, ps.fr, ppf, T, ps.nb, pcm, st_, nf_dev_ + ps.row0);
    Q3_CHECK(ppf == up_, 7, "internal error: codec upsampling mismatch");
}

// run_main_h1 over one chunk of a stream (run_tail_stream's rules: every tensor in its own persistent buffer of hist + chunk frames
// per row, float16 here, the ones a k7 / transposed conv reads back carrying the previous chunk's last frames in their margin).
// in: h32 = the last ConvNeXt stage's fp32 output (stream layout, history kept by the caller's sbuf).
void CodecRunner::run_main_h1_stream(const Pass& ps, const float* h32, int T, int ppf, float* pcm) {
    const CodecW& w = m_.codec;
    const Stream& S = stream_;
    const int H = ps.hist_frames;
    const size_t nblk = w.blocks.size();
    auto hb = [&](size_t frame_halves, bool keeps) { return reinterpret_cast<uint16_t*>(sbuf(frame_halves / 2, keeps)); };
    uint16_t* h = nullptr;   // the residual stream
    uint16_t* ys = nullptr;  // SnakeBeta of the previous stage's output = the next transposed conv's input
    size_t fh = size_t(ppf) * w.init_conv.N;  // float16 elements per frame of the current tensor
    Q3_CHECK(fh % 2 == 0, 7, "internal error: odd float16 frame size in a streamed decode");
    {
        uint16_t* y = hb(fh, nblk == 0);
        ys = nblk ? hb(fh, true) : nullptr;  // transposed conv: one row back
        conv_h1(ps, w.init_conv, h32, true, T, ppf, y, nullptr, nblk ? &w.blocks[0].snake : nullptr, ys);
        h = y;
    }
    for (size_t i = 0; i < nblk; ++i) {
        const auto& Bk = w.blocks[i];
        const SnakeW* after = i + 1 < nblk ? &w.blocks[i + 1].snake : nullptr;
        const bool lastb = i + 1 == nblk;
        bool fused = !no_fuse_ && resunit_h1_supported(Bk.Cout, Bk.res[0].conv1.K, 9);
        for (int j = 0; j < 3; ++j)
            fused = fused && Bk.res[j].conv1.w1 && Bk.res[j].conv2.w1p && Bk.res[j].conv1.N == Bk.Cout && Bk.res[j].conv2.K == 1 &&
                    resunit_h1_supported(Bk.Cout, Bk.res[j].conv1.K, Bk.res[j].conv1.dil);
        fh = size_t(ppf) * Bk.stride * Bk.Cout;
        Q3_CHECK(fh % 2 == 0, 7, "internal error: odd float16 frame size in a streamed decode");
        uint16_t* hs_next = after ? hb(fh, true) : nullptr;
        if (fused) {
            uint16_t* yb[4];
            for (int j = 0; j < 3; ++j) yb[j] = hb(fh, true);   // inputs of the three units (k7, dilated)
            yb[3] = hb(fh, lastb);                               // block output; the last one feeds outConv (k7)
            conv_h1(ps, Bk.tconv, ys, false, T, ppf, yb[0], nullptr, nullptr, nullptr);
            T *= Bk.stride;
            ppf *= Bk.stride;
            for (int j = 0; j < 3; ++j) {
                ResUnitH1Args r{};
                r.y = yb[j] + size_t(H) * fh; r.out = yb[j + 1] + size_t(H) * fh;
                if (j == 2 && after) { r.out2 = hs_next + size_t(H) * fh; r.post_ea = after->ea16; r.post_ib = after->ib16; }
                r.b1 = Bk.res[j].conv1.bias; r.b2 = Bk.res[j].conv2.bias;
                r.w1 = Bk.res[j].conv1.w1; r.w2p = Bk.res[j].conv2.w1p;
                r.ea1 = Bk.res[j].act1.ea16; r.ib1 = Bk.res[j].act1.ib16; r.ea2 = Bk.res[j].act2.ea16; r.ib2 = Bk.res[j].act2.ib16;
                r.frames = ps.fr; r.ppf = ppf; r.Tmax = T; r.B = ps.nb; r.C = Bk.Cout; r.dil = Bk.res[j].conv1.dil;
                r.hist = H * ppf;
                if (!S.dry) launch_resunit_h1(r, st_);
            }
            h = yb[3];
        } else {
            uint16_t* y = hb(fh, lastb);
            uint16_t* ya[3];
            for (int j = 0; j < 3; ++j) ya[j] = hb(fh, true);  // act1_j(y): conv1_j's input (k7, dilated)
            uint16_t* t1 = hb(fh, false);
            conv_h1(ps, Bk.tconv, ys, false, T, ppf, y, nullptr, &Bk.res[0].act1, ya[0]);
            T *= Bk.stride;
            ppf *= Bk.stride;
            for (int j = 0; j < 3; ++j) {
                conv_h1(ps, Bk.res[j].conv1, ya[j], false, T, ppf, nullptr, nullptr, &Bk.res[j].act2, t1);
                const SnakeW* next = j < 2 ? &Bk.res[j + 1].act1 : after;
                conv_h1(ps, Bk.res[j].conv2, t1, false, T, ppf, y, y, next, j < 2 ? ya[j + 1] : hs_next);
            }
            h = y;
        }
        ys = hs_next;
    }
    if (!S.dry)
        launch_out_conv_h1(h + size_t(H) * fh, w.out_C, w.out_snake.ea16, w.out_snake.ib16, w.out_w, w.out_b, ps.fr, ppf, T, ps.nb,
                           pcm + size_t(H) * ppf, st_, nf_dev_, H * ppf);
    Q3_CHECK(ppf == up_, 7, "internal error: codec upsampling mismatch");
}

int CodecRunner::decode(const int32_t* codes_dev, int code_stride_frames, const std::vector<int>& frames, float** pcm_dev,
                        const std::string& stage, std::vector<float>* stage_out, int* stage_T, int* stage_C, int32_t* nonfinite_host,
                        bool force_fp32) {
    struct Fp32Scope {  // the override ends with the call, however it ends
        bool& flag;
        bool old;
        ~Fp32Scope() { flag = old; }
    } fp32_scope{fp32_mfma_, fp32_mfma_};
    if (force_fp32) fp32_mfma_ = true;
    const int B = int(frames.size());
    Q3_CHECK(B <= kMaxRows, 3, "Invalid input: too many rows in one codec decode");
    Q3_HIP(hipMemsetAsync(nf_dev_, 0, size_t(B) * 4, st_));
    int Fmax = 0;
    for (int f : frames) Fmax = std::max(Fmax, f);
    Q3_CHECK(Fmax > 0, 3, "Invalid input: no frames to decode");
    const size_t per_frame = floats_per_frame();
    const size_t pcm_floats = size_t(B) * Fmax * up_;
    int rows_per_chunk = int(std::max<size_t>(1, kScratchBudget / (4 * per_frame * Fmax * sizeof(float))));
    rows_per_chunk = std::min(rows_per_chunk, B);
    const size_t big = align_up(size_t(rows_per_chunk) * Fmax * per_frame * sizeof(float), 256);
    ensure(align_up(pcm_floats * sizeof(float), 256) + 4 * big);
    float* pcm = reinterpret_cast<float*>(buf_);
    float* bufs[4];
    for (int i = 0; i < 4; ++i) bufs[i] = reinterpret_cast<float*>(buf_ + align_up(pcm_floats * sizeof(float), 256) + size_t(i) * big);
    upload_lens(frames.data(), B);
    for (int r0 = 0; r0 < B; r0 += rows_per_chunk) {
        Pass ps{};
        ps.nb = std::min(rows_per_chunk, B - r0);
        ps.row0 = r0;
        ps.fr = lens_dev_ + r0;
        ps.stage = &stage; ps.stage_out = stage_out; ps.stage_T = stage_T; ps.stage_C = stage_C;
        run_front(ps, codes_dev + size_t(r0) * code_stride_frames * 16, code_stride_frames, Fmax, bufs);
        run_tail(ps, Fmax, bufs, pcm + size_t(r0) * Fmax * up_);
    }
    if (nonfinite_host) Q3_HIP(hipMemcpyAsync(nonfinite_host, nf_dev_, size_t(B) * 4, hipMemcpyDeviceToHost, st_));
    *pcm_dev = pcm;
    return Fmax;
}

// f1: the decode in pieces. The pre-transformer runs once over every frame (it is bidirectional); everything behind it is
// causal, so the tail is evaluated chunk by chunk: chunk [f0, f1) takes the pre-transformer frames [f0 - H, f1) with
// H = tail_context_frames() and keeps the last (f1 - f0) * upsample samples. Every kept sample sees exactly the inputs it
// sees in the one-shot decode, through the same kernels and summation order: the PCM is bit-identical
// (tests/test_streaming.py). Each chunk's samples are copied to pcm_host ([B][Fmax * upsample], pinned) at their final
// place and chunk_done[k] is recorded behind the copy.
int CodecRunner::decode_chunked(const int32_t* codes_dev, int code_stride_frames, const std::vector<int>& frames, int chunk_frames,
                                float* pcm_host, std::vector<hipEvent_t>& chunk_done, int32_t* nonfinite_host, int32_t* nf_chunks_host) {
    const CodecDecoderConfig& dc = m_.cfg.codec;
    const int B = int(frames.size());
    int Fmax = 0;
    for (int f : frames) Fmax = std::max(Fmax, f);
    Q3_CHECK(Fmax > 0 && chunk_frames > 0, 3, "Invalid input: no frames to decode");
    Q3_CHECK(B <= kMaxRows, 3, "Invalid input: too many rows in one codec decode");
    Q3_HIP(hipMemsetAsync(nf_dev_, 0, size_t(B) * 4, st_));
    const int H = tail_context_frames();
    const int n_chunks = ceil_div(Fmax, chunk_frames);
    const int Tc = std::min(Fmax, chunk_frames + H);  // frames per tail pass, at most
    const size_t per_frame = floats_per_frame();
    const size_t front_bytes = align_up(size_t(B) * Fmax * dc.latent_dim * sizeof(float), 256);
    // front: four buffers of [B][Fmax] x (widest front tensor); tail: four of [B][Tc] x per_frame; x_all keeps the front's result
    size_t front_pf = std::max<size_t>(size_t(2) * m_.codec.inner, size_t(dc.codebook_dim));
    front_pf = std::max(front_pf, size_t(3) * dc.num_attention_heads * 64);
    front_pf = std::max(front_pf, size_t(2) * dc.intermediate_size);
    front_pf = std::max(front_pf, size_t(dc.latent_dim));
    // rows per pass: like decode(), a batch whose activations exceed the scratch budget goes through in groups of rows (the
    // codes of a finished AR loop must never be lost to a scratch limit); x_all always holds every row
    const size_t per_row = std::max(size_t(Fmax) * front_pf, size_t(Tc) * per_frame) * sizeof(float);
    const int G = int(std::min<size_t>(size_t(B), std::max<size_t>(1, kScratchBudget / (4 * per_row))));
    const size_t big = align_up(size_t(G) * per_row, 256);
    const size_t pcm_bytes = align_up(size_t(G) * Tc * up_ * sizeof(float), 256);
    ensure(pcm_bytes + front_bytes + 4 * big);
    float* pcm = reinterpret_cast<float*>(buf_);
    float* x_all = reinterpret_cast<float*>(buf_ + pcm_bytes);
    float* bufs[4];
    for (int i = 0; i < 4; ++i) bufs[i] = reinterpret_cast<float*>(buf_ + pcm_bytes + front_bytes + size_t(i) * big);
    // row lengths: the whole rows for the front, then per chunk the frames of [h0, f1) each row still has
    std::vector<int32_t> lens((size_t)(B) * (1 + n_chunks));
    for (int b = 0; b < B; ++b) lens[size_t(b)] = frames[size_t(b)];
    for (int k = 0; k < n_chunks; ++k) {
        const int f0 = k * chunk_frames, f1 = std::min(Fmax, f0 + chunk_frames), h0 = std::max(0, f0 - H);
        for (int b = 0; b < B; ++b) lens[size_t(1 + k) * B + b] = std::max(0, std::min(frames[size_t(b)], f1) - h0);
    }
    upload_lens(lens.data(), int(lens.size()));
    while (int(chunk_done.size()) < n_chunks) {
        hipEvent_t e = nullptr;
        Q3_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        chunk_done.push_back(e);
    }
    const std::string none;
    const size_t row_in = size_t(dc.latent_dim) * sizeof(float);
    for (int r0 = 0; r0 < B; r0 += G) {
        Pass ps{};
        ps.nb = std::min(G, B - r0); ps.row0 = r0; ps.fr = lens_dev_ + r0; ps.stage = &none;
        run_front(ps, codes_dev + size_t(r0) * code_stride_frames * 16, code_stride_frames, Fmax, bufs);
        Q3_HIP(hipMemcpyAsync(x_all + size_t(r0) * Fmax * dc.latent_dim, bufs[0], size_t(ps.nb) * Fmax * row_in, hipMemcpyDeviceToDevice, st_));
    }
    for (int k = 0; k < n_chunks; ++k) {
        const int f0 = k * chunk_frames, f1 = std::min(Fmax, f0 + chunk_frames), h0 = std::max(0, f0 - H), T = f1 - h0;
        for (int r0 = 0; r0 < B; r0 += G) {
            Pass ps{};
            ps.nb = std::min(G, B - r0); ps.row0 = r0; ps.fr = lens_dev_ + size_t(1 + k) * B + r0; ps.stage = &none;
            Q3_HIP(hipMemcpy2DAsync(bufs[0], size_t(T) * row_in, x_all + (size_t(r0) * Fmax + h0) * dc.latent_dim, size_t(Fmax) * row_in,
                                    size_t(T) * row_in, size_t(ps.nb), hipMemcpyDeviceToDevice, st_));
            run_tail(ps, T, bufs, pcm);
            Q3_HIP(hipMemcpy2DAsync(pcm_host + (size_t(r0) * Fmax + f0) * up_, size_t(Fmax) * up_ * sizeof(float), pcm + size_t(f0 - h0) * up_,
                                    size_t(T) * up_ * sizeof(float), size_t(f1 - f0) * up_ * sizeof(float), size_t(ps.nb),
                                    hipMemcpyDeviceToHost, st_));
        }
        if (nf_chunks_host) Q3_HIP(hipMemcpyAsync(nf_chunks_host + size_t(k) * B, nf_dev_, size_t(B) * 4, hipMemcpyDeviceToHost, st_));
        Q3_HIP(hipEventRecord(chunk_done[size_t(k)], st_));
    }
    if (nonfinite_host) Q3_HIP(hipMemcpyAsync(nonfinite_host, nf_dev_, size_t(B) * 4, hipMemcpyDeviceToHost, st_));
    return n_chunks;
}

}  // namespace q3

// ---------------------------------------------------------------------------------------------------------------------
// streamed decode (row f1)
// ---------------------------------------------------------------------------------------------------------------------
namespace q3 {

// Frames of history a tail tensor must carry so that every causal conv finds its halo in it: (K - 1) * dilation rows at the
// tensor's rate, rounded up to whole frames (SpeechTokenizer.swift:298-301: left padding only).
int CodecRunner::hist_frames() const {
    const CodecDecoderConfig& dc = m_.cfg.codec;
    int need = 1, ppf = 1;
    for (int r : dc.upsampling_ratios) {
        ppf *= r;
        need = std::max(need, ceil_div(6, ppf));  // ConvNeXt depthwise k7
    }
    need = std::max(need, ceil_div(6, ppf));      // initConv k7
    for (int r : dc.upsample_rates) {
        need = std::max(need, 1);                 // transposed conv: one earlier input row
        ppf *= r;
        need = std::max(need, ceil_div(6 * 9, ppf));  // residual units, k7 with dilation up to 9
    }
    need = std::max(need, ceil_div(6, ppf));      // outConv k7
    return need;
}

float* CodecRunner::sbuf(size_t frame_floats, bool keeps_history) {
    Stream& S = stream_;
    S.off = align_up(S.off, 256);
    float* p = S.dry ? nullptr : reinterpret_cast<float*>(S.arena + S.off);
    S.off += size_t(S.cfg.rows) * S.Tal * frame_floats * sizeof(float);
    if (keeps_history && !S.dry) S.rolls.emplace_back(p, frame_floats);
    return p;
}

void CodecRunner::stream_open(const StreamCfg& cfg) {
    Stream& S = stream_;
    Q3_CHECK(!S.open, 3, "Invalid input: a streamed decode is already open on this model");
    const CodecDecoderConfig& dc = m_.cfg.codec;
    S.cfg = cfg;
    S.hist = hist_frames();
    Q3_CHECK(cfg.rows >= 1 && cfg.max_frames >= 1 && cfg.lookahead >= 0, 3, "Invalid input: streamed decode geometry");
    Q3_CHECK(cfg.chunk_frames >= S.hist, 3, "Invalid input: audio_chunk_frames of a streamed decode must be at least " + std::to_string(S.hist));
    S.Tal = S.hist + cfg.chunk_frames;
    S.next_chunk = 0;
    S.front_done = false;
    S.lens_used = 0;
    const int n_chunks = ceil_div(cfg.max_frames, cfg.chunk_frames);
    // front scratch: the widest front tensor over the longest window
    const int Fwin = cfg.window < 0 ? cfg.max_frames : std::min(cfg.max_frames, cfg.window + cfg.chunk_frames + cfg.lookahead);
    size_t front_pf = std::max<size_t>(size_t(2) * m_.codec.inner, size_t(dc.codebook_dim));
    front_pf = std::max(front_pf, size_t(3) * dc.num_attention_heads * 64);
    front_pf = std::max(front_pf, size_t(2) * dc.intermediate_size);
    front_pf = std::max(front_pf, size_t(dc.latent_dim));
    S.fbuf_floats = size_t(cfg.rows) * Fwin * front_pf;
    // layout pass (no launches: the walk below and run_tail_stream only count), then one allocation
    auto take = [&](size_t floats) {
        S.off = align_up(S.off, 256);
        float* p = S.arena ? reinterpret_cast<float*>(S.arena + S.off) : nullptr;
        S.off += floats * sizeof(float);
        return p;
    };
    // Whatever leaves this function -- the arena allocation failing, a check inside run_tail_stream -- the counting mode
    // ends with it: conv() launches nothing while `dry` is set, and a runner left in that state would turn every later
    // decode of the model into launches of the glue kernels alone (finite garbage, status OK).
    struct DryOff {
        Stream& s;
        ~DryOff() {
            s.dry = false;
            if (!s.open) s.rolls.clear();
        }
    } dry_off{S};
    S.dry = true;
    S.off = 0;
    for (int i = 0; i < 4; ++i) (void)take(S.fbuf_floats);
    if (cfg.window < 0) (void)take(size_t(cfg.rows) * cfg.max_frames * dc.latent_dim);
    (void)sbuf(size_t(dc.latent_dim), false);
    (void)sbuf(size_t(up_), false);
    {
        Pass ps{};
        ps.nb = cfg.rows;
        ps.hist_frames = S.hist;
        run_tail_stream(ps, nullptr, nullptr);
    }
    const size_t need = align_up(S.off, 256);
    if (need > S.arena_bytes) {
        Q3_HIP(hipStreamSynchronize(st_));
        if (S.arena) Q3_HIP(hipFree(S.arena));
        S.arena = nullptr;
        S.arena_bytes = 0;
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&S.arena), need));
        S.arena_bytes = need;
    }
    S.dry = false;
    S.off = 0;
    S.rolls.clear();
    for (auto& f : S.fbufs) f = take(S.fbuf_floats);
    S.x_all = cfg.window < 0 ? take(size_t(cfg.rows) * cfg.max_frames * dc.latent_dim) : nullptr;
    S.lat = sbuf(size_t(dc.latent_dim), false);
    S.pcm = sbuf(size_t(up_), false);
    S.dry = false;
    // history margins start as zeros: the causal left padding of the first chunk
    Q3_HIP(hipMemsetAsync(S.arena, 0, need, st_));
    const size_t slots = size_t(2) * n_chunks + 2;
    if (slots * cfg.rows > S.lens_slots) {
        if (S.lens_host) Q3_HIP(hipHostFree(S.lens_host));
        if (S.lens_dev) Q3_HIP(hipFree(S.lens_dev));
        S.lens_host = nullptr;
        S.lens_dev = nullptr;
        Q3_HIP(hipHostMalloc(reinterpret_cast<void**>(&S.lens_host), slots * cfg.rows * 4, hipHostMallocDefault));
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&S.lens_dev), slots * cfg.rows * 4));
        S.lens_slots = slots * cfg.rows;
    }
    Q3_CHECK(cfg.rows <= kMaxRows, 3, "Invalid input: too many rows in one codec decode");
    Q3_HIP(hipMemsetAsync(nf_dev_, 0, size_t(cfg.rows) * 4, st_));
    S.open = true;
}

void CodecRunner::stream_close(int32_t* nonfinite_host) {
    if (stream_.open && nonfinite_host)
        Q3_HIP(hipMemcpyAsync(nonfinite_host, nf_dev_, size_t(stream_.cfg.rows) * 4, hipMemcpyDeviceToHost, st_));
    stream_.open = false;  // the arena stays for the next stream of the same shape
}

int CodecRunner::stream_push(const int32_t* codes_dev, int code_stride_frames, const int* avail, const uint8_t* final_rows, float* pcm_host,
                             size_t pcm_row_stride, std::vector<hipEvent_t>& chunk_done, int32_t* nf_chunks_host) {
    Stream& S = stream_;
    Q3_CHECK(S.open, 3, "Invalid input: no streamed decode is open");
    const CodecDecoderConfig& dc = m_.cfg.codec;
    const int B = S.cfg.rows, C = S.cfg.chunk_frames, W = S.cfg.window, L = S.cfg.lookahead;
    const size_t lat = size_t(dc.latent_dim);
    auto lens_slot = [&](const std::vector<int32_t>& v) {
        Q3_CHECK((S.lens_used + 1) * B <= S.lens_slots, 7, "internal error: streamed decode ran out of length slots");
        int32_t* h = S.lens_host + S.lens_used * B;
        int32_t* d = S.lens_dev + S.lens_used * B;
        std::memcpy(h, v.data(), size_t(B) * 4);
        Q3_HIP(hipMemcpyAsync(d, h, size_t(B) * 4, hipMemcpyHostToDevice, st_));
        ++S.lens_used;
        return d;
    };
    const std::string none;
    for (;;) {
        const int k = S.next_chunk, f0 = k * C, f1 = f0 + C;
        if (f0 >= S.cfg.max_frames) break;
        // decodable: every row either has its frames up to f1 + lookahead or will get no more; at least one row has a frame in it
        bool ready = true, any = false, all_final = true;
        int have = 0;
        for (int b = 0; b < B; ++b) {
            const bool fin = final_rows && final_rows[b];
            all_final = all_final && fin;
            if (!fin && avail[b] < std::min(S.cfg.max_frames, f1 + (W < 0 ? S.cfg.max_frames : L))) ready = false;
            any = any || avail[b] > f0;
            have = std::max(have, avail[b]);
        }
        if (!ready) break;
        if (!any) {
            if (all_final) break;  // nothing left anywhere
            break;
        }
        std::vector<int32_t> v((size_t)(B));
        Pass ps{};
        ps.nb = B;
        ps.stage = &none;
        // ---- pre-transformer over the window (or, window < 0, once over everything) ----
        int w0 = 0;
        if (W < 0) {
            if (!S.front_done) {
                for (int b = 0; b < B; ++b) v[size_t(b)] = avail[b];
                ps.fr = lens_slot(v);
                run_front(ps, codes_dev, code_stride_frames, S.cfg.max_frames, S.fbufs);
                Q3_HIP(hipMemcpyAsync(S.x_all, S.fbufs[0], size_t(B) * S.cfg.max_frames * lat * 4, hipMemcpyDeviceToDevice, st_));
                S.front_done = true;
            }
            Q3_HIP(hipMemcpy2DAsync(S.lat + size_t(S.hist) * lat, size_t(S.Tal) * lat * 4, S.x_all + size_t(f0) * lat,
                                    size_t(S.cfg.max_frames) * lat * 4, size_t(std::min(C, S.cfg.max_frames - f0)) * lat * 4, size_t(B),
                                    hipMemcpyDeviceToDevice, st_));
        } else {
            w0 = std::max(0, f0 - W);
            const int w1 = std::min(have, f1 + L), Fw = w1 - w0;
            for (int b = 0; b < B; ++b) v[size_t(b)] = std::max(0, std::min(avail[b], w1) - w0);
            ps.fr = lens_slot(v);
            run_front(ps, codes_dev + size_t(w0) * 16, code_stride_frames, Fw, S.fbufs);
            Q3_HIP(hipMemcpy2DAsync(S.lat + size_t(S.hist) * lat, size_t(S.Tal) * lat * 4, S.fbufs[0] + size_t(f0 - w0) * lat,
                                    size_t(Fw) * lat * 4, size_t(std::min(C, w1 - f0)) * lat * 4, size_t(B), hipMemcpyDeviceToDevice, st_));
        }
        // ---- the causal tail over the chunk, state carried in the tensors' margins ----
        for (int b = 0; b < B; ++b) v[size_t(b)] = std::max(0, std::min(avail[b], f1) - f0);
        ps.fr = lens_slot(v);
        ps.hist_frames = S.hist;
        S.off = 0;       // the same walk over the arena as in stream_open
        S.rolls.clear();
        {
            auto skip = [&](size_t floats) {
                S.off = align_up(S.off, 256);
                S.off += floats * sizeof(float);
            };
            for (int i = 0; i < 4; ++i) skip(S.fbuf_floats);
            if (W < 0) skip(size_t(B) * S.cfg.max_frames * lat);
            (void)sbuf(lat, false);
            (void)sbuf(size_t(up_), false);
        }
        Q3_HIP(hipMemsetAsync(S.pcm, 0, size_t(B) * S.Tal * up_ * 4, st_));
        run_tail_stream(ps, S.lat, S.pcm);
        for (auto& r : S.rolls)
            launch_roll_history(r.first + size_t(S.hist) * r.second, int64_t(S.Tal) * int64_t(r.second), int64_t(S.hist) * int64_t(r.second),
                                int64_t(C) * int64_t(r.second), B, st_);
        Q3_HIP(hipMemcpy2DAsync(pcm_host + size_t(f0) * up_, pcm_row_stride * sizeof(float), S.pcm + size_t(S.hist) * up_,
                                size_t(S.Tal) * up_ * sizeof(float), size_t(std::min(C, S.cfg.max_frames - f0)) * up_ * sizeof(float), size_t(B),
                                hipMemcpyDeviceToHost, st_));
        while (int(chunk_done.size()) <= k) {
            hipEvent_t e = nullptr;
            Q3_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            chunk_done.push_back(e);
        }
        if (nf_chunks_host) Q3_HIP(hipMemcpyAsync(nf_chunks_host + size_t(k) * B, nf_dev_, size_t(B) * 4, hipMemcpyDeviceToHost, st_));
        Q3_HIP(hipEventRecord(chunk_done[size_t(k)], st_));
        ++S.next_chunk;
    }
    return S.next_chunk;
}

// run_tail (steps 5-7, SpeechTokenizer.swift:767-781) over one chunk of a stream. The launches and their arguments are
// run_tail's; what differs is where the tensors live: each in its own persistent buffer of Tal = hist + chunk frames per row
// whose first `hist` frames are the previous chunk's last ones (rolled in by stream_push), so a conv's causal halo comes
// from memory instead of being zero (first chunk: the margins are zero, i.e. exactly the reference's left padding).
void CodecRunner::run_tail_stream(const Pass& ps, float* lat, float* pcm) {
    const CodecDecoderConfig& dc = m_.cfg.codec;
    const CodecW& w = m_.codec;
    const Stream& S = stream_;
    const int nb = ps.nb;
    const int32_t* fr = ps.fr;
    const int H = ps.hist_frames;
    int T = S.Tal, ppf = 1;  // T: rows of an allocation at the current rate
    float* h = lat;
    // 5. upsample stages
    for (size_t i = 0; i < w.ups.size(); ++i) {
        const auto& U = w.ups[i];
        const int C = U.tconv.N / U.stride;
        const size_t ff = size_t(ppf) * U.stride * C;  // floats per frame behind the transposed conv
        float* y = sbuf(ff, true);                     // dwconv reads six rows back
        float* t1 = sbuf(ff, false);
        float* t2 = sbuf(size_t(ppf) * U.stride * U.pw1.N, false);
        const bool last = i + 1 == w.ups.size();
        float* yo = sbuf(ff, last);                    // the last stage feeds initConv (k7)
        conv(ps, U.tconv, h, T, ppf, y, nullptr, nullptr, 0);
        T *= U.stride;
        ppf *= U.stride;
        if (!S.dry)
            launch_dwconv_ln(y + size_t(H) * ff, U.dw_w, U.dw_b, U.ln_w, U.ln_b, 1e-6f, C, fr, ppf, T, nb, t1 + size_t(H) * ff, st_, H * ppf);
        conv(ps, U.pw1, t1, T, ppf, t2, nullptr, nullptr, 1);
        conv(ps, U.pw2, t2, T, ppf, yo, nullptr, y, 0);  // out of place: y keeps the values dwconv needs as history
        h = yo;
    }
    // 6. MainDecoder
    const size_t nblk = w.blocks.size();
    if (w.f16_main && !fp32_mfma_ && !no_h1_) {  // a float16 speech tokenizer: float16 from initConv on, as in run_tail
        run_main_h1_stream(ps, h, T, ppf, pcm);
        return;
    }
    float* ys = nullptr;  // SnakeBeta of the previous stage's output = the next transposed conv's input
    {
        const size_t ff = size_t(ppf) * w.init_conv.N;
        float* y = sbuf(ff, nblk == 0);
        ys = nblk ? sbuf(ff, true) : nullptr;  // transposed conv: one row back
        conv(ps, w.init_conv, h, T, ppf, y, nullptr, nullptr, 0, nblk ? &w.blocks[0].snake : nullptr, ys);
        h = y;
    }
    for (size_t i = 0; i < nblk; ++i) {
        const auto& Bk = w.blocks[i];
        const SnakeW* after = i + 1 < nblk ? &w.blocks[i + 1].snake : nullptr;
        const bool lastb = i + 1 == nblk;
        bool fused = !fp32_mfma_ && !no_fuse_ && resunit_supported(Bk.Cout, Bk.res[0].conv1.K, 9);
        for (int j = 0; j < 3; ++j)
            fused = fused && Bk.res[j].conv1.wh && Bk.res[j].conv2.whp &&
                    Bk.res[j].conv1.N == Bk.Cout && Bk.res[j].conv2.K == 1 &&
                    resunit_supported(Bk.Cout, Bk.res[j].conv1.K, Bk.res[j].conv1.dil);
        const size_t ff = size_t(ppf) * Bk.stride * Bk.Cout;
        float* hs_next = after ? sbuf(ff, true) : nullptr;
        if (fused) {
            float* yb[4];
            for (int j = 0; j < 3; ++j) yb[j] = sbuf(ff, true);   // inputs of the three units (k7, dilated)
            yb[3] = sbuf(ff, lastb);                               // block output; the last one feeds outConv (k7)
            conv(ps, Bk.tconv, ys, T, ppf, yb[0], nullptr, nullptr, 0);
            T *= Bk.stride;
            ppf *= Bk.stride;
            for (int j = 0; j < 3; ++j) {
                ResUnitArgs r{};
                r.y = yb[j] + size_t(H) * ff; r.out = yb[j + 1] + size_t(H) * ff;
                if (j == 2 && after) { r.out2 = hs_next + size_t(H) * ff; r.post_ea = after->ea; r.post_ib = after->ib; }
                r.b1 = Bk.res[j].conv1.bias; r.b2 = Bk.res[j].conv2.bias;
                r.w1h = Bk.res[j].conv1.wh; r.w2ph = Bk.res[j].conv2.whp; r.wsc1 = Bk.res[j].conv1.wsc; r.wsc2 = Bk.res[j].conv2.wsc;
                r.ea1 = Bk.res[j].act1.ea; r.ib1 = Bk.res[j].act1.ib; r.ea2 = Bk.res[j].act2.ea; r.ib2 = Bk.res[j].act2.ib;
                r.frames = fr; r.ppf = ppf; r.Tmax = T; r.B = nb; r.C = Bk.Cout; r.K = Bk.res[j].conv1.K; r.dil = Bk.res[j].conv1.dil;
                r.hist = H * ppf;
                if (!S.dry) launch_resunit(r, st_);
            }
            h = yb[3];
        } else {
            float* y = sbuf(ff, lastb);
            float* ya[3];
            for (int j = 0; j < 3; ++j) ya[j] = sbuf(ff, true);  // act1_j(y): conv1_j's input (k7, dilated)
            float* t1 = sbuf(ff, false);
            conv(ps, Bk.tconv, ys, T, ppf, y, nullptr, nullptr, 0, &Bk.res[0].act1, ya[0]);
            T *= Bk.stride;
            ppf *= Bk.stride;
            for (int j = 0; j < 3; ++j) {
                conv(ps, Bk.res[j].conv1, ya[j], T, ppf, nullptr, nullptr, nullptr, 0, &Bk.res[j].act2, t1);
                const SnakeW* next = j < 2 ? &Bk.res[j + 1].act1 : after;
                conv(ps, Bk.res[j].conv2, t1, T, ppf, y, nullptr, y, 0, next, j < 2 ? ya[j + 1] : hs_next);
            }
            h = y;
        }
        ys = hs_next;
    }
    // 7. outSnake -> outConv -> clip
    if (!S.dry) {
        const size_t ff = size_t(ppf) * w.out_C;
        launch_out_conv(h + size_t(H) * ff, w.out_C, w.out_snake.ea, w.out_snake.ib, w.out_w, w.out_b, fr, ppf, T, nb,
                        pcm + size_t(H) * ppf, st_, H * ppf, nf_dev_);
    }
    Q3_CHECK(ppf == up_, 7, "internal error: codec upsampling mismatch");
    (void)dc;
}

}  // namespace q3
