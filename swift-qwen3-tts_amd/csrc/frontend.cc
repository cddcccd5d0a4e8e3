// frontend.cc -- voice-clone front end pipelines. See frontend.h.
#include "frontend.h"

#include <algorithm>
#include <cmath>

#include "codec_kernels.h"

namespace q3 {

namespace {

// Output length of a causal StreamableConv1d (SpeechTokenizerEncoder.swift:114-118, 163-186), Float arithmetic as there.
int streamable_out_len(int T, int k, int stride, int dil) {
    const int eff = (k - 1) * dil + 1, ptotal = eff - stride;
    const float nframes = float(std::max(T + ptotal - eff, 0)) / float(stride) + 1.0f;
    const int ideal = (int(std::ceil(nframes)) - 1) * stride + eff - ptotal;
    const int extra = std::max(0, ideal - T);
    return (T + ptotal + extra - eff) / stride + 1;
}

struct Carver {  // bump allocator over the scratch block; first pass (base == nullptr) only sizes it
    uint8_t* base = nullptr;
    size_t off = 0;
    float* f32(size_t n) {
        off = align_up(off, 256);
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += n * sizeof(float);
        return p;
    }
};

}  // namespace

VoiceFrontEnd::VoiceFrontEnd(const Model& m, hipStream_t st) : m_(m), st_(st) {
    std::vector<int32_t> ones(size_t(kMaxClips), 1);
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&ones_dev_), ones.size() * 4));
    Q3_HIP(hipMemcpy(ones_dev_, ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
}

VoiceFrontEnd::~VoiceFrontEnd() {
    if (buf_) (void)hipFree(buf_);
    if (ones_dev_) (void)hipFree(ones_dev_);
}

void VoiceFrontEnd::ensure(size_t bytes) {
    if (bytes <= buf_bytes_) return;
    if (buf_) Q3_HIP(hipFree(buf_));
    buf_ = nullptr;
    buf_bytes_ = 0;
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&buf_), bytes));
    buf_bytes_ = bytes;
}

void VoiceFrontEnd::capture(StageCapture* cap, const char* name, const float* t, int T, int C, int ld) {
    if (!cap || cap->name != name) return;
    Q3_HIP(hipStreamSynchronize(st_));
    cap->data.resize(size_t(T) * C);
    Q3_HIP(hipMemcpy2D(cap->data.data(), size_t(C) * 4, t, size_t(ld) * 4, size_t(C) * 4, size_t(T), hipMemcpyDeviceToHost));
    cap->T = T;
    cap->C = C;
}

int VoiceFrontEnd::encoded_frames(int64_t n_samples) const {
    const CodecEncW& e = m_.codec_enc;
    int T = int(n_samples);
    for (auto& L : e.layers) T = streamable_out_len(T, 2 * L.ratio, L.ratio, 1);
    return streamable_out_len(T, 2 * e.ds, e.ds, 1);
}

// One conv_gemm launch over B utterances of T positions each (row b at x + b * x_bstride).
static void convB(const ConvW& w, const float* x, int ldx, int64_t x_bs, int T, int B, float* out, int ldo, int64_t out_bs,
                  hipStream_t st, const int32_t* ones, int act = 0, int pre_act = 0, const float* res = nullptr, int ldr = 0,
                  int64_t res_bs = 0, int shift = 0, int reflect = 0, const float* x2 = nullptr, int ldx2 = 0) {
    ConvGemmArgs a{};
    a.x = x; a.ldx = ldx; a.x_bstride = x_bs; a.w = w.w; a.bias = w.bias; a.scale = w.scale;
    a.res = res; a.ldr = ldr; a.res_bstride = res_bs;
    a.out = out; a.ldo = ldo; a.out_bstride = out_bs; a.frames = ones; a.ppf = T; a.Tmax = T; a.B = B;
    a.Cin = w.Cin; a.N = w.N; a.K = w.K; a.dil = w.dil; a.act = act; a.pre_act = pre_act;
    a.shift = shift; a.reflect = reflect; a.x2 = x2; a.ldx2 = ldx2;
    launch_conv_gemm(a, st);
}
static void conv1(const ConvW& w, const float* x, int ldx, int T, float* out, int ldo, hipStream_t st, const int32_t* one,
                  int act = 0, int pre_act = 0, const float* res = nullptr, int ldr = 0, int shift = 0, int reflect = 0,
                  const float* x2 = nullptr, int ldx2 = 0) {
    convB(w, x, ldx, 0, T, 1, out, ldo, 0, st, one, act, pre_act, res, ldr, 0, shift, reflect, x2, ldx2);
}

int VoiceFrontEnd::encode(const float* audio_dev, int64_t n_samples, int32_t* codes_dev, StageCapture* cap) {
    const int64_t off = 0;
    encode_batch(audio_dev, 1, n_samples, &n_samples, &off, codes_dev, cap);
    return encoded_frames(n_samples);
}

// The encoder is causal end to end (SEANet convs, transformer mask, downsample conv; the RVQ search is per frame), so
// clips of different lengths can share one pass: each is zero-padded on the right to the longest (the reference pads
// its own clip with zeros on the right as well, SpeechTokenizerEncoder.swift:184) and only its first valid_T[b] frames
// are searched and written. Every launch then covers all clips: the small weight-bound GEMMs of the transformer and the
// late SEANet stages stream their weights once instead of once per clip.
void VoiceFrontEnd::encode_batch(const float* audio_dev, int B, int64_t n_samples, const int64_t* clip_samples,
                                 const int64_t* code_off, int32_t* codes_dev, StageCapture* cap) {
    const CodecEncW& e = m_.codec_enc;
    Q3_CHECK(m_.has_codec_encoder, 1, "Model not initialized: Speech tokenizer encoder not available");  // Qwen3.swift:432-434
    Q3_CHECK(n_samples >= 1 && n_samples <= (int64_t(1) << 24), 3, "Invalid input: reference audio must hold 1 .. 2^24 samples");
    Q3_CHECK(B >= 1 && B <= kMaxClips, 3, "Invalid input: too many clips in one encoder pass");
    const int S = int(n_samples);
    // ---- lengths per stage
    std::vector<int> Ts{S};
    for (auto& L : e.layers) {
        const int Tn = streamable_out_len(Ts.back(), 2 * L.ratio, L.ratio, 1);
        Q3_CHECK(Tn == (Ts.back() + L.ratio - 1) / L.ratio, 7, "internal error: strided conv length");
        Ts.push_back(Tn);
    }
    const int Te = Ts.back();  // transformer positions
    const int Tq = streamable_out_len(Te, 2 * e.ds, e.ds, 1);
    Q3_CHECK(Tq == (Te + e.ds - 1) / e.ds, 7, "internal error: downsample length");
    Q3_CHECK(Te <= e.max_T, 3, "Invalid input: reference audio longer than the encoder's max_position_embeddings");
    const int H = e.hidden, I = e.tlayers.empty() ? H : e.tlayers[0].fc1.N;
    // ---- scratch: per-clip strides leave room for one padded row group behind the data
    size_t pp = 0;
    {
        int C = e.init_C;
        for (size_t i = 0; i < e.layers.size(); ++i) {
            pp = std::max(pp, (size_t(Ts[i]) + e.layers[i].ratio) * C);
            C = e.layers[i].down.N;
        }
        pp = std::max(pp, (size_t(Te) + e.ds) * std::max(C, H));
    }
    Carver cv;
    float *bufA, *bufB, *xn, *qkv, *ao, *h1, *proj;
    int32_t* valid_dev;  // [stages + 2][B]: each clip's own length at every stage
    int64_t* off_dev;
    const int NS = int(e.layers.size()) + 2;
    auto carve = [&]() {
        bufA = cv.f32(pp * B);
        bufB = cv.f32(pp * B);
        xn = cv.f32(size_t(Te) * H * B);
        qkv = cv.f32(size_t(Te) * 3 * H * B);
        ao = cv.f32(size_t(Te) * H * B);
        h1 = cv.f32(size_t(Te) * I * B);
        proj = cv.f32(size_t(Tq) * 2 * e.dim * B);
        valid_dev = reinterpret_cast<int32_t*>(cv.f32(size_t(B) * NS));
        off_dev = reinterpret_cast<int64_t*>(cv.f32(size_t(2) * B));
    };
    carve();
    ensure(cv.off);
    cv = Carver{buf_, 0};
    carve();
    std::vector<int32_t> valid(size_t(B) * NS);  // stage 0: samples; 1..4: after each SEANet layer; last: output frames
    for (int b = 0; b < B; ++b) {
        Q3_CHECK(clip_samples[b] >= 1 && clip_samples[b] <= n_samples, 7, "internal error: clip longer than the padded batch");
        int v = int(clip_samples[b]);
        valid[size_t(b)] = v;
        for (size_t i = 0; i < e.layers.size(); ++i) {
            v = streamable_out_len(v, 2 * e.layers[i].ratio, e.layers[i].ratio, 1);
            valid[(i + 1) * B + b] = v;
        }
        valid[size_t(NS - 1) * B + b] = streamable_out_len(v, 2 * e.ds, e.ds, 1);
    }
    Q3_HIP(hipMemcpyAsync(valid_dev, valid.data(), valid.size() * 4, hipMemcpyHostToDevice, st_));
    Q3_HIP(hipMemcpyAsync(off_dev, code_off, size_t(B) * 8, hipMemcpyHostToDevice, st_));
    Q3_HIP(hipStreamSynchronize(st_));  // caller memory
    const int64_t bs = int64_t(pp);     // clip stride of the two big buffers, in floats
    // rows behind each clip's own end, up to the whole stride group the next strided conv reads
    auto zero_tail = [&](float* base, int stage, int Tpad, int C) { launch_mask_tail(base, bs, valid_dev + size_t(stage) * B, Tpad, C, B, st_); };

    // ---- SEANet (SpeechTokenizerEncoder.swift:436-443)
    float *cur = bufA, *oth = bufB;
    launch_enc_init_conv(audio_dev, S, B, e.init_w, e.init_b, e.init_C, e.init_K, cur, bs, st_);
    capture(cap, "init_conv", cur, S, e.init_C, e.init_C);
    for (size_t i = 0; i < e.layers.size(); ++i) {
        const auto& L = e.layers[i];
        const int T = Ts[i], Tn = Ts[i + 1], C = L.C;
        convB(L.res1, cur, C, bs, T, B, oth, L.res1.N, bs, st_, ones_dev_, 0, /*ELU*/ 1);                 // :338-339, k3
        convB(L.res2, oth, L.res1.N, bs, T, B, cur, C, bs, st_, ones_dev_, 0, 1, /*skip*/ cur, C, bs);     // k1 + residual (:345)
        zero_tail(cur, int(i), Tn * L.ratio, C);  // right zero padding up to a whole stride group (:114-118, :184)
        convB(L.down, cur, L.ratio * C, bs, Tn, B, oth, L.down.N, bs, st_, ones_dev_, 0, 1);              // ELU, k=2r stride r (:389)
        std::swap(cur, oth);
        if (cap && cap->name == "layer" + std::to_string(i)) capture(cap, cap->name.c_str(), cur, Tn, L.down.N, L.down.N);
    }
    convB(e.final_conv, cur, e.final_conv.Cin, bs, Te, B, oth, H, bs, st_, ones_dev_, 0, 1);  // ELU + k3 (:441-442)
    std::swap(cur, oth);
    capture(cap, "seanet", cur, Te, H, H);

    // ---- causal transformer (:571-590), x lives in `cur` (clip stride bs); the other tensors are dense [B][Te][.]
    const int64_t ts = int64_t(Te);
    for (const auto& L : e.tlayers) {
        launch_layernorm_f32(cur, bs, L.ln1_w, L.ln1_b, 1e-5f, H, Te, B, xn, ts * H, st_);
        convB(L.qkv, xn, H, ts * H, Te, B, qkv, 3 * H, ts * 3 * H, st_, ones_dev_);
        launch_rope_qk_f32(qkv, e.heads, Te, B, e.rope_cos, e.rope_sin, st_);
        launch_attn_causal_f32(qkv, e.heads, Te, B, ao, st_);
        convB(L.o, ao, H, ts * H, Te, B, cur, H, bs, st_, ones_dev_, 0, 0, cur, H, bs);  // x + layer_scale_1 * o_proj(attn)
        launch_layernorm_f32(cur, bs, L.ln2_w, L.ln2_b, 1e-5f, H, Te, B, xn, ts * H, st_);
        convB(L.fc1, xn, H, ts * H, Te, B, h1, I, ts * I, st_, ones_dev_, /*gelu tanh*/ 2);
        convB(L.fc2, h1, I, ts * I, Te, B, cur, H, bs, st_, ones_dev_, 0, 0, cur, H, bs);  // x + layer_scale_2 * linear2(...)
    }
    capture(cap, "transformer", cur, Te, H, H);

    // ---- stride-ds conv (no bias) and the two input projections (:1049-1052, :870-875)
    zero_tail(cur, int(e.layers.size()), Tq * e.ds, H);
    convB(e.down, cur, e.ds * H, bs, Tq, B, oth, H, bs, st_, ones_dev_);
    capture(cap, "downsample", oth, Tq, H, H);
    const int64_t ps = int64_t(Tq) * 2 * e.dim;
    convB(e.rvq_in, oth, H, bs, Tq, B, proj, 2 * e.dim, ps, st_, ones_dev_);
    capture(cap, "rvq_first_in", proj, Tq, e.dim, 2 * e.dim);
    capture(cap, "rvq_rest_in", proj + e.dim, Tq, e.dim, 2 * e.dim);
    // ---- nearest-neighbour search: the semantic layer on its projection, the acoustic layers on theirs (:934-941)
    const int32_t* vq = valid_dev + size_t(NS - 1) * B;
    launch_rvq_encode(proj, 2 * e.dim, ps, Tq, B, vq, off_dev, e.dim, e.bins, e.cb_dev, e.c2_dev, 1, 0, codes_dev, st_);
    if (e.n_layers > 1)
        launch_rvq_encode(proj + e.dim, 2 * e.dim, ps, Tq, B, vq, off_dev, e.dim, e.bins, e.cb_dev + 1, e.c2_dev + 1,
                          e.n_layers - 1, 1, codes_dev, st_);
}

void VoiceFrontEnd::speaker_embedding(const float* audio_dev, int64_t n_samples, float* emb_dev, StageCapture* cap) {
    const SpeakerEncW& s = m_.speaker;
    Q3_CHECK(m_.has_speaker_encoder, 1, "Model not initialized: Speaker encoder not available for this model");  // Qwen3.swift:227-229
    Q3_CHECK(n_samples >= 1 && n_samples <= (int64_t(1) << 24), 3, "Invalid input: reference audio must hold 1 .. 2^24 samples");
    const int S = int(n_samples);
    const int P = S + s.n_fft;                      // zero padding n_fft/2 on both sides (SpeakerEncoder.swift:430-431)
    const int T = (P - s.n_fft) / s.hop + 1;        // :469
    int max_pad = (s.b0.K - 1) * s.b0.dil / 2;
    for (auto& B : s.blocks) max_pad = std::max(max_pad, (B.res[0].K - 1) * B.res[0].dil / 2);
    Q3_CHECK(T > max_pad, 3, "Invalid input: reference audio too short for the speaker encoder");
    const int C = s.blocks[0].C, C4 = s.mfa.N, A = s.asp_tdnn.N, SE = s.blocks[0].se1.N;
    Carver cv;
    float *padded, *spec, *mel, *h0, *cat, *b1, *b2, *vec, *mfa, *att_in, *a1, *a2, *pooled;
    auto carve = [&]() {
        padded = cv.f32(size_t(P));
        spec = cv.f32(size_t(T) * s.dft.N);
        mel = cv.f32(size_t(T) * s.n_mels);
        h0 = cv.f32(size_t(T) * C);
        cat = cv.f32(size_t(T) * 3 * C);
        b1 = cv.f32(size_t(T) * C);
        b2 = cv.f32(size_t(T) * C);
        vec = cv.f32(size_t(4) * std::max({C, C4, SE}) + 64);
        mfa = cv.f32(size_t(T) * C4);
        att_in = cv.f32(size_t(T) * 3 * C4);
        a1 = cv.f32(size_t(T) * A);
        a2 = cv.f32(size_t(T) * C4);
        pooled = cv.f32(size_t(2) * C4);
    };
    carve();
    ensure(cv.off);
    cv = Carver{buf_, 0};
    carve();
    const int vstride = std::max({C, C4, SE});
    float *v_mean = vec, *v_s1 = vec + vstride, *v_se = vec + 2 * vstride, *v_std = vec + 3 * vstride;

    // ---- log-mel (SpeakerEncoder.swift:410-456): frames are overlapping rows of the padded signal (row stride = hop)
    Q3_HIP(hipMemsetAsync(padded, 0, size_t(P) * sizeof(float), st_));
    Q3_HIP(hipMemcpyAsync(padded + s.n_fft / 2, audio_dev, size_t(S) * sizeof(float), hipMemcpyDeviceToDevice, st_));
    conv1(s.dft, padded, s.hop, T, spec, s.dft.N, st_, ones_dev_);
    launch_log_mel(spec, s.dft.N, T, s.nfreq, s.mel_fb, s.n_mels, mel, st_);
    capture(cap, "mel", mel, T, s.n_mels, s.n_mels);

    // ---- ECAPA-TDNN (SpeakerEncoder.swift:364-394). TimeDelayNetBlock = reflect pad + conv + ReLU (:62-69)
    auto tdnn = [&](const ConvW& w, const float* x, int ldx, float* out, int ldo, const float* x2 = nullptr, int ldx2 = 0, int act = 3) {
        const int pad = (w.K - 1) * w.dil / 2;
        conv1(w, x, ldx, T, out, ldo, st_, ones_dev_, act, 0, nullptr, 0, pad, pad > 0 ? 1 : 0, x2, ldx2);
    };
    tdnn(s.b0, mel, s.n_mels, h0, C);
    capture(cap, "h0", h0, T, C, C);
    const float* hin = h0;
    int ld_in = C;
    for (int bi = 0; bi < 3; ++bi) {  // SqueezeExcitationRes2NetBlock (:204-211)
        const auto& B = s.blocks[bi];
        const int cs = C / s.scale;
        tdnn(B.tdnn1, hin, ld_in, b1, C);
        launch_copy2d_f32(b1, C, b2, C, T, cs, st_);  // Res2NetBlock chunk 0 passes through (:105-106)
        for (int i = 1; i < s.scale; ++i)             // chunk i: conv(chunk_i [+ previous output]) (:107-111)
            tdnn(B.res[size_t(i - 1)], b1 + i * cs, C, b2 + i * cs, C, i >= 2 ? b2 + (i - 1) * cs : nullptr, C);
        tdnn(B.tdnn2, b2, C, b1, C);
        // SqueezeExcitationBlock (:143-155): mean over time -> conv1 + ReLU -> conv2 + sigmoid
        launch_time_stats(b1, C, T, C, v_mean, nullptr, 0.f, st_);
        conv1(B.se1, v_mean, C, 1, v_s1, SE, st_, ones_dev_, 3);
        conv1(B.se2, v_s1, SE, 1, v_se, C, st_, ones_dev_, 4);
        float* hout = cat + bi * C;
        launch_scale_res(b1, C, v_se, hin, ld_in, hout, 3 * C, T, C, st_);  // x * se + residual (:154, :210)
        if (cap && cap->name == "h" + std::to_string(bi + 1)) capture(cap, cap->name.c_str(), hout, T, C, 3 * C);
        hin = hout;
        ld_in = 3 * C;
    }
    tdnn(s.mfa, cat, 3 * C, mfa, C4);  // concat of the three block outputs is the layout of `cat` (:379-380)
    capture(cap, "mfa", mfa, T, C4, C4);
    // ---- AttentiveStatisticsPooling (:238-272)
    launch_time_stats(mfa, C4, T, C4, v_mean, v_std, 1e-12f, st_);
    launch_asp_concat(mfa, v_mean, v_std, T, C4, att_in, st_);
    tdnn(s.asp_tdnn, att_in, 3 * C4, a1, A, nullptr, 0, /*tanh(relu)*/ 5);
    conv1(s.asp_conv, a1, A, T, a2, C4, st_, ones_dev_);
    launch_asp_pool(a2, mfa, T, C4, 1e-12f, pooled, st_);
    capture(cap, "pooled", pooled, 1, 2 * C4, 2 * C4);
    conv1(s.fc, pooled, 2 * C4, 1, emb_dev, s.enc_dim, st_, ones_dev_);  // :385-391
}

}  // namespace q3
