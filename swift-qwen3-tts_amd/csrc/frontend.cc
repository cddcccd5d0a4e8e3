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
    const int32_t one = 1;
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&one_dev_), 4));
    Q3_HIP(hipMemcpy(one_dev_, &one, 4, hipMemcpyHostToDevice));
}

VoiceFrontEnd::~VoiceFrontEnd() {
    if (buf_) (void)hipFree(buf_);
    if (one_dev_) (void)hipFree(one_dev_);
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

// One conv_gemm launch over a single utterance of T positions.
static void conv1(const ConvW& w, const float* x, int ldx, int T, float* out, int ldo, hipStream_t st, const int32_t* one,
                  int act = 0, int pre_act = 0, const float* res = nullptr, int ldr = 0, int shift = 0, int reflect = 0,
                  const float* x2 = nullptr, int ldx2 = 0) {
    ConvGemmArgs a{};
    a.x = x; a.ldx = ldx; a.w = w.w; a.bias = w.bias; a.scale = w.scale; a.res = res; a.ldr = ldr;
    a.out = out; a.ldo = ldo; a.frames = one; a.ppf = T; a.Tmax = T; a.B = 1;
    a.Cin = w.Cin; a.N = w.N; a.K = w.K; a.dil = w.dil; a.act = act; a.pre_act = pre_act;
    a.shift = shift; a.reflect = reflect; a.x2 = x2; a.ldx2 = ldx2;
    launch_conv_gemm(a, st);
}

int VoiceFrontEnd::encode(const float* audio_dev, int64_t n_samples, int32_t* codes_dev, StageCapture* cap) {
    const CodecEncW& e = m_.codec_enc;
    Q3_CHECK(m_.has_codec_encoder, 1, "Model not initialized: Speech tokenizer encoder not available");  // Qwen3.swift:432-434
    Q3_CHECK(n_samples >= 1 && n_samples <= (int64_t(1) << 24), 3, "Invalid input: reference audio must hold 1 .. 2^24 samples");
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
    // ---- scratch
    size_t pp = 0;  // floats of the larger SEANet tensor, with room for one padded row group
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
    auto carve = [&]() {
        bufA = cv.f32(pp);
        bufB = cv.f32(pp);
        xn = cv.f32(size_t(Te) * H);
        qkv = cv.f32(size_t(Te) * 3 * H);
        ao = cv.f32(size_t(Te) * H);
        h1 = cv.f32(size_t(Te) * I);
        proj = cv.f32(size_t(Tq) * 2 * e.dim);
    };
    carve();
    ensure(cv.off);
    cv = Carver{buf_, 0};
    carve();

    // ---- SEANet (SpeechTokenizerEncoder.swift:436-443)
    float *cur = bufA, *oth = bufB;
    launch_enc_init_conv(audio_dev, S, e.init_w, e.init_b, e.init_C, e.init_K, cur, st_);
    capture(cap, "init_conv", cur, S, e.init_C, e.init_C);
    for (size_t i = 0; i < e.layers.size(); ++i) {
        const auto& L = e.layers[i];
        const int T = Ts[i], Tn = Ts[i + 1], C = L.C;
        conv1(L.res1, cur, C, T, oth, L.res1.N, st_, one_dev_, 0, /*ELU*/ 1);              // :338-339, k3
        conv1(L.res2, oth, L.res1.N, T, cur, C, st_, one_dev_, 0, 1, /*skip*/ cur, C);        // k1 + residual (:345)
        if (Tn * L.ratio > T)  // right zero padding up to a whole stride group (:114-118, :184)
            Q3_HIP(hipMemsetAsync(cur + size_t(T) * C, 0, size_t(Tn * L.ratio - T) * C * sizeof(float), st_));
        conv1(L.down, cur, L.ratio * C, Tn, oth, L.down.N, st_, one_dev_, 0, 1);              // ELU, k=2r stride r (:389)
        std::swap(cur, oth);
        if (cap && cap->name == "layer" + std::to_string(i)) capture(cap, cap->name.c_str(), cur, Tn, L.down.N, L.down.N);
    }
    conv1(e.final_conv, cur, e.final_conv.Cin, Te, oth, H, st_, one_dev_, 0, 1);  // ELU + k3 (:441-442)
    std::swap(cur, oth);
    capture(cap, "seanet", cur, Te, H, H);

    // ---- causal transformer (:571-590), x lives in `cur`
    for (const auto& L : e.tlayers) {
        launch_layernorm_f32(cur, L.ln1_w, L.ln1_b, 1e-5f, H, Te, xn, st_);
        conv1(L.qkv, xn, H, Te, qkv, 3 * H, st_, one_dev_);
        launch_rope_qk_f32(qkv, e.heads, Te, e.rope_cos, e.rope_sin, st_);
        launch_attn_causal_f32(qkv, e.heads, Te, ao, st_);
        conv1(L.o, ao, H, Te, cur, H, st_, one_dev_, 0, 0, cur, H);  // x + layer_scale_1 * o_proj(attn)
        launch_layernorm_f32(cur, L.ln2_w, L.ln2_b, 1e-5f, H, Te, xn, st_);
        conv1(L.fc1, xn, H, Te, h1, I, st_, one_dev_, /*gelu tanh*/ 2);
        conv1(L.fc2, h1, I, Te, cur, H, st_, one_dev_, 0, 0, cur, H);  // x + layer_scale_2 * linear2(...)
    }
    capture(cap, "transformer", cur, Te, H, H);

    // ---- stride-ds conv (no bias) and the two input projections (:1049-1052, :870-875)
    if (Tq * e.ds > Te) Q3_HIP(hipMemsetAsync(cur + size_t(Te) * H, 0, size_t(Tq * e.ds - Te) * H * sizeof(float), st_));
    conv1(e.down, cur, e.ds * H, Tq, oth, H, st_, one_dev_);
    capture(cap, "downsample", oth, Tq, H, H);
    conv1(e.rvq_in, oth, H, Tq, proj, 2 * e.dim, st_, one_dev_);
    capture(cap, "rvq_first_in", proj, Tq, e.dim, 2 * e.dim);
    capture(cap, "rvq_rest_in", proj + e.dim, Tq, e.dim, 2 * e.dim);
    // ---- nearest-neighbour search: the semantic layer on its projection, the acoustic layers on theirs (:934-941)
    launch_rvq_encode(proj, 2 * e.dim, Tq, e.dim, e.bins, e.cb_dev, e.c2_dev, 1, codes_dev, st_);
    if (e.n_layers > 1)
        launch_rvq_encode(proj + e.dim, 2 * e.dim, Tq, e.dim, e.bins, e.cb_dev + 1, e.c2_dev + 1, e.n_layers - 1,
                          codes_dev + Tq, st_);
    return Tq;
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
    conv1(s.dft, padded, s.hop, T, spec, s.dft.N, st_, one_dev_);
    launch_log_mel(spec, s.dft.N, T, s.nfreq, s.mel_fb, s.n_mels, mel, st_);
    capture(cap, "mel", mel, T, s.n_mels, s.n_mels);

    // ---- ECAPA-TDNN (SpeakerEncoder.swift:364-394). TimeDelayNetBlock = reflect pad + conv + ReLU (:62-69)
    auto tdnn = [&](const ConvW& w, const float* x, int ldx, float* out, int ldo, const float* x2 = nullptr, int ldx2 = 0, int act = 3) {
        const int pad = (w.K - 1) * w.dil / 2;
        conv1(w, x, ldx, T, out, ldo, st_, one_dev_, act, 0, nullptr, 0, pad, pad > 0 ? 1 : 0, x2, ldx2);
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
        conv1(B.se1, v_mean, C, 1, v_s1, SE, st_, one_dev_, 3);
        conv1(B.se2, v_s1, SE, 1, v_se, C, st_, one_dev_, 4);
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
    conv1(s.asp_conv, a1, A, T, a2, C4, st_, one_dev_);
    launch_asp_pool(a2, mfa, T, C4, 1e-12f, pooled, st_);
    capture(cap, "pooled", pooled, 1, 2 * C4, 2 * C4);
    conv1(s.fc, pooled, 2 * C4, 1, emb_dev, s.enc_dim, st_, one_dev_);  // :385-391
}

}  // namespace q3
