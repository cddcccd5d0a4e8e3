// codec.cc -- codec decoder pipeline (codes -> PCM). See codec.h.
#include "codec.h"

#include <algorithm>
#include <cstdlib>
#include <utility>

#include "codec_kernels.h"

namespace q3 {

namespace {
constexpr size_t kScratchBudget = size_t(24) << 30;  // activation scratch per chunk of rows (of 288 GB HBM)
}

CodecRunner::CodecRunner(const Model& m, hipStream_t st) : m_(m), st_(st) {
    up_ = m.cfg.codec.total_upsample();
    Q3_CHECK(m.cfg.codec.head_dim == 64, 6, "codec transformer head_dim must be 64");
    const char* e = std::getenv("Q3TTS_CODEC_FP32");
    fp32_mfma_ = e && e[0] == '1';
    const char* nf = std::getenv("Q3TTS_CODEC_NO_FUSE");
    no_fuse_ = nf && nf[0] == '1';
}

CodecRunner::~CodecRunner() {
    if (buf_) (void)hipFree(buf_);
    if (lens_dev_) (void)hipFree(lens_dev_);
}

void CodecRunner::ensure(size_t bytes) {
    if (bytes <= buf_bytes_) return;
    if (buf_) Q3_HIP(hipFree(buf_));
    buf_ = nullptr;
    buf_bytes_ = 0;
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&buf_), bytes));
    buf_bytes_ = bytes;
}

int CodecRunner::decode(const int32_t* codes_dev, int code_stride_frames, const std::vector<int>& frames, float** pcm_dev,
                        const std::string& stage, std::vector<float>* stage_out, int* stage_T, int* stage_C) {
    const CodecDecoderConfig& dc = m_.cfg.codec;
    const CodecW& w = m_.codec;
    const int B = int(frames.size());
    int Fmax = 0;
    for (int f : frames) Fmax = std::max(Fmax, f);
    Q3_CHECK(Fmax > 0, 3, "Invalid input: no frames to decode");
    // floats per frame of the largest intermediate tensor
    size_t per_frame = std::max<size_t>(size_t(2) * w.inner, size_t(dc.codebook_dim));
    per_frame = std::max(per_frame, size_t(3) * dc.num_attention_heads * 64);
    per_frame = std::max(per_frame, size_t(2) * dc.intermediate_size);
    {
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
    }
    const size_t pcm_floats = size_t(B) * Fmax * up_;
    int rows_per_chunk = int(std::max<size_t>(1, kScratchBudget / (4 * per_frame * Fmax * sizeof(float))));
    rows_per_chunk = std::min(rows_per_chunk, B);
    const size_t big = align_up(size_t(rows_per_chunk) * Fmax * per_frame * sizeof(float), 256);
    ensure(align_up(pcm_floats * sizeof(float), 256) + 4 * big);
    float* pcm = reinterpret_cast<float*>(buf_);
    float* bufs[4];
    for (int i = 0; i < 4; ++i) bufs[i] = reinterpret_cast<float*>(buf_ + align_up(pcm_floats * sizeof(float), 256) + size_t(i) * big);
    if (lens_cap_ < B) {
        if (lens_dev_) Q3_HIP(hipFree(lens_dev_));
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&lens_dev_), size_t(B) * 4));
        lens_cap_ = B;
    }
    Q3_HIP(hipMemcpyAsync(lens_dev_, frames.data(), size_t(B) * 4, hipMemcpyHostToDevice, st_));
    Q3_HIP(hipStreamSynchronize(st_));  // `frames` is caller memory

    auto capture = [&](const char* name, const float* t, int T, int C, int nb) {
        if (!stage_out || stage != name) return;
        Q3_HIP(hipStreamSynchronize(st_));
        stage_out->resize(size_t(nb) * T * C);
        Q3_HIP(hipMemcpy(stage_out->data(), t, stage_out->size() * 4, hipMemcpyDeviceToHost));
        if (stage_T) *stage_T = T;
        if (stage_C) *stage_C = C;
    };

    for (int r0 = 0; r0 < B; r0 += rows_per_chunk) {
        const int nb = std::min(rows_per_chunk, B - r0);
        const int32_t* fr = lens_dev_ + r0;
        const int32_t* codes = codes_dev + size_t(r0) * code_stride_frames * 16;
        // `post`: also (or, with out == nullptr, only) write SnakeBeta_post(result) to out2 for the next conv
        auto conv = [&](const ConvW& cw, const float* x, int Tmax, int ppf, float* out, const SnakeW* sn, const float* res,
                        int act, const SnakeW* post = nullptr, float* out2 = nullptr) {
            ConvGemmArgs a{};
            a.x = x; a.ldx = cw.Cin; a.x_bstride = int64_t(Tmax) * cw.Cin;
            a.w = cw.w; a.w3 = fp32_mfma_ ? nullptr : cw.w3; a.bias = cw.bias; a.scale = cw.scale;
            a.res = res; a.ldr = cw.N; a.res_bstride = int64_t(Tmax) * cw.N;
            a.out = out; a.ldo = cw.N; a.out_bstride = int64_t(Tmax) * cw.N;
            a.snake_ea = sn ? sn->ea : nullptr; a.snake_ib = sn ? sn->ib : nullptr;
            if (post) { a.out2 = out2; a.post_ea = post->ea; a.post_ib = post->ib; a.post_C = post->C; }
            a.frames = fr; a.ppf = ppf; a.Tmax = Tmax; a.B = nb;
            a.Cin = cw.Cin; a.N = cw.N; a.K = cw.K; a.dil = cw.dil; a.act = act;
            launch_conv_gemm(a, st_);
        };
        int T = Fmax, ppf = 1;
        // 1-2. Split-RVQ dequantisation (SpeechTokenizer.swift:214-226)
        launch_rvq_gather(codes, code_stride_frames, w.cb_first, w.cb_rest_dev, int(w.cb_rest.size()), w.inner, fr, Fmax, nb,
                          bufs[0], st_);
        conv(w.rvq_out, bufs[0], T, ppf, bufs[1], nullptr, nullptr, 0);
        capture("quantizer", bufs[1], T, w.rvq_out.N, nb);
        // 3. pre_conv (:759)
        conv(w.pre_conv, bufs[1], T, ppf, bufs[0], nullptr, nullptr, 0);
        capture("pre_conv", bufs[0], T, w.pre_conv.N, nb);
        // 4. pre_transformer (:629-643)
        {
            const int hid = dc.hidden_size, heads = dc.num_attention_heads, I = dc.intermediate_size;
            float *x = bufs[1], *t1 = bufs[2], *t2 = bufs[3];
            conv(w.t_in, bufs[0], T, ppf, x, nullptr, nullptr, 0);
            for (auto& L : w.tlayers) {
                launch_rmsnorm_f32(x, L.ln1, dc.rms_norm_eps, hid, fr, ppf, T, nb, t1, st_);
                conv(L.qkv, t1, T, ppf, t2, nullptr, nullptr, 0);
                launch_attn_full_f32(t2, heads, fr, T, nb, t1, st_);
                conv(L.o, t1, T, ppf, x, nullptr, x, 0);  // x = x + layer_scale * o_proj(attn)  (:589-592)
                launch_rmsnorm_f32(x, L.ln2, dc.rms_norm_eps, hid, fr, ppf, T, nb, t1, st_);
                conv(L.gateup, t1, T, ppf, t2, nullptr, nullptr, 0);
                launch_silu_mul_f32(t2, I, fr, ppf, T, nb, t1, st_);
                conv(L.down, t1, T, ppf, x, nullptr, x, 0);  // (:594-598)
            }
            launch_rmsnorm_f32(x, w.t_norm, dc.rms_norm_eps, hid, fr, ppf, T, nb, t1, st_);
            conv(w.t_out, t1, T, ppf, bufs[0], nullptr, nullptr, 0);
        }
        capture("pre_transformer", bufs[0], T, w.t_out.N, nb);
        int cur = 0;
        // 5. upsample stages: transposed conv (k = stride) + ConvNeXt (:767-775)
        for (size_t i = 0; i < w.ups.size(); ++i) {
            const auto& U = w.ups[i];
            const int C = U.tconv.N / U.stride;
            float *h = bufs[cur], *y = bufs[(cur + 1) & 3], *t1 = bufs[(cur + 2) & 3], *t2 = bufs[(cur + 3) & 3];
            conv(U.tconv, h, T, ppf, y, nullptr, nullptr, 0);  // [T][s*C] == [T*s][C]
            T *= U.stride;
            ppf *= U.stride;
            launch_dwconv_ln(y, U.dw_w, U.dw_b, U.ln_w, U.ln_b, 1e-6f, C, fr, ppf, T, nb, t1, st_);
            conv(U.pw1, t1, T, ppf, t2, nullptr, nullptr, 1);
            conv(U.pw2, t2, T, ppf, y, nullptr, y, 0);  // y = y + gamma * (pwconv2(...) + b)  (:396-400)
            cur = (cur + 1) & 3;
            capture(("upsample" + std::to_string(i)).c_str(), bufs[cur], T, C, nb);
        }
        // 6. MainDecoder (:681-690). Every SnakeBeta sits in front of a conv; it is evaluated in the epilogue of the
        // conv that PRODUCES the tensor (one sinf per element) and the activated copy is what the next conv stages.
        const size_t nblk = w.blocks.size();
        {
            float *y = bufs[(cur + 1) & 3], *ys = bufs[(cur + 2) & 3];
            conv(w.init_conv, bufs[cur], T, ppf, y, nullptr, nullptr, 0, nblk ? &w.blocks[0].snake : nullptr, ys);
            cur = (cur + 1) & 3;  // bufs[cur] = init_conv output, bufs[cur + 1] = snake_0 of it
            capture("init_conv", bufs[cur], T, w.init_conv.N, nb);
        }
        for (size_t i = 0; i < nblk; ++i) {
            const auto& Bk = w.blocks[i];
            // in: bufs[cur + 1] = snake_i(previous stage). y (raw residual stream), ya = act1(y) / next snake(y), t1 = act2(conv1)
            float *hs = bufs[(cur + 1) & 3], *y = bufs[(cur + 2) & 3], *ya = bufs[(cur + 3) & 3], *t1 = bufs[cur];
            const SnakeW* after = i + 1 < nblk ? &w.blocks[i + 1].snake : nullptr;
            bool fused = !fp32_mfma_ && !no_fuse_ && resunit_supported(Bk.Cout, Bk.res[0].conv1.K, 9);
            for (int j = 0; j < 3; ++j)
                fused = fused && Bk.res[j].conv1.w3 && Bk.res[j].conv2.w3p && Bk.res[j].conv1.N == Bk.Cout && Bk.res[j].conv2.K == 1 &&
                        resunit_supported(Bk.Cout, Bk.res[j].conv1.K, Bk.res[j].conv1.dil);
            if (fused) {
                // narrow blocks: each residual unit is one launch, y ping-pongs between two buffers (codec_conv.hip)
                conv(Bk.tconv, hs, T, ppf, y, nullptr, nullptr, 0);  // snake (already applied by the producer) -> transposed conv
                T *= Bk.stride;
                ppf *= Bk.stride;
                float *yin = y, *yout = t1;
                for (int j = 0; j < 3; ++j) {
                    ResUnitArgs r{};
                    r.y = yin; r.out = yout;
                    if (j == 2 && after) { r.out2 = hs; r.post_ea = after->ea; r.post_ib = after->ib; }
                    r.w1 = Bk.res[j].conv1.w3; r.b1 = Bk.res[j].conv1.bias; r.w2p = Bk.res[j].conv2.w3p; r.b2 = Bk.res[j].conv2.bias;
                    r.ea1 = Bk.res[j].act1.ea; r.ib1 = Bk.res[j].act1.ib; r.ea2 = Bk.res[j].act2.ea; r.ib2 = Bk.res[j].act2.ib;
                    r.frames = fr; r.ppf = ppf; r.Tmax = T; r.B = nb; r.C = Bk.Cout; r.K = Bk.res[j].conv1.K; r.dil = Bk.res[j].conv1.dil;
                    launch_resunit(r, st_);
                    std::swap(yin, yout);
                }
                // three units: the result sits in t1 = bufs[cur], its activated copy (if any) in hs = bufs[cur + 1]
                capture(("block" + std::to_string(i)).c_str(), bufs[cur], T, Bk.Cout, nb);
                continue;
            }
            conv(Bk.tconv, hs, T, ppf, y, nullptr, nullptr, 0, &Bk.res[0].act1, ya);  // snake -> transposed conv (:474-475)
            T *= Bk.stride;
            ppf *= Bk.stride;
            for (int j = 0; j < 3; ++j) {  // DecoderResidualUnit (:430-437): y += conv2(act2(conv1(act1(y))))
                conv(Bk.res[j].conv1, ya, T, ppf, nullptr, nullptr, nullptr, 0, &Bk.res[j].act2, t1);
                const SnakeW* next = j < 2 ? &Bk.res[j + 1].act1 : after;
                conv(Bk.res[j].conv2, t1, T, ppf, y, nullptr, y, 0, next, ya);
            }
            cur = (cur + 2) & 3;  // bufs[cur] = y, bufs[cur + 1] = next block's snake of it
            capture(("block" + std::to_string(i)).c_str(), bufs[cur], T, Bk.Cout, nb);
        }
        // 7. outSnake -> outConv -> clip (:687-688, :781)
        launch_out_conv(bufs[cur], w.out_C, w.out_snake.ea, w.out_snake.ib, w.out_w, w.out_b, fr, ppf, T, nb,
                        pcm + size_t(r0) * Fmax * up_, st_);
        Q3_CHECK(T == Fmax * up_, 7, "internal error: codec upsampling mismatch");
    }
    *pcm_dev = pcm;
    return Fmax;
}

}  // namespace q3
