// codec_kernels.h -- launchers of the codec-decoder kernels (kernels/codec_conv.hip, codec_misc.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace q3 {

// Rows are addressed as base + b*bstride + t*ld (channels-last). Row b has frames[b]*ppf valid
// positions; the grid covers Tmax positions and skips the rest.
struct ConvGemmArgs {
    const float* x;
    int ldx;
    int64_t x_bstride;
    const float* w;      // [N][K][Cin]
    const uint16_t* wh;  // optional: w * 2^s[n] split into two fp16 planes (hi | lo), [K][ceil(Cin/32)][N][2][32]; selects the
    const float* wsc;    // fp16x2 kernel (three fp16 MFMAs per product block); wsc[n] = 2^-s[n] rescales the accumulators
    const float* bias;   // [N] or nullptr
    const float* scale;  // [N] or nullptr
    const float* res;    // residual or nullptr
    int ldr;
    int64_t res_bstride;
    float* out;          // may be nullptr when only out2 is wanted
    int ldo;
    int64_t out_bstride;
    float* out2;         // optional second output, same geometry as out: SnakeBeta(post_ea, post_ib) of the result, for
    const float* post_ea;  // the conv that consumes it (its own snake_ea then stays nullptr). Parameters are indexed by
    const float* post_ib;  // n % post_C (a transposed conv's N is stride * channels)
    int post_C;
    const float* snake_ea;  // SnakeBeta prologue on x (exp(alpha), 1/(exp(beta)+1e-9)) or nullptr
    const float* snake_ib;
    const int32_t* frames;  // [B]
    int ppf;                // positions per frame at this stage
    int Tmax;
    int B, Cin, N, K, dil;
    int act;                // 0: none, 1: GELU (erf), 2: GELU (tanh form), 3: ReLU, 4: sigmoid, 5: tanh(ReLU)
    // voice-clone front end (SpeechTokenizerEncoder.swift, SpeakerEncoder.swift); all zero for the decoder
    int pre_act;            // 1: ELU(alpha 1) on the input while staging (SeanetResnetBlock :338, :389, :441)
    const float* x2;        // optional second input added to x before the conv (Res2NetBlock, SpeakerEncoder.swift:110)
    int ldx2;
    int shift;              // input row = t - (K-1)*dil + shift + tap*dil: "same" convs use shift = (K-1)*dil/2
    int reflect;            // 1: rows outside [0,T) are mirrored (reflectPad1d, SpeakerEncoder.swift:26-40), else zero
    int hist;               // streamed decode: rows -hist .. -1 in front of x are the previous chunk's last rows (else zero padding)
};
void launch_conv_gemm(const ConvGemmArgs& a, hipStream_t st);

// The MainDecoder's convs on a float16 speech tokenizer (kernels/codec_conv_h1.hip): float16 activations in HBM, one matrix-core
// product per block, one float16 rounding per op of the reference (conv, + bias, residual +, SnakeBeta's five).
struct ConvH1Args {
    const void* x;         // channels-last [b][t][Cin]: float16, or fp32 when x_f32 (rounded to float16 while staged)
    int x_f32;
    int ldx;               // elements
    int64_t x_bstride;     // elements
    const uint16_t* w1;    // [K][ceil(Cin/32)][N][32] fp16 (model.cc attach_h1)
    const float* bias;     // [N] (float16-exact values) or nullptr
    const uint16_t* res;   // float16 residual or nullptr
    int ldr;
    int64_t res_bstride;
    uint16_t* out;         // float16, may be nullptr when only out2 is wanted
    int ldo;
    int64_t out_bstride;
    uint16_t* out2;        // optional: SnakeBeta(post_ea, post_ib) of the result, same geometry as out
    const float* post_ea;  // SnakeW::ea16 / ib16, indexed by n % post_C
    const float* post_ib;
    int post_C;
    const int32_t* frames;
    int ppf, Tmax, B, Cin, N, K, dil;
    int hist;              // streamed decode: rows -hist .. -1 in front of x are the previous chunk's last rows (else zero padding)
};
void launch_conv_gemm_h1(const ConvH1Args& a, hipStream_t st);
// DecoderResidualUnit (SpeechTokenizer.swift:430-437) on float16 tensors in one launch (kernels/codec_conv_h1.hip resunit_h1_kernel):
// out = y + conv2(act2(conv1(act1(y)))), C = 96, conv1 seven taps. y is read once (+ halo), the sum written once.
struct ResUnitH1Args {
    const uint16_t* y;    // [B][Tmax][C] float16
    uint16_t* out;        // != y
    uint16_t* out2;       // optional: SnakeBeta(post_ea, post_ib) of the result (the next block's input) or nullptr
    const float* post_ea; // SnakeW::ea16 / ib16 arrays throughout
    const float* post_ib;
    const float* b1;      // [C] or nullptr
    const float* b2;
    const uint16_t* w1;   // conv1 [7][C/32][C][32] fp16 (attach_h1)
    const uint16_t* w2p;  // conv2 [C/32][C][32] fp16 in the fused k order (attach_h1_perm)
    const float* ea1;
    const float* ib1;
    const float* ea2;
    const float* ib2;
    const int32_t* frames;
    int ppf, Tmax, B, C, dil;
    int hist;             // as ConvH1Args::hist, for y
};
bool resunit_h1_supported(int C, int K, int dil);
void launch_resunit_h1(const ResUnitH1Args& a, hipStream_t st);
// SnakeBeta -> k7 conv C -> 1 -> clip on a float16 tensor (kernels/codec_conv_h1.hip)
void launch_out_conv_h1(const uint16_t* x, int C, const float* ea16, const float* ib16, const float* w, const float* bias,
                        const int32_t* frames, int ppf, int Tmax, int B, float* pcm, hipStream_t st, int32_t* nonfinite, int hist = 0);

// DecoderResidualUnit (SpeechTokenizer.swift:430-437) in one launch: out = y + conv2(act2(conv1(act1(y)))) with conv1
// k taps / dilation `dil`, conv2 pointwise, C channels on both (C = 32, 64 or 96). Neither act1(y) nor conv1's
// output touch HBM: y is read once (+ halo) and the sum written once, to a different buffer than y.
struct ResUnitArgs {
    const float* y;       // [B][Tmax][C]
    float* out;           // [B][Tmax][C], != y
    float* out2;          // optional: SnakeBeta(post_ea, post_ib) of the result (next block's input) or nullptr
    const float* post_ea;
    const float* post_ib;
    const float* b1;      // [C] or nullptr
    const float* b2;
    const uint16_t* w1h;  // conv1 / conv2 as two fp16 planes (model.cc attach_h2 / attach_h2_perm): [K][C/32][C][2][32] and, in the
    const uint16_t* w2ph; // fused kernel's k order, [C/32][C][2][32]
    const float* wsc1;    // [C] 2^-s of conv1's / conv2's rows
    const float* wsc2;
    const float* ea1;     // act1: exp(alpha), 1/(exp(beta)+1e-9)
    const float* ib1;
    const float* ea2;     // act2
    const float* ib2;
    const int32_t* frames;
    int ppf, Tmax, B, C, K, dil;
    int hist;             // as ConvGemmArgs::hist, for y
    int wdb;              // set by launch_resunit: conv1's weight tiles double-buffered in LDS
};
bool resunit_supported(int C, int K, int dil);
void launch_resunit(const ResUnitArgs& a, hipStream_t st);

// Split-RVQ gather (SpeechTokenizer.swift:214-226, 81-96): out[b][f] = [cb_first[c0] | sum_j cb_rest[j][c_{j+1}]]
void launch_rvq_gather(const int32_t* codes, int code_stride_frames, const float* cb_first,
                       const float* const* cb_rest, int n_rest, int inner, const int32_t* frames, int Fmax, int B,
                       float* out, int rows_first, int rows_rest, hipStream_t st);  // rows: a code is clamped into its table
// fp32 RMSNorm over the last dim (SpeechTokenizer.swift:581-582,626): out = (x*rstd)*w
void launch_rmsnorm_f32(const float* x, const float* w, float eps, int C, const int32_t* frames, int ppf, int Tmax,
                        int B, float* out, hipStream_t st);
// depthwise causal conv k7 + LayerNorm (ConvNeXtBlock, SpeechTokenizer.swift:389-393)
void launch_dwconv_ln(const float* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                      float eps, int C, const int32_t* frames, int ppf, int Tmax, int B, float* out, hipStream_t st, int hist = 0);
// out[t][i] = silu(gu[t][i]) * gu[t][I+i] (DecoderMLP, SpeechTokenizer.swift:560-562)
void launch_silu_mul_f32(const float* gu, int I, const int32_t* frames, int ppf, int Tmax, int B, float* out,
                         hipStream_t st);
// full bidirectional attention without positions or mask (SpeechTokenizer.swift:512-528).
// qkv [B][Tmax][3*heads*64] (q | k | v) -> out [B][Tmax][heads*64]
void launch_attn_full_f32(const float* qkv, int heads, const int32_t* frames, int Tmax, int B, float* out,
                          hipStream_t st);
// SnakeBeta -> k7 conv C->1 -> clip(-1,1) (MainDecoder tail, SpeechTokenizer.swift:687-688,781)
void launch_out_conv(const float* x, int C, const float* ea, const float* ib, const float* w, const float* bias,
                     const int32_t* frames, int ppf, int Tmax, int B, float* pcm, hipStream_t st, int hist = 0,
                     int32_t* nonfinite = nullptr);  // nonfinite[b] |= 1 when row b's pre-clip waveform holds an inf / NaN
// streamed decode: rows [chunk_rows - keep_rows, chunk_rows) of every batch row move to [-keep_rows, 0) (history of the next chunk)
void launch_roll_history(float* cur, int64_t bstride, int64_t keep_floats, int64_t chunk_floats, int B, hipStream_t st);

// ---- voice-clone front end (kernels/voice_frontend.hip) ------------------------------------------
// first SEANet conv: 1 -> C channels, causal k taps (SpeechTokenizerEncoder.swift:404-414). w [C][K], out [S][C]
void launch_enc_init_conv(const float* audio, int64_t S, int B, const float* w, const float* bias, int C, int K, float* out,
                          int64_t out_bstride, hipStream_t st);
// LayerNorm with bias over the last dim (EncoderTransformerLayer norm1/norm2, :559-560)
void launch_layernorm_f32(const float* x, int64_t x_bstride, const float* w, const float* b, float eps, int C, int T, int B,
                          float* out, int64_t out_bstride, hipStream_t st);
// MLXNN.RoPE(traditional: false) on the q and k thirds of qkv [T][3*heads*64], in place (:505-508).
// cos/sin [T][32] fp32 tables.
void launch_rope_qk_f32(float* qkv, int heads, int T, int B, const float* cos_t, const float* sin_t, hipStream_t st);
// causal variant of launch_attn_full_f32 (mask built at :1039-1043)
void launch_attn_causal_f32(const float* qkv, int heads, int T, int B, float* out, hipStream_t st);
// EncoderResidualVectorQuantization.encode (:816-829) over `n_layers` codebooks: r -= emb[argmin(c2 - r.emb)].
// x [T][ldx] (dim columns used), cb = device table of TRANSPOSED codebooks [dim][bins], c2 table, codes out [j * T + t].
// Batch: row b has valid[b] frames (the batch is padded to Tmax) and writes codes[code_off[b] + (layer0 + j) * valid[b] + t].
void launch_rvq_encode(const float* x, int ldx, int64_t x_bstride, int Tmax, int B, const int32_t* valid, const int64_t* code_off,
                       int dim, int bins, const float* const* cb, const float* const* c2, int n_layers, int layer0,
                       int32_t* codes, hipStream_t st);
// power spectrum -> mel -> log (SpeakerEncoder.swift:437-452). spec [T][ld]: re in [0,nfreq), im in [nfreq,2*nfreq)
void launch_log_mel(const float* spec, int ld, int T, int nfreq, const float* fb, int n_mels, float* out, hipStream_t st);
// per-channel mean / variance over time (SqueezeExcitationBlock :146, AttentiveStatisticsPooling :243-245)
void launch_time_stats(const float* x, int ld, int T, int C, float* mean, float* std_or_null, float eps, hipStream_t st);
// out[t][c] = x[t][c] * se[c] + res[t][c]   (SE gate + block residual, SpeakerEncoder.swift:154, :210)
void launch_scale_res(const float* x, int ldx, const float* se, const float* res, int ldr, float* out, int ldo, int T, int C,
                      hipStream_t st);
// out[t] = [x[t] | mean | std]   (AttentiveStatisticsPooling :248-252)
void launch_asp_concat(const float* x, const float* mean, const float* stdv, int T, int C, float* out, hipStream_t st);
// softmax over time of att [T][C], weighted mean / std of x -> pooled [2C] (:262-270)
void launch_asp_pool(const float* att, const float* x, int T, int C, float eps, float* pooled, hipStream_t st);
// rows valid[b] .. Tpad-1 of clip b <- 0 (clips of a padded batch end at different positions)
void launch_mask_tail(float* x, int64_t bstride, const int32_t* valid, int Tpad, int C, int B, hipStream_t st);
void launch_copy2d_f32(const float* src, int lds, float* dst, int ldd, int T, int C, hipStream_t st);

}  // namespace q3
