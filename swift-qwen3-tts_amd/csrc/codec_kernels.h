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
    const float* bias;   // [N] or nullptr
    const float* scale;  // [N] or nullptr
    const float* res;    // residual or nullptr
    int ldr;
    int64_t res_bstride;
    float* out;
    int ldo;
    int64_t out_bstride;
    const float* snake_ea;  // SnakeBeta prologue on x (exp(alpha), 1/(exp(beta)+1e-9)) or nullptr
    const float* snake_ib;
    const int32_t* frames;  // [B]
    int ppf;                // positions per frame at this stage
    int Tmax;
    int B, Cin, N, K, dil;
    int act;                // 0: none, 1: GELU (erf)
};
void launch_conv_gemm(const ConvGemmArgs& a, hipStream_t st);

// Split-RVQ gather (SpeechTokenizer.swift:214-226, 81-96): out[b][f] = [cb_first[c0] | sum_j cb_rest[j][c_{j+1}]]
void launch_rvq_gather(const int32_t* codes, int code_stride_frames, const float* cb_first,
                       const float* const* cb_rest, int n_rest, int inner, const int32_t* frames, int Fmax, int B,
                       float* out, hipStream_t st);
// fp32 RMSNorm over the last dim (SpeechTokenizer.swift:581-582,626): out = (x*rstd)*w
void launch_rmsnorm_f32(const float* x, const float* w, float eps, int C, const int32_t* frames, int ppf, int Tmax,
                        int B, float* out, hipStream_t st);
// depthwise causal conv k7 + LayerNorm (ConvNeXtBlock, SpeechTokenizer.swift:389-393)
void launch_dwconv_ln(const float* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                      float eps, int C, const int32_t* frames, int ppf, int Tmax, int B, float* out, hipStream_t st);
// out[t][i] = silu(gu[t][i]) * gu[t][I+i] (DecoderMLP, SpeechTokenizer.swift:560-562)
void launch_silu_mul_f32(const float* gu, int I, const int32_t* frames, int ppf, int Tmax, int B, float* out,
                         hipStream_t st);
// full bidirectional attention without positions or mask (SpeechTokenizer.swift:512-528).
// qkv [B][Tmax][3*heads*64] (q | k | v) -> out [B][Tmax][heads*64]
void launch_attn_full_f32(const float* qkv, int heads, const int32_t* frames, int Tmax, int B, float* out,
                          hipStream_t st);
// SnakeBeta -> k7 conv C->1 -> clip(-1,1) (MainDecoder tail, SpeechTokenizer.swift:687-688,781)
void launch_out_conv(const float* x, int C, const float* ea, const float* ib, const float* w, const float* bias,
                     const int32_t* frames, int ppf, int Tmax, int B, float* pcm, hipStream_t st);

}  // namespace q3
