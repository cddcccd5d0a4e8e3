// model.h -- device-resident model: one contiguous weight arena in HBM plus typed views.
//
// Load path = Qwen3TTSModel.fromPretrained + sanitize + postLoadHook + sanitizeSpeechTokenizerWeights
// (/root/reference/Sources/Qwen3TTS/Models/Qwen3.swift:1382-1495, 1219-1260, 1498-1750), re-hosted:
// host mmap -> (GPU repack kernels) -> arena. The arena layout is a pure function of the config, so
// replicas can receive it by one RCCL broadcast (q3tts_model_arena).
#pragma once
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "config.h"
#include "safetensors.h"

namespace q3 {

struct LinearW {          // Linear in the streaming tile layout (gemm_decode.hip): bf16 tiles or packed int4 tiles
    const uint16_t* w = nullptr;
    const uint32_t* sb = nullptr;    // int4 path: {scale, bias} per (tile, chunk, lane); nullptr for bf16
    const uint16_t* bias = nullptr;  // [N] or nullptr
    int N = 0, K = 0;                // logical sizes
    int Np = 0, Kp = 0;              // padded to 16 / 128
};

struct LayerW {
    const uint16_t* ln1 = nullptr;
    const uint16_t* ln2 = nullptr;
    const uint16_t* qn = nullptr;
    const uint16_t* kn = nullptr;
    LinearW qkv;     // rows: q | k | v
    LinearW o;
    LinearW gateup;  // 16-row tiles interleaved: gate tile j, up tile j; N = intermediate
    LinearW down;
    int inter = 0, inter_p = 0;
};

struct StackW {
    int hidden = 0, n_heads = 0, n_kv = 0, head_dim = 0;
    float eps = 1e-6f, rope_base = 1e6f;
    std::vector<LayerW> layers;
    const uint16_t* final_norm = nullptr;
    const uint16_t* rope_cos = nullptr;  // [max_pos][128] bf16
    const uint16_t* rope_sin = nullptr;
    int max_pos = 0;
    int max_inter_p = 0;
};

struct ConvW {  // fp32 conv as GEMM: w[N][K][Cin]; transposed convs are stored in polyphase form
    const float* w = nullptr;
    const uint16_t* wh = nullptr;  // codec decoder only: w * 2^s[n] as two fp16 planes (hi | lo), [K][ceil(Cin/32)][N][2][32], and
    const float* wsc = nullptr;    // wsc[n] = 2^-s[n] (codec_conv.hip conv_gemm_h2_kernel; model.cc attach_h2)
    const uint16_t* whp = nullptr; // pointwise conv behind a k7 conv (DecoderResidualUnit conv2): wh with the 32 input channels of a
                                   // chunk in the order the fused kernel's accumulators hold them (codec_conv.hip resunit_h2_kernel)
    const uint16_t* w1 = nullptr;  // MainDecoder convs of a float16 checkpoint: the weights themselves, [K][ceil(Cin/32)][N][32] fp16
                                   // (codec_conv_h1.hip; model.cc attach_h1)
    const uint16_t* w1p = nullptr; // ... and a residual unit's pointwise conv2 in the fused kernel's k order (attach_h1_perm), [Cin/32][N][32]
    const float* bias = nullptr;   // [N] or nullptr
    const float* scale = nullptr;  // per-output-channel scale (LayerScale / ConvNeXt gamma) or nullptr
    int Cin = 0, N = 0, K = 1, dil = 1;
};
struct SnakeW {
    const float* ea = nullptr;  // exp(alpha)
    const float* ib = nullptr;  // 1 / (exp(beta) + 1e-9)
    const float* ea16 = nullptr;  // float16 checkpoints: the same, rounded where MLX rounds on float16 arrays
    const float* ib16 = nullptr;
    int C = 0;
};

struct CodecW {
    // RVQ (SpeechTokenizer.swift:175-227): codebooks [size][inner], output projections fused
    const float* cb_first = nullptr;               // [semantic_size][inner]
    std::vector<const float*> cb_rest;             // 15 x [codebook_size][inner]
    const float* const* cb_rest_dev = nullptr;     // device array of the pointers above
    int inner = 0;
    int cb_first_rows = 0, cb_rest_rows = 0;       // rows of the tables as loaded (the smallest of the 15): what a code may index
    ConvW rvq_out;   // [codebook_dim][1][2*inner]  (semantic | acoustic)
    ConvW pre_conv;
    ConvW t_in, t_out;
    struct TLayer {
        const float* ln1 = nullptr;
        const float* ln2 = nullptr;
        ConvW qkv, o, gateup, down;  // o and down carry the LayerScale as `scale`
    };
    std::vector<TLayer> tlayers;
    const float* t_norm = nullptr;
    struct Up {
        ConvW tconv;                  // polyphase k=s transposed conv: N = s*C, K = 1
        const float* dw_w = nullptr;  // depthwise [C][7]
        const float* dw_b = nullptr;
        const float* ln_w = nullptr;
        const float* ln_b = nullptr;
        ConvW pw1, pw2;               // pw2.scale = gamma
        int stride = 2;
    };
    std::vector<Up> ups;
    ConvW init_conv;
    struct Res {
        SnakeW act1, act2;
        ConvW conv1, conv2;
    };
    struct Block {
        SnakeW snake;
        ConvW tconv;  // polyphase: N = s*Cout, K = 2
        int stride = 1, Cout = 0;
        Res res[3];
    };
    std::vector<Block> blocks;
    SnakeW out_snake;
    bool f16_main = false;         // float16 speech tokenizer: the MainDecoder runs on codec_conv_h1.hip
    const float* out_w = nullptr;  // [1][7][C]
    const float* out_b = nullptr;
    int out_C = 0;
};

// Codec encoder (SpeechTokenizerEncoder.swift): SEANet -> causal transformer -> stride-2 conv -> split RVQ search.
struct CodecEncW {
    const float* init_w = nullptr;  // [C0][K] (one input channel)
    const float* init_b = nullptr;
    int init_C = 0, init_K = 0;
    struct Layer {
        ConvW res1, res2;  // k3 C->C/2, k1 C/2->C (ELU before each, skip connection)
        ConvW down;        // k=2r stride r as a K=2 causal conv over the [T/r][r*C] view (same bytes as [N][2r][C])
        int ratio = 1, C = 0;
    };
    std::vector<Layer> layers;
    ConvW final_conv;
    struct TLayer {
        const float *ln1_w = nullptr, *ln1_b = nullptr, *ln2_w = nullptr, *ln2_b = nullptr;
        ConvW qkv, o, fc1, fc2;  // o / fc2 carry the LayerScale
    };
    std::vector<TLayer> tlayers;
    int heads = 0, hidden = 0;
    const float* rope_cos = nullptr;  // [max_T][32]
    const float* rope_sin = nullptr;
    int max_T = 0;
    ConvW down;  // stride ds, same view trick
    int ds = 2;
    ConvW rvq_in;  // rows: rvq_first.input_proj | rvq_rest.input_proj
    int dim = 0, bins = 0, n_layers = 0;  // layers that reach the output (16: 1 semantic + 15 acoustic)
    std::vector<const float*> cb, c2;  // cb: TRANSPOSED codebooks [dim][bins] (voice_frontend.hip rvq_encode_kernel)
    const float* const* cb_dev = nullptr;
    const float* const* c2_dev = nullptr;
};

// ECAPA-TDNN speaker encoder + log-mel front end (SpeakerEncoder.swift)
struct SpeakerEncW {
    ConvW dft;                    // windowed DFT basis [2*nfreq (+pad)][1][n_fft]: Hann * cos | Hann * sin
    const float* mel_fb = nullptr;  // [nfreq][n_mels]
    int n_fft = 1024, hop = 256, nfreq = 513, n_mels = 128;
    ConvW b0;
    struct Block {
        ConvW tdnn1, tdnn2, se1, se2;
        std::vector<ConvW> res;  // scale-1 convs on C/scale channels
        int C = 0;
    };
    Block blocks[3];
    ConvW mfa, asp_tdnn, asp_conv, fc;
    int scale = 8, enc_dim = 0;
};

struct Model {
    ModelConfig cfg;
    int device = 0;
    uint8_t* arena = nullptr;
    size_t arena_bytes = 0;

    StackW talker, cp;
    const uint16_t* codec_emb = nullptr;  // [V][H]
    const uint16_t* text_emb = nullptr;   // [TV'][TH]
    const int32_t* token_map = nullptr;   // [text_vocab] or nullptr (pruned vocabulary)
    int64_t text_emb_rows = 0;
    LinearW fc1, fc2, codec_head, cp_proj;
    bool has_cp_proj = false;
    std::vector<LinearW> lm_head;           // 15
    std::vector<const uint16_t*> cp_emb;    // 15 x [Vcp][H]
    const uint16_t* const* cp_emb_dev = nullptr;
    bool has_codec = false;
    CodecW codec;
    bool has_codec_encoder = false;    // Qwen3TTSSpeechTokenizer.hasEncoder (SpeechTokenizer.swift:816-818)
    CodecEncW codec_enc;
    bool has_speaker_encoder = false;  // Qwen3TTSModel.hasVoiceCloning (Qwen3.swift:61-63)
    SpeakerEncW speaker;
    int64_t step_weight_bytes = 0;  // distinct weight bytes read by one frame step (roofline)
    std::vector<void*> side_allocs; // pointer tables (absolute addresses; not part of the arena)
    // small_to_mtp_projection applied to every row of the code predictor's embedding tables: built by the first Engine
    // with its own decode GEMM (Engine::build_cp_proj_tables), shared by all lanes. [table][Vcp][CH] bf16, [table][Vcp][CH/16]
    mutable std::vector<const uint16_t*> cp_pe;
    mutable std::vector<const float*> cp_pss;
    mutable std::vector<void*> lazy_allocs;
    mutable std::mutex lazy_mutex;

    ~Model();
};

struct LoadOptions {
    int device = 0;
    int max_pos_talker = 4096;
    bool skip_tensor_data = false;  // weights_from_broadcast
};

std::unique_ptr<Model> load_model(const std::string& dir, const LoadOptions& opt);

// GPU repack helpers (repack.hip)
void launch_tile_weights(const uint16_t* src, int N, int K, uint16_t* dst, int KC, int tile_off, int tile_stride,
                         hipStream_t st, int rows_per_tile = 16, int row_off = 0);
void launch_tile_int4(const uint32_t* wq, const uint16_t* scales, const uint16_t* biases, int N, int K, void* dq,
                      uint32_t* dsb, int KC, int tile_off, int tile_stride, hipStream_t st, int rows_per_tile = 16, int row_off = 0);

}  // namespace q3
