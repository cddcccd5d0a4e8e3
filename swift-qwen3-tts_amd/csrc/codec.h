// codec.h -- neural codec decoder runner (codes -> 24 kHz PCM) on the HIP kernels in
// kernels/codec_conv.hip and kernels/codec_misc.hip. Mirrors Qwen3TTSSpeechTokenizerDecoder
// (/root/reference/Sources/Qwen3TTS/Models/SpeechTokenizer.swift:754-784).
#pragma once
#include <string>
#include <vector>

#include "model.h"

namespace q3 {

class CodecRunner {
  public:
    CodecRunner(const Model& m, hipStream_t st, bool fp32_convs = false);
    ~CodecRunner();
    // codes_dev: [B][code_stride_frames][16] int32 on the device; rows decode frames[b] frames.
    // pcm_dev receives [B][Fmax*upsample] float32 (Fmax = max(frames)); returns Fmax.
    // If `stage` is non-empty the activation after that stage is copied to stage_out ([B][T][C]).
    // nonfinite_host (pinned, one int per row, optional): set to 1 behind the decode for rows whose waveform came out
    // non-finite -- with the default fp16 two-plane convs that is what an activation beyond 65504 turns into (codec_conv.hip);
    // the fp32 matrix-core path (q3tts_load_opts.codec_fp32) has the reference's range.
    // force_fp32: this call contracts on the fp32 matrix cores whatever the runner's default (the engine re-decodes rows
    // that left the fp16 range this way, so that a checkpoint with large activations still gets the reference's waveform).
    int decode(const int32_t* codes_dev, int code_stride_frames, const std::vector<int>& frames, float** pcm_dev,
               const std::string& stage = std::string(), std::vector<float>* stage_out = nullptr, int* stage_T = nullptr,
               int* stage_C = nullptr, int32_t* nonfinite_host = nullptr, bool force_fp32 = false);
    bool fp32_convs() const { return fp32_mfma_; }
    // The same decode with the causal tail (everything behind the pre-transformer) evaluated `chunk_frames` frames at a
    // time (row f1 of SURVEY 8f): each chunk's samples land in pcm_host ([B][Fmax * upsample], pinned host memory) at
    // their final place and chunk_done[k] is recorded behind chunk k's copy. Bit-identical to decode(). Returns the
    // number of chunks; chunk k covers frames [k * chunk_frames, min(Fmax, (k + 1) * chunk_frames)).
    int decode_chunked(const int32_t* codes_dev, int code_stride_frames, const std::vector<int>& frames, int chunk_frames,
                       float* pcm_host, std::vector<hipEvent_t>& chunk_done, int32_t* nonfinite_host = nullptr,
                       int32_t* nf_chunks_host = nullptr);  // (per-chunk flags as in stream_push)
    int tail_context_frames() const;
    static void set_scratch_budget(size_t bytes);  // test hook: forces the row-group paths of decode / decode_chunked (0: default)

    // ---- streamed decode (row f1: audio while tokens are still being generated) -------------------------------------
    // The causal tail keeps its own state between chunks: every tensor a causal conv reads lives in a persistent buffer
    // with `hist_frames()` frames of margin in front, into which the last frames of a chunk are rolled, so chunk k + 1
    // reads exactly the rows the one-shot decode would -- nothing is recomputed and the tail is bit-identical to decode().
    // The pre-transformer is bidirectional over the whole utterance in the reference (SpeechTokenizer.swift:763); a stream
    // cannot wait for the end, so chunk [f0, f1) is computed from a window of frames [f0 - window, f1 + lookahead) (what
    // exists of it): an approximation whose distance from the one-shot decode tests/test_streaming.py measures and bounds.
    // window < 0: the pre-transformer runs ONCE over all frames (needs every code up front; exact, for tests and offline use).
    struct StreamCfg {
        int rows = 0, chunk_frames = 0, window = 0, lookahead = 0, max_frames = 0;
    };
    void stream_open(const StreamCfg& cfg);
    // Rows have avail[b] frames so far (final[b]: the row will get no more). Decodes every chunk that has become decodable;
    // chunk k's samples go to pcm_host + b * pcm_row_stride + k * chunk_frames * upsample() and chunk_done[k] is recorded
    // behind the copy (events are created as needed). Returns the number of chunks issued so far. codes: device
    // [rows][code_stride_frames][16], frames below avail[b] final. No host synchronisation.
    // nf_chunks_host (pinned, optional): [chunk][rows] -- the rows' non-finite flags as they stand behind chunk k, copied in
    // front of chunk_done[k], so that a caller can hold a row's pieces back from the first chunk that left the fp16 range.
    int stream_push(const int32_t* codes_dev, int code_stride_frames, const int* avail, const uint8_t* final_rows, float* pcm_host,
                    size_t pcm_row_stride, std::vector<hipEvent_t>& chunk_done, int32_t* nf_chunks_host = nullptr);
    void stream_close(int32_t* nonfinite_host = nullptr);
    bool streaming() const { return stream_.open; }
    int hist_frames() const;
    int upsample() const { return up_; }
    hipStream_t stream() const { return st_; }
    void set_stream(hipStream_t st) { st_ = st; }  // the caller drains the old stream first (shared scratch)

  private:
    struct Pass {  // one pass of kernels over `nb` rows
        int nb = 0;
        int row0 = 0;         // first row of this pass in the call's batch (non-finite flags)
        int hist_frames = 0;  // streamed decode: tensors carry this many frames of history in front of their first row
        const int32_t* fr = nullptr;  // device: valid frames per row
        const std::string* stage = nullptr;
        std::vector<float>* stage_out = nullptr;
        int* stage_T = nullptr;
        int* stage_C = nullptr;
    };
    void conv(const Pass& ps, const struct ConvW& cw, const float* x, int Tmax, int ppf, float* out, const struct SnakeW* sn,
              const float* res, int act, const struct SnakeW* post = nullptr, float* out2 = nullptr);
    void capture(const Pass& ps, const char* name, const float* t, int T, int C);
    void run_front(const Pass& ps, const int32_t* codes, int code_stride_frames, int Fmax, float* const* bufs);
    void run_tail(const Pass& ps, int Tframes, float* const* bufs, float* pcm);
    // the MainDecoder (initConv .. outConv) of a float16 speech tokenizer on codec_conv_h1.hip: float16 tensors in the same
    // four scratch buffers; in: bufs[cur] = the last ConvNeXt stage's fp32 output at T positions
    void run_main_h1(const Pass& ps, int T, int ppf, int cur, float* const* bufs, float* pcm);
    void conv_h1(const Pass& ps, const struct ConvW& cw, const void* x, bool x_f32, int Tmax, int ppf, uint16_t* out, const uint16_t* res,
                 const struct SnakeW* post, uint16_t* out2);
    void capture_h(const Pass& ps, const char* name, const uint16_t* t, int T, int C);
    // the tail over one chunk of a stream: `lat` = the chunk's pre-transformer frames (stream layout), pcm out (stream layout)
    void run_tail_stream(const Pass& ps, float* lat, float* pcm);
    void run_main_h1_stream(const Pass& ps, const float* h32, int T, int ppf, float* pcm);  // its float16 half (run_main_h1's twin)
    float* sbuf(size_t frame_floats, bool keeps_history);  // next persistent tensor of the stream (same order every chunk)
    struct Stream {
        bool open = false, dry = false;
        StreamCfg cfg;
        int hist = 0, Tal = 0;       // margin frames, frames per allocation (hist + chunk)
        int next_chunk = 0;
        bool front_done = false;     // window < 0: the pre-transformer ran over all frames
        uint8_t* arena = nullptr;
        size_t arena_bytes = 0, off = 0;
        std::vector<std::pair<float*, size_t>> rolls;  // (allocation base, frame floats) of the tensors with history
        float *lat = nullptr, *pcm = nullptr, *x_all = nullptr;
        float* fbufs[4] = {nullptr, nullptr, nullptr, nullptr};
        size_t fbuf_floats = 0;
        int32_t* lens_host = nullptr;  // pinned, one slot per (chunk, kind): never reused inside a stream
        int32_t* lens_dev = nullptr;
        size_t lens_slots = 0, lens_used = 0;
    } stream_;
    size_t floats_per_frame() const;
    void upload_lens(const int32_t* lens, int n);
    int32_t* lens_host_ = nullptr;
    int32_t* nf_dev_ = nullptr;  // [kMaxRows] non-finite flags of the decode in flight (out_conv)
    static constexpr int kMaxRows = 4096;
    const Model& m_;
    hipStream_t st_;
    int up_ = 1920;
    bool no_h1_ = false;      // Q3TTS_CODEC_NO_F16=1: a float16 speech tokenizer through the up-cast (fp32-equivalent) path
    bool no_fuse_ = false;    // Q3TTS_CODEC_NO_FUSE=1: residual units of the narrow blocks as two launches each
    bool fp32_mfma_ = false;  // Q3TTS_CODEC_FP32=1: contract on the fp32 matrix-core path instead of the split one
    uint8_t* buf_ = nullptr;
    size_t buf_bytes_ = 0;
    int32_t* lens_dev_ = nullptr;
    int lens_cap_ = 0;
    void ensure(size_t bytes);
};

}  // namespace q3
