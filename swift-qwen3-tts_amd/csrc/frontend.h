// frontend.h -- voice-clone front end on the GPU: reference waveform -> 16 codec code
// rows (Qwen3TTSSpeechTokenizerEncoder.encode, /root/reference/Sources/Qwen3TTS/Models/SpeechTokenizerEncoder.swift:
// 1031-1056) and -> speaker x-vector (extractSpeakerEmbedding, Qwen3.swift:222-249 over SpeakerEncoder.swift).
// One utterance per call; both run once per request before the autoregressive loop.
#pragma once
#include <string>
#include <vector>

#include "model.h"

namespace q3 {

struct StageCapture {      // test hook: copy of one named intermediate activation
    std::string name;
    std::vector<float> data;
    int T = 0, C = 0;
};

class VoiceFrontEnd {
  public:
    VoiceFrontEnd(const Model& m, hipStream_t st);
    ~VoiceFrontEnd();
    // frames the encoder will produce for n_samples (StreamableConv1d padding rule, SpeechTokenizerEncoder.swift:114-118)
    int encoded_frames(int64_t n_samples) const;
    // audio_dev [n_samples] fp32 on the device -> codes_dev [16][T] int32 (row-major by code row); returns T
    int encode(const float* audio_dev, int64_t n_samples, int32_t* codes_dev, StageCapture* cap = nullptr);
    // B clips in one pass: audio_dev [B][n_samples], every clip zero-padded on the right to n_samples; clip b holds
    // clip_samples[b] samples and its encoded_frames(clip_samples[b]) frames are written as [16][frames] at
    // codes_dev + code_off[b] (host arrays)
    static constexpr int kMaxClips = 64;
    void encode_batch(const float* audio_dev, int B, int64_t n_samples, const int64_t* clip_samples, const int64_t* code_off,
                      int32_t* codes_dev, StageCapture* cap = nullptr);
    // audio_dev -> emb_dev [enc_dim] fp32
    void speaker_embedding(const float* audio_dev, int64_t n_samples, float* emb_dev, StageCapture* cap = nullptr);

  private:
    const Model& m_;
    hipStream_t st_;
    uint8_t* buf_ = nullptr;
    size_t buf_bytes_ = 0;
    int32_t* ones_dev_ = nullptr;  // frames[b] = 1: conv_gemm addresses rows as frames[b] * ppf
    void ensure(size_t bytes);
    void capture(StageCapture* cap, const char* name, const float* t, int T, int C, int ld);
};

}  // namespace q3
