/*
 * q3tts.h -- C ABI of the MI355X-native Qwen3-TTS engine (libq3tts_hip.so).
 *
 * The reference (AtomGradient/swift-qwen3-tts) has no FFI boundary of its own: Swift calls the
 * mlx-swift API directly. This header is the boundary a Swift shim binds
 * to so that `Qwen3TTSModel` / `.generate` / `.generateStream` keep their signatures while the
 * MLX/Metal backend is replaced by hand-written HIP. Each entry point cites the reference
 * interface it replaces (paths relative to /root/reference/Sources/Qwen3TTS/). The Swift-side
 * binding is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, opaque handle, int status codes, no exceptions and no
 * torch/HIP types across the boundary. Tokenisation stays on the caller's side
 * (swift-transformers in the reference, Qwen3.swift:274-275): the engine takes token ids.
 * One q3tts_model per GPU; calls on one handle are serialised by the caller (the reference
 * model object is not re-entrant either). The library checks this: a call that finds the handle
 * inside another thread's call returns Q3TTS_ERR_INVALID_INPUT at once (nested calls from the
 * same thread, e.g. from an event callback running on the calling thread, are fine).
 */
#ifndef Q3TTS_H
#define Q3TTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Q3TTS_ABI_VERSION 3

typedef struct q3tts_model q3tts_model;

/* Status codes 1..5 map 1:1 to AudioGenerationError (Core/GenerationTypes.swift:63-84). */
typedef enum {
    Q3TTS_OK = 0,
    Q3TTS_ERR_MODEL_NOT_INITIALIZED = 1, /* .modelNotInitialized  "Model not initialized: ..." */
    Q3TTS_ERR_GENERATION_FAILED = 2,     /* .generationFailed     "Generation failed: ..."     */
    Q3TTS_ERR_INVALID_INPUT = 3,         /* .invalidInput         "Invalid input: ..."         */
    Q3TTS_ERR_AUDIO_DECODING_FAILED = 4, /* .audioDecodingFailed                               */
    Q3TTS_ERR_AUDIO_ENCODING_FAILED = 5, /* .audioEncodingFailed                               */
    Q3TTS_ERR_IO = 6,     /* checkpoint/config read or parse failure (thrown Foundation errors) */
    Q3TTS_ERR_DEVICE = 7  /* HIP runtime failure; the engine never falls back to the CPU        */
} q3tts_status;

typedef struct {
    int32_t device;     /* HIP device ordinal */
    int32_t max_batch;  /* rows per q3tts_generate call, 1..64 (reference: always 1) */
    int32_t max_frames; /* upper bound on codec frames per row; sizes the paged KV pool (default 2048) */
    int32_t max_prompt; /* upper bound on prompt positions per row (default 512) */
    int32_t use_graph;  /* 1: replay the per-frame step as a hipGraph (default); 0: eager launches */
    int32_t weights_from_broadcast; /* 1: allocate the weight arena but do not read tensor data from
                                       disk; the caller fills it (RCCL broadcast from rank 0) via
                                       q3tts_model_arena before the first generate */
    int32_t n_streams;  /* lanes the batch is split over (own HIP stream + hipGraph + host thread each);
                           0 = default (1). Results do not depend on it (rows are independent). With more than one
                           lane q3tts_generate_begin runs its job to completion (TOKEN, INFO and AUDIO events fire inside
                           begin) and q3tts_generate_end only hands the results over: nothing overlaps */
    int32_t codec_overlap_cus; /* compute units a codec decode is confined to while the NEXT batch's frame loop runs beside it
                           (q3tts_generate_begin with more_follows != 0); multiple of 8; 0 = default (tuned for a 1.7B
                           talker at batch 32), -1 = never confine. Results do not depend on it */
    int32_t codec_fp32;  /* 1: the codec decoder's convolutions on the fp32 matrix cores (the reference's arithmetic range, 2.4x the
                           time) instead of fp16 matrix cores with every fp32 operand split into two fp16 planes (same accuracy,
                           activations limited to |x| < 65504: a decode that leaves that range reports
                           Q3TTS_ERR_AUDIO_DECODING_FAILED for the row instead of a waveform). Default 0 */
} q3tts_load_opts;

void q3tts_default_load_opts(q3tts_load_opts* o);

/* Qwen3TTSModel.fromPretrained(_:) (Models/Qwen3.swift:1382-1452) + postLoadHook (:1455-1495,
 * minus the text tokenizer, which stays with the caller). */
q3tts_status q3tts_model_load(const char* model_dir, const q3tts_load_opts* opts, q3tts_model** out);
void q3tts_model_free(q3tts_model* m);

/* Message of the last failure on this handle (NULL handle: last load failure on this thread).
 * Texts match the reference's error descriptions (GenerationTypes.swift:70-83). */
const char* q3tts_last_error(const q3tts_model* m);

/* Device weight arena (one contiguous allocation; layout is a pure function of the config, so
 * rank 0 can broadcast it to replicas at load: SURVEY.md section 8e). */
q3tts_status q3tts_model_arena(q3tts_model* m, void** device_ptr, size_t* bytes);

typedef struct {
    char tts_model_type[32]; /* Qwen3TTSModel.ttsModelType (Qwen3.swift:1269-1271) */
    int32_t sample_rate;     /* .sampleRate (Qwen3.swift:1262-1264) */
    int32_t supports_voice_cloning; /* .supportsVoiceCloning (Qwen3.swift:1210-1214) */
    int32_t has_voice_cloning;      /* .hasVoiceCloning (Qwen3.swift:61-63) */
    int32_t hidden_size, num_layers, vocab_size, text_vocab_size, num_code_groups;
    int32_t cp_hidden_size, cp_num_layers, cp_vocab_size;
    int32_t codec_eos_token_id;
    int32_t samples_per_frame; /* decodeUpsampleRate, 1920 */
    int32_t max_batch;
    int64_t weight_bytes;      /* distinct bytes streamed per decode step (roofline accounting) */
    int32_t speaker_embedding_dim; /* enc_dim of the speaker encoder, 0 when absent */
} q3tts_model_info;
q3tts_status q3tts_model_get_info(const q3tts_model* m, q3tts_model_info* out);

/* ---- multi-GPU (new: the reference is single-device, Qwen3.swift:1382-1470) ----------------------------------------------------
 * Utterances are independent, so a job shards by rows: one process (or handle) and one full replica per GPU, NO collective on
 * the data path. The only exchange is at load: rank `root` reads the checkpoint, every other rank loads with
 * q3tts_load_opts.weights_from_broadcast = 1 (config only, empty arena) and receives the arena -- whose layout is a pure
 * function of the config -- in one RCCL broadcast over xGMI. Recipe for a host without Python:
 *   rank 0: q3tts_comm_get_unique_id(&id); ship the 128 bytes to the other ranks (file, socket, MPI, environment ...)
 *   all   : q3tts_model_broadcast(model, &id, rank, world, 0);   // collective: every rank must call it
 *   rank r: q3tts_generate(rows [r * B, (r + 1) * B), q3tts_sampling.row_base = r * B)   // draws what one big call would
 * RCCL is opened at run time (librccl.so.1); single-GPU callers never need it. Status DEVICE (7) carries RCCL's message. */
typedef struct { char bytes[128]; } q3tts_comm_id; /* == ncclUniqueId */
q3tts_status q3tts_comm_get_unique_id(q3tts_comm_id* out);
q3tts_status q3tts_model_broadcast(q3tts_model* m, const q3tts_comm_id* id, int32_t rank, int32_t world, int32_t root);
/* 64-bit sum of the arena's 32-bit words (device-side reduction): equal on every rank after the broadcast. */
q3tts_status q3tts_model_arena_checksum(q3tts_model* m, uint64_t* out);

/* Qwen3TTSModel.supportedSpeakers, sorted (Qwen3.swift:965-971). */
int32_t q3tts_model_num_speakers(const q3tts_model* m);
const char* q3tts_model_speaker_name(const q3tts_model* m, int32_t i);

/* One utterance. Mirrors the arguments of generate(text:speaker:instruct:language:...)
 * (Qwen3.swift:1291-1301) after tokenisation:
 *   text_ids      = tokens of "<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n" (:274-275)
 *   instruct_ids  = tokens of "<|im_start|>user\n{instruct}<|im_end|>\n" or NULL (:364-365)
 *   target_token_count = tokens of {text} alone, for the max-token cap (:822-823) */
typedef struct {
    const int32_t* text_ids;
    int32_t n_text_ids;
    const int32_t* instruct_ids;
    int32_t n_instruct_ids;
    int32_t target_token_count;
    const char* speaker;  /* NULL = none */
    const char* language; /* NULL = "auto" */
    int32_t max_tokens;   /* 0 = 2048 (reference default) */
    /* Voice clone -- generateVoiceClone(text:referenceAudio:referenceText:language:...) (Qwen3.swift:1009-1020).
     * ref_audio != NULL selects it; speaker / instruct_ids are then ignored, as in the reference.
     *   ref_audio     = reference waveform, 24 kHz mono float32 (host memory, read during the call; a NaN or infinite
     *                   sample is Q3TTS_ERR_INVALID_INPUT)
     *   ref_text_ids  = tokens of "<|im_start|>assistant\n{referenceText}<|im_end|>\n" (:448-449)
     * The reference's default repetition penalty on this path is 1.5 (:1017): set it in q3tts_sampling.
     * Result: pcm = audio of the target text only (reference part cut proportionally, :1195-1199),
     * codes = generated frames only. */
    const float* ref_audio;
    int64_t n_ref_samples;
    const int32_t* ref_text_ids;
    int32_t n_ref_text_ids;
    int32_t route; /* 0: generate() -- the prompt builder is chosen by tts_model_type and its requirements are enforced
                      (Qwen3.swift:1302-1372). 1: generateVoiceDesign called directly (:587-597): no speaker, instruct optional,
                      whatever the checkpoint's type. 2: generateCustomVoice called directly (:783-794): the speaker is
                      required and validated against talker_config.spk_id (:803-811), instruct optional. Ignored for
                      voice-clone requests (ref_audio != NULL) */
} q3tts_request;

/* Defaults as generate(): 0.9 / 50 / 1.0 / 1.05 (Qwen3.swift:1296-1299). */
typedef struct {
    float temperature; /* <= 0: greedy argmax (Qwen3.swift:182-185) */
    int32_t top_k;
    float top_p;
    float repetition_penalty;
    uint64_t seed;        /* new: the reference draws from MLX's global key and has no seed API */
    int32_t force_frames; /* bench only: mask EOS and emit exactly this many frames per row */
    int32_t audio_chunk_frames; /* new (the reference decodes one-shot, README.md:140): > 0 delivers the waveform in pieces of
                                   this many codec frames through AUDIO_CHUNK events as the causal tail of the decoder
                                   produces them, before INFO / AUDIO; the samples are bit-identical to the one-shot decode */
    int32_t audio_window_frames; /* 0 (default): the chunks above are cut after the last token, from the exact decode.
                                   > 0 (with audio_chunk_frames > 0): audio leaves WHILE tokens are still being generated. The
                                   decoder's pre-transformer is bidirectional over the whole utterance (SpeechTokenizer.swift:763),
                                   so a chunk is then computed from the frames that exist: this many frames of left context and
                                   audio_lookahead_frames to the right; everything behind the pre-transformer is causal and
                                   carries its state from chunk to chunk (exact). The arithmetic is pinned against the oracle's
                                   restatement of exactly this definition (OracleModel.codec_decode_streamed, PCM within 1e-4:
                                   tests/test_streaming.py). Its distance from the ONE-SHOT waveform is a property of the
                                   checkpoint -- how much of the signal the pre-transformer's attention carries -- and is
                                   not guaranteed: 1.6e-2 max / 1 % of the signal r.m.s. at window 32 / lookahead 4 on the synthetic
                                   full-width checkpoint (reported by the test), zero with a window over
                                   everything. AUDIO then carries the concatenation of the chunks -- all generated frames: the
                                   reference's end trim to count(code0 > 0) frames (SpeechTokenizer.swift:831-833) cannot apply
                                   to samples that have already left. audio_chunk_frames must be at
                                   least the causal tail's history (3 frames for the shipped decoder geometry; checked before any
                                   GPU work). Not combined with voice-clone rows (those fall back to 0) */
    int32_t audio_lookahead_frames; /* frames to the right of a chunk that must exist before it is decoded (default 4) */
    uint32_t row_base;    /* new: global index of reqs[0] in a job whose rows are sharded over several processes (one
                             replica per GPU). A row's random stream is keyed by (seed, row_base + row index), so a sharded
                             job draws what the same rows would draw in one call. Default 0 */
} q3tts_sampling;
void q3tts_default_sampling(q3tts_sampling* s);

/* AudioGenerationInfo (Core/GenerationTypes.swift:15-21). */
typedef struct {
    int32_t prompt_token_count;
    int32_t generation_token_count;
    double prefill_time;
    double generate_time;
    double tokens_per_second;
    double peak_memory_usage; /* GB */
} q3tts_gen_info;

/* enum AudioGeneration { token, info, audio } (Core/GenerationTypes.swift:51-58). Per request the
 * order is TOKEN* (EOS is not reported: Qwen3.swift:868-871), INFO, AUDIO -- the order
 * generateStream yields them (Qwen3+Streaming.swift:24-27,118-120). */
typedef enum {
    Q3TTS_EVENT_TOKEN = 0, Q3TTS_EVENT_INFO = 1, Q3TTS_EVENT_AUDIO = 2,
    Q3TTS_EVENT_AUDIO_CHUNK = 3 /* only with q3tts_sampling.audio_chunk_frames > 0: samples [sample_offset, +n_samples) of
                                   the request's final audio, in order, between the last TOKEN and INFO */
} q3tts_event_kind;
typedef struct {
    q3tts_event_kind kind;
    int32_t request_index;
    int32_t token;              /* TOKEN */
    const q3tts_gen_info* info; /* INFO  */
    const float* pcm;           /* AUDIO, AUDIO_CHUNK: valid during the callback */
    int64_t n_samples;
    int64_t sample_offset;      /* AUDIO_CHUNK: position of pcm[0] in the request's final audio */
} q3tts_event;
typedef void (*q3tts_event_cb)(void* user, const q3tts_event* ev);

typedef struct {
    q3tts_status status; /* per-request outcome (e.g. GENERATION_FAILED "No tokens generated") */
    float* pcm;          /* 24 kHz mono float32, trimmed as Qwen3.swift:954-959; engine-owned */
    int64_t n_samples;
    int32_t* codes;      /* [n_frames][num_code_groups]; engine-owned */
    int32_t n_frames;
    q3tts_gen_info info;
} q3tts_result;

/* generate / generateStream (Qwen3.swift:1291-1373, Qwen3+Streaming.swift:8-125) for n_reqs
 * utterances at once (row-independent: each row equals the batch-1 result for that request).
 * `cb` may be NULL. `results` has n_reqs entries, released with q3tts_result_free. */
q3tts_status q3tts_generate(q3tts_model* m, const q3tts_request* reqs, int32_t n_reqs,
                            const q3tts_sampling* sampling, q3tts_event_cb cb, void* user,
                            q3tts_result* results);
void q3tts_result_free(q3tts_result* results, int32_t n);

/* q3tts_generate in two halves, for callers with a queue of batches (the reference has neither batches nor a queue:
 * its generate() decodes one-shot after the loop, Qwen3.swift:943-959, which is what each job still does).
 *   begin: prompt assembly, prefill and the AR loop of this batch (TOKEN events fire here); returns once the codes exist
 *          and their codec decode has been queued on the engine's second HIP stream.
 *   end:   waits for that decode, fills `results` (n_reqs entries of the begin call), fires INFO and AUDIO events, and
 *          releases the job.
 * A second begin may be issued before the first job's end: its AR loop (a latency-bound chain of small launches) then
 * overlaps the first job's decode (matrix-core work). At most 2 jobs may be outstanding per model handle; results do
 * not depend on the interleaving (the decode reads job-owned copies of the codes). q3tts_generate == begin + end.
 * `more_follows` != 0 says that another begin will be issued before this job's end: the decode is then confined to part
 * of the chip so that the next batch's launch chain keeps room (a decode that fills every CU stalls that chain and
 * nothing is gained); 0 (the last batch of a queue) lets the decode use the whole chip. Results do not depend on it. */
typedef struct q3tts_job q3tts_job;
q3tts_status q3tts_generate_begin(q3tts_model* m, const q3tts_request* reqs, int32_t n_reqs,
                                  const q3tts_sampling* sampling, q3tts_event_cb cb, void* user, int32_t more_follows,
                                  q3tts_job** job);
q3tts_status q3tts_generate_end(q3tts_model* m, q3tts_job* job, q3tts_result* results);

/* Qwen3TTSSpeechTokenizer.decode (Models/SpeechTokenizer.swift:823-836): codes
 * [batch][max_frames][num_code_groups] -> pcm [batch][max_frames*1920] (caller-allocated),
 * audio_lengths[batch] = count(code0 > 0) * 1920. n_frames[b] <= max_frames are the valid rows. Every code of a valid row is
 * a row of its RVQ table: one outside it is Q3TTS_ERR_INVALID_INPUT (checked on the host before anything is uploaded). */
q3tts_status q3tts_codec_decode(q3tts_model* m, const int32_t* codes, const int32_t* n_frames,
                                int32_t batch, int32_t max_frames, float* pcm, int64_t* audio_lengths);

/* The same decode the way a stream produces it (q3tts_sampling.audio_window_frames): chunks of `chunk_frames` frames, the causal
 * tail carrying its state between them, the pre-transformer over [chunk start - window, chunk end + lookahead) -- or, with
 * window < 0, once over all frames, which makes the result bit-identical to q3tts_codec_decode. For tests and for callers that
 * hold a code sequence and want the bounded-latency arithmetic. */
q3tts_status q3tts_codec_decode_streamed(q3tts_model* m, const int32_t* codes, const int32_t* n_frames, int32_t batch,
                                         int32_t max_frames, int32_t chunk_frames, int32_t window, int32_t lookahead, float* pcm);

/* Qwen3TTSSpeechTokenizer.encode (Models/SpeechTokenizer.swift:841-846 -> SpeechTokenizerEncoder.swift:1031-1056):
 * 24 kHz mono float32 waveform -> codes [16][*n_frames] int32 (code row major, as the reference returns
 * [1, 16, time]); cap_frames = capacity of `codes` in frames (q3tts_codec_encoded_frames gives the exact count). */
q3tts_status q3tts_codec_encode(q3tts_model* m, const float* audio, int64_t n_samples, int32_t* codes,
                                int32_t cap_frames, int32_t* n_frames);
int32_t q3tts_codec_encoded_frames(const q3tts_model* m, int64_t n_samples);

/* Qwen3TTSModel.extractSpeakerEmbedding(_:sampleRate:) (Models/Qwen3.swift:222-249): log-mel (n_fft 1024, hop 256,
 * 128 mels) -> ECAPA-TDNN (Models/SpeakerEncoder.swift:364-394). out [enc_dim] float32; sample_rate must be 24000
 * (:223-225). */
q3tts_status q3tts_speaker_embedding(q3tts_model* m, const float* audio, int64_t n_samples, int32_t sample_rate,
                                     float* out, int32_t cap);

/* 16-bit PCM as the reference's CLI writes it (Sources/Qwen3TTSDemo/main.swift:134-165): each sample is clamped to
 * [-1, 1], multiplied by 32767 in Float and converted with Int16(_:), i.e. truncated toward zero (:158-162).
 * q3tts_write_wav writes the same 44-byte RIFF/WAVE header (PCM, mono, 16 bit) followed by those samples. Host code. */
void q3tts_pcm_to_int16(const float* pcm, int64_t n_samples, int16_t* out);
q3tts_status q3tts_write_wav(const char* path, const float* pcm, int64_t n_samples, int32_t sample_rate);

/* Text tokeniser: the Qwen2 byte-level BPE that the checkpoints ship as tokenizer.json (or vocab.json + merges.txt),
 * which the reference loads through swift-transformers (`AutoTokenizer.from(modelFolder:)`, Models/Qwen3.swift:1458) and
 * calls at :274-275, :364-365, :448-457, :822. Optional: callers that tokenise themselves never touch it. `path` is a
 * model directory or a tokenizer.json file. encode = tokenizer.encode(text:) (no special tokens are added by the Qwen2
 * post-processor); ids == NULL only counts. Host code, no GPU. */
typedef struct q3tts_tokenizer q3tts_tokenizer;
q3tts_status q3tts_tokenizer_load(const char* path, q3tts_tokenizer** out);
void q3tts_tokenizer_free(q3tts_tokenizer* t);
q3tts_status q3tts_tokenizer_encode(const q3tts_tokenizer* t, const char* utf8, int32_t* ids, int32_t cap, int32_t* n);

/* Timing of the last q3tts_generate / q3tts_codec_decode on this handle, measured with HIP events
 * on the engine's own stream (bench.py's roofline object reads these). */
typedef struct {
    double prefill_ms;
    double decode_ms;       /* all frame steps */
    double codec_ms;
    int32_t frame_steps;    /* frame-step launches in decode_ms */
    int32_t rows;
    int64_t kv_bytes_read;  /* algorithmic KV bytes read over all frame steps */
    double frontend_ms;     /* voice clone: codec encoder + speaker encoder over all rows of the call */
    double first_audio_ms;  /* streamed decode: request in -> first AUDIO_CHUNK samples on the host (0 when nothing streamed) */
    int32_t launches_per_frame_step; /* kernel launches (graph nodes) of ONE frame step at this call's batch size: the chain decode_ms
                                        is made of is frame_steps x this many dependent launches */
} q3tts_timing;
q3tts_status q3tts_last_timing(const q3tts_model* m, q3tts_timing* out);

/* ---------------------------------------------------------------------------------------------
 * Test hooks: block-level entry points used by tests/ to compare each stage with the oracle.
 * Not part of the drop-in surface. All bf16 buffers are raw uint16 bit patterns, host memory.
 * ------------------------------------------------------------------------------------------- */

/* prepareGenerationInputs (Qwen3.swift:259-409). Outputs (caller-allocated, capacities in rows):
 * input_embeds [*n_prompt][H], trailing [*n_trailing][H], tts_pad [H]. */
q3tts_status q3tts_debug_prepare_inputs(q3tts_model* m, const q3tts_request* req,
                                        uint16_t* input_embeds, int32_t cap_prompt, int32_t* n_prompt,
                                        uint16_t* trailing, int32_t cap_trailing, int32_t* n_trailing,
                                        uint16_t* tts_pad);

/* Teacher-forced generation: same kernels as q3tts_generate, but the tokens fed back are
 * forced_codes [n_reqs][n_frames][groups]; per-frame logits are returned:
 * talker_logits [n_reqs][n_frames][V], cp_logits [n_reqs][n_frames][groups-1][Vcp] (either may be
 * NULL). Sampled tokens (what the sampler would have chosen) go to sampled [n_reqs][n_frames][groups]. */
q3tts_status q3tts_debug_generate_forced(q3tts_model* m, const q3tts_request* reqs, int32_t n_reqs,
                                         const q3tts_sampling* sampling, const int32_t* forced_codes,
                                         int32_t n_frames, uint16_t* talker_logits, uint16_t* cp_logits,
                                         int32_t* sampled);

/* sampleToken (Qwen3.swift:130-213) on caller-supplied logits [rows][V] (bf16) with the engine's
 * sampler kernel. seen [rows][V] uint8 may be NULL. */
q3tts_status q3tts_debug_sample(q3tts_model* m, const uint16_t* logits, int32_t rows, int32_t V,
                                const q3tts_sampling* sampling, const uint8_t* seen,
                                int32_t suppress_lo, int32_t suppress_hi, int32_t eos_id,
                                uint32_t row0, uint32_t draw, int32_t* tokens);

/* Skinny bf16 GEMM used by every Linear on the decode path (Talker.swift:183-186,413-415):
 * y[M][N] = x[M][K] W[N][K]^T (+bias), M <= 64. */
q3tts_status q3tts_debug_linear(q3tts_model* m, const uint16_t* x, const uint16_t* W, const uint16_t* bias,
                                int32_t M, int32_t K, int32_t N, uint16_t* y);

/* Codec decoder with intermediate activations (SpeechTokenizer.swift:754-784) for one utterance:
 * stage names: "quantizer","pre_conv","pre_transformer","upsample0","upsample1","init_conv",
 * "block0".."block3". Output is channels-last [T][C] float32; *T,*C receive the shape. */
q3tts_status q3tts_debug_codec_stage(q3tts_model* m, const int32_t* codes, int32_t n_frames,
                                     const char* stage, float* out, int64_t cap_floats, int32_t* T, int32_t* C);

/* Activation scratch the codec decoder may use per pass (default 24 GB; 0 restores it): a small value forces the paths that
 * take a large batch through in groups of rows. Process-wide. */
void q3tts_debug_set_codec_scratch(uint64_t bytes);

/* The launchers' diagnostic switches (Q3TTS_GEMM_NO_ROW_SPLIT, Q3TTS_NO_TALL_GEMM, Q3TTS_PF, ...: csrc/kernels.h DebugEnv)
 * are read from the environment once per q3tts_model_load, not per launch; a test that changes one on a live model calls
 * this afterwards. Process-wide; not to be called while a generate call is running. */
void q3tts_debug_reload_env(void);

/* Voice-clone front end with intermediate activations for one waveform. Codec encoder stages
 * (SpeechTokenizerEncoder.swift:1031-1056): "init_conv","layer0".."layer3","seanet","transformer","downsample",
 * "rvq_first_in","rvq_rest_in"; speaker encoder stages (SpeakerEncoder.swift:364-394): "mel","h0".."h3","mfa",
 * "pooled". Output is channels-last [T][C] float32. */
q3tts_status q3tts_debug_frontend_stage(q3tts_model* m, const float* audio, int64_t n_samples, const char* stage,
                                        float* out, int64_t cap_floats, int32_t* T, int32_t* C);

#ifdef __cplusplus
}
#endif
#endif /* Q3TTS_H */
