// kernels.h -- host-callable launchers of the HIP kernels (definitions in kernels/*.hip).
// Every launcher only enqueues work on `st` (no allocation, no sync), so the whole frame step can
// be captured into a hipGraph (cdna_hip_programming.md Guideline 9).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels/prefetch.h"

namespace q3 {

// Diagnostic switches of the launchers, read from the environment ONCE (first use) and again only on request
// (q3tts_debug_reload_env; the Python mirror calls it at every model load): a frame step is ~550 launches, and getenv per
// launch both costs host time on the eager path and races with a test's setenv on another lane's thread.
struct DebugEnv {
    bool gemm_no_row_split;  // Q3TTS_GEMM_NO_ROW_SPLIT: row blocks of narrow layers stay in one workgroup
    bool gemm_one_pair;      // Q3TTS_GEMM_ONE_PAIR: one weight tile per workgroup everywhere
    bool no_tall_gemm;       // Q3TTS_NO_TALL_GEMM: prefill chunks through the skinny kernels
    int tall_shape;          // Q3TTS_TALL_SHAPE = 2 | 3: force one tile shape of the tall GEMM (0: automatic)
    int chunk_qsplit;        // Q3TTS_CHUNK_QSPLIT: workgroups sharing a chunk's queries (0: automatic)
    bool conv_no_pw;         // Q3TTS_CONV_NO_PW: pointwise convs through the general conv kernel
    bool nt_off;             // Q3TTS_NT=0: no non-temporal loads on the talker's weights / cache
    int prefetch;            // Q3TTS_PF: 0 = launches touch nothing ahead, 1 (default) = next-launch weight touch (prefetch.h;
                             // only builds with Q3_PF_MODE != 0 carry the touch code -- the shipped one does not)
    int pf_budget_kb;        // Q3TTS_PF_BUDGET_KB: bytes per XCD that may sit touched-ahead in its L2
    int pf_ahead;            // Q3TTS_PF_AHEAD: how many launches ahead a launch may look for a stream to touch
    int pf_skip;             // Q3TTS_PF_SKIP: pass over this many candidate streams first (experiments: which stream pays)
};
const DebugEnv& debug_env();
void debug_env_reload();

constexpr int kPageTokens = 64;  // tokens per KV page
constexpr int kHeadDim = 128;    // Qwen3-TTS talker / code predictor head_dim (Config.swift:154,301)

// ---- RMSNorm of whole rows (lm_misc.hip): only where a normalised vector must be materialised --
// (the code predictor's first input when there is no small_to_mtp_projection). In the layers the
// norm lives in the consumer GEMM's prologue.
struct NormRowsArgs {
    const uint16_t* h;  // fragment-major [M][H]
    int hMB;
    const uint16_t* w;
    float eps;
    uint16_t* out;      // fragment-major
    int outMB;
    float* ss_out;      // [ss_ld]: sum(out^2) per row (first partial of the consumer's norm) or nullptr
    int M, H;
    // optional: sum(h^2) per row from the producer's per-tile partials [ss_count][ss_ld], added up exactly as the GEMM's
    // norm prologue adds them (gemm_body.inc), so that this kernel + a GEMM without prologue == the GEMM with it, bit for bit
    const float* ss_in;
    int ss_count, ss_ld;
};
void launch_norm_rows(const NormRowsArgs& a, hipStream_t st);

// ---- skinny GEMM (gemm_decode.hip) -------------------------------------------------------------
struct GemmArgs {
    const uint16_t* W;  // tiled weights (repack.hip): bf16 tiles, or packed int4 tiles when Wsb != nullptr
    const uint32_t* Wsb;  // int4 path: per (tile, chunk, lane) {bf16 scale, bf16 bias} of the lane's 64-wide group
    const uint16_t* x;  // fragment-major activations (common.h: act_tiled_offset), rows >= M ignored
    int xMB;            // row blocks of the x allocation
    int M, Mpad, N, K;  // N, K are the padded (tiled) sizes
    int epi;            // 0: y=bf16(acc+bias) (+silu)   2: gate/up -> silu(g)*u   3: hidden-state store (+residual)
    uint16_t* y;        // epi 0: row-major [M][ldy], or fragment-major when y_tiled; epi 2/3: fragment-major
    int ldy;
    int y_tiled, yMB;
    const uint16_t* bias;
    int act_silu;
    // RMSNorm prologue (norm_w != nullptr): x is the raw residual stream
    const uint16_t* norm_w;  // [K] row-major
    const float* ss_in;      // [ss_count][ss_ld] per-tile sums of squares of x rows
    int ss_count, ss_ld, norm_dim;
    float norm_eps;
    // epi 3
    int nt_weights;          // 1: load the weight tiles non-temporally (read once per step, working set >> Infinity Cache)
    int resid;               // 1: y = bf16(y_old + bf16(acc+bias)); 0: y = bf16(acc+bias)
    float* ss_out;           // [N/16][ss_ld] this tile's share of sum(y^2) per row, or nullptr
    PfArgs pf;               // weight tiles of a later launch to touch (prefetch.h); decode-shaped launches only
};
void launch_gemm_skinny(const GemmArgs& a, hipStream_t st);
// The launch launch_gemm_skinny makes for these arguments -- ONE place decides it (the plain launch, the launch with riders
// and the engine's prefetch planner all read it here).
struct SkinnyGeom {
    bool tall;         // the tall (prefill) kernel takes it: none of the rest applies
    int split, mbw;    // grid.y, row blocks per workgroup
    int nw, ch, np;    // waves, k chunks per wave (0: streamed), weight tiles per workgroup
    int gx;            // grid.x
    bool ntw;          // non-temporal weight loads
    bool touches;      // the kernel carries the touch-ahead code (decode shapes: at most two row blocks, chunks in registers)
    int threads() const { return nw * 64; }
    // bytes of W that workgroup x streams, contiguous from x * span_bytes
    size_t span_bytes(const GemmArgs& a) const { return size_t(np) * size_t(a.K / 128) * (a.Wsb ? 1024 : 4096); }
};
SkinnyGeom skinny_geometry(const GemmArgs& a);
constexpr bool gemm_touches(int mb, int ch) { return Q3_PF_MODE != 0 && Q3_PF_GEMM && mb <= 2 && ch > 0; }  // which instantiations carry the touch-ahead code
// The same launch with riders (kernels/row_jobs.h): n.M extra workgroups, each the RMSNorm of one row -- a job that reads what
// this GEMM reads and nothing it writes, so it shares the launch instead of paying its own. Only the shapes of the talker's
// codec_head (EPI 0 with norm prologue, K = 1024 or 2048, at most 32 rows per workgroup); returns false (nothing launched)
// otherwise. A kernel of its own: the riders' dispatch test must not sit in front of every other GEMM's first load (as a
// field of GemmArgs it cost 0.28 us per launch, 500 launches per frame step).
// More than 64 rows, plain bf16 weights, no prologue / bias: the tall form (gemm_prefill.hip), bit-identical to the skinny one.
// launch_gemm_skinny tries it first; false = nothing launched (shape or options it does not take).
bool launch_gemm_tall(const GemmArgs& a, hipStream_t st);
bool gemm_tall_takes(const GemmArgs& a);  // whether launch_gemm_tall would launch (same test, nothing enqueued)
bool launch_gemm_skinny_with_norm_rows(const GemmArgs& a, const NormRowsArgs& n, hipStream_t st);
bool gemm_norm_rows_rides(const GemmArgs& a, const NormRowsArgs& n);  // whether the call above would launch


// ---- QK-norm + RoPE + KV append + paged decode attention (attn_decode.hip) ---------------------
struct AttnArgs {
    const uint16_t* qkv;  // [B][ld] : q heads | k heads | v heads
    int ld;
    const uint16_t* qn_w;
    const uint16_t* kn_w;
    float eps;
    const uint16_t* rope_cos;  // [max_pos][128] bf16
    const uint16_t* rope_sin;
    uint16_t* kpool;           // this layer: [n_pages][n_kv][kPageTokens][128]
    uint16_t* vpool;
    const int32_t* block_table;  // [B][max_pages]
    int max_pages;
    const int32_t* kv_len;       // [B] tokens already cached (= position of the new token)
    const uint8_t* active;       // [B] or nullptr (always): append this token to the cache
    uint16_t* out;               // fragment-major [B][n_heads*128] (o_proj's x operand)
    int outMB;
    int n_heads, n_kv, B;
    float scale;
    int fixed_len;       // >= 0: every row's cache holds exactly this many tokens (code predictor: the pass index is known
                         // when the launch is enqueued), so nothing has to be loaded before the cache rows are requested
    int identity_pages;  // 1: row b owns page b (the code predictor's one-page-per-row cache)
    // chunk > 1: `chunk` consecutive positions per batch row in this launch; qkv / out row of element p of row b is
    // p * B + b. chunk_n_prompt != nullptr: right-aligned prompt chunks, element p of row b is prompt position
    // chunk_r_base + chunk_n_prompt[b] + p and is skipped while that is negative; nullptr: every element is live.
    int chunk;
    const int32_t* chunk_n_prompt;
    int chunk_r_base;
    int nt_kv;           // 1: cache rows are loaded non-temporally (long caches read once per step)
    PfArgs pf;           // weight tiles of a later launch to touch (prefetch.h); one-position launches only
};
void launch_attn_decode(const AttnArgs& a, hipStream_t st);
int attn_decode_threads(const AttnArgs& a);  // threads per workgroup of the one-position launch for these arguments

struct FrameEndArgs {  // Qwen3.swift:919-935 + loop bookkeeping
    const int32_t* cur_codes;     // [B][16]
    const uint16_t* codec_emb;    // talker codec_embedding [V][H]
    const uint16_t* const* cp_emb;  // device array of 15 tables [Vcp][H]
    const uint16_t* trailing;     // [B][Tmax][H]
    const int32_t* n_trailing;    // [B]
    int32_t* trailing_idx;        // [B]
    int Tmax;
    const uint16_t* tts_pad;      // [H]
    uint16_t* h;                  // fragment-major [B][H] next talker input
    int hMB, H, B, groups;
    float* ss_out;                // [B] sum of squares of the new input
    int32_t* n_frames;
    const int32_t* max_frames;
    uint8_t* finished;
    uint8_t* active;
    int32_t* cp_len;              // [B] reset to 0
};
void launch_frame_end(const FrameEndArgs& a, hipStream_t st);

// ---- sampler (sampler.hip) ---------------------------------------------------------------------
struct SamplingParams {  // lives in device memory so the captured graph does not depend on it
    float temperature;
    int top_k;
    float top_p;
    float rep_penalty;
    uint64_t seed;
    uint32_t row0;
    int mask_eos;
};

struct SamplerArgs {
    const uint16_t* logits;  // [B][ldl]
    int ldl, V;
    const SamplingParams* sp;
    int is_talker;           // 1: suppress range + repetition penalty + EOS handling; 0: plain (code predictor)
    int suppress_lo, suppress_hi, eos_id;
    uint8_t* seen;           // [B][V] (talker only)
    int cb;                  // codebook index written
    const int32_t* n_frames; // [B] frames completed so far (draw index = frame*16 + cb)
    const int32_t* max_frames;  // [B]
    uint8_t* finished;       // [B]
    uint8_t* active;         // [B] talker: cleared when the row finishes
    int32_t* kv_len;         // [B] advanced by `advance` when the row consumed this step
    int advance;             // 1: kv_len[b] += (active ? 1 : 0) -- end of a talker / predictor pass
    const uint8_t* advance_gate;  // nullptr: unconditional
    int32_t* cur_codes;      // [B][16] codes of the frame in flight
    int32_t* codes;          // [B][Fmax][16]
    int Fmax;
    const int32_t* forced;   // [B][forced_frames][16] teacher forcing (tests) or nullptr
    int forced_frames;
    int32_t* sampled;        // [B][forced_frames][16] what the sampler chose (tests) or nullptr
    const uint16_t* emb;     // embedding table of the sampled id -> next code-predictor input
    int emb_ld;
    uint16_t* next_x;        // fragment-major [B][H] or nullptr
    int next_MB;
    int next_row0;           // row b is written at row b + next_row0 of next_x (second position of predictor step 0)
    float* next_ss;          // [B] sum of squares of the gathered row (norm prologue of the consumer) or nullptr
    const float* emb_ss;     // optional [V][nss]: precomputed per-tile sums of squares of every table row (projected tables);
    int nss, next_ss_ld;     // copied to next_ss[j * next_ss_ld + row] instead of the single sum
    int H;
    int B;
    uint16_t* logits_dump;   // [B][forced_frames][dump_ld] (tests) or nullptr
    int dump_ld, dump_off;
};
void launch_sampler(const SamplerArgs& a, hipStream_t st);
// The frame's last draw with its row's end-of-frame job riding along (kernels/row_jobs.h frame_end_job): the sampler is that
// job's only predecessor and works on the same row. Code-predictor draws only (a.is_talker == 0, V <= 2048).
void launch_sampler_with_frame_end(const SamplerArgs& a, const FrameEndArgs& fe, hipStream_t st);
// dst[r][j] = ss[j * ss_ld + r]: per-tile sums of squares of `rows` GEMM output rows -> table rows
void launch_ss_to_table(const float* ss, int ss_ld, float* dst, int nss, int rows, hipStream_t st);

// ---- embedding plumbing (lm_misc.hip) ----------------------------------------------------------
// gather rows of a bf16 table: out[i] = table[ids[i]] (optionally through a token map)
// out_MB > 0: out is fragment-major with that many row blocks (ldo ignored)
void launch_gather_rows(const uint16_t* table, int ld, const int32_t* ids, const int32_t* token_map,
                        int n, int dim, uint16_t* out, int ldo, int out_MB, hipStream_t st);
// row-major <-> fragment-major conversions (rows x dim, dim % 128 == 0)
void launch_tile_rows(const uint16_t* src, int lds, uint16_t* dst, int dstMB, int rows, int dim, hipStream_t st);
void launch_untile_rows(const uint16_t* src, int srcMB, uint16_t* dst, int ldd, int rows, int dim, hipStream_t st);
// out = a + b (bf16, one rounding), rows x dim; b_stride 0 broadcasts one row
void launch_add_rows(const uint16_t* a, int lda, const uint16_t* b, int ldb, int rows, int dim,
                     uint16_t* out, int ldo, hipStream_t st);

struct PrefillLoadArgs {
    const uint16_t* prompt;   // [B][Pmax][H]
    const int32_t* n_prompt;  // [B]
    int Pmax, step, H, B;
    uint16_t* h;              // fragment-major [B][H]
    int hMB;
    float* ss_out;            // [B] sum of squares of the loaded row
    uint8_t* active;          // [B]
};
void launch_prefill_load(const PrefillLoadArgs& a, hipStream_t st);
void launch_advance_len(int32_t* kv_len, const uint8_t* active, int B, hipStream_t st);
void launch_stamp(unsigned long long* acc, unsigned long long* last, int k, hipStream_t st);  // diagnostics
// chunked prefill: C positions per row per step; a.step carries r_base (prompt index of element 0 = r_base + n_prompt[b])
void launch_prefill_chunk_load(const PrefillLoadArgs& a, int C, hipStream_t st);
void launch_advance_len_chunk(int32_t* kv_len, const int32_t* n_prompt, int r_base, int C, int B, hipStream_t st);


// prompt assembly (Qwen3.swift:371-406, 505-510): dst[dst_row[i]] = proj[a[i]] when b[i] == -1, else
// bf16(proj[a[i]] + other) with other = table[b[i]] (b >= 0) or extra[-2 - b[i]] (voice-clone rows)
void launch_compose_rows(const uint16_t* proj, int ldp, const uint16_t* table, int ldt, const uint16_t* extra, int lde,
                         const int32_t* a, const int32_t* b, const int32_t* dst_row, uint16_t* dst, int ldd, int n, int dim,
                         hipStream_t st);
// sum of the 16 codebook embeddings of reference frames (Qwen3.swift:485-491); codes [groups][T]
void launch_ref_embed_rows(const int32_t* codes, int T, int groups, const uint16_t* codec_emb, const uint16_t* const* cp_emb,
                           int H, uint16_t* out, int ldo, hipStream_t st);
void launch_f32_to_bf16(const float* x, uint16_t* out, int n, hipStream_t st);
// out [Tref + F][16] = transpose(ref [16][Tref]) ++ gen [F][16]   (Qwen3.swift:1176-1180)
void launch_build_decode_codes(const int32_t* ref, int Tref, const int32_t* gen, int F, int32_t* out, hipStream_t st);

// copies rows (bf16) between strided buffers: dst[r][0..dim) = src[r][0..dim)
void launch_copy_rows(const uint16_t* src, int lds, uint16_t* dst, int ldd, int rows, int dim, hipStream_t st);

}  // namespace q3
