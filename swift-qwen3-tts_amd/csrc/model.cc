// model.cc -- checkpoint loader: restates the reference's load + sanitise steps and lays the
// weights out in HBM for the HIP kernels. See model.h.
#include "model.h"

#include "kernels.h"

#include <climits>
#include <cmath>
#include <functional>
#include <map>

namespace q3 {

Model::~Model() {
    if (arena) (void)hipFree(arena);
    for (void* p : side_allocs) (void)hipFree(p);
    for (void* p : lazy_allocs) (void)hipFree(p);
}

namespace {

// ------------------------------------------------------------------------------------------------
// host tensors for the speech tokenizer (fp32, small enough to sanitise on the host)
// ------------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;  // empty when only the shape is known (skip_tensor_data)
    bool f16 = false;         // stored as float16 in the checkpoint (the "lite" speech tokenizers, docs/paper.tex:207)
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

std::vector<float> to_f32(const TensorView& tv) {
    std::vector<float> out((size_t)(tv.numel()));
    if (tv.dtype == DType::F32) {
        std::memcpy(out.data(), tv.data, out.size() * 4);
    } else if (tv.dtype == DType::BF16) {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(tv.data);
        for (size_t i = 0; i < out.size(); ++i) out[i] = bf16_to_f32_host(p[i]);
    } else if (tv.dtype == DType::F16) {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(tv.data);
        for (size_t i = 0; i < out.size(); ++i) {  // fp16 "lite" codecs (docs/paper.tex:207)
            uint32_t h = p[i], sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
            if (e == 0) {
                if (m == 0) u = sign;
                else {
                    int sh = 0;
                    while (!(m & 0x400)) { m <<= 1; ++sh; }
                    u = sign | ((uint32_t)(113 - sh) << 23) | ((m & 0x3ff) << 13);
                }
            } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
            else u = sign | ((e + 112) << 23) | (m << 13);
            std::memcpy(&out[i], &u, 4);
        }
    } else {
        throw Error(6, "unsupported dtype for a float tensor");
    }
    return out;
}

// checkArrayShapeQwen3 (Qwen3.swift:1246-1260): "is this 3-D conv weight already in MLX layout?"
bool is_mlx_conv_layout(const std::vector<int64_t>& s) {
    const int64_t d2 = s[1], d3 = s[2];
    if (d2 == 1) return d3 > 64;
    if (d3 == 1) return d2 <= 64;
    return d2 < d3;
}

HostTensor permute3(const HostTensor& t, int a, int b, int c) {
    HostTensor o;
    const int64_t s[3] = {t.shape[0], t.shape[1], t.shape[2]};
    o.shape = {s[a], s[b], s[c]};
    if (t.data.empty()) return o;
    o.data.resize(t.data.size());
    int64_t idx[3];
    for (idx[0] = 0; idx[0] < s[0]; ++idx[0])
        for (idx[1] = 0; idx[1] < s[1]; ++idx[1])
            for (idx[2] = 0; idx[2] < s[2]; ++idx[2]) {
                const int64_t src = (idx[0] * s[1] + idx[1]) * s[2] + idx[2];
                const int64_t dst = (idx[a] * s[b] + idx[b]) * s[c] + idx[c];
                o.data[size_t(dst)] = t.data[size_t(src)];
            }
    return o;
}

void replace_all(std::string& s, const std::string& a, const std::string& b) {
    size_t pos = 0;
    while ((pos = s.find(a, pos)) != std::string::npos) {
        s.replace(pos, a.size(), b);
        pos += b.size();
    }
}
bool contains(const std::string& s, const std::string& a) { return s.find(a) != std::string::npos; }
bool starts_with(const std::string& s, const std::string& a) { return s.rfind(a, 0) == 0; }
bool ends_with(const std::string& s, const std::string& a) {
    return s.size() >= a.size() && s.compare(s.size() - a.size(), a.size(), a) == 0;
}

// Encoder half of sanitizeSpeechTokenizerWeights (Qwen3.swift:1592-1700): key remaps of the SEANet / transformer /
// quantizer trees and the conv-weight transposes to [out][k][in].
void sanitize_encoder_entry(std::string& nk, HostTensor& t) {
    static const std::pair<const char*, const char*> kSeanet[] = {  // :1517-1528
        {"encoder.encoder.layers.0.", "encoder.encoder.init_conv1d."},
        {"encoder.encoder.layers.1.", "encoder.encoder.layers.0.residuals.0."},
        {"encoder.encoder.layers.3.", "encoder.encoder.layers.0.downsample."},
        {"encoder.encoder.layers.4.", "encoder.encoder.layers.1.residuals.0."},
        {"encoder.encoder.layers.6.", "encoder.encoder.layers.1.downsample."},
        {"encoder.encoder.layers.7.", "encoder.encoder.layers.2.residuals.0."},
        {"encoder.encoder.layers.9.", "encoder.encoder.layers.2.downsample."},
        {"encoder.encoder.layers.10.", "encoder.encoder.layers.3.residuals.0."},
        {"encoder.encoder.layers.12.", "encoder.encoder.layers.3.downsample."},
        {"encoder.encoder.layers.14.", "encoder.encoder.final_conv1d."}};
    const size_t nd = t.shape.size();
    bool transposed = false;
    for (auto& m : kSeanet)  // :1594-1599
        if (starts_with(nk, m.first)) {
            nk = std::string(m.second) + nk.substr(std::strlen(m.first));
            break;
        }
    if (contains(nk, ".residuals.")) {  // :1603-1607
        replace_all(nk, ".block.1.", ".block.0.");
        replace_all(nk, ".block.3.", ".block.1.");
    }
    const bool is_seanet = starts_with(nk, "encoder.encoder.") && !contains(nk, "encoder_transformer") && !contains(nk, "quantizer") &&
                           (contains(nk, ".conv.weight") || contains(nk, ".conv.bias"));  // :1612-1615
    if (is_seanet) {
        replace_all(nk, ".conv.weight", ".conv.conv.weight");
        replace_all(nk, ".conv.bias", ".conv.conv.bias");
        if (ends_with(nk, ".weight") && nd == 3) transposed = true;  // forced, :1622-1624
    }
    if (contains(nk, "encoder.encoder_transformer.layers.")) {  // :1629-1649
        replace_all(nk, "encoder.encoder_transformer.layers.", "encoder.encoder_transformer.transformer.layers.");
        replace_all(nk, ".input_layernorm.", ".norm1.");
        replace_all(nk, ".post_attention_layernorm.", ".norm2.");
        replace_all(nk, ".mlp.fc1.", ".gating.linear1.");
        replace_all(nk, ".mlp.fc2.", ".gating.linear2.");
        replace_all(nk, ".self_attn_layer_scale.", ".layer_scale_1.");
        replace_all(nk, ".mlp_layer_scale.", ".layer_scale_2.");
    }
    if (starts_with(nk, "encoder.downsample.conv.") && !contains(nk, "encoder.downsample.conv.conv.")) {  // :1652-1659
        const bool is_w = ends_with(nk, ".weight");
        replace_all(nk, "encoder.downsample.conv.", "encoder.downsample.conv.conv.conv.");
        if (is_w && nd == 3) transposed = true;
    }
    if (contains(nk, "encoder.quantizer.")) {  // :1664-1676
        replace_all(nk, ".semantic_residual_vector_quantizer.", ".rvq_first.");
        replace_all(nk, ".acoustic_residual_vector_quantizer.", ".rvq_rest.");
        replace_all(nk, ".rvq_first.layers.", ".rvq_first.vq.layers.");
        replace_all(nk, ".rvq_rest.layers.", ".rvq_rest.vq.layers.");
    }
    const bool was_seanet_w = starts_with(nk, "encoder.encoder.") && !contains(nk, "encoder_transformer") &&
                              !contains(nk, "quantizer") && ends_with(nk, ".conv.conv.weight");  // :1682-1685
    const bool is_proj = (contains(nk, "input_proj.weight") || contains(nk, "output_proj.weight")) && contains(nk, "quantizer");
    if (is_proj && nd == 3) transposed = true;  // :1688-1692
    // :1696-1700 re-derives the value from the ORIGINAL tensor: a transpose when the heuristic says "not MLX
    // layout", otherwise whatever was decided above stays
    if (contains(nk, "conv.weight") && nd == 3 && !is_proj && !was_seanet_w && !is_mlx_conv_layout(t.shape)) transposed = true;
    if (transposed) t = permute3(t, 0, 2, 1);
}

// sanitizeSpeechTokenizerWeights (Qwen3.swift:1498-1750), decoder and encoder halves.
std::map<std::string, HostTensor> sanitize_speech_tokenizer(const SafetensorsDir& st, bool with_data) {
    static const std::pair<const char*, const char*> kIndex[] = {
        {"decoder.decoder.0", "decoder.decoder.initConv"}, {"decoder.decoder.1", "decoder.decoder.block0"},
        {"decoder.decoder.2", "decoder.decoder.block1"},   {"decoder.decoder.3", "decoder.decoder.block2"},
        {"decoder.decoder.4", "decoder.decoder.block3"},   {"decoder.decoder.5", "decoder.decoder.outSnake"},
        {"decoder.decoder.6", "decoder.decoder.outConv"}};
    std::map<std::string, HostTensor> out;
    std::map<std::string, std::pair<const TensorView*, const TensorView*>> cb;   // base -> (usage, sum)
    std::map<std::string, std::pair<const TensorView*, const TensorView*>> ecb;  // encoder codebooks
    for (auto& kv : st.all()) {
        const std::string& key = kv.first;
        const TensorView& tv = kv.second;
        if (contains(key, "._codebook.cluster_usage") || contains(key, "._codebook.embedding_sum")) {  // :1532-1543
            std::string base = key.substr(0, key.find("._codebook."));
            if (contains(key, "cluster_usage")) cb[base].first = &tv;
            else cb[base].second = &tv;
            continue;
        }
        if (starts_with(key, "encoder.quantizer.") && contains(key, ".codebook.")) {  // :1546-1565
            const size_t pos = key.find(".codebook.");
            const std::string field = key.substr(pos + 10);
            if (key.find(".codebook.", pos + 1) == std::string::npos && (field == "embed_sum" || field == "cluster_usage")) {
                auto& e = ecb[key.substr(0, pos)];
                (field == "cluster_usage" ? e.first : e.second) = &tv;
                continue;
            }
            if (contains(key, ".initialized")) continue;
        }
        if (starts_with(key, "encoder.")) {
            HostTensor t;
            t.shape = tv.shape;
            if (with_data) t.data = to_f32(tv);
            std::string nk = key;
            sanitize_encoder_entry(nk, t);
            out[nk] = std::move(t);
            continue;
        }
        std::string nk = key;
        for (auto& m : kIndex) {  // :1573-1578
            std::string pre = std::string(m.first) + ".";
            if (starts_with(key, pre)) {
                nk = std::string(m.second) + key.substr(pre.size() - 1);
                break;
            }
        }
        if (starts_with(nk, "decoder.")) {  // :1581-1588
            replace_all(nk, ".block.0.", ".snake.");
            replace_all(nk, ".block.1.", ".upsample.");
            replace_all(nk, ".block.2.", ".res1.");
            replace_all(nk, ".block.3.", ".res2.");
            replace_all(nk, ".block.4.", ".res3.");
        }
        HostTensor t;
        t.shape = tv.shape;
        t.f16 = tv.dtype == DType::F16;
        if (with_data) t.data = to_f32(tv);
        const bool is_proj = (contains(nk, "input_proj.weight") || contains(nk, "output_proj.weight")) && contains(nk, "quantizer");
        HostTensor nv;
        bool changed = false;
        if (is_proj && t.shape.size() == 3) {  // :1688-1692
            nv = permute3(t, 0, 2, 1);
            changed = true;
        }
        if (contains(nk, "conv.weight") && t.shape.size() == 3 && !is_proj) {  // :1696-1700
            if (!is_mlx_conv_layout(t.shape)) {
                nv = permute3(t, 0, 2, 1);
                changed = true;
            }
        }
        const bool is_tr = (contains(nk, "upsample") && contains(nk, ".0.conv.weight")) ||
                           (contains(nk, "decoder.decoder.block") && contains(nk, "upsample.conv.weight"));  // :1704-1711
        if (is_tr && t.shape.size() == 3 && !is_mlx_conv_layout(t.shape)) {
            nv = permute3(t, 1, 2, 0);
            changed = true;
        }
        if (changed) nv.f16 = t.f16;
        out[nk] = changed ? std::move(nv) : std::move(t);
    }
    for (auto& kv : cb) {  // codebook = embedding_sum / clip(cluster_usage, 1e-5) (:1716-1724)
        if (!kv.second.first || !kv.second.second) continue;
        HostTensor e;
        e.shape = kv.second.second->shape;
        if (with_data) {
            std::vector<float> usage = to_f32(*kv.second.first), sum = to_f32(*kv.second.second);
            const int64_t rows = e.shape[0], dim = e.shape[1];
            e.data.resize(sum.size());
            for (int64_t r = 0; r < rows; ++r) {
                const float u = std::max(usage[size_t(r)], 1e-5f);
                for (int64_t d = 0; d < dim; ++d) e.data[size_t(r * dim + d)] = sum[size_t(r * dim + d)] / u;
            }
        }
        out[kv.first + ".codebook.embed.weight"] = std::move(e);
    }
    for (auto& kv : ecb) {  // raw sums and usage under the Swift property names (:1727-1747)
        if (!kv.second.first || !kv.second.second) continue;
        std::string nb = kv.first;
        replace_all(nb, ".semantic_residual_vector_quantizer.", ".rvq_first.");
        replace_all(nb, ".acoustic_residual_vector_quantizer.", ".rvq_rest.");
        replace_all(nb, ".rvq_first.layers.", ".rvq_first.vq.layers.");
        replace_all(nb, ".rvq_rest.layers.", ".rvq_rest.vq.layers.");
        HostTensor sum, usage;
        sum.shape = kv.second.second->shape;
        usage.shape = kv.second.first->shape;
        if (with_data) {
            sum.data = to_f32(*kv.second.second);
            usage.data = to_f32(*kv.second.first);
        }
        out[nb + ".codebook.embeddingSum"] = std::move(sum);
        out[nb + ".codebook.clusterUsage"] = std::move(usage);
    }
    return out;
}

// Qwen3TTSModel.sanitize (Qwen3.swift:1219-1243) for the only 3-d tensors of the main checkpoint: the speaker
// encoder's Conv1d weights, PyTorch [out][in][k] -> MLX [out][k][in] unless the shape heuristic says otherwise.
std::map<std::string, HostTensor> speaker_encoder_tensors(const SafetensorsDir& main, bool with_data) {
    std::map<std::string, HostTensor> out;
    for (auto& kv : main.all()) {
        const std::string& key = kv.first;
        if (!starts_with(key, "speaker_encoder.")) continue;
        HostTensor t;
        t.shape = kv.second.shape;
        if (with_data) t.data = to_f32(kv.second);  // bf16 parameters meet fp32 activations: exact upcast
        const bool is_conv_w = (contains(key, "conv") || contains(key, "speaker_encoder.fc")) && contains(key, "weight");
        if (is_conv_w && t.shape.size() == 3 && !is_mlx_conv_layout(t.shape)) t = permute3(t, 0, 2, 1);
        out[key] = std::move(t);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// arena builder: the same construction code runs twice (size pass, then fill pass)
// ------------------------------------------------------------------------------------------------
struct Builder {
    bool dry = true;
    bool fill = true;  // false: allocate only (weights arrive by broadcast)
    uint8_t* base = nullptr;
    size_t off = 0;
    uint8_t* staging = nullptr;   // device staging for raw matrices before tiling
    size_t max_staging = 0;       // bytes, collected in the dry pass
    bool split3 = false;          // codec decoder: convs also get the two-plane fp16 copy of their weights (attach_h2)
    bool h1 = false;              // float16 speech tokenizer: the MainDecoder's convs also get their exact one-plane tiles (attach_h1)

    template <class T>
    T* alloc(size_t n) {
        off = align_up(off, 256);
        T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
    template <class T>
    const T* put(const T* host, size_t n) {
        T* p = alloc<T>(n);
        if (!dry && fill && host) Q3_HIP(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
        return p;
    }
    // pointer tables hold absolute device addresses, so they stay outside the broadcastable arena
    std::vector<void*>* side = nullptr;
    template <class T>
    const T* put_side(const T* host, size_t n) {
        if (dry) return nullptr;
        T* p = nullptr;
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
        side->push_back(p);
        Q3_HIP(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
        return p;
    }
    const float* put_f32(const HostTensor& t) {
        return put<float>(t.data.empty() ? nullptr : t.data.data(), size_t(t.numel()));
    }
};

struct MainTensors {
    const SafetensorsDir& st;
    const TensorView& bf16(const std::string& k, std::vector<int64_t> shape) const {
        const TensorView& tv = st.at(k);
        Q3_CHECK(tv.dtype == DType::BF16, 6, "tensor '" + k + "' must be bf16 (only Linear weights and embedding tables may be int4-quantised: .weight uint32 + .scales / .biases)");
        Q3_CHECK(tv.shape == shape, 6, "tensor '" + k + "' has an unexpected shape");
        return tv;
    }
};

// A Linear's weight as stored in the checkpoint: bf16 [N][K], or MLX affine int4 (packed uint32 [N][K/8] plus
// bf16 scales/biases [N][K/64]) when `<name>.scales` exists (Qwen3.swift:1402-1425).
struct LinSrc {
    const TensorView* w = nullptr;
    const TensorView* scales = nullptr;
    const TensorView* biases = nullptr;
    int N = 0, K = 0;
    bool quant() const { return scales != nullptr; }
};
LinSrc lin_src(const SafetensorsDir& st, const std::string& name, int N, int K) {
    LinSrc s;
    s.N = N;
    s.K = K;
    s.w = &st.at(name + ".weight");
    if (st.has(name + ".scales")) {
        s.scales = &st.at(name + ".scales");
        s.biases = &st.at(name + ".biases");
        Q3_CHECK(s.w->dtype == DType::U32 && s.w->shape == std::vector<int64_t>({N, K / 8}), 6,
                 "quantised weight '" + name + "' must be uint32 [out][in/8] (4-bit)");
        Q3_CHECK(s.scales->dtype == DType::BF16 && s.biases->dtype == DType::BF16 &&
                     s.scales->shape == std::vector<int64_t>({N, K / 64}) && s.biases->shape == s.scales->shape,
                 6, "quantised '" + name + "' needs bf16 scales/biases [out][in/64] (group size 64)");
    } else {
        Q3_CHECK(s.w->dtype == DType::BF16 && s.w->shape == std::vector<int64_t>({N, K}), 6,
                 "tensor '" + name + ".weight' has an unexpected dtype or shape");
    }
    return s;
}

const uint16_t* put_bf16(Builder& b, const TensorView& tv) {
    return b.put<uint16_t>(reinterpret_cast<const uint16_t*>(tv.data), size_t(tv.numel()));
}

// rows of several [N_i][K] matrices -> one tiled weight (tile index = tile_off + tile_stride * j)
struct RowSrc {
    LinSrc src;
    int tile_off, tile_stride;
    int rows_per_tile = 16, row_off = 0;  // 8 / {0, 8}: two matrices share every tile (gate rows 0..7, up rows 8..15)
};
LinearW put_linear(Builder& b, const std::vector<RowSrc>& srcs, int N, int K, int total_tiles, const TensorView* bias) {
    LinearW L;
    L.N = N;
    L.K = K;
    L.Kp = int(align_up(size_t(K), 128));
    L.Np = total_tiles * 16;
    const bool quant = srcs[0].src.quant();
    for (auto& s : srcs) Q3_CHECK(s.src.quant() == quant && s.src.K == K, 6, "fused linears must share dtype and inner size");
    const int KC = L.Kp / 128;
    const size_t wbytes = quant ? size_t(L.Np) * L.Kp / 2 : size_t(L.Np) * L.Kp * 2;
    uint8_t* dst = b.alloc<uint8_t>(wbytes);
    uint32_t* dsb = quant ? b.alloc<uint32_t>(size_t(total_tiles) * KC * 64) : nullptr;
    for (auto& s : srcs) b.max_staging = std::max(b.max_staging, s.src.w->nbytes + (quant ? 2 * s.src.scales->nbytes : 0) + 1024);
    if (!b.dry && b.fill) {
        Q3_HIP(hipMemsetAsync(dst, 0, wbytes, nullptr));
        if (dsb) Q3_HIP(hipMemsetAsync(dsb, 0, size_t(total_tiles) * KC * 64 * 4, nullptr));
        for (auto& s : srcs) {
            const int n = s.src.N;
            Q3_HIP(hipMemcpyAsync(b.staging, s.src.w->data, s.src.w->nbytes, hipMemcpyHostToDevice, nullptr));
            if (quant) {
                uint8_t* ds = b.staging + align_up(s.src.w->nbytes, 256);
                uint8_t* db = ds + align_up(s.src.scales->nbytes, 256);
                Q3_HIP(hipMemcpyAsync(ds, s.src.scales->data, s.src.scales->nbytes, hipMemcpyHostToDevice, nullptr));
                Q3_HIP(hipMemcpyAsync(db, s.src.biases->data, s.src.biases->nbytes, hipMemcpyHostToDevice, nullptr));
                launch_tile_int4(reinterpret_cast<const uint32_t*>(b.staging), reinterpret_cast<const uint16_t*>(ds),
                                 reinterpret_cast<const uint16_t*>(db), n, K, dst, dsb, KC, s.tile_off, s.tile_stride, nullptr,
                                 s.rows_per_tile, s.row_off);
            } else {
                launch_tile_weights(reinterpret_cast<const uint16_t*>(b.staging), n, K, reinterpret_cast<uint16_t*>(dst), KC,
                                    s.tile_off, s.tile_stride, nullptr, s.rows_per_tile, s.row_off);
            }
            Q3_HIP(hipStreamSynchronize(nullptr));
        }
    }
    L.w = reinterpret_cast<const uint16_t*>(dst);
    L.sb = dsb;
    if (bias) {
        // bias padded to Np so the epilogue may read any row of a padded tile
        std::vector<uint16_t> tmp(size_t(L.Np), 0);
        std::memcpy(tmp.data(), bias->data, size_t(bias->numel()) * 2);
        L.bias = b.put<uint16_t>(tmp.data(), tmp.size());
    }
    return L;
}

// Embedding table, dequantised at load when the checkpoint holds a QuantizedEmbedding (`.scales` present:
// Qwen3.swift:1402-1406, 1419-1422): row = bf16(q * scale + bias), exactly what a dequantising gather returns.
const uint16_t* put_embedding(Builder& b, const SafetensorsDir& st, const std::string& name, int64_t* rows_out, int dim) {
    const TensorView& w = st.at(name + ".weight");
    if (!st.has(name + ".scales")) {
        Q3_CHECK(w.dtype == DType::BF16 && w.shape.size() == 2 && w.shape[1] == dim, 6, "unexpected embedding '" + name + "'");
        if (rows_out) *rows_out = w.shape[0];
        return put_bf16(b, w);
    }
    const TensorView& sc = st.at(name + ".scales");
    const TensorView& bi = st.at(name + ".biases");
    Q3_CHECK(w.shape.size() == 2 && dim % 64 == 0, 6, "unexpected quantised embedding '" + name + "'");
    const int64_t rows = w.shape[0];
    Q3_CHECK(w.dtype == DType::U32 && w.shape[1] == dim / 8, 6, "quantised embedding '" + name + "' must be uint32 [rows][dim/8] (4-bit)");
    Q3_CHECK(sc.dtype == DType::BF16 && bi.dtype == DType::BF16 && sc.shape == std::vector<int64_t>({rows, dim / 64}) &&
                 bi.shape == sc.shape,
             6, "quantised embedding '" + name + "' needs bf16 scales/biases [rows][dim/64] (group size 64)");
    if (rows_out) *rows_out = rows;
    std::vector<uint16_t> deq;
    if (!b.dry && b.fill) {
        deq.resize(size_t(rows) * dim);
        const uint32_t* q = reinterpret_cast<const uint32_t*>(w.data);
        const uint16_t* s = reinterpret_cast<const uint16_t*>(sc.data);
        const uint16_t* z = reinterpret_cast<const uint16_t*>(bi.data);
        for (int64_t r = 0; r < rows; ++r)
            for (int k = 0; k < dim; ++k) {
                const float qv = float((q[r * (dim / 8) + k / 8] >> (4 * (k & 7))) & 15u);
                const float prod = qv * bf16_to_f32_host(s[r * (dim / 64) + k / 64]);
                deq[size_t(r) * dim + k] = f32_to_bf16_host(prod + bf16_to_f32_host(z[r * (dim / 64) + k / 64]));
            }
    }
    return b.put<uint16_t>(deq.empty() ? nullptr : deq.data(), size_t(rows) * dim);
}

void build_stack(Builder& b, const MainTensors& mt, const std::string& prefix, StackW& s, int hidden,
                 const std::vector<int>& inter, int n_heads, int n_kv, int head_dim, float eps, float base, int max_pos) {
    s.hidden = hidden;
    s.n_heads = n_heads;
    s.n_kv = n_kv;
    s.head_dim = head_dim;
    s.eps = eps;
    s.rope_base = base;
    s.max_pos = max_pos;
    s.layers.resize(inter.size());
    const int qd = n_heads * head_dim, kd = n_kv * head_dim;
    for (size_t l = 0; l < inter.size(); ++l) {
        LayerW& L = s.layers[l];
        const std::string p = prefix + ".layers." + std::to_string(l);
        const int I = inter[l];
        L.inter = I;
        L.inter_p = int(align_up(size_t(I), 128));  // down_proj reads it as K
        s.max_inter_p = std::max(s.max_inter_p, L.inter_p);
        L.ln1 = put_bf16(b, mt.bf16(p + ".input_layernorm.weight", {hidden}));
        L.ln2 = put_bf16(b, mt.bf16(p + ".post_attention_layernorm.weight", {hidden}));
        L.qn = put_bf16(b, mt.bf16(p + ".self_attn.q_norm.weight", {head_dim}));
        L.kn = put_bf16(b, mt.bf16(p + ".self_attn.k_norm.weight", {head_dim}));
        const SafetensorsDir& st = mt.st;
        L.qkv = put_linear(b, {{lin_src(st, p + ".self_attn.q_proj", qd, hidden), 0, 1},
                               {lin_src(st, p + ".self_attn.k_proj", kd, hidden), qd / 16, 1},
                               {lin_src(st, p + ".self_attn.v_proj", kd, hidden), (qd + kd) / 16, 1}},
                           qd + 2 * kd, hidden, (qd + 2 * kd) / 16, nullptr);
        L.o = put_linear(b, {{lin_src(st, p + ".self_attn.o_proj", hidden, qd), 0, 1}}, hidden, qd, hidden / 16, nullptr);
        // gate and up share every 16-row tile: rows 0..7 = eight gate rows, rows 8..15 = the up rows of the same eight
        // columns, so a tile is a self-contained unit of the SwiGLU epilogue (8 outputs) and a workgroup may take any
        // number of them: 6144 columns = 768 tiles = 256 workgroups x 3 (tile PAIRS would be 384 = 1.5 per CU)
        const int it = L.inter_p / 8;  // padded so that act has inter_p columns
        L.gateup = put_linear(b, {{lin_src(st, p + ".mlp.gate_proj", I, hidden), 0, 1, 8, 0}, {lin_src(st, p + ".mlp.up_proj", I, hidden), 0, 1, 8, 8}},
                              I, hidden, it, nullptr);
        L.gateup.Np = L.inter_p;  // logical output columns (8 per tile)
        L.down = put_linear(b, {{lin_src(st, p + ".mlp.down_proj", hidden, I), 0, 1}}, hidden, I, hidden / 16, nullptr);
    }
    s.final_norm = put_bf16(b, mt.bf16(prefix + ".norm.weight", {hidden}));
    // RoPE tables (Talker.swift:42-44,103-117; CodePredictor.swift:38-39,44-56): fp32 angles,
    // cos/sin rounded to bf16; computed with the host libm exactly like the oracle.
    std::vector<uint16_t> cs((size_t)(max_pos) * head_dim), sn((size_t)(max_pos) * head_dim);
    if (!b.dry && b.fill) {
        const int half = head_dim / 2;
        for (int p = 0; p < max_pos; ++p)
            for (int i = 0; i < half; ++i) {
                const float inv = 1.0f / powf(base, float(2 * i) / float(head_dim));
                const float ang = float(p) * inv;
                const uint16_t c = f32_to_bf16_host(cosf(ang)), sv = f32_to_bf16_host(sinf(ang));
                cs[size_t(p) * head_dim + i] = c;
                cs[size_t(p) * head_dim + half + i] = c;
                sn[size_t(p) * head_dim + i] = sv;
                sn[size_t(p) * head_dim + half + i] = sv;
            }
    }
    s.rope_cos = b.put<uint16_t>(cs.data(), cs.size());
    s.rope_sin = b.put<uint16_t>(sn.data(), sn.size());
}

// ---- codec helpers -----------------------------------------------------------------------------
using TMap = std::map<std::string, HostTensor>;
const HostTensor& need(const TMap& t, const std::string& k) {
    auto it = t.find(k);
    Q3_CHECK(it != t.end(), 6, "missing speech tokenizer tensor '" + k + "'");
    return it->second;
}
const HostTensor* maybe(const TMap& t, const std::string& k) {
    auto it = t.find(k);
    return it == t.end() ? nullptr : &it->second;
}

// ---- fp16x2 form (codec_conv.hip conv_gemm_h2_kernel) ----
inline uint16_t f32_to_f16_bits(float x) {  // round to nearest even (the compiler's conversion; |x| < 65520 here)
    const _Float16 h = static_cast<_Float16>(x);
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}
inline float f16_bits_to_f32(uint16_t u) {
    _Float16 h;
    std::memcpy(&h, &u, 2);
    return static_cast<float>(h);
}
// 2^s[n] brings the largest |w| of output row n into [2^13, 2^14): hi = fp16(w 2^s) is then a normal number for every
// element within 2^-27 of the row maximum, and lo = fp16(w 2^s - hi) (about 2^-12 of the element, no further scaling)
// for every element within 2^-15 of it; smaller elements lose bits that sit 2^-38 below the row maximum.
inline std::vector<int> row_shifts(const ConvW& c, const std::vector<float>& w) {
    std::vector<int> s(size_t(c.N), 0);
    const size_t per = size_t(c.K) * c.Cin;
    for (int nn = 0; nn < c.N; ++nn) {
        float mx = 0.f;
        for (size_t i = 0; i < per; ++i) mx = std::max(mx, std::fabs(w[size_t(nn) * per + i]));
        if (mx > 0.f && std::isfinite(mx)) {
            int e;
            std::frexp(mx, &e);  // mx = f 2^e, f in [0.5, 1)
            s[size_t(nn)] = std::min(100, std::max(-100, 14 - e));
        }
    }
    return s;
}
inline void split_h2(float w, int s, uint16_t& hi, uint16_t& lo) {
    const float ws = std::ldexp(w, s);
    hi = f32_to_f16_bits(ws);
    lo = f32_to_f16_bits(ws - f16_bits_to_f32(hi));
}
void put_row_scales(Builder& b, ConvW& c, const std::vector<int>& s, bool have) {
    std::vector<float> sc;
    if (have) {
        sc.resize(size_t(c.N));
        for (int nn = 0; nn < c.N; ++nn) sc[size_t(nn)] = std::ldexp(1.0f, -s[size_t(nn)]);
    }
    c.wsc = b.put<float>(sc.empty() ? nullptr : sc.data(), size_t(c.N));
}

// w [N][K][Cin] fp32 -> [K][chunks of 32 input channels][N][2 planes][32] fp16 (zero beyond Cin) + 2^-s per row
void attach_h2(Builder& b, ConvW& c, const std::vector<float>& w) {
    if (!b.split3) return;
    const int chunks = (c.Cin + 31) / 32;
    const size_t n = size_t(c.K) * chunks * c.N * 64;
    const bool have = !b.dry && b.fill && !w.empty();
    std::vector<uint16_t> p;
    std::vector<int> s;
    if (have) {
        s = row_shifts(c, w);
        p.assign(n, 0);
        for (int nn = 0; nn < c.N; ++nn)
            for (int tap = 0; tap < c.K; ++tap)
                for (int ci = 0; ci < c.Cin; ++ci) {
                    uint16_t* d = &p[((size_t(tap) * chunks + ci / 32) * c.N + nn) * 64 + (ci % 32)];
                    split_h2(w[(size_t(nn) * c.K + tap) * c.Cin + ci], s[size_t(nn)], d[0], d[32]);
                }
    }
    c.wh = b.put<uint16_t>(p.empty() ? nullptr : p.data(), n);
    put_row_scales(b, c, s, have);
}

// conv2 of a residual unit for the fused kernel: k slot s = 8q + 4e + j of chunk m holds input channel 32m + 16e + 4q + j,
// i.e. what lane group q of an MFMA accumulator pair (tiles 2m, 2m+1) carries in registers j of tile e; shares wsc with
// attach_h2's copy
void attach_h2_perm(Builder& b, ConvW& c, const std::vector<float>& w) {
    if (!b.split3 || c.K != 1 || c.Cin % 32 != 0) return;
    const int chunks = c.Cin / 32;
    const size_t n = size_t(chunks) * c.N * 64;
    std::vector<uint16_t> p;
    if (!b.dry && b.fill && !w.empty()) {
        const std::vector<int> s = row_shifts(c, w);
        p.assign(n, 0);
        for (int nn = 0; nn < c.N; ++nn)
            for (int m = 0; m < chunks; ++m)
                for (int sl = 0; sl < 32; ++sl) {
                    const int q = sl >> 3, e = (sl >> 2) & 1, j = sl & 3;
                    const int ci = 32 * m + 16 * e + 4 * q + j;
                    uint16_t* d = &p[(size_t(m) * c.N + nn) * 64 + sl];
                    split_h2(w[size_t(nn) * c.Cin + ci], s[size_t(nn)], d[0], d[32]);
                }
    }
    c.whp = b.put<uint16_t>(p.empty() ? nullptr : p.data(), n);
}

// float16 checkpoints: w [N][K][Cin] (float16-exact values) -> [K][chunks of 32 input channels][N][32] fp16, zero beyond Cin
// (codec_conv_h1.hip conv_gemm_h1_kernel: one matrix-core product per block, no row scaling -- the values ARE float16)
void attach_h1(Builder& b, ConvW& c, const std::vector<float>& w) {
    if (!b.h1) return;
    const int chunks = (c.Cin + 31) / 32;
    const size_t n = size_t(c.K) * chunks * c.N * 32;
    std::vector<uint16_t> p;
    if (!b.dry && b.fill && !w.empty()) {
        p.assign(n, 0);
        for (int nn = 0; nn < c.N; ++nn)
            for (int tap = 0; tap < c.K; ++tap)
                for (int ci = 0; ci < c.Cin; ++ci)
                    p[((size_t(tap) * chunks + ci / 32) * c.N + nn) * 32 + (ci % 32)] = f32_to_f16_bits(w[(size_t(nn) * c.K + tap) * c.Cin + ci]);
    }
    c.w1 = b.put<uint16_t>(p.empty() ? nullptr : p.data(), n);
}

// conv2 of a residual unit for the fused float16 kernel (codec_conv_h1.hip resunit_h1_kernel): k slot s = 8q + 4e + j of chunk m
// holds input channel 32m + 16e + 4q + j -- what lane group q of an accumulator pair (tiles 2m, 2m + 1) carries in registers j of
// tile e (the order of attach_h2_perm)
void attach_h1_perm(Builder& b, ConvW& c, const std::vector<float>& w) {
    if (!b.h1 || c.K != 1 || c.Cin % 32 != 0) return;
    const int chunks = c.Cin / 32;
    const size_t n = size_t(chunks) * c.N * 32;
    std::vector<uint16_t> p;
    if (!b.dry && b.fill && !w.empty()) {
        p.assign(n, 0);
        for (int nn = 0; nn < c.N; ++nn)
            for (int m = 0; m < chunks; ++m)
                for (int sl = 0; sl < 32; ++sl) {
                    const int q = sl >> 3, e = (sl >> 2) & 1, j = sl & 3;
                    p[(size_t(m) * c.N + nn) * 32 + sl] = f32_to_f16_bits(w[size_t(nn) * c.Cin + 32 * m + 16 * e + 4 * q + j]);
                }
    }
    c.w1p = b.put<uint16_t>(p.empty() ? nullptr : p.data(), n);
}

ConvW put_conv(Builder& b, const TMap& t, const std::string& name, int dil = 1) {
    const HostTensor& w = need(t, name + ".weight");
    ConvW c;
    if (w.shape.size() == 3) {
        c.N = int(w.shape[0]);
        c.K = int(w.shape[1]);
        c.Cin = int(w.shape[2]);
    } else {
        c.N = int(w.shape[0]);
        c.K = 1;
        c.Cin = int(w.shape[1]);
    }
    c.dil = dil;
    c.w = b.put_f32(w);
    attach_h2(b, c, w.data);
    if (starts_with(name, "decoder.decoder.")) attach_h1(b, c, w.data);
    if (const HostTensor* bias = maybe(t, name + ".bias")) c.bias = b.put_f32(*bias);
    return c;
}

// rows of several [N_i][K] fp32 matrices stacked
ConvW put_linear_concat(Builder& b, const TMap& t, const std::vector<std::string>& names) {
    HostTensor cat;
    int K = 0, N = 0;
    bool have = true;
    for (auto& n : names) {
        const HostTensor& w = need(t, n + ".weight");
        K = int(w.shape[1]);
        N += int(w.shape[0]);
        have = have && !w.data.empty();
    }
    cat.shape = {N, K};
    if (have)
        for (auto& n : names) {
            const HostTensor& w = need(t, n + ".weight");
            cat.data.insert(cat.data.end(), w.data.begin(), w.data.end());
        }
    ConvW c;
    c.N = N;
    c.K = 1;
    c.Cin = K;
    c.w = b.put_f32(cat);
    attach_h2(b, c, cat.data);
    return c;
}

// ConvTransposed1d [Cout][K][Cin] (K = taps*stride) -> causal conv with N = stride*Cout, taps = K/stride:
// y[t*s + r][co] = sum_ci x[t][ci] W[co][r][ci] + x[t-1][ci] W[co][r+s][ci]  (SpeechTokenizer.swift:330-352)
ConvW put_tconv(Builder& b, const TMap& t, const std::string& name, int stride) {
    const HostTensor& w = need(t, name + ".weight");
    const int Cout = int(w.shape[0]), K = int(w.shape[1]), Cin = int(w.shape[2]);
    Q3_CHECK(K % stride == 0 && (K / stride == 1 || K / stride == 2), 6, "unsupported transposed conv geometry: " + name);
    const int taps = K / stride;
    HostTensor p;
    p.shape = {int64_t(stride) * Cout, taps, Cin};
    if (!w.data.empty()) {
        p.data.resize(size_t(p.numel()));
        for (int r = 0; r < stride; ++r)
            for (int co = 0; co < Cout; ++co)
                for (int tap = 0; tap < taps; ++tap) {
                    // causal tap index `tap` reads x[t - (taps-1-tap)]; the current sample pairs with kernel tap r
                    const int k = r + (taps - 1 - tap) * stride;
                    const float* src = &w.data[(size_t(co) * K + k) * Cin];
                    float* dst = &p.data[((size_t(r) * Cout + co) * taps + tap) * Cin];
                    std::memcpy(dst, src, size_t(Cin) * 4);
                }
    }
    ConvW c;
    c.N = stride * Cout;
    c.K = taps;
    c.Cin = Cin;
    c.dil = 1;
    c.w = b.put_f32(p);
    attach_h2(b, c, p.data);
    if (starts_with(name, "decoder.decoder.")) attach_h1(b, c, p.data);
    if (const HostTensor* bias = maybe(t, name + ".bias")) {
        HostTensor bb;
        bb.shape = {int64_t(stride) * Cout};
        if (!bias->data.empty()) {
            bb.data.resize(size_t(stride) * Cout);
            for (int r = 0; r < stride; ++r) std::memcpy(&bb.data[size_t(r) * Cout], bias->data.data(), size_t(Cout) * 4);
        }
        c.bias = b.put_f32(bb);
    }
    return c;
}

SnakeW put_snake(Builder& b, const TMap& t, const std::string& name) {
    const HostTensor& al = need(t, name + ".alpha");
    const HostTensor& be = need(t, name + ".beta");
    HostTensor ea, ib;
    ea.shape = al.shape;
    ib.shape = be.shape;
    if (!al.data.empty()) {
        ea.data.resize(al.data.size());
        ib.data.resize(be.data.size());
        for (size_t i = 0; i < al.data.size(); ++i) {  // SnakeBeta, SpeechTokenizer.swift:246-253
            ea.data[i] = expf(al.data[i]);
            ib.data[i] = 1.0f / (expf(be.data[i]) + 1e-9f);
        }
    }
    SnakeW s;
    s.C = int(al.shape[0]);
    s.ea = b.put_f32(ea);
    s.ib = b.put_f32(ib);
    if (b.h1) {
        // the same parameters the way MLX forms them on float16 arrays (SpeechTokenizer.swift:247-248, 252): exp(alpha) and
        // exp(beta) rounded to float16, + eps (a Float 1e-9 is 0 in float16), 1 / . rounded again (codec_conv_h1.hip snake_h)
        auto r16 = [](float v) { return f16_bits_to_f32(f32_to_f16_bits(v)); };
        HostTensor ea16 = ea, ib16 = ib;
        for (size_t i = 0; i < ea16.data.size(); ++i) {
            ea16.data[i] = r16(expf(r16(al.data[i])));
            ib16.data[i] = r16(1.0f / r16(expf(r16(be.data[i]))));
        }
        s.ea16 = b.put_f32(ea16);
        s.ib16 = b.put_f32(ib16);
    }
    return s;
}

void build_codec(Builder& b, const TMap& t, const CodecDecoderConfig& dc, CodecW& c) {
    const std::string q = "decoder.quantizer.";
    const HostTensor& cb0 = need(t, q + "rvq_first.vq.layers.0.codebook.embed.weight");
    Q3_CHECK(cb0.shape.size() == 2 && cb0.shape[0] >= 1 && cb0.shape[1] >= 1, 6, "unexpected semantic codebook shape");
    c.inner = int(cb0.shape[1]);
    c.cb_first_rows = int(cb0.shape[0]);
    Q3_CHECK(dc.num_semantic_quantizers == 1, 6, "only one semantic quantizer is supported");
    c.cb_first = b.put_f32(cb0);
    const int nrest = dc.num_quantizers - dc.num_semantic_quantizers;
    // a frame is 16 codes everywhere (engine, kernels, the ABI's codes[F][16]): one semantic + at most 15 acoustic tables
    Q3_CHECK(nrest >= 1 && nrest <= 15, 6, "the speech tokenizer must have 2..16 quantizers");
    c.cb_rest.resize(size_t(nrest));
    c.cb_rest_rows = INT_MAX;
    for (int i = 0; i < nrest; ++i) {
        const HostTensor& cb = need(t, q + "rvq_rest.vq.layers." + std::to_string(i) + ".codebook.embed.weight");
        // the gather indexes every table with the semantic table's row length
        Q3_CHECK(cb.shape.size() == 2 && cb.shape[0] >= 1 && cb.shape[1] == c.inner, 6, "unexpected acoustic codebook shape");
        c.cb_rest_rows = std::min(c.cb_rest_rows, int(cb.shape[0]));
        c.cb_rest[size_t(i)] = b.put_f32(cb);
    }
    c.cb_rest_dev = b.put_side<const float*>(c.cb_rest.data(), c.cb_rest.size());
    {  // fused output projection: [cd][1][inner] x2 -> [cd][1][2*inner]
        const HostTensor& w1 = need(t, q + "rvq_first.output_proj.weight");
        const HostTensor& w2 = need(t, q + "rvq_rest.output_proj.weight");
        const int cd = int(w1.shape[0]), in = int(w1.shape[2]);
        HostTensor f;
        f.shape = {cd, 1, 2 * in};
        if (!w1.data.empty()) {
            f.data.resize(size_t(cd) * 2 * in);
            for (int n = 0; n < cd; ++n) {
                std::memcpy(&f.data[size_t(n) * 2 * in], &w1.data[size_t(n) * in], size_t(in) * 4);
                std::memcpy(&f.data[size_t(n) * 2 * in + in], &w2.data[size_t(n) * in], size_t(in) * 4);
            }
        }
        c.rvq_out.N = cd;
        c.rvq_out.K = 1;
        c.rvq_out.Cin = 2 * in;
        c.rvq_out.w = b.put_f32(f);
        attach_h2(b, c.rvq_out, f.data);
    }
    c.pre_conv = put_conv(b, t, "decoder.pre_conv.conv");
    const std::string pt = "decoder.pre_transformer";
    c.t_in = put_conv(b, t, pt + ".input_proj");
    c.t_out = put_conv(b, t, pt + ".output_proj");
    c.tlayers.resize(size_t(dc.num_hidden_layers));
    for (int l = 0; l < dc.num_hidden_layers; ++l) {
        auto& L = c.tlayers[size_t(l)];
        const std::string p = pt + ".layers." + std::to_string(l);
        L.ln1 = b.put_f32(need(t, p + ".input_layernorm.weight"));
        L.ln2 = b.put_f32(need(t, p + ".post_attention_layernorm.weight"));
        L.qkv = put_linear_concat(b, t, {p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"});
        L.o = put_conv(b, t, p + ".self_attn.o_proj");
        L.o.scale = b.put_f32(need(t, p + ".self_attn_layer_scale.scale"));
        L.gateup = put_linear_concat(b, t, {p + ".mlp.gate_proj", p + ".mlp.up_proj"});
        L.down = put_conv(b, t, p + ".mlp.down_proj");
        L.down.scale = b.put_f32(need(t, p + ".mlp_layer_scale.scale"));
    }
    c.t_norm = b.put_f32(need(t, pt + ".norm.weight"));
    c.ups.resize(dc.upsampling_ratios.size());
    for (size_t i = 0; i < dc.upsampling_ratios.size(); ++i) {
        auto& U = c.ups[i];
        U.stride = dc.upsampling_ratios[i];
        const std::string p = "decoder.upsample." + std::to_string(i);
        U.tconv = put_tconv(b, t, p + ".0.conv", U.stride);
        const HostTensor& dw = need(t, p + ".1.dwconv.conv.weight");  // [C][7][1]
        Q3_CHECK(dw.shape.size() == 3 && dw.shape[1] == 7 && dw.shape[2] == 1, 6, "unexpected depthwise conv shape");
        U.dw_w = b.put_f32(dw);
        U.dw_b = b.put_f32(need(t, p + ".1.dwconv.conv.bias"));
        U.ln_w = b.put_f32(need(t, p + ".1.norm.weight"));
        U.ln_b = b.put_f32(need(t, p + ".1.norm.bias"));
        U.pw1 = put_conv(b, t, p + ".1.pwconv1");
        U.pw2 = put_conv(b, t, p + ".1.pwconv2");
        U.pw2.scale = b.put_f32(need(t, p + ".1.gamma"));
    }
    c.init_conv = put_conv(b, t, "decoder.decoder.initConv.conv");
    c.blocks.resize(dc.upsample_rates.size());
    for (size_t i = 0; i < dc.upsample_rates.size(); ++i) {
        auto& B = c.blocks[i];
        const std::string p = "decoder.decoder.block" + std::to_string(i);
        B.stride = dc.upsample_rates[i];
        B.snake = put_snake(b, t, p + ".snake");
        B.tconv = put_tconv(b, t, p + ".upsample.conv", B.stride);
        B.Cout = B.tconv.N / B.stride;
        const int dils[3] = {1, 3, 9};  // SpeechTokenizer.swift:468-470
        for (int j = 0; j < 3; ++j) {
            const std::string rp = p + ".res" + std::to_string(j + 1);
            B.res[j].act1 = put_snake(b, t, rp + ".act1");
            B.res[j].conv1 = put_conv(b, t, rp + ".conv1.conv", dils[j]);
            B.res[j].act2 = put_snake(b, t, rp + ".act2");
            B.res[j].conv2 = put_conv(b, t, rp + ".conv2.conv");
            attach_h2_perm(b, B.res[j].conv2, need(t, rp + ".conv2.conv.weight").data);
            attach_h1_perm(b, B.res[j].conv2, need(t, rp + ".conv2.conv.weight").data);
        }
    }
    c.out_snake = put_snake(b, t, "decoder.decoder.outSnake");
    const HostTensor& ow = need(t, "decoder.decoder.outConv.conv.weight");
    Q3_CHECK(ow.shape.size() == 3 && ow.shape[0] == 1, 6, "unexpected output conv shape");
    c.out_C = int(ow.shape[2]);
    c.out_w = b.put_f32(ow);
    c.out_b = b.put_f32(need(t, "decoder.decoder.outConv.conv.bias"));
}

// ---- voice-clone front end ------------------------------------------------------------------------
// A strided causal conv (k = 2r, stride r) over channels-last data is a K = 2 causal conv over the
// [T/r][r*C] view of the same buffer: W[n][2r][C] read as [n][2][r*C] needs no reordering.
ConvW put_strided_conv(Builder& b, const TMap& t, const std::string& name, int stride) {
    ConvW c = put_conv(b, t, name);
    Q3_CHECK(c.K == 2 * stride, 6, "unexpected strided conv geometry: " + name);
    c.K = 2;
    c.Cin *= stride;
    return c;
}

void build_codec_encoder(Builder& b, const TMap& t, const CodecEncoderConfig& ec, CodecEncW& e) {
    Q3_CHECK(ec.num_residual_layers == 1 && ec.upsampling_ratios.size() == 4, 6,
             "codec encoder: the reference's key mapping fixes one residual layer and four ratios (Qwen3.swift:1517-1528)");
    Q3_CHECK(!ec.use_conv_shortcut && ec.use_causal_conv && ec.audio_channels == 1, 6, "codec encoder: unsupported SEANet variant");
    Q3_CHECK(ec.num_attention_heads * 64 == ec.hidden_size && ec.num_key_value_heads == ec.num_attention_heads, 6,
             "codec encoder: attention must be heads x 64 without grouped keys");
    const std::string se = "encoder.encoder.";
    const HostTensor& iw = need(t, se + "init_conv1d.conv.conv.weight");  // [C0][K][1]
    Q3_CHECK(iw.shape.size() == 3 && iw.shape[2] == 1 && iw.shape[1] == ec.kernel_size, 6, "unexpected first SEANet conv");
    e.init_C = int(iw.shape[0]);
    e.init_K = int(iw.shape[1]);
    e.init_w = b.put_f32(iw);
    e.init_b = b.put_f32(need(t, se + "init_conv1d.conv.conv.bias"));
    e.layers.resize(4);
    int C = e.init_C;
    for (int i = 0; i < 4; ++i) {
        auto& L = e.layers[size_t(i)];
        L.ratio = ec.upsampling_ratios[size_t(3 - i)];  // ratios reversed (SpeechTokenizerEncoder.swift:417)
        L.C = C;
        const std::string p = se + "layers." + std::to_string(i);
        L.res1 = put_conv(b, t, p + ".residuals.0.block.0.conv.conv");
        L.res2 = put_conv(b, t, p + ".residuals.0.block.1.conv.conv");
        Q3_CHECK(L.res1.Cin == C && L.res1.K == ec.residual_kernel_size && L.res2.K == 1 && L.res2.N == C && L.res2.Cin == L.res1.N, 6,
                 "unexpected SEANet residual block shapes");
        L.down = put_strided_conv(b, t, p + ".downsample.conv.conv", L.ratio);
        Q3_CHECK(L.down.Cin == C * L.ratio, 6, "unexpected SEANet downsample shape");
        C = L.down.N;
    }
    e.final_conv = put_conv(b, t, se + "final_conv1d.conv.conv");
    Q3_CHECK(e.final_conv.Cin == C && e.final_conv.N == ec.hidden_size, 6, "unexpected last SEANet conv");
    e.hidden = ec.hidden_size;
    e.heads = ec.num_attention_heads;
    e.tlayers.resize(size_t(ec.num_hidden_layers));
    for (int l = 0; l < ec.num_hidden_layers; ++l) {
        auto& L = e.tlayers[size_t(l)];
        const std::string p = "encoder.encoder_transformer.transformer.layers." + std::to_string(l);
        L.ln1_w = b.put_f32(need(t, p + ".norm1.weight"));
        L.ln1_b = b.put_f32(need(t, p + ".norm1.bias"));
        L.ln2_w = b.put_f32(need(t, p + ".norm2.weight"));
        L.ln2_b = b.put_f32(need(t, p + ".norm2.bias"));
        L.qkv = put_linear_concat(b, t, {p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"});
        L.o = put_conv(b, t, p + ".self_attn.o_proj");
        L.o.scale = b.put_f32(need(t, p + ".layer_scale_1.scale"));
        L.fc1 = put_conv(b, t, p + ".gating.linear1");
        L.fc2 = put_conv(b, t, p + ".gating.linear2");
        L.fc2.scale = b.put_f32(need(t, p + ".layer_scale_2.scale"));
    }
    {  // MLXNN.RoPE(dimensions: 64, traditional: false, base) at offset 0 (SpeechTokenizerEncoder.swift:494): fp32 tables
       // of double-precision angles, as in the oracle
        e.max_T = ec.max_position_embeddings;
        HostTensor cs, sn;
        cs.shape = sn.shape = {e.max_T, 32};
        if (!b.dry && b.fill) {
            cs.data.resize(size_t(e.max_T) * 32);
            sn.data.resize(size_t(e.max_T) * 32);
            for (int i = 0; i < 32; ++i) {
                const double inv = std::pow(double(ec.rope_theta), -double(i) / 32.0);
                for (int p = 0; p < e.max_T; ++p) {
                    cs.data[size_t(p) * 32 + i] = float(std::cos(double(p) * inv));
                    sn.data[size_t(p) * 32 + i] = float(std::sin(double(p) * inv));
                }
            }
        }
        e.rope_cos = b.put_f32(cs);
        e.rope_sin = b.put_f32(sn);
    }
    e.ds = ec.downsample_stride();
    e.down = put_strided_conv(b, t, "encoder.downsample.conv.conv.conv", e.ds);
    Q3_CHECK(e.down.N == ec.hidden_size && e.down.bias == nullptr, 6, "unexpected encoder downsample conv");
    {  // the two 1x1 input projections side by side (EncoderConv1dProj, :889-903)
        const HostTensor& w1 = need(t, "encoder.quantizer.rvq_first.input_proj.weight");  // [dim][1][hidden]
        const HostTensor& w2 = need(t, "encoder.quantizer.rvq_rest.input_proj.weight");
        Q3_CHECK(w1.shape.size() == 3 && w1.shape[1] == 1 && w1.shape == w2.shape && w1.shape[2] == ec.hidden_size, 6,
                 "unexpected quantizer input projection");
        e.dim = int(w1.shape[0]);
        HostTensor cat;
        cat.shape = {2 * e.dim, 1, ec.hidden_size};
        if (!w1.data.empty()) {
            cat.data = w1.data;
            cat.data.insert(cat.data.end(), w2.data.begin(), w2.data.end());
        }
        e.rvq_in.N = 2 * e.dim;
        e.rvq_in.K = 1;
        e.rvq_in.Cin = ec.hidden_size;
        e.rvq_in.w = b.put_f32(cat);
    }
    // only the first 16 code rows reach the caller (validNumQuantizers, :957, :1055): 1 semantic + 15 acoustic
    e.n_layers = std::min(16, ec.num_quantizers);
    e.bins = ec.codebook_size;
    e.cb.resize(size_t(e.n_layers));
    e.c2.resize(size_t(e.n_layers));
    for (int j = 0; j < e.n_layers; ++j) {
        const std::string p = "encoder.quantizer." + (j == 0 ? std::string("rvq_first.vq.layers.0") : "rvq_rest.vq.layers." + std::to_string(j - 1)) +
                              ".codebook";
        const HostTensor& sum = need(t, p + ".embeddingSum");
        const HostTensor& usage = need(t, p + ".clusterUsage");
        Q3_CHECK(sum.shape.size() == 2 && sum.shape[0] == e.bins && sum.shape[1] == e.dim && usage.numel() == e.bins, 6,
                 "unexpected encoder codebook shape");
        HostTensor emb, c2;
        emb.shape = sum.shape;
        c2.shape = {e.bins};
        if (!sum.data.empty()) {  // EncoderEuclideanCodebook.updateInPlace (:738-743)
            emb.data.resize(sum.data.size());
            c2.data.resize(size_t(e.bins));
            for (int r = 0; r < e.bins; ++r) {
                const float u = std::max(usage.data[size_t(r)], 1e-5f);
                double acc = 0.0;
                for (int d = 0; d < e.dim; ++d) {
                    const float v = sum.data[size_t(r) * e.dim + d] / u;
                    emb.data[size_t(r) * e.dim + d] = v;
                    const float sq = v * v;
                    acc += double(sq);
                }
                c2.data[size_t(r)] = float(acc) / 2.0f;
            }
        }
        HostTensor embt;  // [dim][bins]: the search kernel reads one dimension of 64 consecutive codes per wave load
        embt.shape = {e.dim, e.bins};
        if (!emb.data.empty()) {
            embt.data.resize(emb.data.size());
            for (int r = 0; r < e.bins; ++r)
                for (int d = 0; d < e.dim; ++d) embt.data[size_t(d) * e.bins + r] = emb.data[size_t(r) * e.dim + d];
        }
        e.cb[size_t(j)] = b.put_f32(embt);
        e.c2[size_t(j)] = b.put_f32(c2);
    }
    e.cb_dev = b.put_side<const float*>(e.cb.data(), e.cb.size());
    e.c2_dev = b.put_side<const float*>(e.c2.data(), e.c2.size());
}

// melFilterbank (SpeakerEncoder.swift:493-550), Float arithmetic as in the Swift source
std::vector<float> mel_filterbank(int nfft, int n_mels, int sr, float fmin, float fmax) {
    const int nfreq = nfft / 2 + 1;
    const float mel_min = 2595.0f * log10f(1.0f + fmin / 700.0f), mel_max = 2595.0f * log10f(1.0f + fmax / 700.0f);
    std::vector<int> bins((size_t)(n_mels + 2));
    for (int i = 0; i <= n_mels + 1; ++i) {
        const float mel = mel_min + float(i) * (mel_max - mel_min) / float(n_mels + 1);
        const float hz = 700.0f * (powf(10.0f, mel / 2595.0f) - 1.0f);
        bins[size_t(i)] = int(floorf(float(nfft + 1) * hz / float(sr)));
    }
    std::vector<float> fb((size_t)(nfreq) * n_mels, 0.f);
    for (int m = 0; m < n_mels; ++m) {
        const int left = bins[size_t(m)], center = bins[size_t(m + 1)], right = bins[size_t(m + 2)];
        for (int k = left; k < center; ++k)
            if (k < nfreq && center > left) fb[size_t(k) * n_mels + m] = float(k - left) / float(center - left);
        for (int k = center; k < right; ++k)
            if (k < nfreq && right > center) fb[size_t(k) * n_mels + m] = float(right - k) / float(right - center);
    }
    return fb;
}

void build_speaker_encoder(Builder& b, const TMap& t, const SpeakerEncoderConfig& sc, SpeakerEncW& s) {
    Q3_CHECK(sc.enc_channels.size() == 5 && sc.enc_kernel_sizes.size() == 5 && sc.enc_dilations.size() == 5, 6,
             "speaker encoder: five stages expected (SpeakerEncoder.swift:283-297)");
    Q3_CHECK(sc.mel_dim == 128, 6, "speaker encoder: mel_dim must be 128 (Qwen3.swift:232-241 hard-codes it)");
    s.n_fft = 1024; s.hop = 256; s.nfreq = 513; s.n_mels = 128;  // extractSpeakerEmbedding, Qwen3.swift:232-241
    {  // STFT as a GEMM: row k = Hann[n] * cos(2 pi k n / N), row nfreq + k = Hann[n] * sin(...); |X|^2 ignores the sign.
       // Hann window with the (N - 1) denominator (SpeakerEncoder.swift:459-462).
        const int N = s.n_fft, rows = int(align_up(size_t(2 * s.nfreq), 4));
        HostTensor w;
        w.shape = {rows, 1, N};
        if (!b.dry && b.fill) {
            w.data.assign(size_t(rows) * N, 0.f);
            std::vector<float> win((size_t)(N));
            for (int n = 0; n < N; ++n) win[size_t(n)] = 0.5f * (1.0f - cosf(2.0f * 3.14159265358979323846f * float(n) / float(N - 1)));
            for (int k = 0; k < s.nfreq; ++k)
                for (int n = 0; n < N; ++n) {
                    const double ang = 2.0 * 3.14159265358979323846 * double((int64_t(k) * n) % N) / double(N);
                    w.data[size_t(k) * N + n] = float(double(win[size_t(n)]) * std::cos(ang));
                    w.data[size_t(s.nfreq + k) * N + n] = float(double(win[size_t(n)]) * std::sin(ang));
                }
        }
        s.dft.N = rows; s.dft.K = 1; s.dft.Cin = N;
        s.dft.w = b.put_f32(w);
    }
    {
        HostTensor fb;
        fb.shape = {s.nfreq, s.n_mels};
        if (!b.dry && b.fill) fb.data = mel_filterbank(s.n_fft, s.n_mels, 24000, 0.f, 12000.f);
        s.mel_fb = b.put_f32(fb);
    }
    const auto& ch = sc.enc_channels;
    const auto& ks = sc.enc_kernel_sizes;
    const auto& dl = sc.enc_dilations;
    s.scale = sc.enc_res2net_scale;
    s.enc_dim = sc.enc_dim;
    s.b0 = put_conv(b, t, "speaker_encoder.blocks.0.conv", dl[0]);
    Q3_CHECK(s.b0.K == ks[0] && s.b0.Cin == sc.mel_dim && s.b0.N == ch[0], 6, "unexpected speaker encoder input conv");
    for (int bi = 1; bi <= 3; ++bi) {
        auto& B = s.blocks[bi - 1];
        const std::string p = "speaker_encoder.blocks." + std::to_string(bi);
        B.C = ch[size_t(bi)];
        Q3_CHECK(B.C == ch[size_t(bi - 1)] && B.C % (4 * s.scale) == 0, 6,
                 "speaker encoder: SE-Res2Net blocks need equal widths (residual add) divisible by 4*scale");
        B.tdnn1 = put_conv(b, t, p + ".tdnn1.conv");
        for (int j = 0; j < s.scale - 1; ++j) {
            B.res.push_back(put_conv(b, t, p + ".res2net_block.blocks." + std::to_string(j) + ".conv", dl[size_t(bi)]));
            Q3_CHECK(B.res.back().K == ks[size_t(bi)] && B.res.back().Cin == B.C / s.scale, 6, "unexpected Res2Net conv shape");
        }
        B.tdnn2 = put_conv(b, t, p + ".tdnn2.conv");
        B.se1 = put_conv(b, t, p + ".se_block.conv1");
        B.se2 = put_conv(b, t, p + ".se_block.conv2");
    }
    s.mfa = put_conv(b, t, "speaker_encoder.mfa.conv", dl[4]);
    Q3_CHECK(s.mfa.K == ks[4] && s.mfa.Cin == ch[1] + ch[2] + ch[3] && s.mfa.N == ch[4], 6, "unexpected speaker encoder MFA conv");
    s.asp_tdnn = put_conv(b, t, "speaker_encoder.asp.tdnn.conv");
    s.asp_conv = put_conv(b, t, "speaker_encoder.asp.conv");
    s.fc = put_conv(b, t, "speaker_encoder.fc");
    Q3_CHECK(s.asp_tdnn.Cin == 3 * ch[4] && s.asp_conv.N == ch[4] && s.fc.Cin == 2 * ch[4] && s.fc.N == sc.enc_dim, 6,
             "unexpected speaker encoder pooling shapes");
}

int64_t linear_bytes(const LinearW& L) {  // weight bytes streamed per use: bf16, or 4 bit + {scale,bias} per 64
    return L.sb ? int64_t(L.N) * L.K / 2 + int64_t(L.N) * (L.K / 64) * 4 : int64_t(L.N) * L.K * 2;
}

void build_all(Builder& b, Model& m, const SafetensorsDir& main, const TMap* codec_t, const TMap* spk_t, const LoadOptions& opt) {
    const ModelConfig& cfg = m.cfg;
    const TalkerConfig& t = cfg.talker;
    MainTensors mt{main};
    Q3_CHECK(t.head_dim == kHeadDim && t.cp.head_dim == kHeadDim, 6, "head_dim must be 128");
    const int H = t.hidden_size, TH = t.text_hidden_size;
    m.codec_emb = put_embedding(b, main, "talker.model.codec_embedding", nullptr, H);
    // fewer rows than text_vocab_size when the vocabulary is pruned (docs/paper.tex:160-178)
    m.text_emb = put_embedding(b, main, "talker.model.text_embedding", &m.text_emb_rows, TH);
    if (main.has("talker.model.text_token_map")) {  // Qwen3.swift:1434-1444
        const TensorView& tm = main.at("talker.model.text_token_map");
        Q3_CHECK(tm.dtype == DType::I32, 6, "text_token_map must be int32");
        // ids are validated against text_vocab_size per request; the map must cover them and stay inside the compact table
        Q3_CHECK(tm.numel() >= t.text_vocab_size, 6, "text_token_map is shorter than text_vocab_size");
        if (!b.dry) {
            const int32_t* mp = reinterpret_cast<const int32_t*>(tm.data);
            for (int64_t i = 0; i < tm.numel(); ++i)
                Q3_CHECK(mp[i] >= 0 && mp[i] < m.text_emb_rows, 6, "text_token_map points outside the text embedding table");
        }
        m.token_map = b.put<int32_t>(reinterpret_cast<const int32_t*>(tm.data), size_t(tm.numel()));
    }
    std::vector<int> inter;
    for (int l = 0; l < t.num_hidden_layers; ++l) inter.push_back(t.inter(l));
    build_stack(b, mt, "talker.model", m.talker, H, inter, t.num_attention_heads, t.num_key_value_heads, t.head_dim,
                t.rms_norm_eps, t.rope_theta, opt.max_pos_talker);
    const TensorView& f1b = mt.bf16("talker.text_projection.linear_fc1.bias", {TH});
    const TensorView& f2b = mt.bf16("talker.text_projection.linear_fc2.bias", {H});
    m.fc1 = put_linear(b, {{lin_src(main, "talker.text_projection.linear_fc1", TH, TH), 0, 1}}, TH, TH, TH / 16, &f1b);
    m.fc2 = put_linear(b, {{lin_src(main, "talker.text_projection.linear_fc2", H, TH), 0, 1}}, H, TH, H / 16, &f2b);
    m.codec_head = put_linear(b, {{lin_src(main, "talker.codec_head", t.vocab_size, H), 0, 1}}, t.vocab_size, H, t.vocab_size / 16,
                              nullptr);
    Q3_CHECK(t.has_code_predictor, 6, "code_predictor_config is required");
    const CodePredictorConfig& cp = t.cp;
    const int CH = cp.hidden_size;
    m.has_cp_proj = (CH != H);  // CodePredictor.swift:295-299
    if (m.has_cp_proj) {
        const TensorView& pb = mt.bf16("talker.code_predictor.small_to_mtp_projection.bias", {CH});
        m.cp_proj = put_linear(b, {{lin_src(main, "talker.code_predictor.small_to_mtp_projection", CH, H), 0, 1}}, CH, H, CH / 16, &pb);
    }
    const int ng = cp.num_code_groups - 1;
    m.cp_emb.resize(size_t(ng));
    m.lm_head.resize(size_t(ng));
    for (int i = 0; i < ng; ++i) {
        m.cp_emb[size_t(i)] = put_embedding(b, main, "talker.code_predictor.model.codec_embedding." + std::to_string(i), nullptr, H);
        m.lm_head[size_t(i)] = put_linear(b, {{lin_src(main, "talker.code_predictor.lm_head." + std::to_string(i), cp.vocab_size, CH), 0, 1}},
                                          cp.vocab_size, CH, cp.vocab_size / 16, nullptr);
    }
    m.cp_emb_dev = b.put_side<const uint16_t*>(m.cp_emb.data(), m.cp_emb.size());
    std::vector<int> cp_inter((size_t)(cp.num_hidden_layers), cp.intermediate_size);
    build_stack(b, mt, "talker.code_predictor.model", m.cp, CH, cp_inter, cp.num_attention_heads, cp.num_key_value_heads,
                cp.head_dim, cp.rms_norm_eps, cp.rope_theta, 64);
    if (codec_t) {
        b.split3 = true;
        // a float16 speech tokenizer ("lite" checkpoints): the MainDecoder runs in float16 like the reference's (codec_conv_h1.hip)
        const HostTensor* ic = maybe(*codec_t, "decoder.decoder.initConv.conv.weight");
        b.h1 = ic && ic->f16;
        m.codec.f16_main = b.h1;
        build_codec(b, *codec_t, cfg.codec, m.codec);
        b.split3 = false;
        b.h1 = false;
        m.has_codec = true;
        if (cfg.has_codec_encoder) {  // SpeechTokenizer.swift:808-812
            build_codec_encoder(b, *codec_t, cfg.codec_enc, m.codec_enc);
            m.has_codec_encoder = true;
        }
    }
    if (spk_t) {  // Qwen3.swift:55-57
        Q3_CHECK(cfg.speaker.enc_dim == H, 6, "speaker encoder: enc_dim must equal the talker hidden size (Qwen3.swift:553-558)");
        build_speaker_encoder(b, *spk_t, cfg.speaker, m.speaker);
        m.has_speaker_encoder = true;
    }
    // distinct weight bytes of one frame step
    int64_t wb = linear_bytes(m.codec_head);
    auto gu = [](const LinearW& g) { LinearW t = g; t.N = 2 * g.N; return linear_bytes(t); };
    for (auto& L : m.talker.layers) wb += linear_bytes(L.qkv) + linear_bytes(L.o) + gu(L.gateup) + linear_bytes(L.down);
    for (auto& L : m.cp.layers) wb += linear_bytes(L.qkv) + linear_bytes(L.o) + gu(L.gateup) + linear_bytes(L.down);
    for (auto& L : m.lm_head) wb += linear_bytes(L);
    if (m.has_cp_proj) wb += linear_bytes(m.cp_proj);
    m.step_weight_bytes = wb;
}

}  // namespace

std::unique_ptr<Model> load_model(const std::string& dir, const LoadOptions& opt) {
    auto m = std::make_unique<Model>();
    m->device = opt.device;
    Q3_HIP(hipSetDevice(opt.device));
    {
        std::string txt = read_file(dir + "/config.json");  // Qwen3.swift:1386-1388
        Json j = JsonParser(txt.data(), txt.size()).parse();
        m->cfg.parse(j);
    }
    Q3_CHECK(m->cfg.has_talker, 1, "Talker config is required");  // fatalError at Qwen3.swift:48-50
    m->cfg.validate();
    if (m->cfg.has_quantization)  // MLX affine quantisation as shipped with the "lite" checkpoints (docs/paper.tex:232,239)
        Q3_CHECK(m->cfg.quant_bits == 4 && m->cfg.quant_group_size == 64, 6, "only 4-bit, group-size-64 quantisation is supported");
    SafetensorsDir main;
    main.open_dir(dir);  // Qwen3.swift:1391-1399
    SafetensorsDir st;
    TMap codec_t;
    const std::string st_dir = dir + "/speech_tokenizer";
    const bool have_codec = file_exists(st_dir + "/config.json");  // Qwen3.swift:1462-1463
    if (have_codec) {
        std::string txt = read_file(st_dir + "/config.json");
        Json j = JsonParser(txt.data(), txt.size()).parse();
        m->cfg.parse_speech_tokenizer(j);
        Q3_CHECK(m->cfg.has_codec, 1, "Decoder config is required");  // fatalError at SpeechTokenizer.swift:804
        m->cfg.validate();
        st.open_dir(st_dir);
        codec_t = sanitize_speech_tokenizer(st, !opt.skip_tensor_data);
    }
    TMap spk_t;
    if (m->cfg.has_speaker_encoder) spk_t = speaker_encoder_tensors(main, !opt.skip_tensor_data);
    const TMap* spk_p = m->cfg.has_speaker_encoder ? &spk_t : nullptr;
    Builder dry;
    dry.dry = true;
    {
        Model scratch;
        scratch.cfg = m->cfg;
        build_all(dry, scratch, main, have_codec ? &codec_t : nullptr, spk_p, opt);
    }
    m->arena_bytes = align_up(dry.off, 256);
    Q3_HIP(hipMalloc(reinterpret_cast<void**>(&m->arena), m->arena_bytes));
    Builder real;
    real.dry = false;
    real.fill = !opt.skip_tensor_data;
    real.base = m->arena;
    real.side = &m->side_allocs;
    if (real.fill) {
        Q3_HIP(hipMalloc(reinterpret_cast<void**>(&real.staging), dry.max_staging));
    }
    try {
        build_all(real, *m, main, have_codec ? &codec_t : nullptr, spk_p, opt);
    } catch (...) {
        if (real.staging) (void)hipFree(real.staging);
        throw;
    }
    Q3_HIP(hipDeviceSynchronize());
    if (real.staging) Q3_HIP(hipFree(real.staging));
    Q3_CHECK(real.off == dry.off, 7, "internal error: arena passes disagree");
    return m;
}

}  // namespace q3
