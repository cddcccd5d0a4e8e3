// safetensors.h -- mmap reader for *.safetensors directories (replaces MLX.loadArrays at
// /root/reference/Sources/Qwen3TTS/Models/Qwen3.swift:1391-1399, 1473-1480). Header-only.
#pragma once
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "json.h"

namespace q3 {

enum class DType { F32, BF16, F16, I32, U32, I64, U8, Unknown };

inline size_t dtype_size(DType d) {
    switch (d) {
        case DType::F32: case DType::I32: case DType::U32: return 4;
        case DType::BF16: case DType::F16: return 2;
        case DType::I64: return 8;
        case DType::U8: return 1;
        default: return 0;
    }
}

struct TensorView {
    DType dtype = DType::Unknown;
    std::vector<int64_t> shape;
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

class SafetensorsDir {
  public:
    SafetensorsDir() = default;
    SafetensorsDir(const SafetensorsDir&) = delete;
    SafetensorsDir& operator=(const SafetensorsDir&) = delete;
    ~SafetensorsDir() {
        for (auto& m : maps_) {
            munmap(m.base, m.size);
        }
    }

    // Loads every *.safetensors in `dir` (sorted by name; later files override earlier keys, like
    // the reference's weights.merge(...) { _, new in new }).
    void open_dir(const std::string& dir) {
        std::vector<std::string> files;
        DIR* d = opendir(dir.c_str());
        Q3_CHECK(d != nullptr, 6, "cannot open directory " + dir);
        while (dirent* e = readdir(d)) {
            std::string n = e->d_name;
            if (n.size() > 12 && n.substr(n.size() - 12) == ".safetensors") files.push_back(dir + "/" + n);
        }
        closedir(d);
        std::sort(files.begin(), files.end());
        for (auto& f : files) open_file(f);
    }

    bool has(const std::string& k) const { return t_.count(k) != 0; }
    const TensorView& at(const std::string& k) const {
        auto it = t_.find(k);
        Q3_CHECK(it != t_.end(), 6, "missing tensor '" + k + "' in checkpoint");
        return it->second;
    }
    const std::map<std::string, TensorView>& all() const { return t_; }

  private:
    struct Map { void* base; size_t size; };
    std::vector<Map> maps_;
    std::map<std::string, TensorView> t_;

    static DType parse_dtype(const std::string& s) {
        if (s == "F32") return DType::F32;
        if (s == "BF16") return DType::BF16;
        if (s == "F16") return DType::F16;
        if (s == "I32") return DType::I32;
        if (s == "U32") return DType::U32;
        if (s == "I64") return DType::I64;
        if (s == "U8") return DType::U8;
        return DType::Unknown;
    }

    void open_file(const std::string& path) {
        int fd = ::open(path.c_str(), O_RDONLY);
        Q3_CHECK(fd >= 0, 6, "cannot open " + path);
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 8) {
            ::close(fd);
            throw Error(6, "truncated safetensors file " + path);
        }
        size_t size = size_t(st.st_size);
        void* base = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        Q3_CHECK(base != MAP_FAILED, 6, "mmap failed for " + path);
        maps_.push_back({base, size});
        const uint8_t* p = static_cast<const uint8_t*>(base);
        uint64_t hlen;
        std::memcpy(&hlen, p, 8);
        Q3_CHECK(hlen <= size - 8, 6, "bad safetensors header in " + path);  // (not 8 + hlen <= size: that sum wraps)
        const uint64_t data_bytes = uint64_t(size) - 8 - hlen;
        Json hdr = JsonParser(reinterpret_cast<const char*>(p + 8), size_t(hlen)).parse();
        const uint8_t* data = p + 8 + hlen;
        for (auto& kv : hdr.obj) {
            if (kv.first == "__metadata__") continue;
            TensorView tv;
            tv.dtype = parse_dtype(kv.second.s("dtype", ""));
            const Json* sh = kv.second.get("shape");
            // every number of the header is checked as a number of THIS file before it becomes an integer: a dimension or an
            // offset beyond the file (or negative, or 1e999) is a damaged checkpoint, and the element count may not wrap
            uint64_t numel = 1;
            if (sh) {
                Q3_CHECK(sh->kind == Json::Arr, 6, "bad shape for " + kv.first);
                for (auto& e : sh->arr) {
                    Q3_CHECK(e.kind == Json::Num, 6, "bad shape for " + kv.first);
                    const int64_t dim = Json::to_int(e.num, 0, 1e15, "the shape of " + kv.first);
                    Q3_CHECK(!__builtin_mul_overflow(numel, uint64_t(dim), &numel), 6, "bad shape for " + kv.first);
                    tv.shape.push_back(dim);
                }
            }
            const Json* off = kv.second.get("data_offsets");
            Q3_CHECK(off && off->kind == Json::Arr && off->arr.size() == 2 && off->arr[0].kind == Json::Num && off->arr[1].kind == Json::Num,
                     6, "bad data_offsets for " + kv.first);
            const uint64_t a = uint64_t(Json::to_int(off->arr[0].num, 0, 9.0e18, "data_offsets of " + kv.first));
            const uint64_t b = uint64_t(Json::to_int(off->arr[1].num, 0, 9.0e18, "data_offsets of " + kv.first));
            Q3_CHECK(b <= data_bytes && a <= b, 6, "tensor out of file bounds: " + kv.first);
            tv.data = data + a;
            tv.nbytes = size_t(b - a);
            uint64_t want = 0;
            Q3_CHECK(tv.dtype == DType::Unknown ||
                         (!__builtin_mul_overflow(numel, uint64_t(dtype_size(tv.dtype)), &want) && want == tv.nbytes),
                     6, "shape/bytes mismatch for " + kv.first);
            t_[kv.first] = tv;
        }
    }
};

}  // namespace q3
