// debug_env.cc -- the launchers' diagnostic switches (kernels.h DebugEnv): one read of the environment, repeated on request.
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "kernels.h"

namespace q3 {

namespace {
std::mutex g_mu;
std::atomic<const DebugEnv*> g_env{nullptr};

bool flag(const char* name) { return std::getenv(name) != nullptr; }
int number(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e && *e ? std::atoi(e) : dflt;
}

const DebugEnv* read_env() {
    DebugEnv* d = new DebugEnv{};  // never freed: a reader on another thread may still hold the previous one
    d->gemm_no_row_split = flag("Q3TTS_GEMM_NO_ROW_SPLIT");
    d->gemm_one_pair = flag("Q3TTS_GEMM_ONE_PAIR");
    d->no_tall_gemm = flag("Q3TTS_NO_TALL_GEMM");
    d->tall_shape = number("Q3TTS_TALL_SHAPE", 0);
    d->chunk_qsplit = number("Q3TTS_CHUNK_QSPLIT", 0);
    d->conv_no_pw = flag("Q3TTS_CONV_NO_PW");
    d->nt_off = std::getenv("Q3TTS_NT") && number("Q3TTS_NT", 1) == 0;
    d->prefetch = number("Q3TTS_PF", 1);
    d->pf_budget_kb = number("Q3TTS_PF_BUDGET_KB", 2560);
    d->pf_ahead = number("Q3TTS_PF_AHEAD", 3);
    d->pf_skip = number("Q3TTS_PF_SKIP", 0);
    return d;
}
}  // namespace

const DebugEnv& debug_env() {
    const DebugEnv* e = g_env.load(std::memory_order_acquire);
    if (!e) {
        std::lock_guard<std::mutex> lk(g_mu);
        e = g_env.load(std::memory_order_acquire);
        if (!e) {
            e = read_env();
            g_env.store(e, std::memory_order_release);
        }
    }
    return *e;
}

void debug_env_reload() {
    std::lock_guard<std::mutex> lk(g_mu);
    g_env.store(read_env(), std::memory_order_release);
}

}  // namespace q3
