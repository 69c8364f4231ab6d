#include <cstdlib>
#include <vector>
#include "side.h"
#include "chain.h"

namespace {
hipStream_t g_side = nullptr;
bool g_init = false;
int g_enabled = -1;
bool g_dirty = false;                 // something was queued on the side stream since the last join
bool g_defer = false;                 // side_join() is a no-op; the caller joins explicitly (side_join_now)
std::vector<hipEvent_t> g_pool;
size_t g_next = 0;

hipEvent_t next_event() {
    if (g_pool.empty()) {
        g_pool.resize(64);
        for (auto& e : g_pool)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
    }
    hipEvent_t e = g_pool[g_next];
    g_next = (g_next + 1) % g_pool.size();
    return e;
}
}  // namespace

int side_enabled() {
    if (g_enabled < 0) {
        const char* v = std::getenv("INET_SIDE_STREAM");
        g_enabled = (v && v[0] == '0') ? 0 : 1;
    }
    return g_enabled;
}
void side_set_enabled(int on) { g_enabled = on ? 1 : 0; }
bool side_is(hipStream_t s) { return g_side != nullptr && s == g_side; }

hipStream_t side_fork(hipStream_t main_stream) {
    if (!side_enabled()) return main_stream;
    if (!g_init) {
        g_init = true;
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&g_side, hipStreamNonBlocking, lo) != hipSuccess) g_side = nullptr;
    }
    if (!g_side) return main_stream;
    hipEvent_t e = next_event();
    if (!e || hipEventRecord(e, main_stream) != hipSuccess || hipStreamWaitEvent(g_side, e, 0) != hipSuccess)
        return main_stream;
    g_dirty = true;
    return g_side;
}

void side_set_defer(int on) { g_defer = on != 0; }
int side_join(hipStream_t main_stream) { return g_defer ? 0 : side_join_now(main_stream); }

int side_join_now(hipStream_t main_stream) {
    if (!g_dirty || !g_side) return 0;
    g_dirty = false;
    hipEvent_t e = next_event();
    if (!e) return -2;
    if (hipEventRecord(e, g_side) != hipSuccess) return -2;
    if (hipStreamWaitEvent(main_stream, e, 0) != hipSuccess) return -2;
    return 0;
}

// ---- chain kernels on/off (chain.h) ----
namespace { int g_chain = -1; }
int chain_enabled() {
    if (g_chain < 0) {
        const char* v = std::getenv("INET_CHAIN");
        g_chain = (v && v[0] == '0') ? 0 : 1;
    }
    return g_chain;
}
void chain_set_enabled(int on) { g_chain = on ? 1 : 0; }
unsigned* chain_host_status() {
    static unsigned* p = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void* q = nullptr;
        if (hipHostMalloc(&q, 64, hipHostMallocMapped) == hipSuccess) { p = static_cast<unsigned*>(q); *p = 0; }
    }
    return p;
}
