#include <cstdlib>
#include <utility>
#include <vector>
#include "side.h"
#include "chain.h"

namespace {
// Leaf work (weight-gradient products, bias sums) goes to low-priority side streams, one fork "session" per call of
// side_fork().  Sessions are independent of each other (distinct gradient tensors, scratch private to a session), so
// they rotate over kMaxSide streams: the tail of one 256-workgroup product (skewed finishers, atomics) overlaps the
// start of the next (4.345 -> 4.231 ms per training step with two streams instead of one).
constexpr int kMaxSide = 3;           // measured: 1 -> 4.35, 2 -> 4.24, 3 -> 4.29 ms per step; 4 -> 7.57 (the queues get multiplexed)
hipStream_t g_sides[kMaxSide] = {nullptr, nullptr, nullptr};
bool g_dirty[kMaxSide] = {false, false, false};          // something was queued on stream i since the last join
int g_nside = 0, g_turn = 0;
bool g_init = false;
int g_enabled = -1;
bool g_defer = false;                 // side_join() is a no-op; the caller joins explicitly (side_join_now)
std::vector<hipEvent_t> g_pool;
std::vector<std::pair<const void*, int>> g_dests;         // (tensor, side stream) accumulations queued since the last join
size_t g_next = 0;

hipEvent_t next_event() {
    if (g_pool.empty()) {
        // These events only order streams of ONE device against each other: the system-scope fence an event performs by
        // default when it completes (cache write-back + invalidate, ~7 us of bubble on the recording stream per fork,
        // profiles/r02_v_timeline_full_step.txt) buys nothing here.  (The switch that restored the default was removed in round 5.)
        constexpr bool fence = false;
        g_pool.resize(64);
        for (auto& e : g_pool)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming | (fence ? 0u : hipEventDisableSystemFence)) != hipSuccess &&
                hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
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
namespace { int g_active = 0; }
void side_set_active(int n) { g_active = n > 0 ? n : 0; }
bool side_is(hipStream_t s) {
    for (int i = 0; i < g_nside; ++i)
        if (s == g_sides[i]) return true;
    return false;
}

namespace { void twin_create(); }
hipStream_t side_fork(hipStream_t main_stream) {
    if (!side_enabled()) return main_stream;
    if (!g_init) {
        g_init = true;
        // (the twin stream first, whoever asks first: see twin_fork.  a switch of round 4 created it only when a two-layer LSTM pipeline asked)
        constexpr bool eager = true;
        if (eager) twin_create();
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        int want = 2;                                              // (two rotating side streams: the header's measurement)
        want = want < 1 ? 1 : (want > kMaxSide ? kMaxSide : want);
        if (g_active > 0 && g_active < want) want = g_active;      // (key 13 set before the first fork: no more streams than will be used)
        for (int i = 0; i < want; ++i) {
            if (hipStreamCreateWithPriority(&g_sides[g_nside], hipStreamNonBlocking, lo) != hipSuccess) break;
            ++g_nside;
        }
    }
    if (g_nside == 0) return main_stream;
    const int i = g_turn++ % (g_active > 0 && g_active < g_nside ? g_active : g_nside);
    hipEvent_t e = next_event();
    if (!e || hipEventRecord(e, main_stream) != hipSuccess || hipStreamWaitEvent(g_sides[i], e, 0) != hipSuccess)
        return main_stream;
    g_dirty[i] = true;
    return g_sides[i];
}

// (g_dests is forgotten where the CALLER's stream has just been ordered behind all side work -- the join at the end of a
// *_bwd call, the trainer's switch of the join mode around a step -- not in side_join_now, which data-parallel training
// also uses to order a third stream)
void side_set_defer(int on) { g_defer = on != 0; g_dests.clear(); }
int side_join(hipStream_t main_stream) {
    if (g_defer) return 0;
    g_dests.clear();
    return side_join_now(main_stream);
}

int side_join_now(hipStream_t main_stream) {
    for (int i = 0; i < g_nside; ++i) {
        if (!g_dirty[i]) continue;
        g_dirty[i] = false;
        hipEvent_t e = next_event();
        if (!e || hipEventRecord(e, g_sides[i]) != hipSuccess || hipStreamWaitEvent(main_stream, e, 0) != hipSuccess) return -2;
    }
    return 0;
}

// (`other` may be a stream whose kernels talk to OTHER devices -- the data-parallel exchange's all-reduce stream --, so these
//  events keep the default system-scope fence; the pool's no-fence events only ever order streams of one device's own work)
namespace {
std::vector<hipEvent_t> g_fenced;
size_t g_fenced_next = 0;
hipEvent_t next_fenced_event() {
    if (g_fenced.empty()) {
        g_fenced.resize(16);
        for (auto& e : g_fenced)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
    }
    hipEvent_t e = g_fenced[g_fenced_next];
    g_fenced_next = (g_fenced_next + 1) % g_fenced.size();
    return e;
}
}  // namespace
int side_wait_on(hipStream_t other) {
    for (int i = 0; i < g_nside; ++i) {
        if (!g_dirty[i]) continue;
        hipEvent_t e = next_fenced_event();
        if (!e || hipEventRecord(e, g_sides[i]) != hipSuccess || hipStreamWaitEvent(other, e, 0) != hipSuccess) return -2;
    }
    return 0;
}

// The twin stream is created BEFORE the side streams, whichever of the two is asked for first: the runtime deals its hardware
// queues to streams in creation order, and AnticipationRNN's layer pipelines (caller's stream + twin, both running chains) measured
// 6.8 ms per step in a process whose first library call was theirs and 7.8 ms behind a MeasureVAE step, which creates the side
// streams first (tools/arnn_order.py).
namespace {
hipStream_t g_twin = nullptr; bool g_twin_tried = false, g_twin_dirty = false;
void twin_create() {
    if (g_twin_tried) return;
    g_twin_tried = true;
    if (hipStreamCreateWithFlags(&g_twin, hipStreamNonBlocking) != hipSuccess) g_twin = nullptr;
}
}  // namespace
hipStream_t twin_stream() { twin_create(); return g_twin; }
hipStream_t twin_fork(hipStream_t main_stream) {
    twin_create();
    hipEvent_t e = g_twin ? next_event() : nullptr;
    if (!e || hipEventRecord(e, main_stream) != hipSuccess || hipStreamWaitEvent(g_twin, e, 0) != hipSuccess) return main_stream;
    g_twin_dirty = true;
    return g_twin;
}
int twin_join(hipStream_t main_stream) {
    if (!g_twin_dirty) return 0;
    g_twin_dirty = false;
    hipEvent_t e = next_event();
    if (!e || hipEventRecord(e, g_twin) != hipSuccess || hipStreamWaitEvent(main_stream, e, 0) != hipSuccess) return -2;
    return 0;
}

int stream_wait(hipStream_t waiter, hipStream_t on) {
    if (waiter == on) return 0;
    hipEvent_t e = next_event();
    return e && hipEventRecord(e, on) == hipSuccess && hipStreamWaitEvent(waiter, e, 0) == hipSuccess ? 0 : -2;
}

int side_order_dest(const void* dest, hipStream_t s) {
    int me = -1;
    for (int i = 0; i < g_nside; ++i)
        if (s == g_sides[i]) me = i;
    if (me < 0) return 0;
    bool known = false;
    unsigned waited = 0;
    for (const auto& d : g_dests) {
        if (d.first != dest) continue;
        if (d.second == me) known = true;
        else if (!(waited & (1u << d.second))) {
            waited |= 1u << d.second;
            if (stream_wait(s, g_sides[d.second]) != 0) return -2;
        }
    }
    if (!known) g_dests.emplace_back(dest, me);
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
unsigned* chain_dev_status() {
    static unsigned* p = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void* q = nullptr;
        // (word 0: the status; from kChainDiagWord on: kChainDiagBytes of scratch for instrumented builds, inet_debug_read; behind it
        //  the slow-wait recorder, chain::kRecWord)
        const size_t n = kChainStatusAreaBytes;
        const unsigned polls = chain::kRecDefaultPolls;
        if (hipMalloc(&q, n) == hipSuccess && hipMemset(q, 0, n) == hipSuccess &&
            hipMemcpy(static_cast<unsigned*>(q) + chain::kRecWord + 1, &polls, 4, hipMemcpyHostToDevice) == hipSuccess)
            p = static_cast<unsigned*>(q);
    }
    return p;
}
int chain_status_reset() {
    if (unsigned* h = chain_host_status()) __atomic_store_n(h, 0u, __ATOMIC_RELAXED);
    unsigned* d = chain_dev_status();
    if (d && (hipDeviceSynchronize() != hipSuccess || hipMemset(d, 0, 4) != hipSuccess)) return -2;
    return 0;
}
// ---- token range status (chain.h) ----
unsigned* token_host_status() {
    static unsigned* p = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void* q = nullptr;
        if (hipHostMalloc(&q, 64, hipHostMallocMapped) == hipSuccess) { p = static_cast<unsigned*>(q); *p = 0; }
    }
    return p;
}
namespace { int g_fault = 0; }
void chain_arm_fault(int on) { g_fault = on ? 1 : 0; }
int chain_take_fault() { const int f = g_fault; g_fault = 0; return f; }
int chain_capacity() {
    static const int forced = [] { const char* v = std::getenv("INET_CHAIN_CUS"); return v ? std::atoi(v) : 0; }();
    if (forced > 0) return forced;
    static int cached[64];                         // per device ordinal; 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;   // no device here (CPU-side size queries)
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached[dev] = cus;
    }
    return cached[dev];
}
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
