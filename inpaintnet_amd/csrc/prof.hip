#include <cstdio>
#include <string>
#include <vector>
#include "prof.h"
#include "../../include/inpaintnet_hip.h"

namespace {
struct Rec { int cls; double flops, bytes; hipEvent_t a, b; std::string label; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;

hipEvent_t get_event() {
    if (g_pool_next == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_next++];
}
}  // namespace

ProfScope::ProfScope(int cls, double flops, hipStream_t stream, const char* label, double bytes) : idx(-1), s(stream) {
    if (!g_on) return;
    Rec r{cls, flops, bytes, get_event(), get_event(), label ? label : ""};
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    idx = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_recs[idx].b, s);
}

extern "C" {
int inet_prof_enable(int on) {
    g_on = on != 0;
    if (g_on) { g_recs.clear(); g_pool_next = 0; }
    return 0;
}
int inet_prof_dump(const char* path) {
    FILE* f = std::fopen(path, "w");
    if (!f) return -1;
    std::fprintf(f, "class,label,us,gflop,mbytes\n");
    for (const Rec& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) { std::fclose(f); return -2; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        std::fprintf(f, "%d,%s,%.3f,%.4f,%.4f\n", r.cls, r.label.c_str(), t * 1e3, r.flops * 1e-9, r.bytes * 1e-6);
    }
    std::fclose(f);
    return 0;
}
int inet_prof_read(int cls, int64_t* launches, double* total_ms, double* total_flops) {
    if (cls < 0 || cls >= PROF_NCLASS) return -1;
    int64_t n = 0; double ms = 0, fl = 0;
    for (const Rec& r : g_recs) {
        if (r.cls != cls) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) return -2;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return -2;
        ++n; ms += t; fl += r.flops;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    return 0;
}
}
