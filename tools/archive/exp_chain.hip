// Standalone feasibility experiment for a per-layer "chain kernel" (DESIGN.md section 7, next step 1):
// what does it cost to hand a hidden state from 32 producer workgroups to the same 32 workgroups INSIDE a kernel,
// compared with the ~2 us dispatch + ~2 us cold read of a dependent launch?
//   hipcc --offload-arch=gfx950 -O3 tools/exp_chain.hip -o build/exp_chain
// 256 workgroups = 8 groups of 32.  Per step every workgroup writes its 4 KB slice (64 rows x 16 columns) of the
// group's 128 KB state, the group synchronises on a counter, then every workgroup reads the whole 128 KB.
//   V0 plain stores/loads, no synchronisation            (data movement only; results are garbage by design)
//   V1 16-byte sc1 stores + sc1 loads + relaxed agent-scope counter, group = blockIdx % 8   (same XCD if the
//      dispatcher deals consecutive workgroups round-robin over the XCDs)
//   V2 as V1, group = blockIdx / 32                        (every group spans all XCDs)
//   V3 plain stores, __threadfence(), counter, plain loads (the textbook protocol)
//   V4 plain stores (the line stays in the XCD's L2) + counter + sc1 loads, group = blockIdx % 8
//   V5 plain stores + counter + nt loads, group = blockIdx % 8
// Every reader checks the values it sees (step-stamped), so a protocol that is fast but wrong shows up as mismatches.
// All spins are bounded: a broken protocol reports "timeout", it cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int GROUPS = 8, PER = 32, STEPS = 24;
constexpr int STATE_F = 64 * 512;          // floats per group state (128 KB)
constexpr int SLICE_F = STATE_F / PER;     // 1024 floats = 4 KB per workgroup

__device__ __forceinline__ void st_sc1(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// issue only; wait_all() below makes the values usable
__device__ __forceinline__ f32x4 ld_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f32x4 ld_nt(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void wait_all(f32x4 (&v)[N]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));      // uses cannot move above the wait
}

struct Out { unsigned long long t_begin, t_end; unsigned mismatches, timeouts, xcc; };

template <int V>
__global__ __launch_bounds__(256) void chain(float* state /*[2][GROUPS][STATE_F]*/, unsigned* counters /*[GROUPS]*/,
                                             Out* out, float* sink) {
    const int t = threadIdx.x;
    const int g = (V == 2) ? blockIdx.x / PER : blockIdx.x % GROUPS;
    const int m = (V == 2) ? blockIdx.x % PER : blockIdx.x / GROUPS;     // member index within the group
    unsigned mism = 0, tmo = 0;
    unsigned long long t0 = 0;
    if (t == 0) t0 = wall_clock64();
    float acc = 0.f;
    for (int s = 0; s < STEPS; ++s) {
        float* cur = state + ((size_t)(s & 1) * GROUPS + g) * STATE_F;
        // ---- produce: this workgroup's 4 KB slice, stamped with (step, member) ----
        {
            const f32x4 v = {(float)(s + 1), (float)m, (float)t, 1.f};
            float* p = cur + m * SLICE_F + t * 4;
            if (V == 1 || V == 2) st_sc1(p, v);
            else *reinterpret_cast<f32x4*>(p) = v;
        }
        // ---- group barrier ----
        if (V != 0) {
            if (V == 3) __threadfence();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                __hip_atomic_fetch_add(counters + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(s + 1) * PER;
                int spins = 0;
                while (__hip_atomic_load(counters + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    if (++spins > 200000) { ++tmo; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (V == 3) __threadfence();      // acquire side: invalidate this CU's L1
        }
        // ---- consume: the whole 128 KB state of the group (as the next step's A operand would be) ----
        constexpr int NL = STATE_F / (256 * 4);      // 32 loads of 16 B per thread
#pragma unroll
        for (int i0 = 0; i0 < NL; i0 += 16) {
            f32x4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = ((i0 + i) * 256 + t) * 4;
                if (V == 1 || V == 2 || V == 4) v[i] = ld_sc1(cur + idx);
                else if (V == 5) v[i] = ld_nt(cur + idx);
                else v[i] = *reinterpret_cast<const f32x4*>(cur + idx);
            }
            if (V == 1 || V == 2 || V == 4 || V == 5) wait_all(v);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = ((i0 + i) * 256 + t) * 4;
                if (V != 0 && (v[i][0] != (float)(s + 1) || v[i][1] != (float)(idx / SLICE_F))) ++mism;
                acc += v[i][2];
            }
        }
    }
    if (acc == 123.456f) sink[t] = acc;
    // reduce mismatch counts
    __shared__ unsigned sm[256];
    sm[t] = mism;
    __syncthreads();
    if (t == 0) {
        unsigned tot = 0;
        for (int i = 0; i < 256; ++i) tot += sm[i];
        Out o;
        o.t_begin = t0; o.t_end = wall_clock64(); o.mismatches = tot; o.timeouts = tmo;
        o.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
        out[blockIdx.x] = o;
    }
}

template <int V>
void run(const char* name, float* state, unsigned* counters, Out* out, float* sink) {
    std::vector<Out> h(GROUPS * PER);
    double best = 1e30;
    unsigned mism = 0, tmo = 0;
    int mixed = 0;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipMemset(counters, 0, GROUPS * sizeof(unsigned));
        (void)hipMemset(state, 0, (size_t)2 * GROUPS * STATE_F * sizeof(float));
        hipLaunchKernelGGL(chain<V>, dim3(GROUPS * PER), dim3(256), 0, 0, state, counters, out, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("  %-40s launch failed\n", name); return; }
        (void)hipMemcpy(h.data(), out, h.size() * sizeof(Out), hipMemcpyDeviceToHost);
        unsigned long long b = ~0ull, e = 0;
        mism = tmo = 0;
        for (auto& o : h) { b = std::min(b, o.t_begin); e = std::max(e, o.t_end); mism += o.mismatches; tmo += o.timeouts; }
        best = std::min(best, (double)(e - b) / 100.0 / STEPS);
        mixed = 0;
        for (int g = 0; g < GROUPS; ++g) {
            unsigned first = 99;
            bool mix = false;
            for (int i = 0; i < GROUPS * PER; ++i) {
                const int gg = (V == 2) ? i / PER : i % GROUPS;
                if (gg != g) continue;
                if (first == 99) first = h[i].xcc; else if (h[i].xcc != first) mix = true;
            }
            mixed += mix;
        }
    }
    printf("  %-52s %6.2f us per step   mismatches %u  timeouts %u  groups spanning >1 XCD: %d of %d\n", name, best, mism, tmo,
           mixed, GROUPS);
}

int main() {
    float *state, *sink; unsigned* counters; Out* out;
    (void)hipMalloc(&state, (size_t)2 * GROUPS * STATE_F * sizeof(float));
    (void)hipMalloc(&sink, 4096);
    (void)hipMalloc(&counters, GROUPS * sizeof(unsigned));
    (void)hipMalloc(&out, GROUPS * PER * sizeof(Out));
    printf("in-kernel hand-off of a 128 KB state among 32 workgroups, %d steps, 8 groups (256 workgroups)\n", STEPS);
    run<0>("V0 plain ld/st, no sync (data movement only)", state, counters, out, sink);
    run<1>("V1 sc1 16B st/ld + relaxed counter, group = id % 8", state, counters, out, sink);
    run<2>("V2 sc1 16B st/ld + relaxed counter, group = id / 32", state, counters, out, sink);
    run<3>("V3 plain st + __threadfence + counter + plain ld", state, counters, out, sink);
    run<4>("V4 plain st + relaxed counter + sc1 ld, group = id % 8", state, counters, out, sink);
    run<5>("V5 plain st + relaxed counter + nt ld, group = id % 8", state, counters, out, sink);
    return 0;
}
