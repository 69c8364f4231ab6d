// Chain step, second design ("wave-independent"): the same encoder-shaped problem as tools/exp_chain2.hip (2 directions x
// 256 rows x H = 512, 24 steps; 256 workgroups = 8 tiles of 64 rows x 32 members of 16 hidden units), but
//   * the member's W_hh slice (3 gates x 16 units x K = 512 = 96 KB) lives in LDS, fragment-major, instead of being split over
//     the registers of four K-quarter waves;
//   * wave w of a workgroup owns ROW BLOCK w of the tile for the whole K: no cross-wave reduction, no __syncthreads in the
//     step at all; the gates are computed straight from the accumulators (C layout: 4 rows x 1 unit per lane);
//   * a "group" is one 16-row block x 32 members: every wave polls the counter of its own row block.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I inpaintnet_amd/csrc tools/exp_chain3.hip -o build/exp_chain3 && build/exp_chain3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "chain.h"
#include "ksplit.h"
using namespace ksplit;

#ifndef WAVES_
#define WAVES_ 4                                   // waves (= row blocks) per workgroup; 8: two per SIMD, B = 512
#endif
constexpr int WV = WAVES_;
constexpr int H = 512, B = 64 * WV, T = 24, MEMBERS = 32, S = H / 16, NRB = B / 16;
constexpr int TILES = 4, GROUPS = 2 * TILES, NB = GROUPS * MEMBERS;     // a tile = WV row blocks
#ifndef RING_
#define RING_ 8
#endif
struct Args { float* hx; const float* W; const float* gi; float* out; unsigned* counters; unsigned* status; unsigned long long* stamps; };

template <bool MFMA, bool SYNC>
__global__ __launch_bounds__(64 * WV) void k(Args A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wl = smem;                                  // [3][S][64][4]
    float* const xt = smem + 3 * S * 256;                    // [WV waves][256]
    int group, member;
    chain::decode_block(blockIdx.x, MEMBERS, group, member);
    const int dir = group / TILES, tile = group % TILES;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int rb = tile * WV + w;                            // this wave's row block
    const int j0 = member * 16;
    const int slot_bytes = B * H * 4;
    float* hx = A.hx + (size_t)dir * 2 * B * H;
    for (int i = t; i < 3 * S * 64; i += 64 * WV) {
        const int ln = i & 63, s = (i >> 6) % S, g = i / (64 * S);
        *reinterpret_cast<f32x4*>(wl + (long)i * 4) = ld4u(A.W + (long)(g * H + j0 + (ln & 15)) * H + 16 * s + 4 * (ln >> 4));
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(hx);
    unsigned* counter = A.counters + (dir * NRB + rb) * 64;
    float hp[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned long long* stamp = A.stamps + ((size_t)blockIdx.x * WV + w) * T * 4;
    float* myxt = xt + w * 256;
    const int abase = (rb * S * 256 + lane * 4) * 4;         // byte offset of k-step 0 of this row block
    for (int step = 0; step < T; ++step) {
        if (lane == 0) stamp[step * 4 + 0] = wall_clock64();
        float pg[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < 3; ++g) pg[r][g] = A.gi[((long)step * B + rb * 16 + 4 * q + r) * 3 * H + g * H + j0 + c];
        if (SYNC && step > 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(step * MEMBERS)) {
                if (++spins > 400000) { if (lane == 0) *A.status = 1; return; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (lane == 0) stamp[step * 4 + 1] = wall_clock64();
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (MFMA) {
            const int base = ((step + 1) & 1) * slot_bytes;
            f32x4 Ar[RING_];
#pragma unroll
            for (int d = 0; d < RING_ - 1; ++d) Ar[d] = chain::ld16_sc1(rs, abase + d * 1024, base);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int si = 0; si < S; ++si) {
                if (si + RING_ - 1 < S) Ar[(si + RING_ - 1) % RING_] = chain::ld16_sc1(rs, abase + (si + RING_ - 1) * 1024, base);
                f32x4 Bf[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) Bf[g] = *reinterpret_cast<const f32x4*>(wl + ((g * S + si) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[si % RING_][e], Bf[g][e], acc[g], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (lane == 0) stamp[step * 4 + 2] = wall_clock64();
        float eh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rr = sigmoid_f(acc[0][r] + pg[r][0]), z = sigmoid_f(acc[1][r] + pg[r][1]);
            const float n = tanh_f(pg[r][2] + rr * acc[2][r]);
            const float hn = (1.f - z) * n + z * hp[r];
            hp[r] = hn; eh[r] = hn;
            myxt[(4 * q + r) * 16 + c] = hn;
        }
        // wave-local transpose C layout -> A fragment layout, then one 16-byte write-through store per lane
        const f32x4 v = *reinterpret_cast<const f32x4*>(myxt + (lane & 15) * 16 + (lane >> 4) * 4);
        chain::st16_sc1(rs, (step & 1) * slot_bytes + ((rb * S + member) * 256 + lane * 4) * 4, v);
        if (SYNC) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) stamp[step * 4 + 3] = wall_clock64();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long o = ((long)step * B + rb * 16 + 4 * q + r) * H + j0 + c;
#pragma unroll
            for (int a = 0; a < 7; ++a) A.out[o + (long)a * T * B * H] = eh[r] + a;
        }
    }
}

template <bool MFMA, bool SYNC>
void run(const char* name, Args a) {
    const size_t ns = (size_t)NB * WV * T * 4;
    std::vector<unsigned long long> h(ns);
    double best = 1e30; double ph[4] = {0, 0, 0, 0};
    unsigned stat = 0;
    const size_t lds = (size_t)(3 * S * 256 + WV * 256) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MFMA, SYNC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipMemset(a.counters, 0, 128 * 64 * 4); (void)hipMemset(a.status, 0, 4);
        hipLaunchKernelGGL((k<MFMA, SYNC>), dim3(NB), dim3(64 * WV), lds, 0, a);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
        (void)hipMemcpy(h.data(), a.stamps, ns * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&stat, a.status, 4, hipMemcpyDeviceToHost);
        unsigned long long b0 = ~0ull, e1 = 0;
        for (size_t wv = 0; wv < (size_t)NB * WV; ++wv) { b0 = std::min(b0, h[wv * T * 4]); e1 = std::max(e1, h[wv * T * 4 + (T - 1) * 4 + 3]); }
        const double us = (e1 - b0) / 100.0 / T;
        if (us < best) {
            best = us;
            for (auto& x : ph) x = 0;
            for (size_t wv = 0; wv < (size_t)NB * WV; ++wv) for (int s = 2; s < T; ++s) {
                const unsigned long long* p = &h[(wv * T + s) * 4];
                const unsigned long long prev_end = h[(wv * T + s - 1) * 4 + 3];
                ph[0] += p[0] - prev_end; ph[1] += p[1] - p[0]; ph[2] += p[2] - p[1]; ph[3] += p[3] - p[2];
            }
            for (auto& x : ph) x /= 100.0 * NB * WV * (T - 2);
        }
    }
    printf("%-10s %6.2f us/step | stores+loop %.2f  prefetch+wait %.2f  contract %.2f  gates+publish+arrive %.2f | status %u\n",
           name, best, ph[0], ph[1], ph[2], ph[3], stat);
}

int main() {
    Args a;
    (void)hipMalloc(&a.hx, (size_t)2 * 2 * B * H * 4); (void)hipMemset(a.hx, 0, (size_t)2 * 2 * B * H * 4);
    float* W; (void)hipMalloc(&W, (size_t)3 * H * H * 4); a.W = W;
    float* gi; (void)hipMalloc(&gi, (size_t)T * B * 3 * H * 4); a.gi = gi;
    {
        std::vector<float> hw((size_t)3 * H * H), hg((size_t)T * B * 3 * H);
        unsigned x = 12345u;
        auto rnd = [&] { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xffff) / 32768.f - 1.f; };
        for (auto& v : hw) v = rnd() * 0.05f;
        for (auto& v : hg) v = rnd();
        (void)hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(gi, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    }
    (void)hipMalloc(&a.out, (size_t)7 * T * B * H * 4);
    (void)hipMalloc(&a.counters, 128 * 64 * 4); (void)hipMalloc(&a.status, 4);
    (void)hipMalloc(&a.stamps, (size_t)NB * WV * T * 4 * 8);
    run<true, true>("full", a);
    run<true, false>("no_sync", a);
    run<false, true>("no_mfma", a);
    return 0;
}
