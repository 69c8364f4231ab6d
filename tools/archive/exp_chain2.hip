// Anatomy of one chain-kernel step at the encoder shape (8 groups x 32 members, 64 rows x 16 hidden x 3 gates, K = 512):
// which part of the ~12 us per step is MFMA, operand latency, reduce/epilogue, and the group hand-off.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I inpaintnet_amd/csrc tools/exp_chain2.hip -o build/exp_chain2 && build/exp_chain2
// Variants switch phases off (the results are then meaningless numerically; only the timing matters):
//   full            everything
//   no_mfma         loads + sync, no MFMAs
//   no_loads        MFMAs on stale registers + sync
//   no_sync         loads + MFMAs, no waiting (each WG free-runs)
//   plain_st        hand-off with plain stores instead of sc1 (valid only same-XCD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "chain.h"
#include "ksplit.h"
using namespace ksplit;

#ifndef MS_
#define MS_ 4
#endif
constexpr int MS = MS_, SQ = 8, H = 512, B = 256, T = 24, MEMBERS = 32;
constexpr int TILES = B / (16 * MS), GROUPS = 2 * TILES, NB = GROUPS * MEMBERS;   // MS_=2: 512 workgroups, two per CU
struct Args { float* hx; const float* W; const float* gi; float* out; unsigned* counters; unsigned* status; unsigned long long* stamps; unsigned* where; };

template <bool MFMA, bool LOADS, bool SYNC, bool PLAIN>
__global__ __launch_bounds__(256) void k(Args A) {
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * MS * 256];
    __shared__ __attribute__((aligned(16))) float xt[MS * 256];
    __shared__ unsigned flag[2];
#ifdef PAD_
    __shared__ float pad[PAD_];
    if (threadIdx.x == 0 && A.where == nullptr) pad[0] = 1.f;
#endif
    int group, member;
    chain::decode_block(blockIdx.x, MEMBERS, group, member);
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        A.where[blockIdx.x] = ((xcc & 0xf) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
    }
    const int row0 = (group % TILES) * 16 * MS, t = threadIdx.x, lane = t & 63, S = H >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), i16 = lane & 15, q = lane >> 4;
    const int j0 = member * 16, jc = j0 + (t & 15), rb0 = row0 >> 4, rb_last = (B - 1) >> 4;
    const int slot_bytes = B * H * 4;
    float* hx = A.hx + (size_t)(group / TILES) * 2 * B * H;
    f32x4 Wr[3][SQ];
    for (int g = 0; g < 3; ++g) for (int si = 0; si < SQ; ++si)
        Wr[g][si] = ld4u(A.W + (long)(g * H + j0 + i16) * H + 16 * (w * SQ + si) + 4 * q);
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(hx);
    #ifndef CSTRIDE_
#define CSTRIDE_ 1
#endif
    unsigned* counter = A.counters + group * CSTRIDE_;
    chain::Status st{A.status, nullptr, nullptr};
    float hp[MS] = {};
    unsigned long long* stamp = A.stamps + (size_t)blockIdx.x * T * 6;
    f32x4 acc[MS][4];
    for (int ms = 0; ms < MS; ++ms) for (int a = 0; a < 3; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef PRIO_
    // the two workgroups of a CU at different wave priorities: when both are in their MFMA phase the pipe goes to the
    // high-priority one, the other falls behind and ends up computing while the first one hands off
    if (group >= GROUPS / 2) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(PRIO_);
#endif
#ifdef SKEW_
    // de-phase the two workgroups that share a CU: the second half of the groups (dispatched onto the same CUs as the first
    // half) starts SKEW_ x 10 ns late, so that its MFMA phase falls into the other's hand-off / epilogue phase
    if (group >= GROUPS / 2) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < SKEW_) __builtin_amdgcn_s_sleep(8); }
#endif
    for (int step = 0; step < T; ++step) {
        if (t == 0) stamp[step * 6 + 0] = wall_clock64();
        float pg[MS][3];
        for (int p = 0; p < MS; ++p) for (int g = 0; g < 3; ++g)
            pg[p][g] = A.gi[((long)step * B + row0 + ((t + 256 * p) >> 4)) * 3 * H + g * H + jc];
        if (SYNC && step > 0 && !chain::wait_group(counter, (unsigned)(step * MEMBERS), st, &flag[step & 1])) return;
        if (t == 0) stamp[step * 6 + 1] = wall_clock64();
        if (LOADS && MFMA) {
            for (int ms = 0; ms < MS; ++ms) for (int a = 0; a < 3; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
            chain::contract<MS, 3, SQ>(acc, Wr, rs, ((step + 1) & 1) * slot_bytes, rb0, rb_last, S, w * SQ, lane);
        } else if (LOADS) {
            f32x4 s = {0, 0, 0, 0};
            for (int ms = 0; ms < MS; ++ms) for (int si = 0; si < SQ; ++si)
                s += chain::ld16_sc1(rs, ((step + 1) & 1) * slot_bytes + (((rb0 + ms) * S + w * SQ + si) * 256 + lane * 4) * 4);
            acc[0][0] += s;
        } else if (MFMA) {
            for (int si = 0; si < SQ; ++si) for (int e = 0; e < 4; ++e) for (int ms = 0; ms < MS; ++ms) for (int g = 0; g < 3; ++g)
                acc[ms][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wr[g][si][e], Wr[(g + ms) % 3][si][e], acc[ms][g], 0, 0, 0);
        }
        if (t == 0) stamp[step * 6 + 2] = wall_clock64();
        float v[MS][3];
        reduce_waves<MS, 3>(acc, red, t, v);
        if (t == 0) stamp[step * 6 + 3] = wall_clock64();
        float eh[MS];
        for (int p = 0; p < MS; ++p) {
            const int rl = (t + 256 * p) >> 4;
            const float r = sigmoid_f(v[p][0] + pg[p][0]), z = sigmoid_f(v[p][1] + pg[p][1]);
            const float n = tanh_f(pg[p][2] + r * v[p][2]);
            const float hn = (1.f - z) * n + z * hp[p];
            hp[p] = hn; eh[p] = hn;
            xt[rl * 16 + (t & 15)] = hn;
        }
        __syncthreads();
        if (t < 64 * MS) {
            const int p = t >> 6;
            if (PLAIN) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(xt + (p * 16 + (lane & 15)) * 16 + (lane >> 4) * 4);
                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(hx) + (step & 1) * slot_bytes + (((rb0 + p) * S + member) * 256 + lane * 4) * 4) = vv;
            } else chain::publish_block(rs, (step & 1) * slot_bytes, xt, p, lane, rb0 + p, S, member);
        }
        if (SYNC) chain::arrive(counter); else __syncthreads();
        if (t == 0) stamp[step * 6 + 4] = wall_clock64();
        for (int p = 0; p < MS; ++p) {
            const long o = ((long)step * B + row0 + ((t + 256 * p) >> 4)) * H + jc;
            for (int a = 0; a < 7; ++a) A.out[o + (long)a * T * B * H] = eh[p] + a;
        }
        if (t == 0) stamp[step * 6 + 5] = wall_clock64();
    }
}

template <bool MFMA, bool LOADS, bool SYNC, bool PLAIN>
void run(const char* name, Args a) {
    std::vector<unsigned long long> h((size_t)NB * T * 6);
    double best = 1e30; std::vector<double> ph(6, 0);
    unsigned stat = 0;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipMemset(a.counters, 0, 64 * 64 * 4); (void)hipMemset(a.status, 0, 4);
        hipLaunchKernelGGL((k<MFMA, LOADS, SYNC, PLAIN>), dim3(NB), dim3(256), 0, 0, a);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
        (void)hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&stat, a.status, 4, hipMemcpyDeviceToHost);
        unsigned long long b0 = ~0ull, e1 = 0;
        for (int b = 0; b < NB; ++b) { b0 = std::min(b0, h[(size_t)b * T * 6]); e1 = std::max(e1, h[(size_t)b * T * 6 + (T - 1) * 6 + 5]); }
        const double us = (e1 - b0) / 100.0 / T;
        if (us < best) {
            best = us;
            std::fill(ph.begin(), ph.end(), 0.0);
            for (int b = 0; b < NB; ++b) for (int s = 2; s < T; ++s) {        // skip the first two steps
                const unsigned long long* p = &h[((size_t)b * T + s) * 6];
                const unsigned long long prev_end = h[((size_t)b * T + s - 1) * 6 + 5];
                ph[0] += (p[0] - prev_end); ph[1] += p[1] - p[0]; ph[2] += p[2] - p[1]; ph[3] += p[3] - p[2]; ph[4] += p[4] - p[3]; ph[5] += p[5] - p[4];
            }
            for (auto& x : ph) x /= 100.0 * NB * (T - 2);
        }
    }
    {
        std::vector<unsigned> wh(NB);
        (void)hipMemcpy(wh.data(), a.where, NB * 4, hipMemcpyDeviceToHost);
        std::sort(wh.begin(), wh.end());
        int hist[9] = {0}, cus = 0;
        for (size_t i = 0; i < wh.size();) { size_t j = i; while (j < wh.size() && wh[j] == wh[i]) ++j; hist[std::min<size_t>(j - i, 8)]++; ++cus; i = j; }
        printf("  placement: %d CUs; CUs holding 1/2/3/4 workgroups: %d %d %d %d\n", cus, hist[1], hist[2], hist[3], hist[4]);
    }
    printf("%-10s %6.2f us/step | loop %.2f  prefetch+wait %.2f  contract %.2f  reduce %.2f  gates+publish+arrive %.2f  stores %.2f | status %u\n",
           name, best, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], stat);
}

int main() {
    Args a;
    (void)hipMalloc(&a.hx, (size_t)2 * 2 * B * H * 4); (void)hipMemset(a.hx, 0, (size_t)2 * 2 * B * H * 4);
    float* W; (void)hipMalloc(&W, (size_t)3 * H * H * 4); a.W = W;
    float* gi; (void)hipMalloc(&gi, (size_t)T * B * 3 * H * 4); a.gi = gi;
    {   // random operands: zero-filled ones let the chip clock higher (DVFS) than real data does
        std::vector<float> hw((size_t)3 * H * H), hg((size_t)T * B * 3 * H);
        unsigned x = 12345u;
        auto rnd = [&] { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xffff) / 32768.f - 1.f; };
        for (auto& v : hw) v = rnd() * 0.05f;
        for (auto& v : hg) v = rnd();
        (void)hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(gi, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    }
    (void)hipMalloc(&a.out, (size_t)7 * T * B * H * 4);
    (void)hipMalloc(&a.counters, 64 * 64 * 4); (void)hipMalloc(&a.status, 4);
    (void)hipMalloc(&a.stamps, (size_t)NB * T * 6 * 8);
    (void)hipMalloc(&a.where, NB * 4);
    run<true, true, true, false>("full", a);
#ifdef ALSO_NOSYNC_
    run<true, true, false, false>("no_sync", a);
    run<false, true, true, false>("no_mfma", a);
    run<true, false, true, false>("no_loads", a);
#endif
#ifndef ONLY_FULL_
    run<false, true, true, false>("no_mfma", a);
    run<true, false, true, false>("no_loads", a);
    run<true, true, false, false>("no_sync", a);
    run<false, false, true, false>("sync_only", a);
    run<true, true, true, true>("plain_st", a);
#endif
    return 0;
}
