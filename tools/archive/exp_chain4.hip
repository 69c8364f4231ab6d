// Chain step, third design: tools/exp_chain3.hip (W slice in LDS, one row block per wave, no barrier in the step) with the
// contraction on the bf16 matrix cores at fp32 accuracy: every fp32 operand is split EXACTLY into three bf16 pieces
// (a = a0 + a1 + a2, 3 x 8 mantissa bits), and a*b is the sum of NP in {6, 9} piece products, each exact in the f32
// accumulator's input stage, accumulated in f32 (9: every term, the products are those of fp32 arithmetic; 6: the three terms
// below 2^-24 |ab| dropped).  A bf16 MFMA (16x16x32) issues 16x the MACs per cycle of the f32 one (16x16x4).
// The producer splits its 16 columns once; the exchange carries the three pieces (6 bytes per element instead of 4).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I inpaintnet_amd/csrc tools/exp_chain4.hip -DNP_=6 -o build/exp_chain4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "chain.h"
#include "ksplit.h"
using namespace ksplit;

#ifndef NP_
#define NP_ 6
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 512, B = 256, T = 24, MEMBERS = 32, S32 = H / 32, NRB = B / 16;
constexpr int TILES = B / 64, GROUPS = 2 * TILES, NB = GROUPS * MEMBERS;
constexpr int PIECE_BYTES = B * H * 2;                      // one bf16 piece of a [B,H] state
struct Args { unsigned char* hx; const float* W; const float* gi; float* out; unsigned* counters; unsigned* status; unsigned long long* stamps; };

__device__ __forceinline__ void split3(float x, __bf16& a0, __bf16& a1, __bf16& a2) {
    a0 = (__bf16)x;
    const float r1 = x - (float)a0;
    a1 = (__bf16)r1;
    a2 = (__bf16)(r1 - (float)a1);
}

template <bool MFMA, bool SYNC>
__global__ __launch_bounds__(256) void k(Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const wl = smem;                          // [3 pieces][3 gates][S32][64][16 B]
    float* const xt = reinterpret_cast<float*>(smem + 3 * 3 * S32 * 1024);   // [4 waves][256]
    int group, member;
    chain::decode_block(blockIdx.x, MEMBERS, group, member);
    const int dir = group / TILES, tile = group % TILES;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int rb = tile * 4 + w;
    const int j0 = member * 16;
    const int slot_bytes = 3 * PIECE_BYTES;
    unsigned char* hx = A.hx + (size_t)dir * 2 * slot_bytes;
    // W slice -> three bf16 pieces in B-fragment order: fragment (g, s32): lane (unit = lane % 16, k group = lane / 16) holds
    // W[g*H + j0 + unit][32 s32 + 8 (lane / 16) + 0..7]
    for (int i = t; i < 3 * S32 * 64; i += 256) {
        const int ln = i & 63, s = (i >> 6) % S32, g = i / (64 * S32);
        const float* src = A.W + (long)(g * H + j0 + (ln & 15)) * H + 32 * s + 8 * (ln >> 4);
        bf16x8 p0, p1, p2;
#pragma unroll
        for (int j = 0; j < 8; ++j) { __bf16 a, b, cc; split3(src[j], a, b, cc); p0[j] = a; p1[j] = b; p2[j] = cc; }
        *reinterpret_cast<bf16x8*>(wl + ((0 * 3 + g) * S32 + s) * 1024 + ln * 16) = p0;
        *reinterpret_cast<bf16x8*>(wl + ((1 * 3 + g) * S32 + s) * 1024 + ln * 16) = p1;
        *reinterpret_cast<bf16x8*>(wl + ((2 * 3 + g) * S32 + s) * 1024 + ln * 16) = p2;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(hx);
    unsigned* counter = A.counters + (dir * NRB + rb) * 64;
    float hp[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned long long* stamp = A.stamps + ((size_t)blockIdx.x * 4 + w) * T * 4;
    float* myxt = xt + w * 256;
    const int abase = (rb * S32 * 64 + lane) * 16;           // byte offset of k block 0 of this row block inside a piece
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
            constexpr int R = 4;
            bf16x8 Ar[R][3];
            auto ldA = [&](int s, int slot) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    Ar[slot][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, abase + s * 1024 + p * PIECE_BYTES, base, 16));
            };
#pragma unroll
            for (int d = 0; d < R - 1; ++d) ldA(d, d);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < S32; ++s) {
                if (s + R - 1 < S32) ldA(s + R - 1, (s + R - 1) % R);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    bf16x8 Bf[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) Bf[p] = *reinterpret_cast<const bf16x8*>(wl + ((p * 3 + g) * S32 + s) * 1024 + lane * 16);
                    const bf16x8* a = Ar[s % R];
                    // piece products in ascending order of magnitude dropped last: (0,0) (0,1) (1,0) (0,2) (1,1) (2,0) [(1,2) (2,1) (2,2)]
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], Bf[0], acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], Bf[1], acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], Bf[0], acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], Bf[2], acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], Bf[1], acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], Bf[0], acc[g], 0, 0, 0);
                    if (NP_ == 9) {
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], Bf[2], acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], Bf[1], acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], Bf[2], acc[g], 0, 0, 0);
                    }
                }
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
        // wave-local transpose, then lanes 0..31 split their 8 columns of one row into three bf16 pieces: 3 x 16 bytes
        if (lane < 32) {
            const int row = lane & 15, half = lane >> 4;
            const float* src = myxt + row * 16 + 8 * half;
            bf16x8 p0, p1, p2;
#pragma unroll
            for (int j = 0; j < 8; ++j) { __bf16 a, b, cc; split3(src[j], a, b, cc); p0[j] = a; p1[j] = b; p2[j] = cc; }
            const int off = (step & 1) * slot_bytes + ((rb * S32 + (member >> 1)) * 64 + (2 * (member & 1) + half) * 16 + row) * 16;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p0), rs, off, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p1), rs, off + PIECE_BYTES, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p2), rs, off + 2 * PIECE_BYTES, 0, 16);
        }
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
    const size_t ns = (size_t)NB * 4 * T * 4;
    std::vector<unsigned long long> h(ns);
    double best = 1e30; double ph[4] = {0, 0, 0, 0};
    unsigned stat = 0;
    const size_t lds = (size_t)3 * 3 * S32 * 1024 + 4 * 256 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MFMA, SYNC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipMemset(a.counters, 0, 64 * 64 * 4); (void)hipMemset(a.status, 0, 4);
        hipLaunchKernelGGL((k<MFMA, SYNC>), dim3(NB), dim3(256), lds, 0, a);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", name); return; }
        (void)hipMemcpy(h.data(), a.stamps, ns * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&stat, a.status, 4, hipMemcpyDeviceToHost);
        unsigned long long b0 = ~0ull, e1 = 0;
        for (size_t wv = 0; wv < (size_t)NB * 4; ++wv) { b0 = std::min(b0, h[wv * T * 4]); e1 = std::max(e1, h[wv * T * 4 + (T - 1) * 4 + 3]); }
        const double us = (e1 - b0) / 100.0 / T;
        if (us < best) {
            best = us;
            for (auto& x : ph) x = 0;
            for (size_t wv = 0; wv < (size_t)NB * 4; ++wv) for (int s = 2; s < T; ++s) {
                const unsigned long long* p = &h[(wv * T + s) * 4];
                const unsigned long long prev_end = h[(wv * T + s - 1) * 4 + 3];
                ph[0] += p[0] - prev_end; ph[1] += p[1] - p[0]; ph[2] += p[2] - p[1]; ph[3] += p[3] - p[2];
            }
            for (auto& x : ph) x /= 100.0 * NB * 4 * (T - 2);
        }
    }
    printf("%-10s %6.2f us/step | stores+loop %.2f  prefetch+wait %.2f  contract %.2f  gates+publish+arrive %.2f | status %u\n",
           name, best, ph[0], ph[1], ph[2], ph[3], stat);
}

int main() {
    Args a;
    (void)hipMalloc(&a.hx, (size_t)2 * 2 * 3 * PIECE_BYTES); (void)hipMemset(a.hx, 0, (size_t)2 * 2 * 3 * PIECE_BYTES);
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
    (void)hipMalloc(&a.counters, 64 * 64 * 4); (void)hipMalloc(&a.status, 4);
    (void)hipMalloc(&a.stamps, (size_t)NB * 4 * T * 4 * 8);
    printf("NP = %d piece products per fp32 product\n", NP_);
    run<true, true>("full", a);
    run<true, false>("no_sync", a);
    run<false, true>("no_mfma", a);
    return 0;
}
