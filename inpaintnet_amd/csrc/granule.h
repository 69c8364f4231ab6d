// Register-resident persistent kernels for batch-1 recurrences (round 5): the pieces arnn_gen.hip and decode_b1.hip share.
//
// Vectors move between resident workgroups as 8-byte {value, tick} GRANULES: one relaxed agent-scope 64-bit store per element,
// polled with relaxed agent-scope 64-bit loads (MI355X_MICROARCH.md "handoff-1to1", form R2: the tag travels with the value in one
// naturally aligned store, so there is nothing to order and no flag).  A granule read is a round trip through the memory side
// (~0.8 us): everything a phase needs is REQUESTED before the first answer is looked at, and what is known to arrive early is
// requested a phase ahead.  Spins are bounded; the caller zeroes the granules (tag 0 = nothing yet, ticks count from 1).
#pragma once
#include "chain.h"

namespace granule {

constexpr unsigned kSpin = 500000;               // polls per wait before a workgroup gives up: a poll is a ~0.8-1 us round trip + s_sleep, ~0.5 s

__device__ __forceinline__ void put(unsigned long long* g, float v, unsigned tag) {
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// `near` (uniform over the workgroup): EVERY workgroup that exchanges granules with this one runs on this XCD (same_xcd below has
// checked it).  Then a plain store does: it stays in the XCD's L2, where the readers' L1-bypassing loads find it -- the agent-scope
// store drops the line from L2 and sends the readers through the memory side (MI355X_MICROARCH.md, "stores of each flavour").
// Measured on AnticipationRNN's token pass (13 workgroups on one XCD): 3.59 -> 3.24 us per tick.  Across XCDs a plain store is
// NEVER seen (the same run on consecutive workgroup ids: every wait ran into its bound) -- hence the check instead of a promise
// about placement HIP does not make.
__device__ __forceinline__ void put(unsigned long long* g, float v, unsigned tag, bool near) {
    if (near) {
        const unsigned long long x = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
        asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(g), "v"(x) : "memory");
    } else put(g, v, tag);
}
__device__ __forceinline__ unsigned long long peek(const unsigned long long* g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the slow-wait recorder (chain.h) for granule waits: every lane polls for itself, the first slow lane of the wave files the entry
// (INET_GRANULE_KID: the including kernel file's id for the recorder, chain.h)
__device__ __forceinline__ void note_slow(const chain::Status& st, unsigned tag, unsigned spins, bool gave_up) {
    const unsigned long long m = __ballot(1);
    if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) chain::record_slow<INET_GRANULE_KID>(st, 1u, tag, spins, gave_up);
}
// N granules, `stride` apart, of one tick: ALL of them are requested before the first is looked at (a granule read is a round trip
// through the memory side, ~1 us: four of them one after the other were half of the first build's tick).  `w` may hold an earlier
// request of the same granules (first = false skips the first request).  false: gave up (bounded spin, or the launch was aborted).
template <int N>
__device__ __forceinline__ bool get_n(const unsigned long long* g, int stride, unsigned tag, const chain::Status& st, float (&v)[N],
                                      unsigned long long (&w)[N], bool first = true) {
    unsigned spins = 0;
    for (;;) {
        if (first) {
#pragma unroll
            for (int i = 0; i < N; ++i) w[i] = peek(g + (long)i * stride);
        }
        first = true;
        bool all = true;
#pragma unroll
        for (int i = 0; i < N; ++i) all &= (unsigned)(w[i] >> 32) == tag;
        if (all) {
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = __uint_as_float((unsigned)w[i]);
            if (spins > chain::kGranuleSlowSpins) note_slow(st, tag, spins, false);
            return true;
        }
        if (++spins > kSpin ||
            ((spins & 1023) == 0 && __hip_atomic_load(st.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != chain::ST_OK)) {
            note_slow(st, tag, spins, true);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ bool get_1(const unsigned long long* g, unsigned tag, const chain::Status& st, float& v) {
    float vv[1];
    unsigned long long w[1];
    const bool ok = get_n<1>(g, 0, tag, st, vv, w);
    v = vv[0];
    return ok;
}
// Do the `n` (<= 64) workgroups of a launch share one XCD?  Workgroup `me` publishes its XCC id in slot `me` of `slots` (zeroed by
// the caller, agent-scope stores), wave 0 of every workgroup reads all n slots: everybody sees the same n values and comes to the
// same answer.  One hand-off at the start of the launch (~1 us).  false also when a wait gave up.
constexpr unsigned kXccTag = 0xF0000000u;
__device__ __forceinline__ bool same_xcd(unsigned long long* slots, int me, int n, const chain::Status& st, int* lds_flag) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xfu;
    if (threadIdx.x == 0) put(slots + me, __uint_as_float(xcc), kXccTag);
    if (threadIdx.x < 64) {
        bool same = true;
        if ((int)threadIdx.x < n) {
            float v;
            same = get_1(slots + threadIdx.x, kXccTag, st, v) && __float_as_uint(v) == xcc;
        }
        const unsigned long long m = __ballot(same);
        if (threadIdx.x == 0) *lds_flag = m == ~0ull;
    }
    __syncthreads();
    return *lds_flag != 0;
}
// workgroup barrier that waits for this wave's LDS traffic only: granule requests in flight stay in flight across it (__syncthreads
// would drain vmcnt and put their round trip back on the critical path)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (plain v_fmac_f32 on purpose: left to itself hipcc SLP-packs the sums into v_pk_fma_f32 -- no faster on this part -- and pays two
//  v_mov per pair to line the operands up)
__device__ __forceinline__ void fmac(float& acc, float w, float x) { asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(w), "v"(x)); }
// maximum over the wave without LDS: two quad permutes, the two mirrors of a 16-lane row, then the four rows through SGPRs
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0xB1, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x4E, 0xF, 0xF, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x141, 0xF, 0xF, false)));  // row_half_mirror
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x140, 0xF, 0xF, false)));  // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

}  // namespace granule
