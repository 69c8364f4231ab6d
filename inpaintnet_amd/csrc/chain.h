// "Chain kernels": one persistent launch runs ALL time steps of a recurrent layer.
//
// Why (measured, profiles/r02_c_*): a per-step launch starts with a cold L2 (kernel boundaries write back AND invalidate
// the per-XCD L2s), so every step re-fetches its W_hh slice through the fabric -- rocprofv3 counts 37 MB of memory-side
// reads per two-direction GRU step against 15 MB of algorithmic bytes, 2.4 TB/s for 15 us -- and pays ~2 us of dispatch
// on top.  The weights never change during a sequence: in a chain kernel every workgroup loads its W slice ONCE into
// registers (16 hidden units x gates x K/4 per wave = 64-96 VGPRs) and only the hidden state moves per step.
//
// Geometry: a GROUP = the workgroups that share one row tile (16*MS batch rows) of one problem (direction): H/16
// members, one per 16 hidden units.  Each step every member needs the group's whole previous hidden state, so the
// group synchronises once per step on a monotonic counter.  Groups are independent (no grid barrier).
//
// Hand-off protocol (MI355X_MICROARCH.md "inter-workgroup visibility", cdna_hip_programming.md Guideline 16, form R1;
// measured in tools/exp_chain.hip V1: 4.2 us per step same-XCD incl. moving 128 KB, 0 stale reads):
//   producer: 16-byte `sc1` (write-through) stores of its [16 rows x 16 k] fragment blocks -> EVERY wave drains
//             (s_waitcnt vmcnt(0)) -> __syncthreads -> ONE lane adds 1 to the group counter (relaxed, agent scope);
//   consumer: ONE lane polls the counter (relaxed agent-scope load + s_sleep, bounded) -> __syncthreads -> every wave
//             reads the fragments with 16-byte `sc1` loads (served by L2, never by this CU's L1).
// This is correct under ANY workgroup->XCD placement.  For speed only, block b is given group b % 8 so that, with the
// observed round-robin dispatch, all members of a group share one XCD and the exchange stays inside its L2
// (cross-XCD the same protocol measures 11 us per step).
// Every spin is bounded: on timeout the workgroup raises the status word and leaves; the host side reports it.
// The exchange buffers are fragment-major (ksplit.h pk_offset): a member's 16 hidden units of one 16-row block ARE one
// contiguous 1 KB block = one 16-byte store per lane of one wave, and a consumer wave's MFMA A-fragment is one 16-byte
// load per lane.
#pragma once
#include "common.h"

namespace chain {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kSpinLimit = 400000;          // polls (~1 us each with s_sleep): ~0.4 s before giving up
enum { ST_OK = 0, ST_TIMEOUT = 1 };

// Layout of the process-wide device status area (side.hip chain_dev_status()), in 32-bit words: word 0 = the status; from kDiagWord:
// kDiagBytes of scratch for instrumented builds (inet_debug_read); from kRecWord: the SLOW-WAIT RECORDER -- word 0 = waits of at
// least kSlowSpins polls since the last reset ("noted": a poll of a counter is ~0.4 us, of a granule 0.1-0.3 us; a hand-off in
// steady state takes 3-8; granule waits are not noted below kGranuleSlowSpins), word 1 = the entry threshold in polls (default kRecDefaultPolls ~ 6 ms, inet_set_option key 16), word 2 =
// waits of at least that many polls ("slow"), words 8.. = the first kRecEntries slow ones, 8 words each (record_slow below).
// A wait files its entry when it ENDS -- arrived or given up (always filed) -- so that a lost hand-off or a workgroup that
// became resident very late leaves its coordinates behind instead of a bare status word (four events without
// a trace).  What the first runs showed at once: in EVERY training step the early members of some BPTT chain groups wait 200-300
// polls (~100 us) on one box and 1000-1300 (~0.4 ms) on another for the group's first arrival -- the launch becomes resident
// group by group while the weight-gradient products of the side streams hold CUs (the overlap of DESIGN.md section 8 "leaf
// schedule" seen from inside the chain) -- hence two levels: the noted count says how much such waiting there is, entries are kept
// for waits no overlap explains (the default threshold is several steps' worth of time; diagnosis runs lower it).
// inet_slow_waits() copies it out; ChainTimeoutError, bench.py (`slow_waits`) and the test suite's teardown print it.
constexpr int kDiagWord = 64, kDiagBytes = 16384, kRecWord = kDiagWord + kDiagBytes / 4, kRecEntries = 127, kRecWords = 8 * (1 + kRecEntries);
constexpr unsigned kSlowSpins = 64, kRecDefaultPolls = 16384;
// Granule waits (granule.h) are polled by EVERY thread and a recurrent-side workgroup of the register-resident kernels waits most of
// a tick for its next input as a matter of course: noting those (one atomic on one word per wave and wait, and a dependent load of
// the threshold behind it) quadrupled the decode call -- 0.12 -> 0.4-0.5 ms, every phase of the tick slower, a feedback loop:
// slower ticks, longer waits, more atomics.  Granule waits therefore reach the recorder only from kGranuleSlowSpins polls on.
constexpr unsigned kGranuleSlowSpins = 4096;
#ifndef INET_RECORDER
#define INET_RECORDER 1                          // 0: build without the recorder (A/B of what it costs the chain kernels)
#endif
// kernel ids of the recorder (a template constant of every wait)
enum { K_GRU_FWD = 1, K_GRU_BWD = 2, K_GRU2_FWD = 3, K_GRU2_BWD = 4, K_LSTM_FWD = 5, K_LSTM_BWD = 6, K_LSTM_PIPE = 7, K_DECODE_CHAIN = 8,
       K_ARNN_GEN = 9, K_DECODE_B1 = 10 };

// buffer resource over a (wave-uniform) base pointer; 2 GB window, raw addressing
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
// 16-byte load / store with sc1: L1-bypassing, write-through -- the agent-coherent forms.  The compiler tracks these in
// vmcnt like any other buffer access, so they pipeline normally.
__device__ __forceinline__ f32x4 ld16_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, int s_off = 0) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, s_off, 16));
}
// A VGPR that holds 0 without the compiler knowing: added to a wave-uniform index it keeps the derived addresses in
// vector registers.  (The A-fragment offsets of a contraction are loop-invariant and uniform; hipcc hoists one SGPR per
// (row block, k-step) out of the step loop and, past ~100 of them, spills scalars to VGPR lanes.)
__device__ __forceinline__ int opaque_zero() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}

// Producer side, called by ALL threads after their sc1 stores of the step.
__device__ __forceinline__ void arrive(unsigned* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Consumer side, called by ALL threads: returns false (for every thread) if the group did not arrive in time.
// `flag` is one word of LDS; consecutive waits must use different words (flag[step & 1]): a fast wave may enter the
// next wait while a slow one still reads this one's word.
// `status` = {device word inside the workspace (aborts the other workgroups of the launch quickly), host-mapped word
// (chain_host_status(): what inet_chain_status() reports), process-wide device word (chain_dev_status(): what the
// optimizer kernel reads -- a step whose chain launch timed out never updates the weights)}.
#ifndef INET_CHAIN_POLL_SLEEP
#define INET_CHAIN_POLL_SLEEP 1                   // s_sleep units (64 clocks) between two polls of the group counter
#endif
struct Status { unsigned* dev; unsigned* host; unsigned* gdev; };
// one recorder entry, written by ONE lane behind a wait that was slow: {kernel id | XCC id << 8 | gave up << 15 | site << 16,
// workgroup, expected tag or counter, polls, wall clock (100 MHz) lo, hi, 0, 0}.  The kernel id is a template constant of the
// wait (a Status field cost the chain kernels ten scalar registers and their first scalar spills: tools/kernel_resources.py).
template <int KID>
__device__ __forceinline__ void record_slow(const Status& st, unsigned site, unsigned expected, unsigned spins, bool gave_up) {
    if (!INET_RECORDER || !st.gdev) return;
    // (every temporary of this cold path is derived from a VECTOR zero: as uniform values they would sit in scalar registers,
    //  and the chain kernels have none to spare inside their step loops -- the first build of the recorder added 15-20 scalar
    //  spills to each of them, tools/kernel_resources.py; vector registers are free at a wait, between two steps)
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    unsigned* rec = st.gdev + kRecWord + z;
    __hip_atomic_fetch_add(rec, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!gave_up && spins < __hip_atomic_load(rec + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const unsigned n = __hip_atomic_fetch_add(rec + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n >= (unsigned)kRecEntries) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned* e = rec + 8 * (1 + n);
    e[0] = ((unsigned)KID + z) | ((xcc & 0xfu) << 8) | (gave_up ? 0x8000u : 0u) | (site << 16);
    e[1] = blockIdx.x + z; e[2] = expected + z; e[3] = spins + z;
    const unsigned long long now = wall_clock64();
    e[4] = (unsigned)now; e[5] = (unsigned)(now >> 32); e[6] = 0u; e[7] = 0u;
}
__device__ __forceinline__ void raise_timeout(Status status) {
    __hip_atomic_store(status.dev, (unsigned)ST_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (status.gdev) __hip_atomic_store(status.gdev, (unsigned)ST_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (status.host) __hip_atomic_fetch_add(status.host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int KID>
__device__ __forceinline__ bool wait_group(unsigned* counter, unsigned target, Status status, unsigned* flag) {
    if (threadIdx.x == 0) {
        unsigned ok = 1, spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > kSpinLimit ||
                ((spins & 63) == 0 && __hip_atomic_load(status.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ST_OK)) {
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(INET_CHAIN_POLL_SLEEP);
        }
        if (spins >= kSlowSpins) {                              // (a wait that gave up has polled at least kSlowSpins times)
            if (!ok) raise_timeout(status);
            record_slow<KID>(status, 0u, target, spins, ok == 0);
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;                                          // callers alternate between two flag words
}

// block -> (group, member).  Workgroup ids go round-robin over the 8 XCDs (32 CUs each) and a chain workgroup has its CU to
// itself, so the members of a group must sit on as many XCDs as they need CUs: xpg = ceil(members / 32) XCDs per group
// (1 up to H = 512; 2 for the 64 members of an H = 1024 layer -- with all 64 on ONE XCD half of them never become resident
// and the group runs into its bounded spin).  x = b % 8, k = b / 8:  group = x / xpg + (8 / xpg) * (k / mpx),
// member = (x % xpg) * mpx + k % mpx, mpx = members / xpg.  Launch blocks_for() blocks; those whose group >= groups leave at once.
__host__ __device__ inline int xcds_per_group(int members) { return members > 32 ? (members + 31) / 32 : 1; }
__host__ __device__ inline int blocks_for(int groups, int members) {
    const int xpg = xcds_per_group(members), gpr = 8 / xpg;
    return 8 * (members / xpg) * ((groups + gpr - 1) / gpr);
}
__device__ __forceinline__ void decode_block(int b, int members, int& group, int& member) {
    const int xpg = xcds_per_group(members), mpx = members / xpg, x = b & 7, k = b >> 3;
    group = x / xpg + (8 / xpg) * (k / mpx);
    member = (x % xpg) * mpx + k % mpx;
}

// acc[ms][slot g] += A[16*MS rows of the group's state, this wave's K quarter] x Wr[g]^T, where the A fragments are
// read from the exchange buffer `r` (fragment-major, S k-steps per 16-row block) at byte offset `base` with sc1 loads
// and the B fragments Wr[g][si] (k-step s0 + si) already sit in registers.  Loads run one chunk of 4 k-steps ahead of
// the MFMAs (double buffer, fully unrolled: all indices are compile-time).
template <int MS, int NG, int SQ>
__device__ __forceinline__ void contract(f32x4 (&acc)[MS][4], const f32x4 (&Wr)[NG][SQ], __amdgpu_buffer_rsrc_t r,
                                         int base, int rb0, int rb_last, int S, int s0, int lane) {
    constexpr int CH = SQ < 4 ? SQ : 4, NCH = SQ / CH;
    static_assert(SQ % CH == 0, "k-steps per wave must be a multiple of the chunk");
    f32x4 A[2][MS][CH];
    auto load = [&](int c, int buf) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            const int rb = min(rb0 + ms, rb_last);              // row blocks past the batch are clamped, never stored
#pragma unroll
            for (int i = 0; i < CH; ++i)
                A[buf][ms][i] = ld16_sc1(r, base + ((rb * S + s0 + c * CH + i) * 256 + lane * 4) * 4);
        }
    };
    load(0, 0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) load(c + 1, (c + 1) & 1);
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ms = 0; ms < MS; ++ms)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        acc[ms][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c & 1][ms][i][e], Wr[g][c * CH + i][e], acc[ms][g], 0, 0, 0);
    }
}

// Streamed variant of contract(): acc[ms][g] += A[16*MS rows of the group's state, this wave's K quarter] x Wr[g]^T, where the A fragments are read from
// the exchange buffer `r` (fragment-major, S k-steps per 16-row block) at byte offset `base` with sc1 loads and the B
// fragments Wr[g][si] (k-step s0 + si) already sit in registers.
//
// Schedule (the streamed form of ksplit.h, with B in registers): a ring of R k-steps of A fragments; R-1 k-steps are
// requested up front, then every k-step's MFMAs carry the loads of the k-step R-1 ahead, ONE load dealt in front of
// each row block's 4*NG MFMAs and pinned there with sched_barrier.  Left to itself hipcc sinks each load to its first
// use (load 4 -> wait -> 16 MFMAs -> load 4 ...: latency fully exposed, 12 us per backward step); issued as bursts of a
// whole chunk the loads stall the in-order wave in the issue stage (~100 cycles per 1 KB load while four waves stream)
// and the MFMA pipe idles behind them (13 us).  Dealt out, the issue stalls hide under MFMA execution.
template <int MS, int NG, int SQ, int R = 4>
__device__ __forceinline__ void contract_stream(f32x4 (&acc)[MS][4], const f32x4 (&Wr)[NG][SQ], __amdgpu_buffer_rsrc_t r,
                                         int base, int rb0, int rb_last, int S, int s0, int lane) {
    f32x4 A[R][MS];
    // per-lane byte offset of k-step s0 in row block ms, kept in vector registers on purpose (opaque_zero); k-steps
    // are 1 KB apart and ride in the instruction's immediate, the slot `base` in its scalar offset
    int vo[MS];
    const int oz = opaque_zero();
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) vo[ms] = (((min(rb0 + ms, rb_last) + oz) * S + s0) * 256 + lane * 4) * 4;   // clamped rows are never stored
#pragma unroll
    for (int d = 0; d < R - 1; ++d)
        if (d < SQ) {
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) A[d][ms] = ld16_sc1(r, vo[ms] + d * 1024, base);
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int si = 0; si < SQ; ++si) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            if (si + R - 1 < SQ) A[(si + R - 1) % R][ms] = ld16_sc1(r, vo[ms] + (si + R - 1) * 1024, base);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    acc[ms][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[si % R][ms][e], Wr[g][si][e], acc[ms][g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// One 16x16 block of the group state: rows of sub-tile p held row-major in LDS tile `xt` ([16*MS][16]) -> the member's
// 1 KB fragment block (row block rb, k-step ks) of the exchange buffer, one 16-byte sc1 store per lane of wave p.
__device__ __forceinline__ void publish_block(__amdgpu_buffer_rsrc_t r, int base, const float* xt, int p, int lane,
                                              int rb, int S, int ks) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xt + (p * 16 + (lane & 15)) * 16 + (lane >> 4) * 4);
    st16_sc1(r, base + ((rb * S + ks) * 256 + lane * 4) * 4, v);
}

}  // namespace chain

// runtime switch (inet_set_option key 4; INET_CHAIN=0 in the environment): 0 = per-step launches everywhere
int chain_enabled();
void chain_set_enabled(int on);
// host-mapped failure counter shared by every chain launch of the process (null if it could not be allocated)
unsigned* chain_host_status();
// process-wide device-memory twin of that counter (null if it could not be allocated): non-zero after any timeout, read by
// the Adam kernel (pw_adam) so that a failed step leaves the parameters untouched; inet_chain_status(reset) clears both
unsigned* chain_dev_status();
// behind the process-wide device status word: a scratch area that instrumented builds (INET_CHAIN2_STAMPS) write wall-clock stamps
// into -- chain::Status.gdev + kChainDiagWord -- and inet_debug_read() copies to the host; behind it the slow-wait recorder (chain::kRecWord)
constexpr int kChainDiagWord = chain::kDiagWord;
constexpr int kChainDiagBytes = chain::kDiagBytes;
constexpr size_t kChainStatusAreaBytes = 4 * (size_t)(chain::kRecWord + chain::kRecWords);
int chain_status_reset();
// host-mapped count of token indices outside [0, V) seen by a module's prologue (decoder.py:36-45 check_index raises
// ValueError; here the host raises at its next status read); null if it could not be allocated
unsigned* token_host_status();
inline chain::Status chain_status_for(unsigned* dev_word) { return chain::Status{dev_word, chain_host_status(), chain_dev_status()}; }
// Workgroups of ONE chain launch that are guaranteed to be resident at the same time on the current device: its compute
// units (hipDeviceAttributeMultiprocessorCount, cached per device) x one workgroup -- the chain kernels are built for one
// wave per SIMD (264..512 registers), so a CU never holds two of them.  A launch with more workgroups than this would
// leave members queued behind spinning ones: the *_chain_ok() predicates refuse it and the caller takes the per-step
// path.  (MI355X: 256.  A partition, a smaller part or INET_CHAIN_CUS=n gives n.)  What the query cannot see -- a CU
// mask, another process on the same GPU -- is caught by the bounded spin and reported through inet_chain_status().
int chain_capacity();
// test hook (inet_set_option key 6): arm a fault for the next forward GRU chain launch; chain_take_fault() returns and clears it
void chain_arm_fault(int on);
int chain_take_fault();
