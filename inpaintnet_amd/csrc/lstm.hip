// Fused LSTM step kernels + sequence driver (AnticipationRNN, config 5: torch.nn.LSTM(num_layers=1) cells stacked by
// lstm_with_activations, AnticipationRNN/anticipation_rnn_gauss_reg_model.py:14-39,110-133).
// Same geometry as the GRU step (ksplit.h): tile = 16*MS batch rows x 16 hidden units x {i,f,g,o} gates, the
// recurrent contraction h_prev[B,H] x W_hh[4H,H]^T streamed from L2 into v_mfma_f32_16x16x4_f32 fragments, gate
// math / cell update / backward saves in the epilogue.  Input-side pre-activations gi = x W_ih^T + b_ih are formed for
// all time steps at once by the batched GEMM (gemm.hip).
#include <cstdio>
#include "ksplit.h"
#include "chain.h"
#include "prof.h"
#include "seq.h"
#include "lstm.h"

using namespace ksplit;

namespace {

// sync area: one 256-byte block per group counter (gru_chain.h): 32 forward counters, 32 backward counters, status word
constexpr int kCounterStride = 64;
constexpr int kBwdCounters = 32 * kCounterStride;
constexpr int kStatusWord = 64 * kCounterStride;
constexpr int kSyncWords = kStatusWord + 4;
// chunked pipelines (lstm2_seq_*): one backward counter area per chunk behind the base area, zeroed by ONE memset per call
// instead of a 5 us fill in front of every chunk launch
constexpr int kMaxChunks = 16;
constexpr int kSyncWordsAll = kSyncWords + 60 + kMaxChunks * kBwdCounters;

// One launch in front of a forward chunk instead of three (memset of the counters, memset 0xFF of the armed ring slots,
// pack of the previous state into its slot: 16 us per chunk of a 150 us chain launch): blockIdx.y = job.
__global__ void lstm_chunk_prologue_kernel(unsigned* zero_words, int nzero, unsigned* fill_words, long nfill,
                                           const float* __restrict__ hprev, int B, int H, float* __restrict__ slot) {
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    if (blockIdx.y == 0) {
        for (long i = i0; i < nzero; i += stride) zero_words[i] = 0u;
        for (long i = i0; i < nfill; i += stride) fill_words[i] = 0xffffffffu;
    } else {
        const int S = H >> 4;                                        // pack_frag_kernel's layout (pointwise.hip)
        const long slots = (long)((B + 15) >> 4) * S * 64;
        for (long i = i0; i < slots; i += stride) {
            const int lane = (int)(i & 63);
            const long blk = i >> 6;
            const int sb = (int)(blk % S), rb = (int)(blk / S);
            const int row = 16 * rb + (lane & 15), k = 16 * sb + 4 * (lane >> 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < B) { const float* q = hprev + (long)row * H + k; v = make_float4(q[0], q[1], q[2], q[3]); }
            *reinterpret_cast<float4*>(slot + 4 * i) = v;
        }
    }
}

struct LstmFwdArgs {
    int B, H;
    const float* h_prev; const float* c_prev;     // [B,H]
    const float* W_hh; const float* b_hh;         // [4H,H], [4H]
    const float* gi;                              // [B,4H]
    float* h_new; float* c_new;                   // [B,H]
    float* sv;                                    // 6 x [B,H]: i, f, g, o, c_prev, tanh(c_new); or null
    long sv_stride;
};

struct LstmBwdArgs {
    int B, H;
    const float* dg_next;                         // [B,4H] gate gradients of the step processed before (null: none)
    const float* W_hhT;                           // [H,4H]
    const float* dout; const float* dout2;        // [B,H] external gradients into h(t) (nullable)
    const float* dc_next;                         // [B,H] dLoss/dc(t) carried from the later step (nullable)
    const float* dc_ext;                          // [B,H] external gradient into c(t) (final cell state), nullable
    const float* sv; long sv_stride;              // saves of THIS step (null => only write dh_out / dc_out)
    float* dg;                                    // [B,4H] gate gradients of this step
    float* dc_prev;                               // [B,H] dLoss/dc(t-1)
    float* db_ih; float* db_hh;                   // [4H] accumulated with atomics (nullable)
    float* dh_out;                                // [B,H] (init-gradient mode)
};

template <int MS>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(LstmFwdArgs P) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 4 * MS * 256];
    const int H = P.H, t = threadIdx.x;
    const int j0 = blockIdx.x * TH, row0 = blockIdx.y * (16 * MS);
    f32x4 acc[MS][4];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int brow[4] = {j0, H + j0, 2 * H + j0, 3 * H + j0};
    const int slot[4] = {0, 1, 2, 3};
    // epilogue operands requested from inside the contraction (ksplit.h hook), not after the reduce
    const int jc = j0 + (t & 15);
    float pg[MS][4], pcp[MS], pb[4];
    auto prefetch = [&](auto tag) {
        constexpr int I = decltype(tag)::value;
        if constexpr (I == -1) {
            kernarg_touch(P.gi, P.b_hh, P.c_prev);
        } else if constexpr (I == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a) pb[a] = P.b_hh[a * H + jc];
        } else if constexpr (I <= MS) {
            constexpr int p = I - 1;
            const int b = min(row0 + ((t + 256 * p) >> 4), P.B - 1);
#pragma unroll
            for (int a = 0; a < 4; ++a) pg[p][a] = P.gi[(long)b * 4 * H + a * H + jc];
            pcp[p] = P.c_prev[(long)b * H + jc];
        }
    };
    ksplit_segment<MS, 4>(acc, slot, P.h_prev, (long)H, row0, P.B, P.W_hh, (long)H, brow, H, t, prefetch);
    float v[MS][4];
    reduce_waves<MS, 4>(acc, lds, t, v);
#pragma unroll
    for (int p = 0; p < MS; ++p) {
        const int pos = t + 256 * p;
        const int b = row0 + (pos >> 4), j = jc;
        if (b >= P.B) continue;
        const float i = sigmoid_f(v[p][0] + pg[p][0] + pb[0]);
        const float f = sigmoid_f(v[p][1] + pg[p][1] + pb[1]);
        const float g = tanh_f(v[p][2] + pg[p][2] + pb[2]);
        const float o = sigmoid_f(v[p][3] + pg[p][3] + pb[3]);
        const long q = (long)b * H + j;
        const float cp = pcp[p];
        const float c = f * cp + i * g;
        const float tc = tanh_f(c);
        P.c_new[q] = c;
        P.h_new[q] = o * tc;
        if (P.sv) {
            float* s = P.sv + q;
            const long st = P.sv_stride;
            s[0] = i; s[st] = f; s[2 * st] = g; s[3 * st] = o; s[4 * st] = cp; s[5 * st] = tc;
        }
    }
}

// dh = dg_next W_hh + dout + dout2 ;  do = dh tanh(c) ; dc = dc_next + dc_ext + dh o (1 - tanh(c)^2)
// di = dc g ; df = dc c_prev ; dg = dc i ; dc_prev = dc f ; pre-activation gradients through sigmoid / tanh.
template <int MS>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(LstmBwdArgs P) {
    __shared__ __attribute__((aligned(16))) float lds[(4 * MS * 256 > 1024) ? 4 * MS * 256 : 1024];
    const int H = P.H, t = threadIdx.x;
    const int j0 = blockIdx.x * TH, row0 = blockIdx.y * (16 * MS);
    const int jc = j0 + (t & 15);
    float pe[MS][4], psv[MS][6];
    auto prefetch = [&](auto tag) {
        constexpr int I = decltype(tag)::value;
        if constexpr (I == -1) {
            kernarg_touch(P.dout, P.dout2, P.dc_next, P.dc_ext, P.sv, P.sv_stride);
        } else if constexpr (I >= 1 && I <= MS) {
            constexpr int p = I - 1;
            const int b = min(row0 + ((t + 256 * p) >> 4), P.B - 1);
            const long q = (long)b * H + jc;
            pe[p][0] = P.dout ? P.dout[q] : 0.f;
            pe[p][1] = P.dout2 ? P.dout2[q] : 0.f;
            pe[p][2] = P.dc_next ? P.dc_next[q] : 0.f;
            pe[p][3] = P.dc_ext ? P.dc_ext[q] : 0.f;
#pragma unroll
            for (int a = 0; a < 6; ++a) psv[p][a] = P.sv ? P.sv[q + a * P.sv_stride] : 0.f;
        }
    };
    float v[MS][1];
#pragma unroll
    for (int p = 0; p < MS; ++p) v[p][0] = 0.f;
    if (P.dg_next) {
        f32x4 acc[MS][4];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int brow[1] = {j0};
        const int slot[1] = {0};
        ksplit_segment<MS, 1>(acc, slot, P.dg_next, (long)4 * H, row0, P.B, P.W_hhT, (long)4 * H, brow, 4 * H, t, prefetch);
        reduce_waves<MS, 1>(acc, lds, t, v);
    } else {
        prefetch(HookTag<-1>{});
        hook_pieces<MS + 1>(prefetch);
    }
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < MS; ++p) {
        const int pos = t + 256 * p;
        const int b = row0 + (pos >> 4), j = jc;
        if (b >= P.B) continue;
        const long q = (long)b * H + j;
        const float dh = v[p][0] + pe[p][0] + pe[p][1];
        if (!P.sv) {                       // gradient wrt the initial hidden / cell state
            if (P.dh_out) P.dh_out[q] = dh;
            continue;
        }
        const float i = psv[p][0], f = psv[p][1], g = psv[p][2], o = psv[p][3], cp = psv[p][4], tc = psv[p][5];
        const float dc = dh * o * (1.f - tc * tc) + pe[p][2] + pe[p][3];
        const float di = dc * g * i * (1.f - i);
        const float df = dc * cp * f * (1.f - f);
        const float dgg = dc * i * (1.f - g * g);
        const float dob = dh * tc * o * (1.f - o);
        P.dc_prev[q] = dc * f;
        float* d = P.dg + (long)b * 4 * H;
        d[j] = di; d[H + j] = df; d[2 * H + j] = dgg; d[3 * H + j] = dob;
        bs[0] += di; bs[1] += df; bs[2] += dgg; bs[3] += dob;
    }
    if (P.sv && P.db_ih) {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a) lds[a * 256 + t] = bs[a];
        __syncthreads();
        if (t < 64) {
            const int a = t >> 4, c = t & 15;
            float sum = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) sum += lds[a * 256 + rr * 16 + c];
            unsafeAtomicAdd(P.db_ih + a * H + j0 + c, sum);
            unsafeAtomicAdd(P.db_hh + a * H + j0 + c, sum);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Chain kernels (chain.h): all T steps of one LSTM layer in ONE launch.  W_hh (fwd) / W_hh^T (bwd) slices live in
// registers for the whole sequence, the cell state c and its gradient are per-thread registers, and only h (fwd) /
// the gate gradients (bwd) travel between the H/16 members of a row-tile group, once per step.
// ---------------------------------------------------------------------------------------------------------------------
struct LstmChainFwdArgs {
    int B, H, T, reverse, members;
    const float* gi;                              // [T,B,4H] input-side pre-activations (x W_ih^T + b_ih)
    const float* W_hh; const float* b_hh;         // [4H,H], [4H]
    const float* c0;                              // [B,H]
    float* out; float* cseq;                      // [T,B,H]
    float* sv; long sv_stride;                    // 6 x [T,B,H] (i,f,g,o,c_prev,tanh c) or null
    float* hx;                                    // exchange [2][rows16][H] fragment-major; slot 1 holds h0
    unsigned* counters; chain::Status status;
    int phase;                                    // tagged hand-off: steps the ring has carried before this launch (chunked layers)
    int xrot;                                     // the launch's groups start at XCD xrot (two chains side by side: different XCDs)
};

template <int MS, int SQ>                          // SQ = H/64: k-steps of 16 per wave
__global__ __launch_bounds__(256) void lstm_chain_fwd_kernel(LstmChainFwdArgs P) {
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * MS * 256];
    __shared__ __attribute__((aligned(16))) float xt[MS * 256];
    __shared__ unsigned flag[2];
    int group, member;
    chain::decode_block((blockIdx.x & ~7) | ((blockIdx.x + 8 - P.xrot) & 7), P.members, group, member);
    const int row0 = group * 16 * MS;
    if (row0 >= P.B) return;
    const int H = P.H, B = P.B, S = H >> 4, t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int j0 = member * 16, jc = j0 + (t & 15);
    const int rb0 = row0 >> 4, rb_last = (B - 1) >> 4;
    const int slot_bytes = ((B + 15) >> 4) * 16 * H * 4;
    // this wave's W_hh fragments: 4 gates x SQ k-steps, resident for all T steps
    f32x4 Wr[4][SQ];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int si = 0; si < SQ; ++si)
            Wr[g][si] = ld4u(P.W_hh + (long)(g * H + j0 + i16) * H + 16 * (w * SQ + si) + 4 * q);
    float bh[4], c[MS];
#pragma unroll
    for (int a = 0; a < 4; ++a) bh[a] = P.b_hh[a * H + jc];
#pragma unroll
    for (int p = 0; p < MS; ++p) c[p] = P.c0[(long)min(row0 + ((t + 256 * p) >> 4), B - 1) * H + jc];
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(P.hx);
    for (int step = 0; step < P.T; ++step) {
        const int tt = P.reverse ? P.T - 1 - step : step;
        float pg[MS][4];                           // does not depend on h: requested before the wait
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int b = min(row0 + ((t + 256 * p) >> 4), B - 1);
#pragma unroll
            for (int a = 0; a < 4; ++a) pg[p][a] = P.gi[((long)tt * B + b) * 4 * H + a * H + jc];
        }
        if (step > 0 && !chain::wait_group<chain::K_LSTM_FWD>(P.counters + group * kCounterStride, (unsigned)(step * P.members), P.status, &flag[step & 1])) return;
        f32x4 acc[MS][4];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        chain::contract_stream<MS, 4, SQ>(acc, Wr, rs, ((step + 1) & 1) * slot_bytes, rb0, rb_last, S, w * SQ, lane);
        float v[MS][4];
        reduce_waves<MS, 4>(acc, red, t, v);
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int rl = (t + 256 * p) >> 4;
            const int b = row0 + rl;
            const float i = sigmoid_f(v[p][0] + pg[p][0] + bh[0]);
            const float f = sigmoid_f(v[p][1] + pg[p][1] + bh[1]);
            const float g = tanh_f(v[p][2] + pg[p][2] + bh[2]);
            const float o = sigmoid_f(v[p][3] + pg[p][3] + bh[3]);
            const float cp = c[p];
            const float cn = f * cp + i * g;
            const float tc = tanh_f(cn);
            const float h = o * tc;
            c[p] = cn;
            xt[rl * 16 + (t & 15)] = h;
            if (b < B) {
                const long qo = ((long)tt * B + b) * H + jc;
                P.out[qo] = h;
                P.cseq[qo] = cn;
                if (P.sv) {
                    float* sp = P.sv + qo;
                    const long st = P.sv_stride;
                    sp[0] = i; sp[st] = f; sp[2 * st] = g; sp[3 * st] = o; sp[4 * st] = cp; sp[5 * st] = tc;
                }
            }
        }
        __syncthreads();
        if (t < 64 * MS && rb0 + (t >> 6) <= rb_last)
            chain::publish_block(rs, (step & 1) * slot_bytes, xt, t >> 6, lane, rb0 + (t >> 6), S, member);
        chain::arrive(P.counters + group * kCounterStride);
    }
}

// The same layer with a DATA-DRIVEN hand-off (the default for small tiles): no counter.  The exchange is a ring of 4 slots; a slot that
// is about to receive step t's state holds a sentinel (all bits set: no finite float) in every element, the members poll the
// fragments they need themselves until no lane sees the sentinel, and every producer re-arms the slot of step t + 2 right
// after publishing step t (its previous contents, step t - 2, have been consumed by everybody: a member that publishes step t
// has read all of step t - 1, so all members have finished step t - 2's consumers).  Per step this drops the producer's
// store drain + barrier + atomic and the consumer's counter round trip + barrier: what is left is one store -> load
// latency through L2.  The host arms slots 0 and 1 (memset 0xFF) and packs the initial state into slot 3 -- for a layer's first
// launch.  The chunks of a chunked layer go on with the same ring (P.phase = steps carried so far): the previous launch's last
// step left its state in slot (phase - 1) & 3, fragment-major as the consumers want it, and its last two steps armed slots
// phase & 3 and (phase + 1) & 3 -- nothing to prepare, no launch in front of the chunk (round 4).
// A lane's 16-byte element comes from ONE 16-byte store of one producer lane; all four words are checked, so a torn view of
// that store would only delay the consumer, never feed it a sentinel.
template <int MS, int SQ>                          // SQ = H/64: k-steps of 16 per wave
__global__ __launch_bounds__(256) void lstm_chain_fwd_tag_kernel(LstmChainFwdArgs P) {
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * MS * 256];
    __shared__ __attribute__((aligned(16))) float xt[MS * 256];
    int group, member;
    chain::decode_block((blockIdx.x & ~7) | ((blockIdx.x + 8 - P.xrot) & 7), P.members, group, member);
    const int row0 = group * 16 * MS;
    if (row0 >= P.B) return;
    const int H = P.H, B = P.B, S = H >> 4, t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int j0 = member * 16, jc = j0 + (t & 15);
    const int rb0 = row0 >> 4, rb_last = (B - 1) >> 4;
    const int slot_bytes = ((B + 15) >> 4) * 16 * H * 4;
    f32x4 Wr[4][SQ];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int si = 0; si < SQ; ++si)
            Wr[g][si] = ld4u(P.W_hh + (long)(g * H + j0 + i16) * H + 16 * (w * SQ + si) + 4 * q);
    float bh[4], c[MS];
#pragma unroll
    for (int a = 0; a < 4; ++a) bh[a] = P.b_hh[a * H + jc];
#pragma unroll
    for (int p = 0; p < MS; ++p) c[p] = P.c0[(long)min(row0 + ((t + 256 * p) >> 4), B - 1) * H + jc];
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(P.hx);
    int fo[MS];                                    // byte offset of this lane's fragment of k-step w*SQ in row block ms
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) fo[ms] = ((min(rb0 + ms, rb_last) * S + w * SQ) * 256 + lane * 4) * 4;
    const f32x4 armed = __builtin_bit_cast(f32x4, chain::u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
    const int ph = P.phase & 3;                    // the ring goes on where the layer's previous chunk left it
    for (int step = 0; step < P.T; ++step) {
        const int tt = P.reverse ? P.T - 1 - step : step;
        float pg[MS][4];                           // does not depend on h: requested before the poll
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int b = min(row0 + ((t + 256 * p) >> 4), B - 1);
#pragma unroll
            for (int a = 0; a < 4; ++a) pg[p][a] = P.gi[((long)tt * B + b) * 4 * H + a * H + jc];
        }
        // poll the fragments of the previous step's state (slot (step - 1) & 3) until none of them is armed
        f32x4 A[MS][SQ];
        const int in_base = ((ph + step + 3) & 3) * slot_bytes;
        for (unsigned spins = 0;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int ms = 0; ms < MS; ++ms)
#pragma unroll
                for (int si = 0; si < SQ; ++si) {
                    A[ms][si] = chain::ld16_sc1(rs, fo[ms] + si * 1024, in_base);
                    const chain::u32x4 bits = __builtin_bit_cast(chain::u32x4, A[ms][si]);
                    ok = ok && bits[0] != 0xffffffffu && bits[1] != 0xffffffffu && bits[2] != 0xffffffffu && bits[3] != 0xffffffffu;
                }
            if (__all(ok)) {
                if (spins > chain::kSlowSpins && lane == 0) chain::record_slow<chain::K_LSTM_FWD>(P.status, 3u, (unsigned)step, spins, false);
                break;
            }
            if (spins > chain::kSpinLimit ||
                ((spins & 63) == 63 && __hip_atomic_load(P.status.dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != chain::ST_OK)) {
                if (lane == 0) { chain::raise_timeout(P.status); chain::record_slow<chain::K_LSTM_FWD>(P.status, 3u, (unsigned)step, spins, true); }
                break;                             // carry on with what is there: the host reports the launch as failed
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");         // the loads above must be issued again
        }
        f32x4 acc[MS][4];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int si = 0; si < SQ; ++si)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ms = 0; ms < MS; ++ms)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[ms][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ms][si][e], Wr[g][si][e], acc[ms][g], 0, 0, 0);
        float v[MS][4];
        reduce_waves<MS, 4>(acc, red, t, v);
        float ev[MS][8];
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int rl = (t + 256 * p) >> 4;
            const float i = sigmoid_f(v[p][0] + pg[p][0] + bh[0]);
            const float f = sigmoid_f(v[p][1] + pg[p][1] + bh[1]);
            const float g = tanh_f(v[p][2] + pg[p][2] + bh[2]);
            const float o = sigmoid_f(v[p][3] + pg[p][3] + bh[3]);
            const float cp = c[p];
            const float cn = f * cp + i * g;
            const float tc = tanh_f(cn);
            const float h = o * tc;
            c[p] = cn;
            xt[rl * 16 + (t & 15)] = h;
            ev[p][0] = i; ev[p][1] = f; ev[p][2] = g; ev[p][3] = o; ev[p][4] = cp; ev[p][5] = tc; ev[p][6] = h; ev[p][7] = cn;
        }
        __syncthreads();
        if (t < 64 * MS && rb0 + (t >> 6) <= rb_last) {
            const int rb = rb0 + (t >> 6);
            chain::publish_block(rs, ((ph + step) & 3) * slot_bytes, xt, t >> 6, lane, rb, S, member);
            chain::st16_sc1(rs, ((ph + step + 2) & 3) * slot_bytes + ((rb * S + member) * 256 + lane * 4) * 4, armed);   // re-arm
        }
        // (round 4) what nobody in the launch reads -- output, cell state, the six saves -- is stored BEHIND the hand-off stores:
        // the memory pipe is in order, and eight scalar stores per thread used to go first
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int b = row0 + ((t + 256 * p) >> 4);
            if (b < B) {
                const long qo = ((long)tt * B + b) * H + jc;
                P.out[qo] = ev[p][6];
                P.cseq[qo] = ev[p][7];
                if (P.sv) {
                    float* sp = P.sv + qo;
                    const long st = P.sv_stride;
                    sp[0] = ev[p][0]; sp[st] = ev[p][1]; sp[2 * st] = ev[p][2]; sp[3 * st] = ev[p][3]; sp[4 * st] = ev[p][4]; sp[5 * st] = ev[p][5];
                }
            }
        }
        __syncthreads();                           // xt is rewritten by the next step's gates
    }
}

struct LstmChainBwdArgs {
    int B, H, T, reverse, members;
    const float* W_hhT;                           // [H,4H]
    const float* dout;                            // [T,B,H] or null
    const float* dhT; const float* dcT;           // [B,H] or null: gradients into the final state
    const float* sv; long sv_stride;              // forward saves
    float* dg;                                    // [T,B,4H] gate gradients (row-major: the weight-gradient GEMMs read it)
    float* dh0; float* dc0;                       // [B,H] or null: gradients wrt the initial state
    float* db_ih; float* db_hh;                   // [4H] accumulated (nullable)
    float* gx;                                    // exchange [2][rows16][4H] fragment-major
    unsigned* counters; chain::Status status;
    int xrot;                                     // the launch's groups start at XCD xrot (two chains side by side: different XCDs)
};

template <int MS, int SQ>                          // SQ = H/16: k-steps of 16 per wave over K = 4H
__global__ __launch_bounds__(256) void lstm_chain_bwd_kernel(LstmChainBwdArgs P) {
    __shared__ __attribute__((aligned(16))) float red[4 * MS * 256];
    __shared__ __attribute__((aligned(16))) float xt[4][MS * 256];
    __shared__ unsigned flag[2];
    int group, member;
    chain::decode_block((blockIdx.x & ~7) | ((blockIdx.x + 8 - P.xrot) & 7), P.members, group, member);
    const int row0 = group * 16 * MS;
    if (row0 >= P.B) return;
    const int H = P.H, B = P.B, T = P.T, S4 = (4 * H) >> 4, t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    const int j0 = member * 16, jc = j0 + (t & 15);
    const int rb0 = row0 >> 4, rb_last = (B - 1) >> 4;
    const int slot_bytes = ((B + 15) >> 4) * 16 * 4 * H * 4;
    f32x4 Wr[1][SQ];                               // rows j0..j0+15 of W_hh^T, this wave's quarter of K = 4H
#pragma unroll
    for (int si = 0; si < SQ; ++si) Wr[0][si] = ld4u(P.W_hhT + (long)(j0 + i16) * 4 * H + 16 * (w * SQ + si) + 4 * q);
    const __amdgpu_buffer_rsrc_t rs = chain::make_rsrc(P.gx);
    float dc[MS], bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < MS; ++p) dc[p] = 0.f;
    for (int step = T - 1; step >= -1; --step) {
        // step == -1: only dh0 = dg(first step) . W_hh
        const bool tail = step < 0;
        if (tail && !P.dh0) break;
        const int tt = tail ? 0 : (P.reverse ? T - 1 - step : step);
        float pe[MS][2], psv[MS][6];
        if (!tail) {
#pragma unroll
            for (int p = 0; p < MS; ++p) {
                const int b = min(row0 + ((t + 256 * p) >> 4), B - 1);
                const long qo = ((long)tt * B + b) * H + jc, q2 = (long)b * H + jc;
                pe[p][0] = P.dout ? P.dout[qo] : 0.f;
                pe[p][1] = 0.f;
                if (step == T - 1) {
                    if (P.dhT) pe[p][0] += P.dhT[q2];
                    if (P.dcT) pe[p][1] = P.dcT[q2];
                }
#pragma unroll
                for (int a = 0; a < 6; ++a) psv[p][a] = P.sv[qo + a * P.sv_stride];
            }
        }
        float v[MS][1];
#pragma unroll
        for (int p = 0; p < MS; ++p) v[p][0] = 0.f;
        if (step != T - 1) {
            if (!chain::wait_group<chain::K_LSTM_BWD>(P.counters + group * kCounterStride, (unsigned)((T - 1 - step) * P.members), P.status, &flag[step & 1])) return;
            f32x4 acc[MS][4];
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) acc[ms][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (a ring of 8 k-steps: with one gate block per k-step -- 4 MFMAs, 60 ns -- the GRU chains' ring of 4 covers a third
            //  of the L2 latency; 8: +2 % on the AnticipationRNN step, 16: +1 %)
            chain::contract_stream<MS, 1, SQ, 8>(acc, Wr, rs, ((step + 1) & 1) * slot_bytes, rb0, rb_last, S4, w * SQ, lane);
            reduce_waves<MS, 1>(acc, red, t, v);
        }
        if (tail) {
#pragma unroll
            for (int p = 0; p < MS; ++p) {
                const int b = row0 + ((t + 256 * p) >> 4);
                if (b < B) P.dh0[(long)b * H + jc] = v[p][0];
            }
            break;
        }
        float eg[MS][4];
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int rl = (t + 256 * p) >> 4;
            const float dh = v[p][0] + pe[p][0];
            const float i = psv[p][0], f = psv[p][1], g = psv[p][2], o = psv[p][3], cp = psv[p][4], tc = psv[p][5];
            const float dct = dh * o * (1.f - tc * tc) + dc[p] + pe[p][1];
            const float di = dct * g * i * (1.f - i);
            const float df = dct * cp * f * (1.f - f);
            const float dgg = dct * i * (1.f - g * g);
            const float dob = dh * tc * o * (1.f - o);
            dc[p] = dct * f;
            const int xo = rl * 16 + (t & 15);
            xt[0][xo] = di; xt[1][xo] = df; xt[2][xo] = dgg; xt[3][xo] = dob;
            eg[p][0] = di; eg[p][1] = df; eg[p][2] = dgg; eg[p][3] = dob;
        }
        __syncthreads();
        // publish the 4 gate blocks of every row sub-tile: (4 * MS) KB, one wave per block round-robin
        for (int blk = t >> 6; blk < 4 * MS; blk += 4) {
            const int g = blk / MS, p = blk % MS;
            if (rb0 + p <= rb_last)
                chain::publish_block(rs, (step & 1) * slot_bytes, xt[g], p, lane, rb0 + p, S4, g * (H >> 4) + member);
        }
        chain::arrive(P.counters + group * kCounterStride);
        // the gate gradients nobody in the launch reads leave AFTER the hand-off (round 4: arrive() drains every outstanding store
        // of the wave, and these four per thread used to sit in front of it)
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int b = row0 + ((t + 256 * p) >> 4);
            if (b < B) {
                float* d = P.dg + ((long)tt * B + b) * 4 * H;
                d[jc] = eg[p][0]; d[H + jc] = eg[p][1]; d[2 * H + jc] = eg[p][2]; d[3 * H + jc] = eg[p][3];
                bs[0] += eg[p][0]; bs[1] += eg[p][1]; bs[2] += eg[p][2]; bs[3] += eg[p][3];
            }
        }
    }
    if (P.dc0) {
#pragma unroll
        for (int p = 0; p < MS; ++p) {
            const int b = row0 + ((t + 256 * p) >> 4);
            if (b < B) P.dc0[(long)b * H + jc] = dc[p];
        }
    }
    if (P.db_ih) {                                 // bias gradients: one tile reduction for the whole sequence
        __syncthreads();
        float* lb = &xt[0][0];
#pragma unroll
        for (int a = 0; a < 4; ++a) lb[a * 256 + t] = bs[a];
        __syncthreads();
        if (t < 64) {
            const int a = t >> 4, cc = t & 15;
            float sum = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) sum += lb[a * 256 + rr * 16 + cc];
            unsafeAtomicAdd(P.db_ih + a * H + j0 + cc, sum);
            unsafeAtomicAdd(P.db_hh + a * H + j0 + cc, sum);
        }
    }
}

template <typename K, typename A>
int launch_chain(K kernel, const A& a, int groups, hipStream_t s) {
    hipLaunchKernelGGL(kernel, dim3(chain::blocks_for(groups, a.members)), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// rows per group (16 * MS): the smallest tile that still fits the launch on the chip -- a step of these chains is mostly
// hand-off latency plus the MFMAs of ONE workgroup (B = 32, H = 256: 1.9 of 4.1 us with 32 rows per workgroup), so more,
// smaller groups shorten every step (AnticipationRNN: two 16-row groups instead of one 32-row group)
inline int chain_ms(int B, int H) {
    constexpr int force = 0;
    if (force == 1 || force == 2 || force == 4) return force;
    for (int ms = 1; ms <= 4; ms *= 2) {
        const int groups = (B + 16 * ms - 1) / (16 * ms);
        if (groups * (H / 16) <= chain_capacity() && groups <= 32) return ms;
    }
    return 4;
}
inline bool lstm_chain_ok(int B, int H) {
    if (!chain_enabled() || (H != 256 && H != 512)) return false;
    const int ms = chain_ms(B, H), groups = (B + 16 * ms - 1) / (16 * ms);
    return groups * (H / 16) <= chain_capacity() && groups <= 32;   // every workgroup of the launch must be resident at once
}

int launch_fwd(const LstmFwdArgs& a, hipStream_t s) {
    dim3 grid(a.H / TH, (a.B + 31) / 32);
    ProfScope prof(PROF_GRU_FWD, 2.0 * a.B * 4.0 * a.H * a.H, s);
    hipLaunchKernelGGL(lstm_step_fwd_kernel<2>, grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_bwd(const LstmBwdArgs& a, hipStream_t s) {
    dim3 grid(a.H / TH, (a.B + 31) / 32);
    ProfScope prof(PROF_GRU_BWD, a.dg_next ? 2.0 * a.B * 4.0 * a.H * a.H : 0.0, s);
    hipLaunchKernelGGL(lstm_step_bwd_kernel<2>, grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

struct LstmWs { float *zeros, *cseq, *sv, *whhT, *dc, *hx, *gx, *carry; unsigned* sync; };
size_t lstm_carve(int B, int T, int H, int save, void* base, LstmWs& w) {
    Carver cv(base);
    const size_t BH = (size_t)B * H;
    w.zeros = cv.take<float>(BH);
    w.cseq = cv.take<float>((size_t)T * BH);
    w.sv = save ? cv.take<float>(6 * (size_t)T * BH) : nullptr;
    w.whhT = save ? cv.take<float>((size_t)4 * H * H) : nullptr;
    w.dc = save ? cv.take<float>(2 * BH) : nullptr;
    w.hx = cv.take<float>(4 * pk_floats(B, H));           // 2 slots (counter hand-off) or a ring of 4 (tagged hand-off)
    w.gx = save ? cv.take<float>(2 * pk_floats(B, 4 * H)) : nullptr;
    w.carry = save ? cv.take<float>(4 * BH) : nullptr;        // (dh, dc) handed from one chunk of a chunked backward to the next, x2
    w.sync = cv.take<unsigned>(kSyncWordsAll);
    return cv.bytes();
}

}  // namespace

size_t lstm_ws_bytes(int B, int T, int H, int save) {
    LstmWs w;
    return lstm_carve(B, T, H, save, nullptr, w);
}

// gi [T,B,4H] (time-major, includes b_ih); out [T,B,H]; h0/c0 [B,H] or null; hT/cT [B,H] or null.
int lstm_seq_fwd(int B, int T, int H, const float* gi, const float* W_hh, const float* b_hh, const float* h0,
                 const float* c0, int reverse, float* out, float* hT, float* cT, void* ws, int save, hipStream_t s) {
    LstmWs w;
    lstm_carve(B, T, H, save, ws, w);
    const long BH = (long)B * H, TBH = (long)T * BH;
    if ((!h0 || !c0) && pw_zero(w.zeros, BH, s) != 0) return -2;
    if (lstm_chain_ok(B, H)) {
        const int ms = chain_ms(B, H), groups = (B + 16 * ms - 1) / (16 * ms);
        if (hipMemsetAsync(w.sync, 0, kSyncWords * sizeof(unsigned), s) != hipSuccess) return -2;
        INET_TRY(pw_pack_frag(h0 ? h0 : w.zeros, H, B, H, w.hx + pk_floats(B, H), 0, 1, 0, 0, s));   // slot 1 = h0
        LstmChainFwdArgs a{};
        a.B = B; a.H = H; a.T = T; a.reverse = reverse; a.members = H / 16;
        a.gi = gi; a.W_hh = W_hh; a.b_hh = b_hh; a.c0 = c0 ? c0 : w.zeros;
        a.out = out; a.cseq = w.cseq;
        if (save) { a.sv = w.sv; a.sv_stride = TBH; }
        a.hx = w.hx; a.counters = w.sync; a.status = chain_status_for(w.sync + kStatusWord);
        char label[64];
        std::snprintf(label, sizeof label, "lstm_chain_fwd ms%d T%d B%d H%d", ms, T, B, H);
        ProfScope prof(PROF_GRU_FWD, 2.0 * T * B * 4.0 * H * H, s, label,
                       4.0 * (4.0 * H * H + (double)T * B * H * (4 + 2 + (save ? 6 : 0))));
        int rc;
        if (H == 256) rc = ms == 1 ? launch_chain(lstm_chain_fwd_kernel<1, 4>, a, groups, s)
                           : ms == 2 ? launch_chain(lstm_chain_fwd_kernel<2, 4>, a, groups, s)
                                     : launch_chain(lstm_chain_fwd_kernel<4, 4>, a, groups, s);
        else rc = ms == 1 ? launch_chain(lstm_chain_fwd_kernel<1, 8>, a, groups, s)
                  : ms == 2 ? launch_chain(lstm_chain_fwd_kernel<2, 8>, a, groups, s)
                            : launch_chain(lstm_chain_fwd_kernel<4, 8>, a, groups, s);
        INET_TRY(rc);
    } else
    for (int step = 0; step < T; ++step) {
        const int t = reverse ? T - 1 - step : step;
        const int tp = reverse ? t + 1 : t - 1;
        LstmFwdArgs a{};
        a.B = B; a.H = H;
        a.h_prev = step == 0 ? (h0 ? h0 : w.zeros) : out + (long)tp * BH;
        a.c_prev = step == 0 ? (c0 ? c0 : w.zeros) : w.cseq + (long)tp * BH;
        a.W_hh = W_hh; a.b_hh = b_hh;
        a.gi = gi + (long)t * B * 4 * H;
        a.h_new = out + (long)t * BH; a.c_new = w.cseq + (long)t * BH;
        if (save) { a.sv = w.sv + (long)t * BH; a.sv_stride = TBH; }
        INET_TRY(launch_fwd(a, s));
    }
    const int tl = reverse ? 0 : T - 1;
    if (hT && pw_copy_bytes(hT, out + (long)tl * BH, BH * sizeof(float), s) != 0) return -2;
    if (cT && pw_copy_bytes(cT, w.cseq + (long)tl * BH, BH * sizeof(float), s) != 0) return -2;
    return 0;
}

namespace {

// Forward steps [s_lo, s_lo + nt) of a T-step layer as ONE chain launch that continues from (hprev, cprev) [B,H]
// (the state after step s_lo - 1: rows of `out` / `cseq`, or zeros).  The kernel sees a sequence of nt steps whose
// buffers start at the chunk's lowest time index; the saves keep the full sequence's array stride.
int lstm_chunk_fwd(int B, int T, int H, const float* gi, const float* W_hh, const float* b_hh, const float* hprev,
                   const float* cprev, int reverse, float* out, LstmWs& w, int save, int s_lo, int nt, hipStream_t s, int xrot = 0,
                   bool ring_goes_on = false) {
    const long BH = (long)B * H, TBH = (long)T * BH;
    const int ms = chain_ms(B, H), groups = (B + 16 * ms - 1) / (16 * ms);
    const long t_lo = reverse ? T - (s_lo + nt) : s_lo;
    // tagged hand-off (lstm_chain_fwd_tag_kernel) for the small tiles; the counter protocol otherwise.  (The same
    // for the backward chain -- 16 fragments per lane to poll, four gate blocks per member to wait for -- measured slower
    // than its counter: 9.34 vs 9.23 ms per AnticipationRNN step with both, 8.87 with the forward chains only.)
    constexpr bool tag_on = true;
    const bool tagged = tag_on && H == 256 && chain_ms(B, H) <= 2;
    // counters zeroed; tagged: slots 0, 1 armed and slot 3 = the previous step's h; counter protocol: slot 1 = that h
    // (a tagged chunk behind another chunk of the same layer and call finds all of that in the ring: ring_goes_on)
    const bool cont = tagged && ring_goes_on && s_lo > 0;
    if (!cont) {
        hipLaunchKernelGGL(lstm_chunk_prologue_kernel, dim3(16, 2), dim3(256), 0, s, w.sync, kSyncWords,
                           reinterpret_cast<unsigned*>(w.hx), tagged ? 2L * (long)pk_floats(B, H) : 0L, hprev, B, H,
                           w.hx + (tagged ? 3 : 1) * pk_floats(B, H));
        if (hipGetLastError() != hipSuccess) return -2;
    }
    LstmChainFwdArgs a{};
    a.B = B; a.H = H; a.T = nt; a.reverse = reverse; a.members = H / 16;
    a.gi = gi + t_lo * B * 4 * H; a.W_hh = W_hh; a.b_hh = b_hh; a.c0 = cprev;
    a.out = out + t_lo * BH; a.cseq = w.cseq + t_lo * BH;
    if (save) { a.sv = w.sv + t_lo * BH; a.sv_stride = TBH; }
    a.hx = w.hx; a.counters = w.sync; a.status = chain_status_for(w.sync + kStatusWord); a.xrot = xrot & 7;
    a.phase = cont ? s_lo : 0;
    char label[64];
    std::snprintf(label, sizeof label, "lstm_chain_fwd ms%d T%d B%d H%d", ms, nt, B, H);
    ProfScope prof(PROF_GRU_FWD, 2.0 * nt * B * 4.0 * H * H, s, label,
                   4.0 * (4.0 * H * H + (double)nt * B * H * (4 + 2 + (save ? 6 : 0))));
    if (tagged)
        return ms == 1 ? launch_chain(lstm_chain_fwd_tag_kernel<1, 4>, a, groups, s) : launch_chain(lstm_chain_fwd_tag_kernel<2, 4>, a, groups, s);
    if (H == 256) return ms == 1 ? launch_chain(lstm_chain_fwd_kernel<1, 4>, a, groups, s)
                       : ms == 2 ? launch_chain(lstm_chain_fwd_kernel<2, 4>, a, groups, s)
                                 : launch_chain(lstm_chain_fwd_kernel<4, 4>, a, groups, s);
    return ms == 1 ? launch_chain(lstm_chain_fwd_kernel<1, 8>, a, groups, s)
           : ms == 2 ? launch_chain(lstm_chain_fwd_kernel<2, 8>, a, groups, s)
                     : launch_chain(lstm_chain_fwd_kernel<4, 8>, a, groups, s);
}

// Backward through forward steps [s_lo, s_lo + nt) as one chain launch: (dhT, dcT) = the gradient into the state after
// the chunk's last step (from the chunk that ran before this one, or null), (dh0, dc0) = the gradient into the state in
// front of its first step (for the next chunk, or null).  w.whhT must hold W_hh^T.
// `area` >= 0: the chunk's own pre-zeroed counter area (lstm2_seq_bwd zeroes all of them with one memset); < 0: the base area,
// zeroed here.
int lstm_chunk_bwd(int B, int T, int H, const float* dout, const float* dhT, const float* dcT, int reverse, float* dgi,
                   float* db_ih, float* db_hh, float* dh0, float* dc0, LstmWs& w, int s_lo, int nt, hipStream_t s, int area = -1,
                   int xrot = 0) {
    const long BH = (long)B * H, TBH = (long)T * BH;
    const int ms = chain_ms(B, H), groups = (B + 16 * ms - 1) / (16 * ms);
    const long t_lo = reverse ? T - (s_lo + nt) : s_lo;
    unsigned* const counters = area >= 0 ? w.sync + kSyncWords + 60 + (long)area * kBwdCounters : w.sync + kBwdCounters;
    if (area < 0 && hipMemsetAsync(counters, 0, kBwdCounters * sizeof(unsigned), s) != hipSuccess) return -2;
    LstmChainBwdArgs a{};
    a.B = B; a.H = H; a.T = nt; a.reverse = reverse; a.members = H / 16;
    a.W_hhT = w.whhT; a.dout = dout ? dout + t_lo * BH : nullptr; a.dhT = dhT; a.dcT = dcT;
    a.sv = w.sv + t_lo * BH; a.sv_stride = TBH;
    a.dg = dgi + t_lo * B * 4 * H; a.dh0 = dh0; a.dc0 = dc0;
    a.db_ih = db_ih; a.db_hh = db_hh;
    a.gx = w.gx; a.counters = counters; a.status = chain_status_for(w.sync + kStatusWord); a.xrot = xrot & 7;
    char label[64];
    std::snprintf(label, sizeof label, "lstm_chain_bwd ms%d T%d B%d H%d", ms, nt, B, H);
    ProfScope prof(PROF_GRU_BWD, 2.0 * nt * B * 4.0 * H * H, s, label,
                   4.0 * (4.0 * H * H + (double)nt * B * H * (6 + 4 + 1)));
    if (H == 256) return ms == 1 ? launch_chain(lstm_chain_bwd_kernel<1, 16>, a, groups, s)
                       : ms == 2 ? launch_chain(lstm_chain_bwd_kernel<2, 16>, a, groups, s)
                                 : launch_chain(lstm_chain_bwd_kernel<4, 16>, a, groups, s);
    return ms == 1 ? launch_chain(lstm_chain_bwd_kernel<1, 32>, a, groups, s)
           : ms == 2 ? launch_chain(lstm_chain_bwd_kernel<2, 32>, a, groups, s)
                     : launch_chain(lstm_chain_bwd_kernel<4, 32>, a, groups, s);
}

// XCD the second chain of a two-layer pipeline starts its groups on (the first starts on XCD 0)
int lstm_pipe_xrot() {
    return 4;
}
int lstm_chunk_steps() {
    static const int v = [] { const char* e = std::getenv("INET_LSTM_CHUNK"); return e ? std::atoi(e) : 32; }();
    return v;
}

}  // namespace

bool lstm2_ok(int B, int T, int H) {
    const int CH = lstm_chunk_steps();
    return lstm_chain_ok(B, H) && CH >= 2 && T >= 2 * CH;
}

// Two stacked LSTM layers (zero initial states) as a pipeline over chunks of time steps: layer 1 needs layer 0's output of
// step t only, so while layer 0's chain runs chunk c + 1 on the caller's stream, a second stream projects chunk c
// (gi1 = out0 W_ih1^T + b_ih1) and runs layer 1's chain over it.  A B = 32 chain uses a few dozen CUs and every step of it
// is hand-off latency: the two layers side by side cost little more than one.  gi0 [T,B,4H] (includes b_ih0);
// out0 / out1 [T,B,H]; gi1 [T,B,4H] scratch that the backward pass does not need.
// Returns 1 when the shape does not qualify (caller runs the layers one after the other).
int lstm2_seq_fwd(int B, int T, int H, const float* gi0, const float* W_hh0, const float* b_hh0, const float* W_ih1,
                  const float* b_ih1, const float* W_hh1, const float* b_hh1, int reverse, float* out0, float* gi1,
                  float* out1, void* ws0, void* ws1, int save, hipStream_t s) {
    const int CH = lstm_chunk_steps();
    if (!lstm2_ok(B, T, H)) return 1;
    LstmWs w0, w1;
    lstm_carve(B, T, H, save, ws0, w0);
    lstm_carve(B, T, H, save, ws1, w1);
    const long BH = (long)B * H;
    if (pw_zero(w0.zeros, BH, s) != 0 || pw_zero(w1.zeros, BH, s) != 0) return -2;
    hipStream_t s2 = twin_fork(s);
    constexpr bool third = true;
    constexpr bool ring_on = true;
    for (int s_lo = 0; s_lo < T; s_lo += CH) {
        const int nt = T - s_lo < CH ? T - s_lo : CH;
        const long t_lo = reverse ? T - (s_lo + nt) : s_lo, tp = reverse ? t_lo + nt : t_lo - 1;
        INET_TRY(lstm_chunk_fwd(B, T, H, gi0, W_hh0, b_hh0, s_lo ? out0 + tp * BH : w0.zeros, s_lo ? w0.cseq + tp * BH : w0.zeros,
                                reverse, out0, w0, save, s_lo, nt, s, 0, ring_on));
        // the chunk's projection gi1 = out0 W_ih1^T + b_ih1 on a THIRD stream (a side stream forked behind layer 0's chunk), so
        // that layer 1's queue holds nothing but its chain launches: the product (30 us) runs under layer 1's previous chunk
        hipStream_t s3 = third ? side_fork(s) : s2;
        if (s3 == s2 || s3 == s) { s3 = s2; INET_TRY(stream_wait(s2, s)); }
        INET_TRY(linear_fwd(out0 + t_lo * BH, H, W_ih1, H, b_ih1, gi1 + t_lo * B * 4 * H, 4L * H, nt * B, 4 * H, H, EPI_NONE, s3));
        if (s3 != s2) INET_TRY(stream_wait(s2, s3));
        INET_TRY(lstm_chunk_fwd(B, T, H, gi1, W_hh1, b_hh1, s_lo ? out1 + tp * BH : w1.zeros, s_lo ? w1.cseq + tp * BH : w1.zeros,
                                reverse, out1, w1, save, s_lo, nt, s2, lstm_pipe_xrot(), ring_on));
    }
    return s2 != s ? twin_join(s) : 0;
}

// Backward of lstm2_seq_fwd, pipelined the other way round: layer 1's BPTT chain runs chunk c on the caller's stream, the
// second stream turns its gate gradients into layer 0's output gradient (dout0 = dgi1 W_ih1) and runs layer 0's chain over
// the chunk.  dout1 [T,B,H]; dgi0 / dgi1 [T,B,4H] out; dout0 [T,B,H] scratch; weight / bias gradients accumulated.
int lstm2_seq_bwd(int B, int T, int H, const float* W_hh0, const float* W_ih1, const float* W_hh1, const float* out0,
                  const float* out1, const float* dout1, int reverse, float* dgi0, float* dgi1, float* dout0, float* dW_hh0,
                  float* db_ih0, float* db_hh0, float* dW_ih1, float* dW_hh1, float* db_ih1, float* db_hh1, void* ws0,
                  void* ws1, hipStream_t s) {
    const int CH = lstm_chunk_steps();
    if (!lstm2_ok(B, T, H)) return 1;
    LstmWs w0, w1;
    lstm_carve(B, T, H, 1, ws0, w0);
    lstm_carve(B, T, H, 1, ws1, w1);
    const long BH = (long)B * H, B4H = 4 * BH;
    INET_TRY(pw_transpose(W_hh0, H, w0.whhT, 4L * H, 4 * H, H, s));
    INET_TRY(pw_transpose(W_hh1, H, w1.whhT, 4L * H, 4 * H, H, s));
    hipStream_t s2 = twin_fork(s);
    constexpr bool third = true;
    const int nchunks = (T + CH - 1) / CH;
    const bool areas = nchunks <= kMaxChunks;            // one pre-zeroed counter area per chunk and layer
    if (areas && (hipMemsetAsync(w0.sync + kSyncWords, 0, (kSyncWordsAll - kSyncWords) * sizeof(unsigned), s) != hipSuccess ||
                  hipMemsetAsync(w1.sync + kSyncWords, 0, (kSyncWordsAll - kSyncWords) * sizeof(unsigned), s) != hipSuccess)) return -2;
    if (s2 != s) INET_TRY(stream_wait(s2, s));           // (the second stream's first launch must see those zeros)
    int c = 0;
    for (int s_end = T; s_end > 0; s_end -= CH, ++c) {
        const int nt = s_end < CH ? s_end : CH, s_lo = s_end - nt;
        const long t_lo = reverse ? T - (s_lo + nt) : s_lo;
        float* in1 = w1.carry + (long)((c + 1) & 1) * 2 * BH;   // written by the previous chunk of this layer
        float* ou1 = w1.carry + (long)(c & 1) * 2 * BH;
        float* in0 = w0.carry + (long)((c + 1) & 1) * 2 * BH;
        float* ou0 = w0.carry + (long)(c & 1) * 2 * BH;
        INET_TRY(lstm_chunk_bwd(B, T, H, dout1, c ? in1 : nullptr, c ? in1 + BH : nullptr, reverse, dgi1, db_ih1, db_hh1,
                                s_lo ? ou1 : nullptr, s_lo ? ou1 + BH : nullptr, w1, s_lo, nt, s, areas ? c : -1));
        hipStream_t s3 = third ? side_fork(s) : s2;          // (as in the forward pipeline: the chunk's product on a third stream)
        if (s3 == s2 || s3 == s) { s3 = s2; INET_TRY(stream_wait(s2, s)); }
        INET_TRY(linear_dgrad(dgi1 + t_lo * B4H, 4L * H, W_ih1, H, dout0 + t_lo * BH, H, nt * B, 4 * H, H, EPI_NONE, nullptr, 0,
                              ACC_STORE, s3));
        if (s3 != s2) INET_TRY(stream_wait(s2, s3));
        INET_TRY(lstm_chunk_bwd(B, T, H, dout0, c ? in0 : nullptr, c ? in0 + BH : nullptr, reverse, dgi0, db_ih0, db_hh0,
                                s_lo ? ou0 : nullptr, s_lo ? ou0 + BH : nullptr, w0, s_lo, nt, s2, areas ? c : -1, lstm_pipe_xrot()));
    }
    // Weight gradients, once per layer (handing each chunk's products to an in-order stream as soon as its gate gradients exist
    // was slower -- 9.2 -> 10.5 ms per AnticipationRNN step: beside the chains they slow every hand-off):
    // dW_hh += sum_t dg(t)^T h_prev(t) with h_prev(t) = out(t -/+ 1) (zero initial state); dW_ih1 += dgi1^T out0.
    // Layer 1's two products start when ITS last chunk is done -- the caller's stream, before it joins layer 0's -- and run under
    // layer 0's last chunk; only dW_hh0 is left behind the pipeline (a switch of round 4 put all three behind it).
    constexpr bool early = true;
    auto wgrad1 = [&](hipStream_t ss) -> int {
        INET_TRY(linear_wgrad(reverse ? dgi1 : dgi1 + B4H, 4L * H, reverse ? out1 + BH : out1, H, dW_hh1, H, (T - 1) * B, 4 * H, H, ss));
        INET_TRY(linear_wgrad(dgi1, 4L * H, out0, H, dW_ih1, H, T * B, 4 * H, H, ss));
        return 0;
    };
    if (dW_hh0 && early) INET_TRY(wgrad1(side_fork(s)));
    if (s2 != s) INET_TRY(twin_join(s));
    if (dW_hh0) {
        hipStream_t ss = side_fork(s);
        if (!early) INET_TRY(wgrad1(ss));
        INET_TRY(linear_wgrad(reverse ? dgi0 : dgi0 + B4H, 4L * H, reverse ? out0 + BH : out0, H, dW_hh0, H, (T - 1) * B, 4 * H, H, ss));
    }
    return side_join(s);
}

// dout [T,B,H] (nullable), dhT/dcT [B,H] (nullable) -> dgi [T,B,4H]; dW_hh / db_ih / db_hh accumulated (nullable as a
// group); dh0/dc0 [B,H] (nullable).  `out` is the forward output (h sequence), needed for the W_hh gradient.
int lstm_seq_bwd(int B, int T, int H, const float* W_hh, const float* h0, const float* out, const float* dout,
                 const float* dhT, const float* dcT, int reverse, float* dgi, float* dW_hh, float* db_ih, float* db_hh,
                 float* dh0, float* dc0, void* ws, hipStream_t s) {
    LstmWs w;
    lstm_carve(B, T, H, 1, ws, w);
    const long BH = (long)B * H, TBH = (long)T * BH, B4H = 4 * BH;
    INET_TRY(pw_transpose(W_hh, H, w.whhT, 4L * H, 4 * H, H, s));
    const bool use_chain = lstm_chain_ok(B, H);
    if (use_chain) {
        const int ms = chain_ms(B, H), groups = (B + 16 * ms - 1) / (16 * ms);
        if (hipMemsetAsync(w.sync + kBwdCounters, 0, kBwdCounters * sizeof(unsigned), s) != hipSuccess) return -2;
        LstmChainBwdArgs a{};
        a.B = B; a.H = H; a.T = T; a.reverse = reverse; a.members = H / 16;
        a.W_hhT = w.whhT; a.dout = dout; a.dhT = dhT; a.dcT = dcT;
        a.sv = w.sv; a.sv_stride = TBH;
        a.dg = dgi; a.dh0 = dh0; a.dc0 = dc0;
        a.db_ih = db_ih; a.db_hh = db_hh;
        a.gx = w.gx; a.counters = w.sync + kBwdCounters; a.status = chain_status_for(w.sync + kStatusWord);
        char label[64];
        std::snprintf(label, sizeof label, "lstm_chain_bwd ms%d T%d B%d H%d", ms, T, B, H);
        ProfScope prof(PROF_GRU_BWD, 2.0 * T * B * 4.0 * H * H, s, label,
                       4.0 * (4.0 * H * H + (double)T * B * H * (6 + 4 + 1)));
        int rc;
        if (H == 256) rc = ms == 1 ? launch_chain(lstm_chain_bwd_kernel<1, 16>, a, groups, s)
                           : ms == 2 ? launch_chain(lstm_chain_bwd_kernel<2, 16>, a, groups, s)
                                     : launch_chain(lstm_chain_bwd_kernel<4, 16>, a, groups, s);
        else rc = ms == 1 ? launch_chain(lstm_chain_bwd_kernel<1, 32>, a, groups, s)
                  : ms == 2 ? launch_chain(lstm_chain_bwd_kernel<2, 32>, a, groups, s)
                            : launch_chain(lstm_chain_bwd_kernel<4, 32>, a, groups, s);
        INET_TRY(rc);
    } else
    for (int step = T - 1; step >= 0; --step) {
        const int t = reverse ? T - 1 - step : step;
        const int tn = reverse ? t - 1 : t + 1;
        LstmBwdArgs a{};
        a.B = B; a.H = H;
        if (step != T - 1) {
            a.dg_next = dgi + (long)tn * B4H; a.W_hhT = w.whhT;
            a.dc_next = w.dc + (long)((step + 1) & 1) * BH;
        } else {
            a.dout2 = dhT; a.dc_ext = dcT;
        }
        if (dout) a.dout = dout + (long)t * BH;
        a.sv = w.sv + (long)t * BH; a.sv_stride = TBH;
        a.dg = dgi + (long)t * B4H;
        a.dc_prev = w.dc + (long)(step & 1) * BH;
        a.db_ih = db_ih; a.db_hh = db_hh;
        INET_TRY(launch_bwd(a, s));
    }
    const int t0 = reverse ? T - 1 : 0;
    if (dh0 && !use_chain) {
        LstmBwdArgs a{};
        a.B = B; a.H = H;
        a.dg_next = dgi + (long)t0 * B4H; a.W_hhT = w.whhT;
        a.dh_out = dh0;
        INET_TRY(launch_bwd(a, s));
    }
    if (dc0 && !use_chain && pw_copy_bytes(dc0, w.dc, BH * sizeof(float), s) != 0) return -2;
    if (dW_hh) {
        // dW_hh += sum_t dg(t)^T h_prev(t):  h_prev(t) = out(t -/+ 1), and h0 for the first processed step
        hipStream_t ss = side_fork(s);
        if (T > 1) {
            const float* dg_a = reverse ? dgi : dgi + B4H;              // steps whose h_prev is an output
            const float* hp_a = reverse ? out + BH : out;
            INET_TRY(linear_wgrad(dg_a, 4L * H, hp_a, H, dW_hh, H, (T - 1) * B, 4 * H, H, ss));
        }
        if (h0) INET_TRY(linear_wgrad(dgi + (long)t0 * B4H, 4L * H, h0, H, dW_hh, H, B, 4 * H, H, ss));
    }
    return side_join(s);
}

// ---- AnticipationRNN's free-running pass, the part that is sequential (anticipation_rnn_gauss_reg_model.py:190-259) ---------------
// The generation LSTMs feed back the argmax of BATCH ELEMENT 0 to the whole batch (:253-256) and nothing else of a tick's output:
// the token sequence depends on batch element 0 alone.  This runs those L ticks for that one row -- per tick: input = [embedding of
// the previous token | constraint output of the tick], two LSTM cells, linear_1 + ReLU, the note head, argmax -- as 4 small launches
// per tick queued from here (no host round trip: the token stays on the device), and hands back the L tokens.  With them the
// caller runs the whole batch through the batched (teacher-forced-shaped) kernels: 195 -> 14 ms per training step.
namespace {
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// dot products of ONE row against weight rows, lanes striding over k: every load of a wave is issued before the first multiply (NI =
// ceil(K / 64) is a template bound: a runtime k loop waits for each 64-wide slice in turn -- 7 us per launch instead of 2)
template <int NI>
__device__ __forceinline__ void load_x(float (&xv)[NI], const float* pa, int Ka, const float* pb, int K, int lane) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int k = lane + 64 * i;
        xv[i] = k < Ka ? pa[k] : (k < K ? pb[k - Ka] : 0.f);
    }
}
template <int NI>
__device__ __forceinline__ float dot_row(const float* __restrict__ Wrow, const float (&xv)[NI], int K, int lane) {
    float wv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int k = lane + 64 * i; wv[i] = k < K ? Wrow[k] : 0.f; }
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) a = fmaf(wv[i], xv[i], a);
    return a;
}

// One LSTM cell for ONE row, both products in the launch: gates = W_ih [xa | xb] + b_ih + W_hh h_prev + b_hh.  One wave per hidden
// unit (its four gate rows), four units per workgroup.  xa = the embedding row of *tok (tok null: token 0) when `emb` is given.
template <int NI, int NH>
__global__ __launch_bounds__(256) void lstm_cell_b1_kernel(const float* __restrict__ emb, const long long* __restrict__ tok,
                                                           const float* __restrict__ xa, int Ka, const float* __restrict__ xb, int Kb,
                                                           const float* __restrict__ W_ih, const float* __restrict__ b_ih,
                                                           const float* __restrict__ h_prev, const float* __restrict__ c_prev,
                                                           const float* __restrict__ W_hh, const float* __restrict__ b_hh,
                                                           float* __restrict__ h_new, float* __restrict__ c_new, int H) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    const int K = Ka + Kb;
    const float* pa = emb ? emb + (tok ? *tok : 0) * Ka : xa;
    float xv[NI], hv[NH];
    load_x<NI>(xv, pa, Ka, xb, K, lane);
    load_x<NH>(hv, h_prev, H, nullptr, H, lane);
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        pre[g] = dot_row<NI>(W_ih + (long)(g * H + j) * K, xv, K, lane) + dot_row<NH>(W_hh + (long)(g * H + j) * H, hv, H, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[g] = wave_sum(pre[g]);
    if (lane == 0) {
        const float i = sigmoid_f(pre[0] + b_ih[j] + b_hh[j]);
        const float f = sigmoid_f(pre[1] + b_ih[H + j] + b_hh[H + j]);
        const float g = tanh_f(pre[2] + b_ih[2 * H + j] + b_hh[2 * H + j]);
        const float o = sigmoid_f(pre[3] + b_ih[3 * H + j] + b_hh[3 * H + j]);
        const float c = f * c_prev[j] + i * g;
        c_new[j] = c;
        h_new[j] = o * tanh_f(c);
    }
}

// y[j] = ReLU(W[j,:] . x + b[j]) for ONE row: a wave per output
template <int NI>
__global__ __launch_bounds__(256) void relu_linear_b1_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                             const float* __restrict__ b, float* __restrict__ y, int N, int K) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= N) return;
    float xv[NI];
    load_x<NI>(xv, x, K, nullptr, K, lane);
    const float v = wave_sum(dot_row<NI>(W + (long)j * K, xv, K, lane));
    if (lane == 0) y[j] = fmaxf(v + b[j], 0.f);
}

// does (b2, i2) come before (best, bi) in numpy's argmax order?  NaN > everything, ties to the lower index
__device__ __forceinline__ bool argmax_better(float b2, int i2, float best, int bi) {
    const bool n2 = b2 != b2, n1 = best != best;
    if (n2 || n1) return n2 && (!n1 || i2 < bi);
    return b2 > best || (b2 == best && i2 < bi);
}

// token = argmax_v (W[v,:] . x + b[v]), lowest index on ties, V <= 256: ONE workgroup of 16 waves; a wave's rows (V = 48: three) are all
// requested before the first sum, the logits meet in LDS and the first wave takes the argmax with shuffles
template <int NI>
__global__ __launch_bounds__(1024) void head_argmax_b1_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                              const float* __restrict__ b, long long* __restrict__ tok, int V, int K) {
    __shared__ float lg[256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float xv[NI];
    load_x<NI>(xv, x, K, nullptr, K, lane);
    float part[16];                                            // rows w, w + 16, ...: V <= 256
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int v = w + 16 * r;
        part[r] = v < V ? dot_row<NI>(W + (long)v * K, xv, K, lane) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int v = w + 16 * r;
        if (v < V) {                                           // (wave-uniform)
            const float a = wave_sum(part[r]);
            if (lane == 0) lg[v] = a + b[v];
        }
    }
    __syncthreads();
    if (w == 0) {
        // np.argmax order (anticipation_rnn_gauss_reg_model.py:253): a NaN is the maximum, the lowest index wins among equals -- an
        // all-NaN or all -inf row yields a token INSIDE the vocabulary (the next tick gathers the embedding row by it)
        float best = lane < V ? lg[lane] : -INFINITY;
        int bi = lane < V ? lane : 0x7fffffff;
        for (int v = lane + 64; v < V; v += 64)
            if (argmax_better(lg[v], v, best, bi)) { best = lg[v]; bi = v; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float b2 = __shfl_xor(best, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (argmax_better(b2, i2, best, bi)) { best = b2; bi = i2; }
        }
        if (lane == 0) *tok = bi < V ? bi : 0;
    }
}
}  // namespace

size_t arnn_generate_ws_floats(int L, int E, int Hc, int H, int U, int V) {
    const size_t ticks = (size_t)(E + Hc) + 4 * (size_t)H + 8 * (size_t)H + U + V + 64;
    const size_t pass = arnn_token_pass_ok(H, U, V) ? arnn_token_pass_ws_floats(L, V) : 0;
    return ticks > pass ? ticks : pass;
}

int arnn_generate(int L, int E, int Hc, int H, int U, int V, const float* emb, const float* oc0, long oc_stride, const float* W_ih0,
                  const float* b_ih0, const float* W_hh0, const float* b_hh0, const float* W_ih1, const float* b_ih1,
                  const float* W_hh1, const float* b_hh1, const float* W1, const float* b1, const float* W2, const float* b2,
                  const float* hc_init, const long long* first_tok, long long* tokens, float* ws, hipStream_t s) {
    // the reference's configuration: ONE persistent launch for all L ticks (arnn_gen.hip): 14.3 -> ~3.5 us per tick
    if (arnn_token_pass_ok(H, U, V))
        return arnn_token_pass(L, E, Hc, V, emb, oc0, oc_stride, W_ih0, b_ih0, W_hh0, b_hh0, W_ih1, b_ih1, W_hh1, b_hh1, W1, b1, W2, b2,
                               hc_init, first_tok, tokens, ws, s);
    float* hc = ws;                                            // [layer][h|c][ping-pong][H]
    float* u = hc + 8 * H;
    if (pw_zero(hc, 8L * H, s) != 0) return -2;
    auto H_ = [&](int l, int p) { return hc + ((l * 2 + 0) * 2 + p) * H; };
    auto C_ = [&](int l, int p) { return hc + ((l * 2 + 1) * 2 + p) * H; };
    if (hc_init)                                               // [layer][h | c][H]: the state the ticks go on from (inpainting: after the prefix)
        for (int l = 0; l < 2; ++l)
            if (pw_copy_bytes(H_(l, 0), hc_init + (2 * l) * H, H * sizeof(float), s) != 0 ||
                pw_copy_bytes(C_(l, 0), hc_init + (2 * l + 1) * H, H * sizeof(float), s) != 0) return -2;
    // per tick four launches (round 4's first form had eight: input build, two GEMVs + two cell kernels, two head GEMVs, argmax)
    if (V > 256 || E + Hc > 320 || H > 256 || U > 256) return -1;      // (the template bounds of the one-row kernels)
    for (int t = 0, p = 0; t < L; ++t, p ^= 1) {
        hipLaunchKernelGGL((lstm_cell_b1_kernel<5, 4>), dim3(H / 4), dim3(256), 0, s, emb, t ? tokens + t - 1 : first_tok,
                           (const float*)nullptr, E, oc0 + (long)t * oc_stride, Hc, W_ih0, b_ih0, (const float*)H_(0, p),
                           (const float*)C_(0, p), W_hh0, b_hh0, H_(0, p ^ 1), C_(0, p ^ 1), H);
        hipLaunchKernelGGL((lstm_cell_b1_kernel<4, 4>), dim3(H / 4), dim3(256), 0, s, (const float*)nullptr, (const long long*)nullptr,
                           (const float*)H_(0, p ^ 1), H, (const float*)nullptr, 0, W_ih1, b_ih1, (const float*)H_(1, p),
                           (const float*)C_(1, p), W_hh1, b_hh1, H_(1, p ^ 1), C_(1, p ^ 1), H);
        hipLaunchKernelGGL((relu_linear_b1_kernel<4>), dim3((U + 3) / 4), dim3(256), 0, s, (const float*)H_(1, p ^ 1), W1, b1, u, U, H);
        hipLaunchKernelGGL((head_argmax_b1_kernel<4>), dim3(1), dim3(1024), 0, s, (const float*)u, W2, b2, tokens + t, V, U);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
