// Fused LSTM step kernels + sequence driver (AnticipationRNN, config 5: torch.nn.LSTM(num_layers=1) cells stacked by
// lstm_with_activations, AnticipationRNN/anticipation_rnn_gauss_reg_model.py:14-39,110-133).
// Same geometry as the GRU step (ksplit.h): tile = 16*MS batch rows x 16 hidden units x {i,f,g,o} gates, the
// recurrent contraction h_prev[B,H] x W_hh[4H,H]^T streamed from L2 into v_mfma_f32_16x16x4_f32 fragments, gate
// math / cell update / backward saves in the epilogue.  Input-side pre-activations gi = x W_ih^T + b_ih are formed for
// all time steps at once by the batched GEMM (gemm.hip).
#include "ksplit.h"
#include "prof.h"
#include "seq.h"
#include "lstm.h"

using namespace ksplit;

namespace {

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

struct LstmWs { float *zeros, *cseq, *sv, *whhT, *dc; };
size_t lstm_carve(int B, int T, int H, int save, void* base, LstmWs& w) {
    Carver cv(base);
    const size_t BH = (size_t)B * H;
    w.zeros = cv.take<float>(BH);
    w.cseq = cv.take<float>((size_t)T * BH);
    w.sv = save ? cv.take<float>(6 * (size_t)T * BH) : nullptr;
    w.whhT = save ? cv.take<float>((size_t)4 * H * H) : nullptr;
    w.dc = save ? cv.take<float>(2 * BH) : nullptr;
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

// dout [T,B,H] (nullable), dhT/dcT [B,H] (nullable) -> dgi [T,B,4H]; dW_hh / db_ih / db_hh accumulated (nullable as a
// group); dh0/dc0 [B,H] (nullable).  `out` is the forward output (h sequence), needed for the W_hh gradient.
int lstm_seq_bwd(int B, int T, int H, const float* W_hh, const float* h0, const float* out, const float* dout,
                 const float* dhT, const float* dcT, int reverse, float* dgi, float* dW_hh, float* db_ih, float* db_hh,
                 float* dh0, float* dc0, void* ws, hipStream_t s) {
    LstmWs w;
    lstm_carve(B, T, H, 1, ws, w);
    const long BH = (long)B * H, TBH = (long)T * BH, B4H = 4 * BH;
    INET_TRY(pw_transpose(W_hh, H, w.whhT, 4L * H, 4 * H, H, s));
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
    if (dh0) {
        LstmBwdArgs a{};
        a.B = B; a.H = H;
        a.dg_next = dgi + (long)t0 * B4H; a.W_hhT = w.whhT;
        a.dh_out = dh0;
        INET_TRY(launch_bwd(a, s));
    }
    if (dc0 && pw_copy_bytes(dc0, w.dc, BH * sizeof(float), s) != 0) return -2;
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
