// One GRU time step per launch, fused end to end: the recurrent contraction
// h_prev[B,H] x W_hh[3H,H]^T (and, for the tick decoder's second layer, the
// input contraction x[B,K2] x W_ih[3H,K2]^T in the same pass), the gather of
// embedding-derived gate pre-activations by token index, the sigmoid/tanh gate
// math, the state update, the dropout-masked copy for the next layer and the
// activations the backward pass needs -- no intermediate ever touches HBM.
//
// Why one launch per step and not one persistent kernel: every step needs the
// whole previous hidden state (an all-to-all over the 256 CUs).  On MI355X a
// dependent kernel boundary costs ~1.5 us, an in-kernel grid barrier 4-7 us
// (MI355X_MICROARCH.md price list), so the boundary IS the cheapest barrier;
// the step sequence is captured in a hipGraph by the host to remove launch cost.
//
// Geometry (gfx950): workgroup = 4 wavefronts, output tile = 32 batch rows x 16
// hidden units x {r,z,n} gates, i.e. all three gate columns of the same hidden
// units so the gate math is register-local.  The 4 waves split K (each takes 16
// of every 64-deep chunk; operands staged through LDS in full 256-byte row
// segments, pitch 72 words = conflict-free ds_read_b128), accumulate with
// v_mfma_f32_16x16x4_f32 (exact f32), and are combined through LDS before the
// epilogue.  At B=256,H=512 this is 8 x 32 = 256 workgroups = one per CU, and
// bidirectional layers / the four beats run as blockIdx.z "problems" in the
// same launch.
#include "common.h"
#include "prof.h"

namespace {

constexpr int KC = 64;        // k-chunk depth staged in LDS
constexpr int PITCH = 72;     // LDS row pitch in words (== 8 mod 64)
constexpr int TM_ROWS = 32;   // batch rows per workgroup
constexpr int TH = 16;        // hidden units per workgroup

// One K-segment:  acc[ms][g] += A[32 rows, K] * Bg[16 rows, K]^T   for g < NB.
// A rows beyond `rowsA` and k beyond K are zero filled.  B rows are always valid.
// lds: 2 stages of (32 + NB*16) rows x PITCH words.
template <int NB>
__device__ __forceinline__ void ksplit_segment(f32x4 (&acc)[2][4], const int (&slot)[NB],
                                               const float* __restrict__ A, long lda, int row0, int rowsA,
                                               const float* __restrict__ Bm, long ldb, const int (&brow)[NB],
                                               int K, float* lds, int t) {
    constexpr int STAGE = (TM_ROWS + NB * TH) * PITCH;
    const int lane = t & 63, w = t >> 6;
    const int c4 = t & 15, rr = t >> 4;            // 16 float4 per 64-deep row; 16 rows per pass
    const int nchunks = (K + KC - 1) / KC;
    f32x4 ra[2], rb[NB];

    auto gload = [&](int c) {
        const int k = c * KC + c4 * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = row0 + rr + 16 * i;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (row < rowsA) {
                const float* p = A + (long)row * lda + k;
                if (k + 3 < K) x = ld4u(p);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < K) x[e] = p[e];
                }
            }
            ra[i] = x;
        }
#pragma unroll
        for (int gi = 0; gi < NB; ++gi) {
            const float* p = Bm + (long)(brow[gi] + rr) * ldb + k;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (k + 3 < K) x = ld4u(p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < K) x[e] = p[e];
            }
            rb[gi] = x;
        }
    };
    auto lstore = [&](float* st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(st + (rr + 16 * i) * PITCH + c4 * 4) = ra[i];
#pragma unroll
        for (int gi = 0; gi < NB; ++gi)
            *reinterpret_cast<f32x4*>(st + (TM_ROWS + gi * TH + rr) * PITCH + c4 * 4) = rb[gi];
    };

    gload(0);
    __syncthreads();            // previous users of the LDS stages are done
    lstore(lds);
    __syncthreads();
    const int i16 = lane & 15, q = lane >> 4;
    for (int c = 0; c < nchunks; ++c) {
        const float* st = lds + (c & 1) * STAGE;
        const bool more = c + 1 < nchunks;
        if (more) gload(c + 1);
        // this wave's 16-deep slice of the chunk: lane (i16,q) reads k = 16w + 4q + {0..3}
        const int koff = 16 * w + 4 * q;
        f32x4 fa[2], fb[NB];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) fa[ms] = *reinterpret_cast<const f32x4*>(st + (16 * ms + i16) * PITCH + koff);
#pragma unroll
        for (int gi = 0; gi < NB; ++gi)
            fb[gi] = *reinterpret_cast<const f32x4*>(st + (TM_ROWS + gi * TH + i16) * PITCH + koff);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int gi = 0; gi < NB; ++gi)
                    acc[ms][slot[gi]] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ms][s], fb[gi][s], acc[ms][slot[gi]], 0, 0, 0);
        if (more) lstore(lds + ((c + 1) & 1) * STAGE);
        __syncthreads();
    }
}

// Cross-wave reduction: every wave dumps its partial accumulators, then thread t
// owns output positions t and t+256 of the 32x16 tile (pos = row*16 + col) for
// all NACC accumulators.  C/D map of the 16x16 MFMA: col = lane&15,
// row = 4*(lane>>4) + reg.
template <int NACC>
__device__ __forceinline__ void reduce_waves(const f32x4 (&acc)[2][4], float* red, int t, float (&out)[2][NACC]) {
    const int lane = t & 63, w = t >> 6;
    __syncthreads();
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ms + 4 * (lane >> 4) + r;
                red[(w * NACC + a) * 512 + row * 16 + (lane & 15)] = acc[ms][a][r];
            }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s += red[(ww * NACC + a) * 512 + t + 256 * p];
            out[p][a] = s;
        }
}

constexpr int FWD_LDS_WORDS = 2 * (TM_ROWS + 3 * TH) * PITCH;   // 11520 words; reduction needs 4*4*512 = 8192
constexpr int BWD_LDS_WORDS = 2 * (TM_ROWS + 1 * TH) * PITCH;   // 6912 words;  reduction needs 4*1*512 = 2048

template <bool HAS_X>
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(GruFwdBatch bt) {
    __shared__ __attribute__((aligned(16))) float lds[FWD_LDS_WORDS];
    const GruFwdProb& P = bt.p[blockIdx.z];
    const int H = bt.H;
    const int t = threadIdx.x;
    const int j0 = blockIdx.x * TH;
    const int row0 = blockIdx.y * TM_ROWS;
    if (row0 >= P.B) return;

    f32x4 acc[2][4];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int brow[3] = {j0, H + j0, 2 * H + j0};
    if (HAS_X) {
        const int slotx[3] = {0, 1, 2};          // r, z, gi_n
        ksplit_segment<3>(acc, slotx, P.x, P.ldx, row0, P.B, P.W_ih, P.ld_wih, brow, P.K2, lds, t);
    }
    const int sloth[3] = {0, 1, 3};              // r, z, gh_n
    ksplit_segment<3>(acc, sloth, P.h_prev, P.ld_hprev, row0, P.B, P.W_hh, (long)H, brow, H, lds, t);

    float v[2][4];
    reduce_waves<4>(acc, lds, t, v);

#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int pos = t + 256 * p;
        const int b = row0 + (pos >> 4);
        const int j = j0 + (pos & 15);
        if (b >= P.B) continue;
        float gr = v[p][0], gz = v[p][1], gn = v[p][2], ghn = v[p][3];
        if (HAS_X && P.b_ih) { gr += P.b_ih[j]; gz += P.b_ih[H + j]; gn += P.b_ih[2 * H + j]; }
        if (P.gi_dense) {
            const float* gp = P.gi_dense + (long)b * P.ld_gi;
            gr += gp[j]; gz += gp[H + j]; gn += gp[2 * H + j];
        }
        if (P.gi_table) {
            const float* gp = P.gi_table + (long)P.idx[(long)b * P.idx_stride] * P.ld_table;
            gr += gp[j]; gz += gp[H + j]; gn += gp[2 * H + j];
        }
        if (P.gi_vec) { gr += P.gi_vec[j]; gz += P.gi_vec[H + j]; gn += P.gi_vec[2 * H + j]; }
        gr += P.b_hh[j]; gz += P.b_hh[H + j]; ghn += P.b_hh[2 * H + j];
        const float r = sigmoid_f(gr);
        const float z = sigmoid_f(gz);
        const float n = tanh_f(gn + r * ghn);
        const float hp = P.h_prev[(long)b * P.ld_hprev + j];
        const float hn = (1.f - z) * n + z * hp;
        P.h_new[(long)b * P.ld_hnew + j] = hn;
        if (P.h_copy) P.h_copy[(long)b * P.ld_hc + j] = hn;
        if (P.h_masked) P.h_masked[(long)b * P.ld_hm + j] = P.mask ? hn * P.mask[(long)b * P.ld_mask + j] : hn;
        if (P.sv_r) {
            const long o = (long)b * H + j;
            P.sv_r[o] = r; P.sv_z[o] = z; P.sv_n[o] = n; P.sv_ghn[o] = ghn; P.sv_hprev[o] = hp;
        }
    }
}

// Backward of one step.
//   dh   = dgh_next * W_hh + dhz_next + dout + dout2          (gradient wrt this step's output h)
//   dn   = dh (1-z); dz = dh (hprev - n); dhz = dh z
//   dn_pre = dn (1-n^2); dz_pre = dz z(1-z); dr_pre = dn_pre ghn r(1-r)
//   dgi = [dr_pre, dz_pre, dn_pre]      dgh = [dr_pre, dz_pre, dn_pre r]
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(GruBwdBatch bt) {
    __shared__ __attribute__((aligned(16))) float lds[BWD_LDS_WORDS];
    const GruBwdProb& P = bt.p[blockIdx.z];
    const int H = bt.H;
    const int t = threadIdx.x;
    const int j0 = blockIdx.x * TH;
    const int row0 = blockIdx.y * TM_ROWS;
    if (row0 >= P.B) return;

    float v[2][1] = {{0.f}, {0.f}};
    if (P.dgh_next) {
        f32x4 acc[2][4];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[ms][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int brow[1] = {j0};
        const int slot[1] = {0};
        ksplit_segment<1>(acc, slot, P.dgh_next, P.ld_dgh, row0, P.B, P.W_hhT, (long)3 * H, brow, 3 * H, lds, t);
        reduce_waves<1>(acc, lds, t, v);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int pos = t + 256 * p;
        const int b = row0 + (pos >> 4);
        const int j = j0 + (pos & 15);
        if (b >= P.B) continue;
        const long o = (long)b * H + j;
        float dh = v[p][0];
        if (P.dhz_next) dh += P.dhz_next[o];
        if (P.dout) dh += P.dout[(long)b * P.ld_dout + j];
        if (P.dout2) dh += P.dout2[(long)b * P.ld_dout2 + j];
        if (!P.sv_r) {
            float* d = P.dh_out + (long)b * P.ld_dhout + j;
            *d = P.dh_out_accumulate ? *d + dh : dh;
            continue;
        }
        const float r = P.sv_r[o], z = P.sv_z[o], n = P.sv_n[o], ghn = P.sv_ghn[o], hp = P.sv_hprev[o];
        const float dn_pre = dh * (1.f - z) * (1.f - n * n);
        const float dz_pre = dh * (hp - n) * z * (1.f - z);
        const float dr_pre = dn_pre * ghn * r * (1.f - r);
        P.dhz[o] = dh * z;
        float* gi = P.dgi + (long)b * P.ld_dgi;
        gi[j] = dr_pre; gi[H + j] = dz_pre; gi[2 * H + j] = dn_pre;
        float* gh = P.dgh + (long)b * P.ld_dghout;
        gh[j] = dr_pre; gh[H + j] = dz_pre; gh[2 * H + j] = dn_pre * r;
    }
}

}  // namespace

int launch_gru_fwd(const GruFwdBatch& b, hipStream_t s) {
    if (b.H % TH != 0 || b.nprob < 1 || b.nprob > 4) return -1;
    int maxB = 0;
    bool hasx = b.p[0].x != nullptr;
    for (int i = 0; i < b.nprob; ++i) {
        if (b.p[i].B > maxB) maxB = b.p[i].B;
        if ((b.p[i].x != nullptr) != hasx) return -1;
    }
    if (maxB <= 0) return 0;
    dim3 grid(b.H / TH, (maxB + TM_ROWS - 1) / TM_ROWS, b.nprob);
    double fl = 0;
    for (int i = 0; i < b.nprob; ++i) fl += 2.0 * b.p[i].B * 3.0 * b.H * (b.H + (hasx ? b.p[i].K2 : 0));
    ProfScope prof(PROF_GRU_FWD, fl, s);
    if (hasx) hipLaunchKernelGGL(gru_step_fwd_kernel<true>, grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL(gru_step_fwd_kernel<false>, grid, dim3(256), 0, s, b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_gru_bwd(const GruBwdBatch& b, hipStream_t s) {
    if (b.H % TH != 0 || b.nprob < 1 || b.nprob > 4) return -1;
    int maxB = 0;
    for (int i = 0; i < b.nprob; ++i) if (b.p[i].B > maxB) maxB = b.p[i].B;
    if (maxB <= 0) return 0;
    dim3 grid(b.H / TH, (maxB + TM_ROWS - 1) / TM_ROWS, b.nprob);
    double fl = 0;
    for (int i = 0; i < b.nprob; ++i) if (b.p[i].dgh_next) fl += 2.0 * b.p[i].B * 3.0 * b.H * b.H;
    ProfScope prof(PROF_GRU_BWD, fl, s);
    hipLaunchKernelGGL(gru_step_bwd_kernel, grid, dim3(256), 0, s, b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
