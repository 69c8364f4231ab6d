// extern "C" surface declared in include/inpaintnet_hip.h.
#include <cstring>
#include "seq.h"
#include "layout.h"
#include "vae.h"
#include "lstm.h"
#include "chain.h"
#include "decode_chain.h"
#include "gemm_bf3.h"
#include "gru_step_bf3.h"
#include "side.h"

namespace {

bool cfg_ok(const inet_vae_config* c) {
    return c && c->num_notes > 0 && c->emb_dim > 0 && c->enc_hidden > 0 && c->enc_hidden % 16 == 0 &&
           c->dec_hidden > 0 && c->dec_hidden % 16 == 0 && c->z_dim > 0 && c->beats > 0 && c->beats <= 4 &&
           c->ticks_per_beat > 0;
}

int entry_info(const ArenaBuilder& ab, int index, char* name, int cap, int64_t* off, int64_t* dims, int* ndim) {
    if (index < 0 || index >= (int)ab.entries.size()) return -1;
    const ParamEntry& e = ab.entries[index];
    if (name && cap > 0) { std::strncpy(name, e.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (off) *off = e.offset;
    if (dims) for (int i = 0; i < 4; ++i) dims[i] = e.dims[i];
    if (ndim) *ndim = e.ndim;
    return 0;
}

// pointers to the 16 tensors of a 2-layer bidirectional nn.GRU laid out by ArenaBuilder::add_gru
void bigru_ptrs(const float* w, float* g, int K, int H, GruDirPtr* P) {
    ArenaBuilder ab;
    GruDirOff off[4];
    ab.add_gru("g", K, H, 2, true, off);
    for (int i = 0; i < 4; ++i) {
        P[i].w_ih = w + off[i].w_ih; P[i].w_hh = w + off[i].w_hh; P[i].b_ih = w + off[i].b_ih; P[i].b_hh = w + off[i].b_hh;
        P[i].dw_ih = g ? g + off[i].w_ih : nullptr; P[i].dw_hh = g ? g + off[i].w_hh : nullptr;
        P[i].db_ih = g ? g + off[i].b_ih : nullptr; P[i].db_hh = g ? g + off[i].b_hh : nullptr;
        P[i].K = off[i].K;
    }
}

struct BiWs {
    BiGru2Ws g;
    float *x_tm, *gi0, *gvec, *out_tm, *dout_tm, *dx_tm, *tmp3h;
};
size_t bi_carve(int B, int T, int K, int H, int save, void* base, BiWs& w) {
    Carver cv(base);
    const size_t TB = (size_t)T * B;
    w.x_tm = cv.take<float>(TB * K);
    w.gi0 = cv.take<float>(TB * 6 * H);
    w.gvec = cv.take<float>(6 * H);
    w.out_tm = nullptr;
    bigru2_carve(cv, B, T, H, save, w.g);
    if (save) {
        w.dout_tm = cv.take<float>(TB * 2 * H);
        w.dx_tm = cv.take<float>(TB * K);
        w.tmp3h = cv.take<float>(3 * H + 64);                   // + 64 partial sums of the scalar input's gradient (pw_beat_input_grad)
    } else {
        w.dout_tm = w.dx_tm = w.tmp3h = nullptr;
    }
    return cv.bytes();
}

}  // namespace

extern "C" {

int inet_abi_version(void) { return INET_ABI_VERSION; }

int inet_vae_param_count(const inet_vae_config* cfg) {
    if (!cfg_ok(cfg)) return -1;
    return (int)VaeLayout(*cfg).ab.entries.size();
}
int64_t inet_vae_param_floats(const inet_vae_config* cfg) {
    if (!cfg_ok(cfg)) return -1;
    return VaeLayout(*cfg).ab.total;
}
int inet_vae_param_info(const inet_vae_config* cfg, int index, char* name, int name_cap, int64_t* offset_floats,
                        int64_t* dims, int* ndim) {
    if (!cfg_ok(cfg)) return -1;
    return entry_info(VaeLayout(*cfg).ab, index, name, name_cap, offset_floats, dims, ndim);
}
int inet_latent_param_count(const inet_latent_config* cfg) {
    if (!cfg || cfg->rnn_hidden % 16) return -1;
    return (int)LatentLayout(*cfg).ab.entries.size();
}
int64_t inet_latent_param_floats(const inet_latent_config* cfg) {
    if (!cfg || cfg->rnn_hidden % 16) return -1;
    return LatentLayout(*cfg).ab.total;
}
int inet_latent_param_info(const inet_latent_config* cfg, int index, char* name, int name_cap, int64_t* offset_floats,
                           int64_t* dims, int* ndim) {
    if (!cfg || cfg->rnn_hidden % 16) return -1;
    return entry_info(LatentLayout(*cfg).ab, index, name, name_cap, offset_floats, dims, ndim);
}

int64_t inet_vae_encoder_ws_bytes(const inet_vae_config* cfg, int batch, int save) {
    if (!cfg_ok(cfg) || batch <= 0) return -1;
    return (int64_t)vae_encoder_ws_bytes(*cfg, batch, save);
}
int inet_vae_encoder_fwd(const inet_vae_config* cfg, int batch, const int64_t* tokens, const float* params,
                         const float* mask, float* mu, float* logsigma, void* ws, int64_t ws_bytes, int save,
                         void* stream) {
    if (!cfg_ok(cfg) || batch <= 0 || !tokens || !params || !mu || !logsigma || !ws) return -1;
    if (ws_bytes < (int64_t)vae_encoder_ws_bytes(*cfg, batch, save)) return -1;
    return vae_encoder_fwd(*cfg, batch, (const long long*)tokens, params, mask, mu, logsigma, ws, save, (hipStream_t)stream);
}
int inet_vae_encoder_bwd(const inet_vae_config* cfg, int batch, const int64_t* tokens, const float* params,
                         float* grads, const float* mask, const float* dmu, const float* dlogsigma, void* ws,
                         int64_t ws_bytes, int stage, void* stream) {
    if (!cfg_ok(cfg) || batch <= 0 || !tokens || !params || !grads || !dmu || !dlogsigma || !ws) return -1;
    if (stage < 0 || stage > 2 || ws_bytes < (int64_t)vae_encoder_ws_bytes(*cfg, batch, 1)) return -1;
    return vae_encoder_bwd(*cfg, batch, (const long long*)tokens, params, grads, mask, dmu, dlogsigma, ws,
                           (hipStream_t)stream, stage);
}
int64_t inet_vae_decoder_ws_bytes(const inet_vae_config* cfg, int batch, int save) {
    if (!cfg_ok(cfg) || batch <= 0) return -1;
    return (int64_t)vae_decoder_ws_bytes(*cfg, batch, save);
}
int inet_vae_decoder_fwd(const inet_vae_config* cfg, int batch, const float* z, const int64_t* target,
                         int teacher_forced, const float* params, const float* mask_beat, const float* mask_tick,
                         float* weights, int64_t* samples, void* ws, int64_t ws_bytes, int save,
                         uint64_t multinomial_seed, void* stream) {
    if (!cfg_ok(cfg) || batch <= 0 || !z || !params || !weights || !samples || !ws) return -1;
    if (ws_bytes < (int64_t)vae_decoder_ws_bytes(*cfg, batch, save)) return -1;
    return vae_decoder_fwd(*cfg, batch, z, (const long long*)target, teacher_forced, params, mask_beat, mask_tick,
                           weights, (long long*)samples, ws, save, (hipStream_t)stream, multinomial_seed);
}
int inet_vae_decoder_bwd(const inet_vae_config* cfg, int batch, const float* dweights, const float* weights,
                         const int64_t* tokens_in, const float* params, float* grads, const float* mask_beat,
                         const float* mask_tick, float* dz, void* ws, int64_t ws_bytes, void* stream) {
    if (!cfg_ok(cfg) || batch <= 0 || !dweights || !weights || !tokens_in || !params || !ws) return -1;
    if (ws_bytes < (int64_t)vae_decoder_ws_bytes(*cfg, batch, 1)) return -1;
    return vae_decoder_bwd(*cfg, batch, dweights, weights, (const long long*)tokens_in, params, grads, mask_beat,
                           mask_tick, dz, ws, (hipStream_t)stream);
}

int inet_vae_ws_field(const inet_vae_config* cfg, int batch, int which, const char* name, int64_t* offset_floats,
                      int64_t* count) {
    if (!cfg_ok(cfg) || batch <= 0 || !name) return -1;
    long long off = 0, n = 0;
    const int rc = vae_ws_field(*cfg, batch, which, name, &off, &n);
    if (rc == 0) { if (offset_floats) *offset_floats = off; if (count) *count = n; }
    return rc;
}

int inet_cross_entropy(const float* weights, int64_t ld_w, int rows, int V, const int64_t* targets, float* dW,
                       int64_t ld_dw, float scale, float out_scale, float* loss_sum, float* correct, void* stream) {
    if (!weights || !targets || !loss_sum || !correct || rows <= 0 || V <= 0) return -1;
    return pw_cross_entropy(weights, ld_w, rows, V, (const long long*)targets, dW, ld_dw, scale, out_scale, loss_sum,
                            correct, (hipStream_t)stream);
}
int inet_cross_entropy_ex(const float* weights, int64_t ld_w, int rows, int V, const int64_t* targets, float* dW,
                          int64_t ld_dw, float scale, const float* scale_dev, float out_scale, float* loss_sum,
                          float* correct, const float* add_term, float add_scale, float* fwd_out, float fwd_scale,
                          void* stream) {
    if (!weights || !targets || rows <= 0 || V <= 0 || (!dW && !loss_sum && !correct) || (fwd_out && !scale_dev)) return -1;
    return pw_cross_entropy(weights, ld_w, rows, V, (const long long*)targets, dW, ld_dw, scale, out_scale, loss_sum,
                            correct, (hipStream_t)stream, scale_dev, add_term, add_scale, fwd_out, fwd_scale);
}
int inet_sample_multinomial(const float* weights, int64_t ld_w, int rows, int V, int64_t* out, int64_t stride,
                            uint64_t seed, uint64_t offset, void* stream) {
    if (!weights || !out || rows <= 0 || V <= 0) return -1;
    return pw_sample_multinomial(weights, ld_w, rows, V, (long long*)out, stride, seed, offset, (hipStream_t)stream);
}
int inet_reparam_kl(const float* mu, const float* logsigma, const float* eps, float* z, float* sigma, int64_t n,
                    float* kl_sum, void* stream) {
    if (!mu || !logsigma || n <= 0) return -1;
    return pw_reparam_kl(mu, logsigma, eps, z, sigma, n, kl_sum, (hipStream_t)stream);
}
int inet_latent_bwd(const float* dz, const float* mu, const float* logsigma, const float* eps, float kscale,
                    const float* kscale_dev, float* dmu, float* dlogsigma, int64_t n, void* stream) {
    if (!mu || !logsigma || !dmu || !dlogsigma || n <= 0) return -1;
    return pw_latent_bwd(dz, mu, logsigma, eps, kscale, kscale_dev, dmu, dlogsigma, n, (hipStream_t)stream);
}
int inet_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, int step, float gscale, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return -1;
    return pw_adam(p, g, m, v, n, lr, beta1, beta2, eps, step, gscale, (hipStream_t)stream);
}
int inet_adam_step_ex(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, int step, float gscale, const float* step_flag, uint32_t* report, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return -1;
    return pw_adam(p, g, m, v, n, lr, beta1, beta2, eps, step, gscale, (hipStream_t)stream, step_flag, report);
}
int inet_step_flag_export(float* dst, void* stream) {
    if (!dst) return -1;
    return pw_step_flag_export(dst, (hipStream_t)stream);
}
int inet_token_status(int reset) {
    unsigned* p = token_host_status();
    if (!p) return -2;
    const int v = (int)__atomic_load_n(p, __ATOMIC_RELAXED);
    if (reset) __atomic_store_n(p, 0u, __ATOMIC_RELAXED);
    return v;
}
int inet_epoch_stats_add(float* sums, const float* loss, const float* accuracy, void* stream) {
    if (!sums || !loss) return -1;
    return pw_epoch_stats_add(sums, loss, accuracy, (hipStream_t)stream);
}
int inet_epoch_stats_add_ex(float* sums, const float* loss, const float* accuracy, const float* step_flag, void* stream) {
    if (!sums || !loss) return -1;
    return pw_epoch_stats_add(sums, loss, accuracy, (hipStream_t)stream, step_flag);
}
int inet_dropout_mask(float* out, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
    if (!out || n <= 0 || p < 0.f || p >= 1.f) return -1;
    return pw_dropout_mask(out, n, p, seed, offset, (hipStream_t)stream);
}

int64_t inet_bigru2_ws_bytes(int batch, int T, int K, int H, int save) {
    if (batch <= 0 || T <= 0 || K <= 0 || H <= 0 || H % 16) return -1;
    BiWs w;
    return (int64_t)bi_carve(batch, T, K, H, save, nullptr, w);
}

int inet_bigru2_fwd(int B, int T, int K, int H, const float* x, const float* x_scalar, const float* weights,
                    const float* h0, const float* mask, float* out, float* h_n, void* ws, int64_t ws_bytes, int save,
                    void* stream) {
    if (B <= 0 || T <= 0 || K <= 0 || H <= 0 || H % 16 || !weights || !ws) return -1;
    if (ws_bytes < inet_bigru2_ws_bytes(B, T, K, H, save)) return -1;
    if (!x && !(x_scalar && K == 1)) return -1;
    hipStream_t s = (hipStream_t)stream;
    BiWs w;
    bi_carve(B, T, K, H, save, ws, w);
    GruDirPtr P[4];
    bigru_ptrs(weights, nullptr, K, H, P);
    BiGru2In in{};
    if (x) {
        INET_TRY(pw_swap01(x, B, T, K, w.x_tm, s));                  // [B,T,K] -> [T,B,K]
        for (int dir = 0; dir < 2; ++dir)
            INET_TRY(linear_fwd(w.x_tm, K, P[dir].w_ih, K, P[dir].b_ih, w.gi0 + dir * 3L * H, 6L * H, T * B, 3 * H, K, EPI_NONE, s));
        in.gi0[0] = w.gi0; in.gi0[1] = w.gi0 + 3L * H; in.gi0_ld = 6L * H; in.gi0_ts = (long)B * 6 * H;
    } else {
        for (int dir = 0; dir < 2; ++dir) {
            INET_TRY(pw_axpb(x_scalar, P[dir].w_ih, 1, P[dir].b_ih, w.gvec + dir * 3L * H, 3 * H, s));
            in.gvec[dir] = w.gvec + dir * 3L * H;
        }
    }
    const long BH = (long)B * H;
    float* hn[4] = {nullptr, nullptr, nullptr, nullptr};
    if (h_n) for (int i = 0; i < 4; ++i) hn[i] = h_n + i * BH;
    INET_TRY(bigru2_core_fwd(B, T, H, P, in, h0, mask, h_n ? hn : nullptr, H, w.g, save, s));
    if (out) INET_TRY(pw_swap01(w.g.h1, T, B, 2 * H, out, s));       // [T,B,2H] -> [B,T,2H]
    return 0;
}

int inet_bigru2_bwd(int B, int T, int K, int H, const float* x, const float* x_scalar, const float* weights,
                    float* grads, const float* mask, const float* dout, const float* dh_n, float* dx,
                    float* dx_scalar, float* dh0, void* ws, int64_t ws_bytes, void* stream) {
    if (B <= 0 || T <= 0 || K <= 0 || H <= 0 || H % 16 || !weights || !ws) return -1;
    if (ws_bytes < inet_bigru2_ws_bytes(B, T, K, H, 1)) return -1;
    if (!x && !(x_scalar && K == 1)) return -1;
    hipStream_t s = (hipStream_t)stream;
    BiWs w;
    bi_carve(B, T, K, H, 1, ws, w);
    GruDirPtr P[4];
    bigru_ptrs(weights, grads, K, H, P);
    const long BH = (long)B * H;
    if (dout) INET_TRY(pw_swap01(dout, B, T, 2 * H, w.dout_tm, s));
    const float* dhn[4] = {nullptr, nullptr, nullptr, nullptr};
    if (dh_n) for (int i = 0; i < 4; ++i) dhn[i] = dh_n + i * BH;
    INET_TRY(bigru2_core_bwd(B, T, H, P, mask, dout ? w.dout_tm : nullptr, dh_n ? dhn : nullptr, H, dh0, w.g, s));
    for (int dir = 0; dir < 2; ++dir) {
        const float* dgi = w.g.dgi0 + dir * 3L * H;
        if (x) {
            if (grads) INET_TRY(linear_wgrad(dgi, 6L * H, w.x_tm, K, P[dir].dw_ih, K, T * B, 3 * H, K, side_fork(s)));
            if (dx) INET_TRY(linear_dgrad(dgi, 6L * H, P[dir].w_ih, K, w.dx_tm, K, T * B, 3 * H, K, EPI_NONE, nullptr, 0,
                                          dir == 0 ? ACC_STORE : ACC_ADD, s));
        } else if (grads) {
            INET_TRY(pw_zero(w.tmp3h, 3L * H, s));
            INET_TRY(pw_colsum(dgi, 6L * H, T * B, 3 * H, w.tmp3h, s));
            if (!dx_scalar) return -1;
            INET_TRY(pw_beat_input_grad(w.tmp3h, P[dir].w_ih, 1, x_scalar, P[dir].dw_ih, dx_scalar, 3 * H, dgi, 6L * H, T * B,
                                        w.tmp3h + 3 * H, s));
        }
    }
    if (x && dx) INET_TRY(pw_swap01(w.dx_tm, T, B, K, dx, s));
    return side_join(s);
}

int inet_gemm(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
              int64_t ldc, int M, int N, int K, const float* bias, const float* aux, int64_t ldaux, int epi, int acc,
              void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || epi < 0 || epi > 5 || acc < 0 || acc > 1) return -1;
    return launch_gemm(gemm_args(A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, bias, epi, aux, ldaux, acc),
                       (hipStream_t)stream);
}

int inet_gemm_bf3(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
                  int64_t ldc, int M, int N, int K, const float* bias, int acc, int ksplit, void* stream) {
    if (!A || !B || !C || acc < 0 || acc > 1 || ksplit < 0 || !gemm_bf3_ok(M, N, K)) return -1;
    if ((a_kmajor && M % 64) || (b_kmajor && N % 64)) return -1;
    hipStream_t s = (hipStream_t)stream;
    unsigned char* scratch = nullptr;
    const size_t ab = bf3_bytes(M, K), bb = bf3_bytes(N, K);
    if (hipMalloc(&scratch, ab + bb) != hipSuccess) return -2;
    int rc = bf3_split(A, lda, a_kmajor, M, K, scratch, (long)bf3_piece_bytes(M, K), K / 32, 0, 0, s);
    if (rc == 0) rc = bf3_split(B, ldb, b_kmajor, N, K, scratch + ab, (long)bf3_piece_bytes(N, K), K / 32, 0, 0, s);
    if (rc == 0) {
        Bf3Gemm g{};
        g.A = scratch; g.a_piece = (long)bf3_piece_bytes(M, K); g.a_kb = K / 32;
        g.B = scratch + ab; g.b_piece = (long)bf3_piece_bytes(N, K); g.b_kb = K / 32;
        g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.epi = EPI_NONE; g.acc = acc; g.ksplit = ksplit;
        rc = launch_gemm_bf3(g, s);
    }
    if (hipStreamSynchronize(s) != hipSuccess) rc = -2;
    (void)hipFree(scratch);
    return rc;
}

int inet_gemm_batched(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
                      int64_t ldc, int M, int N, int K, int nbatch, int64_t batchA, int64_t batchB, int64_t batchC,
                      void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || nbatch < 1 || nbatch > 8) return -1;
    GemmArgs g = gemm_args(A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, nullptr, EPI_NONE, nullptr, 0, ACC_ADD);
    g.nbatch = nbatch; g.batchA = batchA; g.batchB = batchB; g.batchC = batchC;
    return launch_gemm(g, (hipStream_t)stream);
}

int inet_linear_fwd(const float* x, const float* W, const float* b, float* y, int M, int N, int K, int epi,
                    void* stream) {
    if (!x || !W || !y || M <= 0 || N <= 0 || K <= 0 || epi < 0 || epi > 2) return -1;
    return linear_fwd(x, K, W, K, b, y, N, M, N, K, epi, (hipStream_t)stream);
}
int inet_linear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db, int M, int N,
                    int K, void* stream) {
    if (!dy || M <= 0 || N <= 0 || K <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        if (!W) return -1;
        // an input width that is not a multiple of 64 (AnticipationRNN: 20 embedding + 256 constraint columns) would send the
        // whole product to the LDS-tiled fallback: the leading K % 64 columns go there alone, the rest to the direct kernels
        const int r = K % 64;
        if (r && K - r >= 128 && M >= 1024) {
            INET_TRY(launch_gemm(gemm_args(dy, N, 0, W, K, 1, dx, K, M, r, N), s));
            INET_TRY(launch_gemm(gemm_args(dy, N, 0, W + r, K, 1, dx + r, K, M, K - r, N), s));
        } else INET_TRY(linear_dgrad(dy, N, W, K, dx, K, M, N, K, EPI_NONE, nullptr, 0, ACC_STORE, s));
    }
    if (dW || db) {
        hipStream_t ss = side_fork(s);                       // leaf work
        if (dW) { if (!x) return -1; INET_TRY(linear_wgrad(dy, N, x, K, dW, K, M, N, K, ss)); }
        if (db) INET_TRY(pw_colsum(dy, N, M, N, db, ss));
        return side_join(s);
    }
    return 0;
}

int64_t inet_lstm_ws_bytes(int batch, int T, int H, int save) {
    if (batch <= 0 || T <= 0 || H <= 0 || H % 16) return -1;
    return (int64_t)lstm_ws_bytes(batch, T, H, save);
}
int inet_lstm_fwd(int B, int T, int H, const float* gi, const float* W_hh, const float* b_hh, const float* h0,
                  const float* c0, int reverse, float* out, float* hT, float* cT, void* ws, int64_t ws_bytes, int save,
                  void* stream) {
    if (B <= 0 || T <= 0 || H <= 0 || H % 16 || !gi || !W_hh || !b_hh || !out || !ws) return -1;
    if (ws_bytes < (int64_t)lstm_ws_bytes(B, T, H, save)) return -1;
    return lstm_seq_fwd(B, T, H, gi, W_hh, b_hh, h0, c0, reverse, out, hT, cT, ws, save, (hipStream_t)stream);
}
int inet_lstm_bwd(int B, int T, int H, const float* W_hh, const float* h0, const float* out, const float* dout,
                  const float* dhT, const float* dcT, int reverse, float* dgi, float* dW_hh, float* db_ih, float* db_hh,
                  float* dh0, float* dc0, void* ws, int64_t ws_bytes, void* stream) {
    if (B <= 0 || T <= 0 || H <= 0 || H % 16 || !W_hh || !out || !dgi || !ws) return -1;
    if ((dW_hh != nullptr) != (db_hh != nullptr) || (dW_hh != nullptr) != (db_ih != nullptr)) return -1;
    if (ws_bytes < (int64_t)lstm_ws_bytes(B, T, H, 1)) return -1;
    return lstm_seq_bwd(B, T, H, W_hh, h0, out, dout, dhT, dcT, reverse, dgi, dW_hh, db_ih, db_hh, dh0, dc0, ws,
                        (hipStream_t)stream);
}
int inet_lstm2_ok(int B, int T, int H) { return B > 0 && T > 0 && H > 0 && H % 16 == 0 && lstm2_ok(B, T, H) ? 1 : 0; }
int inet_lstm2_fwd(int B, int T, int H, const float* gi0, const float* W_hh0, const float* b_hh0, const float* W_ih1,
                   const float* b_ih1, const float* W_hh1, const float* b_hh1, int reverse, float* out0, float* gi1,
                   float* out1, void* ws0, void* ws1, int64_t ws_bytes, int save, void* stream) {
    if (B <= 0 || T <= 0 || H <= 0 || H % 16 || !gi0 || !W_hh0 || !b_hh0 || !W_ih1 || !b_ih1 || !W_hh1 || !b_hh1 || !out0 ||
        !gi1 || !out1 || !ws0 || !ws1 || ws0 == ws1)
        return -1;
    if (ws_bytes < (int64_t)lstm_ws_bytes(B, T, H, save)) return -1;
    return lstm2_seq_fwd(B, T, H, gi0, W_hh0, b_hh0, W_ih1, b_ih1, W_hh1, b_hh1, reverse, out0, gi1, out1, ws0, ws1, save,
                         (hipStream_t)stream);
}
int64_t inet_arnn_generate_ws_floats(int L, int E, int Hc, int H, int U, int V) {
    if (L <= 0 || E <= 0 || Hc < 0 || H <= 0 || H % 16 || U <= 0 || V <= 0) return -1;
    return (int64_t)arnn_generate_ws_floats(L, E, Hc, H, U, V);
}
int inet_arnn_generate(int L, int E, int Hc, int H, int U, int V, const float* emb, const float* oc0, int64_t oc_stride,
                       const float* W_ih0, const float* b_ih0, const float* W_hh0, const float* b_hh0, const float* W_ih1,
                       const float* b_ih1, const float* W_hh1, const float* b_hh1, const float* W1, const float* b1,
                       const float* W2, const float* b2, const float* hc_init, const int64_t* first_tok, int64_t* tokens, float* ws,
                       int64_t ws_floats, void* stream) {
    if (L <= 0 || E <= 0 || Hc < 0 || H <= 0 || H % 16 || U <= 0 || V <= 0 || !emb || (Hc && !oc0) || !W_ih0 || !b_ih0 || !W_hh0 ||
        !b_hh0 || !W_ih1 || !b_ih1 || !W_hh1 || !b_hh1 || !W1 || !b1 || !W2 || !b2 || !tokens || !ws)
        return -1;
    if (ws_floats < (int64_t)arnn_generate_ws_floats(L, E, Hc, H, U, V)) return -1;
    return arnn_generate(L, E, Hc, H, U, V, emb, oc0, (long)oc_stride, W_ih0, b_ih0, W_hh0, b_hh0, W_ih1, b_ih1, W_hh1, b_hh1, W1, b1,
                         W2, b2, hc_init, (const long long*)first_tok, (long long*)tokens, ws, (hipStream_t)stream);
}
int inet_lstm2_bwd(int B, int T, int H, const float* W_hh0, const float* W_ih1, const float* W_hh1, const float* out0,
                   const float* out1, const float* dout1, int reverse, float* dgi0, float* dgi1, float* dout0,
                   float* dW_hh0, float* db_ih0, float* db_hh0, float* dW_ih1, float* dW_hh1, float* db_ih1, float* db_hh1,
                   void* ws0, void* ws1, int64_t ws_bytes, void* stream) {
    if (B <= 0 || T <= 0 || H <= 0 || H % 16 || !W_hh0 || !W_ih1 || !W_hh1 || !out0 || !out1 || !dout1 || !dgi0 || !dgi1 ||
        !dout0 || !ws0 || !ws1 || ws0 == ws1)
        return -1;
    const int given = (dW_hh0 != nullptr) + (db_ih0 != nullptr) + (db_hh0 != nullptr) + (dW_ih1 != nullptr) +
                      (dW_hh1 != nullptr) + (db_ih1 != nullptr) + (db_hh1 != nullptr);
    if (given != 0 && given != 7) return -1;
    if (ws_bytes < (int64_t)lstm_ws_bytes(B, T, H, 1)) return -1;
    return lstm2_seq_bwd(B, T, H, W_hh0, W_ih1, W_hh1, out0, out1, dout1, reverse, dgi0, dgi1, dout0, dW_hh0, db_ih0, db_hh0,
                         dW_ih1, dW_hh1, db_ih1, db_hh1, ws0, ws1, (hipStream_t)stream);
}
int inet_embedding_fwd(const float* table, const int64_t* idx, int64_t rows, int E, float* out, const float* row_scale,
                       void* stream) {
    if (!table || !idx || !out || rows <= 0 || E <= 0) return -1;
    return pw_embedding_fwd(table, (const long long*)idx, rows, E, out, row_scale, (hipStream_t)stream);
}
int inet_embedding_bwd(const float* dout, const int64_t* idx, int64_t rows, int E, float* dtable, const float* row_scale,
                       int num_embeddings, void* stream) {
    if (!dout || !idx || !dtable || rows <= 0 || E <= 0 || num_embeddings < 0) return -1;
    hipStream_t ss = side_fork((hipStream_t)stream);        // leaf work: nothing downstream reads an embedding's gradient
    INET_TRY(pw_embedding_bwd(dout, (const long long*)idx, rows, E, dtable, row_scale, ss, num_embeddings));
    return side_join((hipStream_t)stream);
}
int inet_relu_bwd(const float* dy, const float* y, float* dpre, int64_t n, void* stream) {
    if (!dy || !y || !dpre || n <= 0) return -1;
    return pw_copy2d(dpre, n, dy, n, y, n, 1, (int)n, (hipStream_t)stream);
}
int inet_argmax(const float* w, int64_t ld, int rows, int V, int64_t* out, int64_t stride, void* stream) {
    if (!w || !out || rows <= 0 || V <= 0) return -1;
    return pw_argmax(w, ld, rows, V, (long long*)out, stride, (hipStream_t)stream);
}

int inet_tokens_to_i64(const int32_t* src, int64_t* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return -1;
    return pw_tokens_i32_to_i64(src, (long long*)dst, n, (hipStream_t)stream);
}
int inet_split_score(const int32_t* score, int batch, int n_measures, int measure_len, int n_past, int n_target,
                     int64_t* past, int64_t* target, int64_t* future, void* stream) {
    if (!score || batch <= 0 || n_measures <= 0 || measure_len <= 0 || n_past < 0 || n_target < 0 ||
        n_past + n_target > n_measures)
        return -1;
    if ((n_past > 0 && !past) || (n_target > 0 && !target) || (n_measures - n_past - n_target > 0 && !future)) return -1;
    return pw_split_measures(score, batch, n_measures, measure_len, n_past, n_target, (long long*)past,
                             (long long*)target, (long long*)future, (hipStream_t)stream);
}

int inet_set_option(int key, int value) {
    if (key == 0) { side_set_enabled(value); return 0; }
    if (key == 1) { side_set_defer(value); return 0; }
    if (key == 4) { chain_set_enabled(value); return 0; }
    if (key == 2) { if (value < -1 || value > 4) return -1; gemm_set_force(value, -1); return 0; }
    if (key == 3) { if (value < 0) return -1; gemm_set_force(-2, value); return 0; }
    if (key == 5) { if (value < 0 || value > 4) return -1; gemm_set_direct(value); return 0; }
    if (key == 6) { chain_arm_fault(value); return 0; }
    if (key == 7) { if (value != 0 && value != 9) return -1; chain2_set_mode(value); return 0; }
    if (key == 8) { if (value != 0 && value != 9) return -1; bf3_set_mode(value); return 0; }
    if (key == 9) { if (value < 0 || value > 7) return -1; bf3_set_emit_mask(value); return 0; }
    // (keys 10, 11 -- which layers' weight gradients run on the bf16 pipe; the second-generation BPTT kernel -- were removed in round 4)
    if (key == 12) { gru_step_bf3_set_min_tiles(value); return 0; }
    if (key == 13) { if (value < 0 || value > 3) return -1; side_set_active(value); return 0; }
    if (key == 14) { if (value < 0 || value > 4) return -1; arnn_gen_set_mode(value); return 0; }
    if (key == 15) { if (value < 0 || value > 5) return -1; decode_b1_set_mode(value); return 0; }
    if (key == 16) {                                           // entry threshold of the slow-wait recorder, in polls (chain.h)
        unsigned* d = chain_dev_status();
        const unsigned polls = (unsigned)value;
        if (value < (int)chain::kSlowSpins || !d) return -1;
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(d + chain::kRecWord + 1, &polls, 4, hipMemcpyHostToDevice) != hipSuccess) return -2;
        return 0;
    }
    return -1;
}

int inet_chain_status(int reset) {
    unsigned* p = chain_host_status();
    if (!p) return -2;
    const int v = (int)__atomic_load_n(p, __ATOMIC_RELAXED);
    if (reset && v != 0 && chain_status_reset() != 0) return -2;   // (device-side twin too: needs the device idle)
    return v;
}

int inet_debug_read(void* dst, int64_t nbytes) {
    unsigned* d = chain_dev_status();
    if (!dst || nbytes <= 0 || nbytes > kChainDiagBytes || !d) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpy(dst, d + kChainDiagWord, (size_t)nbytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}

int inet_slow_waits(unsigned* dst, int max_entries, int reset, int64_t* noted) {
    unsigned* d = chain_dev_status();
    if (!d || max_entries < 0 || (max_entries > 0 && !dst)) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    unsigned head[3] = {0, 0, 0};                             // waits noted, entry threshold, slow waits
    if (hipMemcpy(head, d + chain::kRecWord, sizeof head, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    if (noted) *noted = (int64_t)head[0];
    const unsigned slow = head[2];
    const int have = (int)(slow < (unsigned)chain::kRecEntries ? slow : (unsigned)chain::kRecEntries);
    const int n = have < max_entries ? have : max_entries;
    if (n > 0 && hipMemcpy(dst, d + chain::kRecWord + 8, (size_t)n * 32, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    if (reset && head[0] != 0) {                              // (the threshold, word 1, stays)
        if (hipMemset(d + chain::kRecWord, 0, 4) != hipSuccess ||
            hipMemset(d + chain::kRecWord + 2, 0, 4 * (size_t)(chain::kRecWords - 2)) != hipSuccess) return -2;
    }
    return (int)(slow > 0x7fffffffu ? 0x7fffffffu : slow);
}

int inet_decode_b1_plan(int B, int V, int Z, int* out8) { return decode_b1_plan_check(B, V, Z, out8); }

int inet_preload(void) { return preload_kernels(); }
int inet_kernel_count(void) { return preload_kernel_count(); }

int inet_side_join(void* stream) { return side_join_now((hipStream_t)stream); }
int inet_side_wait(void* stream) { return side_wait_on((hipStream_t)stream); }
int inet_twin_stream(void** stream) {
    if (!stream) return -1;
    *stream = (void*)twin_stream();
    return *stream ? 0 : -2;
}

int inet_gru_step(int B, int H, const float* gi, const float* h_prev, const float* W_hh, const float* b_hh,
                  float* h_new, float* sv5, void* stream) {
    if (B <= 0 || H <= 0 || H % 16 || !gi || !h_prev || !W_hh || !b_hh || !h_new) return -1;
    GruFwdBatch bt{};
    bt.H = H; bt.nprob = 1;
    GruFwdProb& P = bt.p[0];
    P.B = B; P.h_prev = h_prev; P.ld_hprev = H; P.W_hh = W_hh; P.b_hh = b_hh;
    P.gi_dense = gi; P.ld_gi = 3L * H; P.h_new = h_new; P.ld_hnew = H;
    if (sv5) {
        const long as = (long)B * H;
        P.sv_r = sv5; P.sv_z = sv5 + as; P.sv_n = sv5 + 2 * as; P.sv_ghn = sv5 + 3 * as; P.sv_hprev = sv5 + 4 * as;
    }
    return launch_gru_fwd(bt, (hipStream_t)stream);
}

}  // extern "C"
