/* inpaintnet_hip.h -- C ABI of the MI355X (gfx950) InpaintNet hot path.
 *
 * Drop-in boundary.  The reference has no FFI: its callers see Python classes
 * (SURVEY.md section 8b).  This header is the C-ABI the repo introduces
 * UNDERNEATH those classes; each entry point cites the reference method whose
 * arithmetic it replaces.  The modules of inpaintnet_amd/ bind it with ctypes and expose
 * the reference's own signatures (MeasureVAE.forward, LatentRNN.forward,
 * Trainer.step ...); INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    stated; the caller owns all memory (parameters, gradients, optimizer
 *    state, workspaces, outputs).  The library never allocates device memory.
 *  - all floating point is fp32; token tensors are int64 (as produced by
 *    utils/helpers.py:17-26 to_cuda_variable_long).
 *  - calls are asynchronous on `stream` (a hipStream_t passed as void*).
 *  - return value: 0 = ok, -1 = invalid argument, -2 = launch/runtime failure, -3 = a *_bwd call on a workspace whose forward
 *    call ran under other library options (inet_set_option keys 4, 7, 8, 9, 12: they decide which kernels run and which piece
 *    buffers exist).
 *  - parameters live in ONE flat fp32 arena per model, laid out in the
 *    reference's state_dict() order; gradients / Adam
 *    moments use arenas of identical layout.  inet_*_param_info() is the
 *    single description of that layout.
 *  - a workspace written by a *_fwd call with save=1 must be handed unchanged to
 *    the matching *_bwd call; every call checks ws_bytes against *_ws_bytes() and
 *    returns -1 rather than touching an undersized workspace.
 */
#ifndef INPAINTNET_HIP_H
#define INPAINTNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INET_ABI_VERSION 1

typedef struct inet_vae_config {
    int32_t num_notes;      /* V  = len(dataset.note2index_dicts[0])   measure_vae.py:56   */
    int32_t emb_dim;        /* E  note_embedding_dim (10)              measure_vae.py:13   */
    int32_t enc_hidden;     /* encoder_hidden_size (512); 2 layers, bidirectional          */
    int32_t z_dim;          /* latent_space_dim (256)                                      */
    int32_t dec_hidden;     /* decoder_hidden_size (512); 2 layers                         */
    int32_t beats;          /* 4, hard-coded at decoder.py:446                             */
    int32_t ticks_per_beat; /* 6, hard-coded at decoder.py:450                             */
} inet_vae_config;

typedef struct inet_latent_config {
    int32_t z_dim;          /* vae_model.latent_space_dim                latent_rnn.py:48  */
    int32_t rnn_hidden;     /* rnn_hidden_size (512): context GRUs; generator uses 2x      */
    int32_t auto_reg;       /* generator input = z (1) or the scalar x_0 (0)  :70-74       */
} inet_latent_config;

int inet_abi_version(void);

/* ---- parameter arena layout -------------------------------------------------------- */
/* MeasureVAE state_dict (encoder.* then decoder.*), MeasureVAE/encoder.py:28-52, decoder.py:335-372 */
int     inet_vae_param_count(const inet_vae_config* cfg);
int64_t inet_vae_param_floats(const inet_vae_config* cfg);
/* name: caller buffer; dims: int64[4]; returns 0 or -1 */
int     inet_vae_param_info(const inet_vae_config* cfg, int index, char* name, int name_cap,
                            int64_t* offset_floats, int64_t* dims, int* ndim);
/* trainable LatentRNN parameters (context_rnn_past/future, generation_rnn, generation_linear, x_0),
 * LatentRNN/latent_rnn.py:53-83 */
int     inet_latent_param_count(const inet_latent_config* cfg);
int64_t inet_latent_param_floats(const inet_latent_config* cfg);
int     inet_latent_param_info(const inet_latent_config* cfg, int index, char* name, int name_cap,
                               int64_t* offset_floats, int64_t* dims, int* ndim);

/* ---- MeasureVAE encoder: Encoder.forward, MeasureVAE/encoder.py:104-134 -------------- */
int64_t inet_vae_encoder_ws_bytes(const inet_vae_config* cfg, int batch, int save);
/* tokens [B,T] int64 row-major (T = beats*ticks_per_beat); params: VAE arena;
 * mask: null, or [T,B,2H] fp32 pre-scaled {0,1/(1-p)} inter-layer dropout mask (time-major);
 * outputs mu, logsigma [B,Z] */
int inet_vae_encoder_fwd(const inet_vae_config* cfg, int batch, const int64_t* tokens, const float* params,
                         const float* mask, float* mu, float* logsigma, void* ws, int64_t ws_bytes, int save,
                         void* stream);
/* autograd of the above (utils/trainer.py:150 loss.backward): accumulates into `grads` (VAE arena layout).
 * stage 0 = the whole pass.  A data-parallel caller may split it: stage 1 = Linear heads + GRU layer 1 (afterwards the
 * arena ranges [weight_ih_l1, note_embedding) and [linear_mean.0.weight, end of the encoder) are final and their
 * all-reduce can start), then stage 2 = GRU layer 0 + embedding on the same `ws`. */
int inet_vae_encoder_bwd(const inet_vae_config* cfg, int batch, const int64_t* tokens, const float* params,
                         float* grads, const float* mask, const float* dmu, const float* dlogsigma, void* ws,
                         int64_t ws_bytes, int stage, void* stream);

/* ---- MeasureVAE decoder: HierarchicalDecoder.forward, MeasureVAE/decoder.py:412-529 -- */
int64_t inet_vae_decoder_ws_bytes(const inet_vae_config* cfg, int batch, int save);
/* z [B,Z]; target [B,T] int64 (read iff teacher_forced); mask_beat [beats,B,H], mask_tick [T,B,H] or null;
 * weights [B,T,V] post-ReLU logits; samples [B,1,T] int64 (= target if teacher forced; else argmax, lowest index on
 * ties, or -- multinomial_seed != 0, decoder.py:506-509 sampling = 'multinomial' -- one draw per tick and row from
 * softmax(weights) with a counter-based generator keyed (multinomial_seed, tick * B + row)) */
int inet_vae_decoder_fwd(const inet_vae_config* cfg, int batch, const float* z, const int64_t* target,
                         int teacher_forced, const float* params, const float* mask_beat, const float* mask_tick,
                         float* weights, int64_t* samples, void* ws, int64_t ws_bytes, int save,
                         uint64_t multinomial_seed, void* stream);
/* dweights [B,T,V] = dLoss/dweights; weights = the forward output; grads may be null (frozen decoder:
 * LatentRNN/latent_rnn.py:42-43) in which case only dz [B,Z] is produced.  `tokens_in` are the tokens that
 * were fed back (= samples of the forward call). */
int inet_vae_decoder_bwd(const inet_vae_config* cfg, int batch, const float* dweights, const float* weights,
                         const int64_t* tokens_in, const float* params, float* grads, const float* mask_beat,
                         const float* mask_tick, float* dz, void* ws, int64_t ws_bytes, void* stream);

/* Test hook: float offset and element count of a named intermediate inside a workspace written with save=1.
 * which 0 = encoder workspace: "a_mu", "a_ls" [B,2H] (SELU outputs of linear_mean.0 / linear_log_std.0, encoder.py:36-52);
 * which 1 = decoder workspace: "hb0" [B,2H] (z_to_beat_rnn_input), "ht0" [beats,B,2H] (beat_emb_to_tick_rnn_hidden),
 * "c_all" [beats,B,H] (beat_emb_to_tick_rnn_input), decoder.py:335-372.  Parity tests read the sign of these SELU outputs
 * to align the derivative branch of near-zero pre-activations with the oracle's. */
int inet_vae_ws_field(const inet_vae_config* cfg, int batch, int which, const char* name, int64_t* offset_floats,
                      int64_t* count);

/* ---- losses: VAETrainer.loss_and_acc_for_batch, vae_trainer.py:16-40,128-139; utils/trainer.py:271-306 */
/* rows of V logits (row stride ld_w); *loss_sum += out_scale * sum_rows (lse - w[target]); *correct += out_scale *
 * #correct (argmax_first) -- out_scale = 1/rows gives the reference's means without a follow-up kernel;
 * dW (nullable, row stride ld_dw) = (softmax - onehot) * scale */
int inet_cross_entropy(const float* weights, int64_t ld_w, int rows, int V, const int64_t* targets, float* dW,
                       int64_t ld_dw, float scale, float out_scale, float* loss_sum, float* correct, void* stream);
/* The same with the glue of VAETrainer.loss_and_acc_for_batch (vae_trainer.py:29-40) folded in: loss_sum / correct may be
 * null (gradient-only call); dW is additionally multiplied by the device scalar scale_dev[0] when given (the upstream
 * gradient of the loss, so that autograd's backward needs no elementwise pass over dW); *loss_sum also receives
 * add_scale * add_term[0] once when add_term is given (the KL term: loss = CE + beta/B * KL complete in one launch);
 * fwd_out[0] = fwd_scale * scale_dev[0] when given (the gradient to hand on to the producer of add_term). */
int inet_cross_entropy_ex(const float* weights, int64_t ld_w, int rows, int V, const int64_t* targets, float* dW,
                          int64_t ld_dw, float scale, const float* scale_dev, float out_scale, float* loss_sum,
                          float* correct, const float* add_term, float add_scale, float* fwd_out, float fwd_scale,
                          void* stream);
/* z = mu + eps*exp(logsigma) (measure_vae.py:119); sigma out (nullable); kl_sum += sum(0.5(s^2+mu^2-1) - ls) */
int inet_reparam_kl(const float* mu, const float* logsigma, const float* eps, float* z, float* sigma, int64_t n,
                    float* kl_sum, void* stream);
/* dmu = dz + k*mu; dlogsigma = dz*eps*sigma + k*(sigma^2-1), k = kscale * (kscale_dev ? *kscale_dev : 1): the gradient of
 * z (dz, nullable) and of the KL sum (a device scalar from autograd, nullable) in one pass */
int inet_latent_bwd(const float* dz, const float* mu, const float* logsigma, const float* eps, float kscale,
                    const float* kscale_dev, float* dmu, float* dlogsigma, int64_t n, void* stream);

/* rows of V (post-ReLU) logits -> out[row*stride] ~ Multinomial(softmax(row))  (decoder.py:506-509): inverse-CDF draw with
 * one counter-based uniform per row keyed (seed, offset + row) */
int inet_sample_multinomial(const float* weights, int64_t ld_w, int rows, int V, int64_t* out, int64_t stride,
                            uint64_t seed, uint64_t offset, void* stream);

/* ---- optimizer: torch.optim.Adam as built at utils/trainer.py:32-35, stepped at :172-177 -------- */
/* p,g,m,v: arenas of n floats; step is 1-based; grads are multiplied by gscale first (1/world_size for DP) */
int inet_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, int step, float gscale, void* stream);
/* The same step for a caller that must know, without stalling the queue, what the kernel decided (round 4):
 *  - step_flag (nullable; TWO device floats): when given, word 0 ALONE decides whether the step is applied -- non-zero = skip.  A
 *    data-parallel caller writes its rank's status there with inet_step_flag_export() -- word 0 = a chain kernel of this rank
 *    timed out, word 1 = a prologue kernel of this rank met a token outside the vocabulary -- and sums the words over ranks
 *    together with the gradients (a 16-byte head in front of the gradient arena: no extra collective), so that every rank skips
 *    a step ANY rank's chain kernels failed in and the replicas stay bit-identical.  Null: this process's own status word decides,
 *    as in inet_adam_step.
 *  - report (nullable): FOUR 32-bit words of caller-owned, device-visible HOST memory (pinned / host-mapped), zeroed by the
 *    caller before the call; the kernel sets [0] = 1 when it has run, [1] = 1 if it skipped the step, [2] = 1 if a parameter
 *    became NaN / inf in this update -- the observable behaviour of MeasureVAE/encoder.py:111-116 and decoder.py:424-429
 *    (ValueError "... has become nan") without a host scan of the weights per forward --, [3] = 1 if step_flag[1] is non-zero
 *    (decoder.py:36-45 check_index: every rank raises its ValueError at the same step).  The caller reads the record behind an
 *    event of its own some steps later: inpaintnet_amd/trainer.py keeps a ring of 16 records per Trainer and reads the one of
 *    step k when it has queued step k + 12 (INET_REPORT_LAG).  The lag is how far the host may run ahead of the GPU and must be
 *    generous: a lag of 2 cost the B = 256 step 5 % (profiles/r04_b_report_lag.txt); Trainer.finish() reads what is outstanding. */
int inet_adam_step_ex(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, int step, float gscale, const float* step_flag, uint32_t* report, void* stream);
/* dst: the TWO floats described above (chain status, token status of this process) */
int inet_step_flag_export(float* dst, void* stream);
/* Number of prologue launches that met a token index outside [0, num_notes) since the last reset (encoder input tokens,
 * teacher-forcing targets): decoder.py:36-45 check_index raises ValueError; here the host-mapped counter is read by the Python
 * layer at its status reads (Trainer.step, the inference wrappers).  reset != 0 clears it.  -2: could not be allocated. */
int inet_token_status(int reset);
/* Epoch statistics of the training loop (utils/trainer.py:124-163 accumulates mean_loss / mean_accuracy per batch):
 * sums[0] += *loss, sums[1] += *accuracy (nullable), sums[2] += 1 -- on the device, and only while no chain launch of this
 * process has timed out since the last inet_chain_status(reset): a step the optimizer kernel skipped (inet_adam_step reads the
 * same flag) does not enter the means either. */
int inet_epoch_stats_add(float* sums, const float* loss, const float* accuracy, void* stream);
/* step_flag as in inet_adam_step_ex: the batch enters the means iff its optimizer step was applied */
int inet_epoch_stats_add_ex(float* sums, const float* loss, const float* accuracy, const float* step_flag, void* stream);


/* ---- dropout masks (nn.GRU inter-layer dropout, encoder.py:32, decoder.py:346,365) --- */
int inet_dropout_mask(float* out, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* ---- generic 2-layer bidirectional GRU: nn.GRU(batch_first, bidirectional, num_layers=2) as used at
 *      LatentRNN/latent_rnn.py:53-82,186-193,228-233 ------------------------------------ */
/* weights: pointer to the 16 tensors of the GRU in state_dict order inside an arena
 * (weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0, *_l0_reverse, *_l1, *_l1_reverse), each 16-byte aligned
 * as produced by inet_latent_param_info.  x [B,T,K] batch-first (or x_scalar: device pointer to ONE float
 * broadcast as a [B,T,1] input when K == 1 and x == null); h0 [4,B,H] or null (zeros);
 * mask null or [T,B,2H]; out (nullable) [B,T,2H]; h_n (nullable) [4,B,H]. */
int64_t inet_bigru2_ws_bytes(int batch, int T, int K, int H, int save);
int inet_bigru2_fwd(int batch, int T, int K, int H, const float* x, const float* x_scalar, const float* weights,
                    const float* h0, const float* mask, float* out, float* h_n, void* ws, int64_t ws_bytes, int save,
                    void* stream);
/* dout (nullable) [B,T,2H], dh_n (nullable) [4,B,H]; grads: same 16-tensor block in the grad arena;
 * dx (nullable) [B,T,K]; dx_scalar (nullable, 1 float, accumulated); dh0 (nullable) [4,B,H] */
int inet_bigru2_bwd(int batch, int T, int K, int H, const float* x, const float* x_scalar, const float* weights,
                    float* grads, const float* mask, const float* dout, const float* dh_n, float* dx,
                    float* dx_scalar, float* dh0, void* ws, int64_t ws_bytes, void* stream);

/* ---- generic fp32 MFMA GEMM (nn.Linear and friends): C[M,N] (op)= epi(A . B^T + bias) -- */
/* a_kmajor/b_kmajor: 0 => operand(row,k) = P[row*ld + k]; 1 => P[k*ld + row].
 * epi: 0 none, 1 SELU, 2 ReLU, 3 *selu'(aux), 4 *aux, 5 *(aux>0).  acc: 0 store, 1 add. */
int inet_gemm(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
              int64_t ldc, int M, int N, int K, const float* bias, const float* aux, int64_t ldaux, int epi,
              int acc, void* stream);
/* `nbatch` such products of one shape in one launch where a batched kernel applies (the two directions of a bi-GRU
 * layer's weight gradients: utils/trainer.py's loss.backward() over encoder.py:53-60), one after the other otherwise:
 * problem i reads A + i*batchA, B + i*batchB and accumulates (acc is always "add") into C + i*batchC (element strides). */
int inet_gemm_batched(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
                      int64_t ldc, int M, int N, int K, int nbatch, int64_t batchA, int64_t batchB, int64_t batchC,
                      void* stream);

/* The same product on the bf16 matrix cores at fp32 accuracy (csrc/gemm_bf3.hip): both operands are split exactly into three
 * bf16 pieces in MFMA fragment order (a scratch allocated for the call: this entry exists for tests and benchmarks; inside the
 * library the pieces live in the callers' workspaces), then C (op)= A . B^T + bias with nine (six) piece products per
 * element product accumulated in f32.  M % 192 == 0, N % 128 == 0, K % 32 == 0 (k-major operands: rows % 64 == 0), else -1.
 * ksplit: 0 = chosen by the library, else the k range is split that many ways over the grid (f32 atomics).  acc: 0 store, 1 add. */
int inet_gemm_bf3(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
                  int64_t ldc, int M, int N, int K, const float* bias, int acc, int ksplit, void* stream);

/* nn.Linear forward / backward (LatentRNN.generation_linear, latent_rnn.py:83,232,250):
 * y[M,N] = epi(x[M,K] W[N,K]^T + b), epi in {0 none, 1 SELU, 2 ReLU};
 * backward (no activation): dx[M,K] = dy W (nullable), dW += dy^T x (nullable), db += colsum(dy) (nullable); dW and db are
 * leaf work: they run on the library's side stream (joined before return unless joins are deferred, inet_set_option 1) */
int inet_linear_fwd(const float* x, const float* W, const float* b, float* y, int M, int N, int K, int epi,
                    void* stream);
int inet_linear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db, int M, int N,
                    int K, void* stream);

/* ---- single-layer LSTM over T steps: torch.nn.LSTM(num_layers=1, batch_first) as stacked by
 *      lstm_with_activations, AnticipationRNN/anticipation_rnn_gauss_reg_model.py:14-39,110-133.
 * gi [T,B,4H] time-major input-side pre-activations (x W_ih^T + b_ih, formed by inet_linear_fwd); gate order i,f,g,o;
 * h0/c0 [B,H] or null (zeros); reverse != 0 processes t = T-1..0 (the constraint LSTM, :467-474); out [T,B,H];
 * hT/cT nullable [B,H].  Backward: dout [T,B,H] (nullable), dhT/dcT (nullable) -> dgi [T,B,4H] (gate gradients: the
 * caller forms dx and dW_ih with inet_linear_bwd); dW_hh/db_ih/db_hh accumulated when all three are given. */
int64_t inet_lstm_ws_bytes(int batch, int T, int H, int save);
int inet_lstm_fwd(int batch, int T, int H, const float* gi, const float* W_hh, const float* b_hh, const float* h0,
                  const float* c0, int reverse, float* out, float* hT, float* cT, void* ws, int64_t ws_bytes, int save,
                  void* stream);
int inet_lstm_bwd(int batch, int T, int H, const float* W_hh, const float* h0, const float* out, const float* dout,
                  const float* dhT, const float* dcT, int reverse, float* dgi, float* dW_hh, float* db_ih, float* db_hh,
                  float* dh0, float* dc0, void* ws, int64_t ws_bytes, void* stream);
/* Two stacked LSTM layers with zero initial states (the loop of lstm_with_activations over lstm_list,
 * anticipation_rnn_gauss_reg_model.py:14-39) as a pipeline over chunks of time steps: layer 1 runs chunk c on a second
 * stream while layer 0 runs chunk c+1.  gi0 [T,B,4H] (x W_ih0^T + b_ih0); out0/out1 [T,B,H]; gi1 [T,B,4H] scratch;
 * ws0/ws1: one inet_lstm_ws_bytes workspace per layer.  Backward: dout1 [T,B,H] -> dgi0, dgi1 [T,B,4H] (the caller forms
 * dx and dW_ih0 from dgi0 with inet_linear_bwd); dout0 [T,B,H] scratch; the weight / bias gradients (all given or all
 * null) are accumulated.  Both return 1 (and do nothing) when the shape does not qualify for the pipeline: the caller
 * then runs inet_lstm_fwd / inet_lstm_bwd layer by layer. */
int inet_lstm2_ok(int batch, int T, int H);       /* 1 when inet_lstm2_fwd / _bwd will take the shape */
int inet_lstm2_fwd(int batch, int T, int H, const float* gi0, const float* W_hh0, const float* b_hh0, const float* W_ih1,
                   const float* b_ih1, const float* W_hh1, const float* b_hh1, int reverse, float* out0, float* gi1,
                   float* out1, void* ws0, void* ws1, int64_t ws_bytes, int save, void* stream);
int inet_lstm2_bwd(int batch, int T, int H, const float* W_hh0, const float* W_ih1, const float* W_hh1,
                   const float* out0, const float* out1, const float* dout1, int reverse, float* dgi0, float* dgi1,
                   float* dout0, float* dW_hh0, float* db_ih0, float* db_hh0, float* dW_ih1, float* dW_hh1,
                   float* db_ih1, float* db_hh1, void* ws0, void* ws1, int64_t ws_bytes, void* stream);

/* The sequential part of AnticipationRNN's free-running pass (AnticipationRNN/anticipation_rnn_gauss_reg_model.py:190-259): the
 * generation LSTMs feed the argmax of BATCH ELEMENT 0 back to the whole batch (:253-256), so the token sequence depends on that one
 * row.  L ticks of [embedding of the previous token (start: token 0) | oc0 + t * oc_stride (the tick's constraint output, Hc floats)]
 * -> LSTM 0 -> LSTM 1 -> ReLU(linear_1) -> note head -> argmax (numpy order: NaN is the maximum, lowest index among equals), without a
 * host round trip: ONE persistent launch of 13 workgroups with their weights in registers for the reference's configuration (H = U =
 * 256, V <= 128; csrc/arnn_gen.hip, round 5: two hand-offs per tick), four small launches per tick otherwise (inet_set_option key 14 /
 * INET_ARNN_GEN: 0 = always the launches, 1 = persistent kernel, 2 = its workgroups on one XCD, 3 = default: 2 + XCD-local granule stores); tokens [L] int64 on the device.  emb [.,E]; W_ih0 [4H, E+Hc]; W_ih1, W_hh* [4H,H]; W1 [U,H]; W2 [V,U].  hc_init
 * (nullable: zeros) = the state the ticks go on from, [layer][h | c][H]; first_tok (nullable: token 0) = device pointer to the token in
 * front of the first tick -- forward_inpaint (:261-346) generates a window behind a teacher-forced prefix.  The caller then runs the
 * whole batch over these tokens with the batched kernels (inpaintnet_amd.arnn._forward_no_tf / forward_inpaint). */
int64_t inet_arnn_generate_ws_floats(int L, int E, int Hc, int H, int U, int V);
int inet_arnn_generate(int L, int E, int Hc, int H, int U, int V, const float* emb, const float* oc0, int64_t oc_stride,
                       const float* W_ih0, const float* b_ih0, const float* W_hh0, const float* b_hh0, const float* W_ih1,
                       const float* b_ih1, const float* W_hh1, const float* b_hh1, const float* W1, const float* b1,
                       const float* W2, const float* b2, const float* hc_init, const int64_t* first_tok, int64_t* tokens, float* ws,
                       int64_t ws_floats, void* stream);
/* nn.Embedding forward / backward (rows of E floats gathered by int64 index; backward accumulates with atomics).
 * row_scale (nullable, [rows]) multiplies each gathered row: the Dropout2d on the shifted note embeddings
 * (drop_input, anticipation_rnn_gauss_reg_model.py:437-442) and the all-zero first time step (:373-376). */
int inet_embedding_fwd(const float* table, const int64_t* idx, int64_t rows, int E, float* out, const float* row_scale,
                       void* stream);
/* num_embeddings: rows of the table (0 = not told): small tables take a segment-sum kernel instead of per-element atomics */
int inet_embedding_bwd(const float* dout, const int64_t* idx, int64_t rows, int E, float* dtable, const float* row_scale,
                       int num_embeddings, void* stream);
/* dpre = dy where y > 0 else 0   (backward of the ReLU fused into inet_linear_fwd epi=2) */
int inet_relu_bwd(const float* dy, const float* y, float* dpre, int64_t n, void* stream);
/* out[r*stride] = argmax_v w[r*ld + v], lowest index on ties (Tensor.max(1) / np.argmax semantics) */
int inet_argmax(const float* w, int64_t ld, int rows, int V, int64_t* out, int64_t stride, void* stream);

/* ---- input feed: the dataset tensors are int32 `score (N,1,384)` (DatasetManager/the_session/folk_dataset.py:852-861),
 *      the models take int64 (utils/helpers.py:17-26 to_cuda_variable_long).  The H2D copy moves the int32 form; these
 *      widen / split on the device. ------------------------------------------------------------------------------ */
/* dst[i] = src[i]: VAETrainer.process_batch_data (MeasureVAE/vae_trainer.py:42-55) -- (B,1,384) viewed as (B*16,24) */
int inet_tokens_to_i64(const int32_t* src, int64_t* dst, int64_t n, void* stream);
/* score [B, n_measures*measure_len] int32 -> past [B,n_past,L], target [B,n_target,L], future [B,rest,L] int64, each
 * contiguous: LatentRNNTrainer.split_score / split_to_measures (LatentRNN/latent_rnn_trainer.py:134-176) */
int inet_split_score(const int32_t* score, int batch, int n_measures, int measure_len, int n_past, int n_target,
                     int64_t* past, int64_t* target, int64_t* future, void* stream);

/* single GRU step (test hook for the fused step kernel; semantics of torch.nn.GRUCell with the input-side
 * gate pre-activations gi [B,3H] already formed) -- r,z,n,ghn,hprev saves are nullable [B,H] */
int inet_gru_step(int batch, int H, const float* gi, const float* h_prev, const float* W_hh, const float* b_hh,
                  float* h_new, float* sv5, void* stream);

/* ---- runtime options: key 0 = overlap the weight-gradient GEMMs of the backward pass with the BPTT chains on
 * a second, lower-priority HIP stream (default 1; also INET_SIDE_STREAM=0 in the environment) */
int inet_set_option(int key, int value);
/* key 2 = force the batched-GEMM tile configuration: value -1 = cost model (default), 0..4 = 64x64, 128x128, 192x64,
 * 192x128, 192x192 block tiles; key 3 = forced split-K factor (0 = none) used while key 2 is forced.  Test hooks: the
 * parity tests drive every tile configuration through the same shapes (keys 2 and 3). */
/* key 1 = deferred joins (default 0).  With 0 every *_bwd entry point makes `stream` wait for the side stream before
 * it returns.  With 1 it does not: the caller must keep every workspace passed to a *_bwd call alive and call
 * inet_side_join(stream) before anything reads the gradient arena (optimizer step, all-reduce) or frees those
 * workspaces.  Lets the leaf GEMMs of one module's backward overlap the next module's BPTT chain. */
/* key 4 = chain kernels (default 1; INET_CHAIN=0): one persistent launch per recurrent layer, weights resident in
 * registers, the hidden state exchanged between workgroups once per step (csrc/chain.h).  0 = one launch per step. */
/* key 5 = LDS-free GEMM kernels (csrc/gemm.hip): 0 = LDS-tiled kernels only, 1 = by shape
 * (default: shared-strip direct kernels for the big products, workgroup split-K for the medium / small ones), 2 = the
 * direct kernels whenever the shape qualifies (test hook), 3 = direct kernels only (no split-K), 4 = split-K first, also for
 * the long weight-gradient products. */
/* key 6 = test hook: value 1 arms ONE injected fault -- the next forward GRU chain launch loses a workgroup, its group runs
 * into the bounded spin (~0.4 s) and inet_chain_status() turns non-zero: lets the failure path (optimizer skip, fallback to
 * per-step kernels) be tested on a healthy GPU. */
/* key 7 = generation of the GRU FORWARD chain kernels: 9 (default; INET_CHAIN2) = second generation (csrc/gru_chain2.hip: one row
 * block per wave, W in LDS, the contraction on the bf16 matrix cores with every fp32 operand split exactly into three bf16 pieces
 * and all nine piece products accumulated in f32 = the products of fp32 arithmetic), 0 = first generation (f32-input MFMA, W in
 * registers).  The BPTT chains always run on the first generation. */
/* key 8 = the encoder's large products (csrc/gemm_bf3.hip; INET_GEMM_BF3): 9 (default) = on the bf16 matrix cores through the same
 * exact three-piece split, nine piece products; 0 = on the f32-input kernels of csrc/gemm.hip.
 * key 9 = which bf16 pieces the chain kernels write themselves (bit 0 the forward chains' rows, 1 their transposed pieces,
 * 2 the BPTT kernel's dgi rows; default 7) -- what they do not write, bf3_split launches make from the f32 arrays: same results.
 * (Round 3's keys 10 and 11 and the value 6 of keys 7 / 8 -- weight-gradient pipe per layer, second-generation BPTT kernel, six
 * piece products -- selected builds that lost their A/Bs or were not fp32 arithmetic; they were removed in round 4: -1.)
 * key 12 = big-batch GRU forward steps on the bf16 pipe (csrc/gru_step_bf3.hip): a layer whose single time
 * step has at least this many tiles of 128 rows x 64 units (default 256 = one per CU: B = 2048 at H = 512, two directions) runs one
 * product per step with the GRU cell as its epilogue instead of chunked chain launches; 0 = never.
 * key 13 = how many of the side streams take leaf work in rotation from now on (0 = all that exist, default; 1: what a process with
 * a gradient exchange beside its steps wants -- inpaintnet_amd.dp sets it: the runtime deals FOUR hardware queues, and caller + two
 * side streams + the exchange's two streams measured 4.94 ms per B = 256 step against 3.87 with one side stream).
 * key 14 = AnticipationRNN's token pass (inet_arnn_generate; INET_ARNN_GEN): 3 (default) = one persistent launch where the shape
 * allows, its 13 workgroups on every 8th workgroup id (one XCD as dispatched today) and -- once they have FOUND themselves on one XCD
 * (they compare XCC ids at the start of the launch) -- granules as plain stores that stay in that XCD's L2; 2 = the same with
 * agent-scope stores; 1 = on 13 consecutive ids; 0 = four launches per tick; 4 = test hook: 3's request on consecutive ids (the
 * XCC-id check must refuse it).
 * key 15 = the free-running decode of ONE to SIXTEEN measures at H = 512 (inference; csrc/decode_b1.hip; INET_DECODE_B1): 4 (default) =
 * 3 with every team's critical workgroups (C + 16 TBi, or the 16 CB of the merged build) on workgroup ids of one residue mod 8 --
 * one XCD under today's round-robin dispatch -- and, once they have CHECKED that they share an XCD, XCD-local copies of h0 / h1
 * written with plain stores next to the agent-scope ones (correct under any placement: a failed check uses the agent-scope copies; 5 = test
 * hook: the same request on consecutive workgroup ids, where the check has to refuse); 3 =
 * one or two measures: ONE register-resident persistent launch for the whole call behind the prologue launch (129 workgroups: 49 for
 * the 24 ticks -- two hand-offs per tick; one where a single workgroup kind can hold layer 1, the head and the argmax: one row with
 * V <= 64 -- and 80 for the beat path); three to sixteen: teams of the 49 tick workgroups in one launch, two rows per team up to ten
 * measures, four beyond -- up to six measures still with the beat path's workgroups in the same launch, beyond behind the beat path's
 * own launches; 2 / 1 = the tick path only, behind the beat path's eight launches, on
 * every 4th workgroup id / on consecutive ids; 0 = decode_chain.hip's 32-member exchange kernel.
 * Keys 4, 7-12, 14 must not change between a forward call and its backward call, nor between sizing a workspace and using it. */
int inet_side_join(void* stream);
/* `stream` -- a THIRD stream, not the one the library calls were issued on -- waits for all side-stream work queued so far.
 * Unlike inet_side_join nothing is consumed: the issuing stream still joins the same work at its own next join (a
 * data-parallel bucket's all-reduce stream orders itself behind the leaf GEMMs this way without the backward pass waiting). */
int inet_side_wait(void* stream);
/* The library's second compute stream ("twin": hipStream_t through *stream; created on first use, before the side streams).  The
 * HIP runtime deals its hardware queues to streams in creation order and a process that keeps more than four of them busy is
 * time-sliced (a fifth stream that carried the optimizer launch made every LatentRNN step 1.7x slower: DESIGN.md section 8), so a
 * caller that wants work beside the library's own -- the trainer's late optimizer launch, a data-parallel bucket's pre-processing --
 * borrows this stream instead of creating one; it orders it against its own stream with events.  The two-layer LSTM pipelines use
 * the same stream inside their calls (never across calls).  -2 if the stream could not be created. */
int inet_twin_stream(void** stream);
/* Number of chain-kernel workgroups that gave up waiting for their group since the last reset (0 = healthy; every
 * in-kernel spin is bounded, so a broken hand-off shows up here instead of hanging the GPU).  Complete after the
 * stream has been synchronised; a non-zero value read earlier is already a definite failure (the counter lives in
 * host-mapped memory: reading it costs nothing and needs no synchronisation).  While it is non-zero inet_adam_step
 * leaves parameters and moments untouched (a device-side twin of the counter is read by the kernel), so a failed step
 * can never reach the weights.  reset != 0 clears both (synchronises the device).  -2: the counter could not be allocated. */
int inet_chain_status(int reset);
/* Diagnostics: copies the first nbytes (<= 16384) of the library's device-side scratch area to the host (synchronises the device).
 * Only instrumented builds write there -- csrc/gru_chain2.hip compiled with -DINET_CHAIN2_STAMPS=1 records the wall-clock stamps of
 * one wave's steps (tools/chain2_anatomy.py) --, a normal build leaves it zero. */
int inet_debug_read(void* dst, int64_t nbytes);
/* The slow-wait recorder.  Every bounded wait on a group / row-block counter inside a persistent kernel that needed at least 64 polls (a
 * steady-state hand-off takes 3-8; a counter poll is ~0.4 us; granule waits, polled by every thread, only from 4096 polls on) is NOTED (*noted, if not null: how many since the last reset); a
 * wait of at least the entry threshold (default 16384 polls ~ 6 ms, inet_set_option key 16: any value >= 16) or one that GAVE UP is SLOW and files one entry
 * when it ends: 8 words {kernel id (1 = gru_chain fwd, 2 = bwd, 3 = gru_chain2 fwd, 5 = lstm fwd, 6 = lstm bwd, 8 = decode_chain,
 * 9 = arnn token pass, 10 = decode_b1) | XCC id << 8 | gave up << 15 | site << 16 (0 = group counter, 1 = granule, 2 = row-block
 * counter, 3 = tagged fragments), workgroup id, expected tag / counter, polls, wall clock lo, hi (100 MHz), 0, 0}.  Copies up to
 * max_entries (<= 127 are kept: the FIRST ones since the last reset) into dst and returns how many waits were slow since the last
 * reset (may exceed what is kept); reset != 0 clears counts and entries (not the threshold).  Synchronises the device.  Noted
 * waits are normal wherever launches overlap -- the members of a chain group wait ~100 us at their first step while the launch
 * becomes resident beside a weight-gradient product --; slow ones are the trace an unexplained timeout or a stalled launch leaves
 * behind (csrc/chain.h record_slow).  -1 bad arguments, -2 runtime failure. */
int inet_slow_waits(unsigned* dst, int max_entries, int reset, int64_t* noted);
/* The launch plan of the register-resident decode (csrc/decode_b1.hip) for a call of B measures with V notes and latent size Z, and
 * a host-side self-check of it -- no GPU needed (tests/test_decode_plan.py).  out8 = {teams, rows per team, shared recurrent groups,
 * critical workgroups per team, placed (1: workgroup ids are mapped to roles so that a team's critical workgroups share a residue mod
 * 8), grid, live workgroups, ok}; ok = every (team, role) the kernel expects appears exactly once among the ids of the grid, every
 * team's critical roles sit on ONE residue, no residue carries more than 32 live workgroups, the grid fits the chip.  0, or -1 for a
 * call the register-resident launch does not take (B > 16, V > 128, ...). */
int inet_decode_b1_plan(int B, int V, int Z, int* out8);
/* Loads every kernel of the library on the CURRENT device (code objects and function objects, which the HIP runtime otherwise
 * builds lazily on the launch path of each kernel's first launch: csrc/preload.hip) without launching anything.  Idempotent per
 * device; returns the number of kernels touched (0 when the device was done already), -2 without a device or on a runtime
 * failure.  The trainers call it at construction, so that no training step pays a first launch. */
int inet_preload(void);
/* Number of __global__ kernels the library registered with the HIP runtime (no GPU needed). */
int inet_kernel_count(void);

/* ---- measurement hooks (bench.py roofline line; not part of the reference surface) --------------- */
/* class 0 = batched MFMA GEMM, 1 = fused GRU/LSTM step forward, 2 = fused GRU/LSTM step backward, 3 = HBM-bound
 * pointwise kernels (the fused Adam update).  While enabled every
 * launch of those kernels is bracketed by hipEventRecord on its stream; read() synchronises and returns the
 * number of launches, the summed event time (ms) and the summed algorithmic FLOPs (2*M*N*K) of the class. */
int inet_prof_enable(int on);
int inet_prof_read(int cls, int64_t* launches, double* total_ms, double* total_flops);
/* per-launch CSV (class,label,us,gflop,mbytes) of everything recorded since inet_prof_enable(1); the label names the
 * kernel instantiation and shape, mbytes = algorithmic HBM bytes of the launch (operands once, results once) */
int inet_prof_dump(const char* path);

#ifdef __cplusplus
}
#endif
#endif /* INPAINTNET_HIP_H */
