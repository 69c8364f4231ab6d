#pragma once
#include <hip/hip_runtime.h>
size_t lstm_ws_bytes(int B, int T, int H, int save);
int lstm_seq_fwd(int B, int T, int H, const float* gi, const float* W_hh, const float* b_hh, const float* h0,
                 const float* c0, int reverse, float* out, float* hT, float* cT, void* ws, int save, hipStream_t s);
int lstm_seq_bwd(int B, int T, int H, const float* W_hh, const float* h0, const float* out, const float* dout,
                 const float* dhT, const float* dcT, int reverse, float* dgi, float* dW_hh, float* db_ih, float* db_hh,
                 float* dh0, float* dc0, void* ws, hipStream_t s);
bool lstm2_ok(int B, int T, int H);
// two stacked layers, pipelined over chunks of time steps on two streams (lstm.hip); return 1 = shape does not qualify
int lstm2_seq_fwd(int B, int T, int H, const float* gi0, const float* W_hh0, const float* b_hh0, const float* W_ih1,
                  const float* b_ih1, const float* W_hh1, const float* b_hh1, int reverse, float* out0, float* gi1,
                  float* out1, void* ws0, void* ws1, int save, hipStream_t s);
int lstm2_seq_bwd(int B, int T, int H, const float* W_hh0, const float* W_ih1, const float* W_hh1, const float* out0,
                  const float* out1, const float* dout1, int reverse, float* dgi0, float* dgi1, float* dout0, float* dW_hh0,
                  float* db_ih0, float* db_hh0, float* dW_ih1, float* dW_hh1, float* db_ih1, float* db_hh1, void* ws0,
                  void* ws1, hipStream_t s);
// the sequential part of AnticipationRNN's free-running pass: L ticks of batch element 0 -> its argmax tokens (lstm.hip)
size_t arnn_generate_ws_floats(int L, int E, int Hc, int H, int U, int V);
int arnn_generate(int L, int E, int Hc, int H, int U, int V, const float* emb, const float* oc0, long oc_stride, const float* W_ih0,
                  const float* b_ih0, const float* W_hh0, const float* b_hh0, const float* W_ih1, const float* b_ih1,
                  const float* W_hh1, const float* b_hh1, const float* W1, const float* b1, const float* W2, const float* b2,
                  const float* hc_init, const long long* first_tok, long long* tokens, float* ws, hipStream_t s);
// ... as ONE persistent launch (arnn_gen.hip, round 5): 13 resident workgroups with their weights in registers, two hand-offs per tick
// on the critical path; H = U = 256, V <= 128 (arnn_generate takes it when it applies; INET_ARNN_GEN=0 / option key 14: never)
bool arnn_token_pass_ok(int H, int U, int V);
void arnn_gen_set_mode(int m);
size_t arnn_token_pass_ws_floats(int L, int V);
int arnn_token_pass(int L, int E, int Hc, int V, const float* emb, const float* oc0, long oc_stride, const float* W_ih0,
                    const float* b_ih0, const float* W_hh0, const float* b_hh0, const float* W_ih1, const float* b_ih1,
                    const float* W_hh1, const float* b_hh1, const float* W1, const float* b1, const float* W2, const float* b2,
                    const float* hc_init, const long long* first_tok, long long* tokens, float* ws, hipStream_t s);
