#pragma once
#include <hip/hip_runtime.h>
size_t lstm_ws_bytes(int B, int T, int H, int save);
int lstm_seq_fwd(int B, int T, int H, const float* gi, const float* W_hh, const float* b_hh, const float* h0,
                 const float* c0, int reverse, float* out, float* hT, float* cT, void* ws, int save, hipStream_t s);
int lstm_seq_bwd(int B, int T, int H, const float* W_hh, const float* h0, const float* out, const float* dout,
                 const float* dhT, const float* dcT, int reverse, float* dgi, float* dW_hh, float* db_ih, float* db_hh,
                 float* dh0, float* dc0, void* ws, hipStream_t s);
