#pragma once
#include <hip/hip_runtime.h>
#include "../../include/inpaintnet_hip.h"

size_t vae_encoder_ws_bytes(const inet_vae_config& c, int B, int save);
int vae_encoder_fwd(const inet_vae_config& c, int B, const long long* tokens, const float* p, const float* mask,
                    float* mu, float* logsigma, void* ws, int save, hipStream_t s);
int vae_encoder_bwd(const inet_vae_config& c, int B, const long long* tokens, const float* p, float* g,
                    const float* mask, const float* dmu, const float* dls, void* ws, hipStream_t s, int stage = 0);
// stage 0 = the whole backward pass.  1 = the Linear heads and GRU layer 1: afterwards their gradients (two contiguous
// ranges of the arena) are final and a data-parallel all-reduce of them can start; 2 = the rest (layer 0, embedding) on
// the state stage 1 left in `ws`.
size_t vae_decoder_ws_bytes(const inet_vae_config& c, int B, int save);
int vae_decoder_fwd(const inet_vae_config& c, int B, const float* z, const long long* target, int teacher_forced,
                    const float* p, const float* mask_beat, const float* mask_tick, float* weights,
                    long long* samples, void* ws, int save, hipStream_t s, uint64_t multinomial_seed = 0);
int vae_decoder_bwd(const inet_vae_config& c, int B, const float* dweights, const float* weights,
                    const long long* tokens_in, const float* p, float* g, const float* mask_beat,
                    const float* mask_tick, float* dz, void* ws, hipStream_t s);
int vae_ws_field(const inet_vae_config& c, int B, int which, const char* name, long long* offset_floats, long long* count);
