"""Thin functional layer over the C-ABI: torch tensors in, torch tensors out.

torch is used here only for device memory and the current stream; every number
is produced by the HIP kernels behind include/inpaintnet_hip.h.  All functions
are asynchronous on torch's current stream.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import VaeConfig, LatentConfig, check, ptr, stream_ptr


# Whether autograd is recording OUTSIDE the custom Function whose forward is running (inside Function.forward grad mode is
# always off, and ctx.needs_input_grad only reflects requires_grad flags -- a trainable model evaluated under
# torch.no_grad() would otherwise save its backward activations and miss the save-free inference kernels).
_OUTER_GRAD = [True]


class TrackedFunction(torch.autograd.Function):
    """autograd.Function whose call() records the caller's grad mode first (read it with ops.outer_grad())."""

    @classmethod
    def call(cls, *args):
        _OUTER_GRAD[0] = torch.is_grad_enabled()
        return cls.apply(*args)


def outer_grad():
    return _OUTER_GRAD[0]


def _f32c(t):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t


def _i64c(t):
    assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t


def vae_config(num_notes, emb_dim=10, enc_hidden=512, z_dim=256, dec_hidden=512, beats=4, ticks_per_beat=6):
    return VaeConfig(num_notes, emb_dim, enc_hidden, z_dim, dec_hidden, beats, ticks_per_beat)


def _entries(cfg, count_fn, info_fn):
    n = count_fn(C.byref(cfg))
    if n < 0:
        raise ValueError("invalid model configuration (hidden sizes must be multiples of 16)")
    out = []
    for i in range(n):
        name = C.create_string_buffer(160)
        off = C.c_int64()
        dims = (C.c_int64 * 4)()
        nd = C.c_int()
        check(info_fn(C.byref(cfg), i, name, 160, C.byref(off), dims, C.byref(nd)), "param_info")
        out.append((name.value.decode(), off.value, tuple(dims[j] for j in range(nd.value))))
    return out


def vae_param_table(cfg):
    """[(state_dict key, offset in floats, shape)], total floats -- from the C library."""
    L = _lib.lib()
    return _entries(cfg, L.inet_vae_param_count, L.inet_vae_param_info), L.inet_vae_param_floats(C.byref(cfg))


def latent_param_table(cfg):
    L = _lib.lib()
    return _entries(cfg, L.inet_latent_param_count, L.inet_latent_param_info), L.inet_latent_param_floats(C.byref(cfg))


def _ws(nbytes, device):
    return torch.empty((int(nbytes) + 3) // 4, dtype=torch.float32, device=device)


# ----------------------------------------------------------------------------- encoder
def encoder_ws(cfg, B, save, device):
    n = _lib.lib().inet_vae_encoder_ws_bytes(C.byref(cfg), B, int(save))
    if n < 0:
        raise ValueError("encoder_ws: invalid arguments")
    return _ws(n, device)


def encoder_fwd(cfg, tokens, params, mask=None, save=False, ws=None):
    """tokens [B,T] int64 -> mu, logsigma [B,Z], ws"""
    _i64c(tokens); _f32c(params)
    B = tokens.shape[0]
    if ws is None:
        ws = encoder_ws(cfg, B, save, tokens.device)
    mu = torch.empty(B, cfg.z_dim, dtype=torch.float32, device=tokens.device)
    ls = torch.empty_like(mu)
    check(_lib.lib().inet_vae_encoder_fwd(C.byref(cfg), B, ptr(tokens), ptr(params), ptr(mask), ptr(mu), ptr(ls),
                                          ptr(ws), ws.numel() * 4, int(save), stream_ptr()), "inet_vae_encoder_fwd")
    return mu, ls, ws


# ----------------------------------------------------------------------------- deferred side-stream joins
_DEFER = False
_HELD = []


def side_defer(on, release=True):
    """Deferred joins (include/inpaintnet_hip.h, inet_set_option key 1): the *_bwd calls stop making the current stream
    wait for the side stream; every tensor they were given is kept alive here until side_join().  release=False when
    leaving deferred mode: the current stream joins, the held tensors stay until release_held() (Trainer.step keeps
    them across the gradient exchange)."""
    global _DEFER
    if not on and _DEFER:
        side_join(release=release)
    _DEFER = bool(on)
    check(_lib.lib().inet_set_option(1, int(_DEFER)), "inet_set_option")


def side_join(release=True):
    """Make the current stream wait for all side-stream work, then release the tensors held for it."""
    check(_lib.lib().inet_side_join(stream_ptr()), "inet_side_join")
    if release:
        _HELD.clear()


def release_held():
    _HELD.clear()


def side_join_on(stream):
    """Make `stream` (a torch.cuda.Stream that is NOT the one the library calls were issued on) wait for all side-stream
    work queued so far.  Nothing is consumed or released (inet_side_wait): the issuing stream has not been ordered after
    that work and still joins it at its own next join (dp.start_bucket uses this to start an all-reduce behind the leaf
    GEMMs without stalling the backward pass)."""
    check(_lib.lib().inet_side_wait(C.c_void_p(stream.cuda_stream)), "inet_side_wait")


def arnn_generate_ok(E, Hc, H, U, V):
    """The shapes the one-row kernels of inet_arnn_generate are built for (csrc/lstm.hip: template bounds); callers keep their
    per-tick loop for anything else."""
    return E + Hc <= 320 and H <= 256 and H % 16 == 0 and U <= 256 and V <= 256


def arnn_generate(emb, oc0, W_ih0, b_ih0, W_hh0, b_hh0, W_ih1, b_ih1, W_hh1, b_hh1, W1, b1, W2, b2, hc_init=None, first_tok=None):
    """inet_arnn_generate: the L argmax tokens of batch element 0 of AnticipationRNN's free-running pass.  oc0 [L,Hc] (rows may be
    strided); hc_init [2,2,H] (layer, h|c) or None (zeros); first_tok: 1-element int64 device tensor or None (token 0);
    -> tokens [L] int64 on the device, no host round trip."""
    L, Hc = oc0.shape
    assert oc0.stride(1) == 1
    E, H, U, V = emb.shape[1], W_hh0.shape[1], W1.shape[0], W2.shape[0]
    for t in (emb, W_ih0, b_ih0, W_hh0, b_hh0, W_ih1, b_ih1, W_hh1, b_hh1, W1, b1, W2, b2):
        _f32c(t)
    nws = int(_lib.lib().inet_arnn_generate_ws_floats(L, E, Hc, H, U, V))
    ws = torch.empty(nws, dtype=torch.float32, device=emb.device)
    tokens = torch.empty(L, dtype=torch.int64, device=emb.device)
    check(_lib.lib().inet_arnn_generate(L, E, Hc, H, U, V, ptr(emb), ptr(oc0), oc0.stride(0), ptr(W_ih0), ptr(b_ih0), ptr(W_hh0),
                                        ptr(b_hh0), ptr(W_ih1), ptr(b_ih1), ptr(W_hh1), ptr(b_hh1), ptr(W1), ptr(b1), ptr(W2),
                                        ptr(b2), ptr(_f32c(hc_init) if hc_init is not None else None),
                                        ptr(_i64c(first_tok) if first_tok is not None else None), ptr(tokens), ptr(ws), nws,
                                        stream_ptr()), "inet_arnn_generate")
    _hold(ws, oc0, hc_init, first_tok)
    if _ARNN_KEEP_WS:                                # diagnostics (tools/arnn_token_pass.py reads the kernel's phase stamps out of it)
        _ARNN_KEEP_WS[:] = [ws]
    return tokens


_ARNN_KEEP_WS = []


_twin = {}


def twin_stream(device=None):
    """The library's second compute stream as a torch stream (inet_twin_stream).  Work that should run beside the caller's stream
    goes HERE, not on a new torch.cuda.Stream(): a fifth busy stream in the process gets every stream time-sliced."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _twin:
        h = C.c_void_p()
        check(_lib.lib().inet_twin_stream(C.byref(h)), "inet_twin_stream")
        _twin[key] = torch.cuda.ExternalStream(h.value, device=dev)
    return _twin[key]


def _hold(*tensors):
    if _DEFER:
        _HELD.append(tensors)


def encoder_bwd(cfg, tokens, params, grads, mask, dmu, dls, ws, stage=0):
    """stage 0: the whole pass; 1 then 2: heads + layer 1, then layer 0 + embedding (include/inpaintnet_hip.h)."""
    B = tokens.shape[0]
    _f32c(dmu); _f32c(dls); _f32c(grads)
    _hold(tokens, params, grads, mask, dmu, dls, ws)
    check(_lib.lib().inet_vae_encoder_bwd(C.byref(cfg), B, ptr(tokens), ptr(params), ptr(grads), ptr(mask), ptr(dmu),
                                          ptr(dls), ptr(ws), ws.numel() * 4, int(stage), stream_ptr()),
          "inet_vae_encoder_bwd")


# ----------------------------------------------------------------------------- decoder
def decoder_ws(cfg, B, save, device):
    n = _lib.lib().inet_vae_decoder_ws_bytes(C.byref(cfg), B, int(save))
    if n < 0:
        raise ValueError("decoder_ws: invalid arguments")
    return _ws(n, device)


def decoder_fwd(cfg, z, target, teacher_forced, params, mask_beat=None, mask_tick=None, save=False, ws=None,
                multinomial_seed=0):
    """z [B,Z] -> weights [B,T,V], samples [B,1,T] int64, ws.  multinomial_seed != 0: the fed-back tokens are drawn from
    softmax(weights) (decoder.py:506-509) instead of the argmax."""
    _f32c(z); _f32c(params)
    B = z.shape[0]
    T = cfg.beats * cfg.ticks_per_beat
    if target is not None:
        _i64c(target)
    if ws is None:
        ws = decoder_ws(cfg, B, save, z.device)
    weights = torch.empty(B, T, cfg.num_notes, dtype=torch.float32, device=z.device)
    samples = torch.empty(B, 1, T, dtype=torch.int64, device=z.device)
    check(_lib.lib().inet_vae_decoder_fwd(C.byref(cfg), B, ptr(z), ptr(target), int(bool(teacher_forced)), ptr(params),
                                          ptr(mask_beat), ptr(mask_tick), ptr(weights), ptr(samples), ptr(ws),
                                          ws.numel() * 4, int(save), int(multinomial_seed) & (2 ** 64 - 1), stream_ptr()),
          "inet_vae_decoder_fwd")
    return weights, samples, ws


def decoder_bwd(cfg, dweights, weights, samples, params, grads, mask_beat, mask_tick, ws, need_dz=True):
    B = weights.shape[0]
    _f32c(dweights); _f32c(weights); _i64c(samples)
    dz = torch.empty(B, cfg.z_dim, dtype=torch.float32, device=weights.device) if need_dz else None
    _hold(dweights, weights, samples, params, grads, mask_beat, mask_tick, ws)
    check(_lib.lib().inet_vae_decoder_bwd(C.byref(cfg), B, ptr(dweights), ptr(weights), ptr(samples), ptr(params),
                                          ptr(grads), ptr(mask_beat), ptr(mask_tick), ptr(dz), ptr(ws), ws.numel() * 4,
                                          stream_ptr()), "inet_vae_decoder_bwd")
    return dz


# ----------------------------------------------------------------------------- losses
def cross_entropy(weights2d, targets1d, out2, dW=None, scale=1.0, out_scale=1.0):
    """weights2d [rows,V] (row stride = stride(0)); out2: 2-float accumulator (loss_sum, correct), both times out_scale."""
    rows, V = weights2d.shape
    assert weights2d.stride(1) == 1
    _i64c(targets1d)
    check(_lib.lib().inet_cross_entropy(ptr(weights2d), weights2d.stride(0), rows, V, ptr(targets1d), ptr(dW),
                                        dW.stride(0) if dW is not None else 0, float(scale), float(out_scale),
                                        ptr(out2[0:1]), ptr(out2[1:2]), stream_ptr()), "inet_cross_entropy")


def cross_entropy_ex(weights2d, targets1d, loss_sum=None, correct=None, dW=None, scale=1.0, scale_dev=None, out_scale=1.0,
                     add_term=None, add_scale=0.0, fwd_out=None, fwd_scale=0.0):
    """inet_cross_entropy_ex: cross_entropy() with nullable outputs, a device-scalar factor on dW and an extra term added to
    the loss (loss_sum / correct / add_term / scale_dev: 1-element float tensors or None)."""
    rows, V = weights2d.shape
    assert weights2d.stride(1) == 1
    _i64c(targets1d)
    check(_lib.lib().inet_cross_entropy_ex(ptr(weights2d), weights2d.stride(0), rows, V, ptr(targets1d), ptr(dW),
                                           dW.stride(0) if dW is not None else 0, float(scale), ptr(scale_dev),
                                           float(out_scale), ptr(loss_sum), ptr(correct), ptr(add_term), float(add_scale),
                                           ptr(fwd_out), float(fwd_scale), stream_ptr()), "inet_cross_entropy_ex")


def sample_multinomial(weights2d, seed, offset=0):
    """One draw per row from softmax(weights2d[row]) (decoder.py:506-509), counter-based generator (seed, offset + row)."""
    rows, V = weights2d.shape
    assert weights2d.stride(1) == 1 and weights2d.dtype == torch.float32
    out = torch.empty(rows, dtype=torch.int64, device=weights2d.device)
    check(_lib.lib().inet_sample_multinomial(ptr(weights2d), weights2d.stride(0), rows, V, ptr(out), 1,
                                             int(seed) & (2 ** 64 - 1), int(offset), stream_ptr()), "inet_sample_multinomial")
    return out


def reparam_kl(mu, ls, eps, kl_sum=None, want_sigma=False):
    _f32c(mu); _f32c(ls)
    z = torch.empty_like(mu)
    sigma = torch.empty_like(mu) if want_sigma else None
    check(_lib.lib().inet_reparam_kl(ptr(mu), ptr(ls), ptr(eps), ptr(z), ptr(sigma), mu.numel(), ptr(kl_sum),
                                     stream_ptr()), "inet_reparam_kl")
    return z, sigma


def latent_bwd(dz, mu, ls, eps, kscale, kscale_dev=None):
    """Gradients of (mu, logsigma) from dz (nullable) and from the KL sum, whose upstream gradient is
    kscale * kscale_dev[0] (kscale_dev: optional device scalar)."""
    dmu = torch.empty_like(mu)
    dls = torch.empty_like(mu)
    _hold(kscale_dev)
    check(_lib.lib().inet_latent_bwd(ptr(dz), ptr(mu), ptr(ls), ptr(eps), float(kscale), ptr(kscale_dev), ptr(dmu),
                                     ptr(dls), mu.numel(), stream_ptr()), "inet_latent_bwd")
    return dmu, dls


def adam_step(p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8, gscale=1.0, step_flag=None, report=None):
    """report: 4 int32 words of PINNED host memory, zeroed by the caller; the kernel sets [0] executed, [1] skipped, [2] a
    parameter became non-finite.  step_flag: 1-element device tensor that alone decides whether the step is applied (the ranks'
    summed chain status, include/inpaintnet_hip.h inet_adam_step_ex)."""
    for t in (p, g, m, v):
        _f32c(t)
    if report is None and step_flag is None:
        check(_lib.lib().inet_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(b1), float(b2),
                                        float(eps), int(step), float(gscale), stream_ptr()), "inet_adam_step")
    else:
        if report is not None:
            assert report.dtype == torch.int32 and report.numel() >= 4 and report.is_pinned() and report.is_contiguous()
        check(_lib.lib().inet_adam_step_ex(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(b1), float(b2),
                                           float(eps), int(step), float(gscale), ptr(step_flag), ptr(report), stream_ptr()),
              "inet_adam_step_ex")


def step_flag_export(dst):
    """dst[0] = 1.0 if a chain launch of this process has timed out since the last reset else 0.0 (on the current stream)."""
    check(_lib.lib().inet_step_flag_export(ptr(_f32c(dst)), stream_ptr()), "inet_step_flag_export")


def token_status(reset=False):
    """Prologue launches that saw a token index outside [0, num_notes) since the last reset (host-mapped counter)."""
    return int(_lib.lib().inet_token_status(int(bool(reset))))


class TokenRangeError(ValueError):
    """decoder.py:36-45 check_index: a token index outside the vocabulary reached the model."""


def check_tokens(what=""):
    if token_status() > 0:
        n = token_status(reset=True)
        raise TokenRangeError(f"Invalid Value of index: {n} launch(es) met a token outside [0, num_notes) "
                              f"({what or 'inet_token_status'}); the results computed from it are not valid")


def dropout_mask(shape, p, seed, offset, device):
    out = torch.empty(shape, dtype=torch.float32, device=device)
    check(_lib.lib().inet_dropout_mask(ptr(out), out.numel(), float(p), int(seed) & (2 ** 64 - 1),
                                       int(offset) & (2 ** 64 - 1), stream_ptr()), "inet_dropout_mask")
    return out


# ----------------------------------------------------------------------------- measurement hooks
PROF_CLASSES = ("gemm_f32_mfma", "gru_step_fwd", "gru_step_bwd", "hbm_pointwise")


def prof_enable(on):
    check(_lib.lib().inet_prof_enable(int(bool(on))), "inet_prof_enable")


def set_option(key, value):
    """inet_set_option (include/inpaintnet_hip.h): 0 side stream, 1 deferred joins, 2 GEMM tile force, 3 GEMM split force, 4 chain kernels,
    5 direct k-major GEMM (0 never / 1 cost model / 2 always)."""
    check(_lib.lib().inet_set_option(int(key), int(value)), "inet_set_option")


def prof_dump(path):
    """Per-launch CSV (class,label,us,gflop) of the launches since prof_enable(True)."""
    check(_lib.lib().inet_prof_dump(str(path).encode()), "inet_prof_dump")


def prof_read():
    """{class: (launches, total_ms, total_flops)} for the launches since prof_enable(True)."""
    out = {}
    for i, name in enumerate(PROF_CLASSES):
        n = C.c_int64()
        ms = C.c_double()
        fl = C.c_double()
        check(_lib.lib().inet_prof_read(i, C.byref(n), C.byref(ms), C.byref(fl)), "inet_prof_read")
        out[name] = (n.value, ms.value, fl.value)
    return out


# ----------------------------------------------------------------------------- generic ops
def gemm(A, B, M, N, K, a_kmajor=False, b_kmajor=False, bias=None, epi=0, aux=None, out=None, accumulate=False):
    """C[M,N] (op)= epi(sum_k A(m,k) B(n,k) + bias).  A, B are 2-D (possibly row-strided) fp32 tensors."""
    assert A.stride(1) == 1 and B.stride(1) == 1
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    check(_lib.lib().inet_gemm(ptr(A), A.stride(0), int(a_kmajor), ptr(B), B.stride(0), int(b_kmajor), ptr(out),
                               out.stride(0), M, N, K, ptr(bias), ptr(aux), aux.stride(0) if aux is not None else 0,
                               int(epi), int(accumulate), stream_ptr()), "inet_gemm")
    return out


def epoch_stats_add(sums, loss, accuracy=None, step_flag=None):
    """sums[:3] += (loss, accuracy, 1) on the device unless the step was skipped (step_flag: the ranks' summed chain status as
    given to adam_step; None: this process's own status word) -- inet_epoch_stats_add_ex."""
    check(_lib.lib().inet_epoch_stats_add_ex(ptr(sums), ptr(loss), ptr(accuracy), ptr(step_flag), stream_ptr()),
          "inet_epoch_stats_add_ex")


def gemm_bf3(A, B, M, N, K, a_kmajor=False, b_kmajor=False, bias=None, out=None, accumulate=False, ksplit=0):
    """The same product through exact three-piece bf16 splits on the bf16 matrix cores (csrc/gemm_bf3.hip); test / bench
    entry: the pieces are made in a scratch allocated for the call."""
    assert A.stride(1) == 1 and B.stride(1) == 1
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    check(_lib.lib().inet_gemm_bf3(ptr(A), A.stride(0), int(a_kmajor), ptr(B), B.stride(0), int(b_kmajor), ptr(out),
                                   out.stride(0), M, N, K, ptr(bias), int(accumulate), int(ksplit), stream_ptr()),
          "inet_gemm_bf3")
    return out


def gemm_batched(A, B, out, M, N, K, nbatch, batchA, batchB, batchC, a_kmajor=True, b_kmajor=True):
    """out_i[M,N] += A_i . B_i^T for nbatch problems of one shape (element strides between problems), one launch where
    a batched kernel applies.  A, B, out: the 2-D views of problem 0."""
    assert A.stride(1) == 1 and B.stride(1) == 1 and out.stride(1) == 1
    check(_lib.lib().inet_gemm_batched(ptr(A), A.stride(0), int(a_kmajor), ptr(B), B.stride(0), int(b_kmajor), ptr(out),
                                       out.stride(0), M, N, K, int(nbatch), int(batchA), int(batchB), int(batchC),
                                       stream_ptr()), "inet_gemm_batched")
    return out


def linear_fwd(x, W, b, epi=0):
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    check(_lib.lib().inet_linear_fwd(ptr(_f32c(x)), ptr(_f32c(W)), ptr(b), ptr(y), M, N, K, int(epi), stream_ptr()),
          "inet_linear_fwd")
    return y


def linear_bwd(dy, x, W, dW=None, db=None, need_dx=True):
    M, N = dy.shape
    K = W.shape[1]
    dx = torch.empty(M, K, dtype=torch.float32, device=dy.device) if need_dx else None
    _hold(dy, x, W, dW, db)                                  # dW / db run on the side stream
    check(_lib.lib().inet_linear_bwd(ptr(_f32c(dy)), ptr(x), ptr(W), ptr(dx), ptr(dW), ptr(db), M, N, K,
                                     stream_ptr()), "inet_linear_bwd")
    return dx


def lstm_ws(B, T, H, save, device):
    n = _lib.lib().inet_lstm_ws_bytes(B, T, H, int(save))
    if n < 0:
        raise ValueError("lstm_ws: invalid arguments")
    return _ws(n, device)


def lstm_fwd(gi_tm, W_hh, b_hh, H, reverse=False, h0=None, c0=None, save=False, want_state=False):
    """gi_tm [T,B,4H] -> out [T,B,H] (+ hT, cT), ws"""
    T, B, _ = gi_tm.shape
    dev = gi_tm.device
    ws = lstm_ws(B, T, H, save, dev)
    out = torch.empty(T, B, H, dtype=torch.float32, device=dev)
    hT = torch.empty(B, H, dtype=torch.float32, device=dev) if want_state else None
    cT = torch.empty(B, H, dtype=torch.float32, device=dev) if want_state else None
    check(_lib.lib().inet_lstm_fwd(B, T, H, ptr(_f32c(gi_tm)), ptr(W_hh), ptr(b_hh), ptr(h0), ptr(c0), int(reverse),
                                   ptr(out), ptr(hT), ptr(cT), ptr(ws), ws.numel() * 4, int(save), stream_ptr()),
          "inet_lstm_fwd")
    return out, hT, cT, ws


def lstm_bwd(W_hh, out, dout, H, reverse, ws, dW_hh=None, db_ih=None, db_hh=None, h0=None, dhT=None, dcT=None,
             want_dstate=False):
    T, B, _ = out.shape
    dev = out.device
    dgi = torch.empty(T, B, 4 * H, dtype=torch.float32, device=dev)
    dh0 = torch.empty(B, H, dtype=torch.float32, device=dev) if want_dstate else None
    dc0 = torch.empty(B, H, dtype=torch.float32, device=dev) if want_dstate else None
    _hold(W_hh, h0, out, dout, dhT, dcT, dgi, dW_hh, db_ih, db_hh, ws)
    check(_lib.lib().inet_lstm_bwd(B, T, H, ptr(W_hh), ptr(h0), ptr(out), ptr(dout), ptr(dhT), ptr(dcT), int(reverse),
                                   ptr(dgi), ptr(dW_hh), ptr(db_ih), ptr(db_hh), ptr(dh0), ptr(dc0), ptr(ws),
                                   ws.numel() * 4, stream_ptr()), "inet_lstm_bwd")
    return dgi, dh0, dc0


def lstm2_ok(B, T, H):
    return bool(_lib.lib().inet_lstm2_ok(int(B), int(T), int(H)))


def lstm2_fwd(gi0_tm, W_hh0, b_hh0, W_ih1, b_ih1, W_hh1, b_hh1, H, reverse=False, save=False):
    """Two stacked layers, zero initial states, pipelined over chunks of time steps (csrc/lstm.hip lstm2_seq_fwd).
    gi0_tm [T,B,4H] -> (out0, out1 [T,B,H], ws0, ws1), or None when the shape does not qualify."""
    T, B, _ = gi0_tm.shape
    dev = gi0_tm.device
    ws0, ws1 = lstm_ws(B, T, H, save, dev), lstm_ws(B, T, H, save, dev)
    out0 = torch.empty(T, B, H, dtype=torch.float32, device=dev)
    out1 = torch.empty(T, B, H, dtype=torch.float32, device=dev)
    gi1 = torch.empty(T, B, 4 * H, dtype=torch.float32, device=dev)
    _hold(gi0_tm, gi1, out0, out1, ws0, ws1)                 # the second stream's work is not ordered by the allocator
    rc = _lib.lib().inet_lstm2_fwd(B, T, H, ptr(_f32c(gi0_tm)), ptr(W_hh0), ptr(b_hh0), ptr(W_ih1), ptr(b_ih1), ptr(W_hh1),
                                   ptr(b_hh1), int(reverse), ptr(out0), ptr(gi1), ptr(out1), ptr(ws0), ptr(ws1),
                                   ws0.numel() * 4, int(save), stream_ptr())
    if rc == 1:
        return None
    check(rc, "inet_lstm2_fwd")
    return out0, out1, ws0, ws1


def lstm2_bwd(W_hh0, W_ih1, W_hh1, out0, out1, dout1, H, reverse, ws0, ws1, grads=None):
    """-> dgi0 [T,B,4H] (dgi1 is consumed inside).  grads: (dW_hh0, db_ih0, db_hh0, dW_ih1, dW_hh1, db_ih1, db_hh1) or None."""
    T, B, _ = out1.shape
    dev = out1.device
    dgi0 = torch.empty(T, B, 4 * H, dtype=torch.float32, device=dev)
    dgi1 = torch.empty(T, B, 4 * H, dtype=torch.float32, device=dev)
    dout0 = torch.empty(T, B, H, dtype=torch.float32, device=dev)
    g = tuple(grads) if grads is not None else (None,) * 7
    _hold(W_hh0, W_ih1, W_hh1, out0, out1, dout1, dgi0, dgi1, dout0, ws0, ws1, *g)
    check(_lib.lib().inet_lstm2_bwd(B, T, H, ptr(W_hh0), ptr(W_ih1), ptr(W_hh1), ptr(out0), ptr(out1), ptr(_f32c(dout1)),
                                    int(reverse), ptr(dgi0), ptr(dgi1), ptr(dout0), *[ptr(x) for x in g], ptr(ws0), ptr(ws1),
                                    ws0.numel() * 4, stream_ptr()), "inet_lstm2_bwd")
    return dgi0


def embedding_fwd(table, idx, row_scale=None):
    rows = idx.numel()
    E = table.shape[1]
    out = torch.empty(rows, E, dtype=torch.float32, device=table.device)
    check(_lib.lib().inet_embedding_fwd(ptr(table), ptr(_i64c(idx)), rows, E, ptr(out), ptr(row_scale), stream_ptr()),
          "inet_embedding_fwd")
    return out


def embedding_bwd(dout, idx, dtable, row_scale=None):
    rows = idx.numel()
    _hold(dout, idx, dtable, row_scale)                      # runs on the side stream
    check(_lib.lib().inet_embedding_bwd(ptr(_f32c(dout)), ptr(_i64c(idx)), rows, dtable.shape[1], ptr(dtable),
                                        ptr(row_scale), int(dtable.shape[0]), stream_ptr()), "inet_embedding_bwd")


def relu_bwd(dy, y):
    out = torch.empty_like(dy)
    check(_lib.lib().inet_relu_bwd(ptr(_f32c(dy)), ptr(_f32c(y)), ptr(out), dy.numel(), stream_ptr()), "inet_relu_bwd")
    return out


def argmax_rows(w2d, out=None):
    rows, V = w2d.shape
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=w2d.device)
    check(_lib.lib().inet_argmax(ptr(w2d), w2d.stride(0), rows, V, ptr(out), 1, stream_ptr()), "inet_argmax")
    return out


def gru_step(gi, h_prev, W_hh, b_hh, save=False):
    B, H = h_prev.shape
    h_new = torch.empty_like(h_prev)
    sv = torch.empty(5, B, H, dtype=torch.float32, device=gi.device) if save else None
    check(_lib.lib().inet_gru_step(B, H, ptr(_f32c(gi)), ptr(_f32c(h_prev)), ptr(_f32c(W_hh)), ptr(_f32c(b_hh)),
                                   ptr(h_new), ptr(sv), stream_ptr()), "inet_gru_step")
    return h_new, sv


def bigru2_ws(B, T, K, H, save, device):
    n = _lib.lib().inet_bigru2_ws_bytes(B, T, K, H, int(save))
    if n < 0:
        raise ValueError("bigru2_ws: invalid arguments")
    return _ws(n, device)


def bigru2_fwd(x, x_scalar, weights, H, B, T, K, h0=None, mask=None, want_out=True, want_hn=True, save=False,
               ws=None):
    """x [B,T,K] or None (then x_scalar: 1-element tensor, K == 1).  weights: view of the arena starting at
    the GRU's weight_ih_l0.  Returns out [B,T,2H] | None, h_n [4,B,H] | None, ws."""
    dev = weights.device
    if ws is None:
        ws = bigru2_ws(B, T, K, H, save, dev)
    out = torch.empty(B, T, 2 * H, dtype=torch.float32, device=dev) if want_out else None
    hn = torch.empty(4, B, H, dtype=torch.float32, device=dev) if want_hn else None
    check(_lib.lib().inet_bigru2_fwd(B, T, K, H, ptr(x), ptr(x_scalar), ptr(weights), ptr(h0), ptr(mask), ptr(out),
                                     ptr(hn), ptr(ws), ws.numel() * 4, int(save), stream_ptr()), "inet_bigru2_fwd")
    return out, hn, ws


def bigru2_bwd(x, x_scalar, weights, grads, H, B, T, K, mask, dout, dhn, ws, want_dx=False, dx_scalar=None,
               want_dh0=False):
    dev = weights.device
    dx = torch.empty(B, T, K, dtype=torch.float32, device=dev) if (want_dx and x is not None) else None
    dh0 = torch.empty(4, B, H, dtype=torch.float32, device=dev) if want_dh0 else None
    _hold(x, x_scalar, weights, grads, mask, dout, dhn, ws)
    check(_lib.lib().inet_bigru2_bwd(B, T, K, H, ptr(x), ptr(x_scalar), ptr(weights), ptr(grads), ptr(mask),
                                     ptr(dout), ptr(dhn), ptr(dx), ptr(dx_scalar), ptr(dh0), ptr(ws), ws.numel() * 4,
                                     stream_ptr()), "inet_bigru2_bwd")
    return dx, dh0


# ----------------------------------------------------------------------------- input feed
def tokens_to_long(t):
    """int32 device tensor -> int64 of the same shape (utils/helpers.py:17-26 on the device)."""
    assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()
    out = torch.empty(t.shape, dtype=torch.int64, device=t.device)
    check(_lib.lib().inet_tokens_to_i64(ptr(t), ptr(out), t.numel(), stream_ptr()), "inet_tokens_to_i64")
    return out


def split_score(score, n_past, n_target, measure_len):
    """score (B,1,L) or (B,L) int32 on the device -> past, future, target int64 (B,n,measure_len), one kernel."""
    assert score.is_cuda and score.dtype == torch.int32 and score.is_contiguous()
    B = score.shape[0]
    L = score.numel() // B
    M = L // measure_len
    n_future = M - n_past - n_target
    mk = lambda n: torch.empty(B, n, measure_len, dtype=torch.int64, device=score.device)
    past, target, future = mk(n_past), mk(n_target), mk(n_future)
    check(_lib.lib().inet_split_score(ptr(score), B, M, measure_len, n_past, n_target, ptr(past), ptr(target),
                                      ptr(future), stream_ptr()), "inet_split_score")
    return past, future, target


def ws_field(cfg, ws, B, which, name):
    """Test hook: view of a named intermediate (flat) inside an encoder (which=0) / decoder (which=1) workspace."""
    off, n = C.c_int64(), C.c_int64()
    check(_lib.lib().inet_vae_ws_field(C.byref(cfg), B, int(which), name.encode(), C.byref(off), C.byref(n)),
          "inet_vae_ws_field")
    return ws[off.value:off.value + n.value]


def chain_status(reset=False):
    """Workgroups of chain kernels that timed out waiting for their group since the last reset (0 = healthy).
    Complete after a synchronisation; a non-zero value read earlier is already a failure (host-mapped counter)."""
    return int(_lib.lib().inet_chain_status(int(bool(reset))))


class ChainTimeoutError(RuntimeError):
    """A persistent (chain) kernel gave up waiting for its group: results computed since are not valid."""


_KERNEL_NAMES = {1: "gru_chain_fwd", 2: "gru_chain_bwd", 3: "gru_chain2_fwd", 4: "gru_chain2_bwd", 5: "lstm_chain_fwd", 6: "lstm_chain_bwd",
                 7: "lstm_pipeline", 8: "decode_chain", 9: "arnn_token_pass", 10: "decode_b1"}
_SITE_NAMES = {0: "group counter", 1: "granule", 2: "row-block counter", 3: "tagged fragments"}


def slow_waits(reset=False, max_entries=127):
    """The library's slow-wait recorder (include/inpaintnet_hip.h inet_slow_waits).  Returns {"count": waits of at least the entry
    threshold (default 16384 polls, set_option(16, polls)) or that gave up, since the last reset; "noted": waits of at least 64 polls
    (normal wherever launches overlap); "entries": the first 127 slow ones, decoded}.  Synchronises the device.  A slow wait is what
    an unexplained timeout or a stalled launch leaves behind: which kernel, which workgroup on which XCD, what it waited for and
    for how many polls (a counter poll is ~0.4 us, a granule poll ~1 us)."""
    import numpy as np
    buf = np.zeros((max(int(max_entries), 1), 8), dtype=np.uint32)
    noted = C.c_int64(0)
    n = int(_lib.lib().inet_slow_waits(C.c_void_p(buf.ctypes.data), int(max_entries), int(bool(reset)), C.byref(noted)))
    if n < 0:
        return {"count": n, "noted": 0, "entries": []}
    out = []
    for e in buf[:min(n, int(max_entries), 127)]:
        w0 = int(e[0])
        out.append({"kernel": _KERNEL_NAMES.get(w0 & 0xff, str(w0 & 0xff)), "xcc": (w0 >> 8) & 0xf, "gave_up": bool(w0 & 0x8000),
                    "site": _SITE_NAMES.get(w0 >> 16, str(w0 >> 16)), "workgroup": int(e[1]), "expected": int(e[2]), "polls": int(e[3]),
                    "clock_10ns": int(e[4]) | (int(e[5]) << 32)})
    return {"count": n, "noted": int(noted.value), "entries": out}


def slow_waits_summary(reset=False, top=6):
    """One line for error messages and logs: the count and the longest few waits of the recorder."""
    r = slow_waits(reset=reset)
    if r["count"] <= 0:
        return f"no slow waits recorded ({r['count']}; {r['noted']} waits of 64+ polls noted)"
    es = sorted(r["entries"], key=lambda e: -e["polls"])[:top]
    return f"{r['count']} slow waits ({r['noted']} of 64+ polls noted); longest: " + "; ".join(
        f"{e['kernel']} wg {e['workgroup']} xcc {e['xcc']} {e['site']} expected {e['expected']} after {e['polls']} polls"
        + (" GAVE UP" if e["gave_up"] else "") for e in es)


def preload():
    """First-touch of every kernel of the library on the current device (csrc/preload.hip): code objects and function objects the HIP
    runtime would otherwise build on the launch path of each kernel's first launch.  Idempotent per device; the trainers and the
    models call it at construction so that no step pays a first launch.  Returns the number of kernels touched (0: done before)."""
    n = int(_lib.lib().inet_preload())
    if n < 0:
        raise _lib.InetError("inet_preload: no HIP device, or a kernel of the library could not be loaded on it")
    return n


def check_chains(what=""):
    """Raise if any chain-kernel workgroup has timed out (reads the host-mapped counter: no synchronisation, ~1 us).
    The inference wrappers call it after their device->host reads (Trainer.step() reads step reports instead: check_steps).
    Also raises TokenRangeError (a ValueError) if a prologue met a token outside the vocabulary (decoder.py:36-45)."""
    n = chain_status()
    if n > 0:
        raise ChainTimeoutError(f"{n} chain-kernel workgroups gave up waiting for their group ({what or 'inet_chain_status'}): "
                                "the results are not valid.  All workgroups of such a launch must be resident at once -- "
                                "is the GPU shared, partitioned or CU-masked?  INET_CHAIN=0 (or ops.set_option(4, 0)) "
                                f"selects the per-step kernels.  Recorder: {slow_waits_summary()}")
    check_tokens(what)
