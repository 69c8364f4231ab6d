"""ctypes binding of libinpaintnet_hip.so (include/inpaintnet_hip.h).

The product path has NO fallback: if the HIP library is missing or a call
fails, this module raises.  Build with `python -c 'import __graft_entry__ as g; g.build()'`
(or inpaintnet_amd._lib.build()).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INET_LIB_PATH") or os.path.join(_HERE, "libinpaintnet_hip.so")
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["preload.hip", "arnn_gen.hip", "decode_b1.hip", "gemm.hip", "gru.hip", "pointwise.hip", "seq.hip", "vae.hip", "api.hip", "prof.hip", "side.hip", "lstm.hip", "gru_chain.hip", "decode_chain.hip", "gru_chain2.hip", "gemm_bf3.hip", "gru_step_bf3.hip"]

_lib = None


class VaeConfig(C.Structure):
    _fields_ = [("num_notes", C.c_int32), ("emb_dim", C.c_int32), ("enc_hidden", C.c_int32),
                ("z_dim", C.c_int32), ("dec_hidden", C.c_int32), ("beats", C.c_int32),
                ("ticks_per_beat", C.c_int32)]


class LatentConfig(C.Structure):
    _fields_ = [("z_dim", C.c_int32), ("rnn_hidden", C.c_int32), ("auto_reg", C.c_int32)]


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into the in-tree shared library: one object per source (in parallel,
    only the stale ones), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, f) for f in SOURCES]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(_HERE), "include", "inpaintnet_hip.h"))
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    objdir = os.path.join(os.path.dirname(_HERE), "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"]
    objs, stale = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            stale.append((src, obj))
    if not stale and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def compile_one(job):
        cmd = [hipcc] + flags + ["-c", job[0], "-o", job[1]]
        if verbose:
            print("[inpaintnet_amd] " + " ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(stale)))) as ex:
        list(ex.map(compile_one, stale))
    # -Bsymbolic-functions: the library's own constructors bind to ITS __hipRegisterFunction (csrc/preload.hip), which files every
    # kernel handle for inet_preload() and passes the call on to the HIP runtime
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic-functions", "-o", LIB_PATH] + objs + ["-ldl"]
    if verbose:
        print("[inpaintnet_amd] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_U = C.c_uint64
_CFG = C.POINTER(VaeConfig)
_LCFG = C.POINTER(LatentConfig)

_SIGNATURES = {
    "inet_abi_version": (C.c_int, []),
    "inet_vae_param_count": (C.c_int, [_CFG]),
    "inet_vae_param_floats": (_L, [_CFG]),
    "inet_vae_param_info": (C.c_int, [_CFG, _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_I)]),
    "inet_latent_param_count": (C.c_int, [_LCFG]),
    "inet_latent_param_floats": (_L, [_LCFG]),
    "inet_latent_param_info": (C.c_int, [_LCFG, _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_I)]),
    "inet_vae_encoder_ws_bytes": (_L, [_CFG, _I, _I]),
    "inet_vae_encoder_fwd": (C.c_int, [_CFG, _I, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "inet_vae_encoder_bwd": (C.c_int, [_CFG, _I, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "inet_vae_decoder_ws_bytes": (_L, [_CFG, _I, _I]),
    "inet_vae_decoder_fwd": (C.c_int, [_CFG, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, C.c_uint64, _P]),
    "inet_vae_decoder_bwd": (C.c_int, [_CFG, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "inet_vae_ws_field": (C.c_int, [_CFG, _I, _I, C.c_char_p, C.POINTER(_L), C.POINTER(_L)]),
    "inet_cross_entropy": (C.c_int, [_P, _L, _I, _I, _P, _P, _L, _F, _F, _P, _P, _P]),
    "inet_cross_entropy_ex": (C.c_int, [_P, _L, _I, _I, _P, _P, _L, _F, _P, _F, _P, _P, _P, _F, _P, _F, _P]),
    "inet_reparam_kl": (C.c_int, [_P, _P, _P, _P, _P, _L, _P, _P]),
    "inet_sample_multinomial": (C.c_int, [_P, _L, _I, _I, _P, _L, C.c_uint64, C.c_uint64, _P]),
    "inet_latent_bwd": (C.c_int, [_P, _P, _P, _P, _F, _P, _P, _P, _L, _P]),
    "inet_adam_step": (C.c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P]),
    "inet_adam_step_ex": (C.c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P, _P, _P]),
    "inet_step_flag_export": (C.c_int, [_P, _P]),
    "inet_token_status": (C.c_int, [_I]),
    "inet_epoch_stats_add_ex": (C.c_int, [_P, _P, _P, _P, _P]),
    "inet_dropout_mask": (C.c_int, [_P, _L, _F, _U, _U, _P]),
    "inet_bigru2_ws_bytes": (_L, [_I, _I, _I, _I, _I]),
    "inet_bigru2_fwd": (C.c_int, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    "inet_bigru2_bwd": (C.c_int, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "inet_gemm": (C.c_int, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _I, _I, _P, _P, _L, _I, _I, _P]),
    "inet_epoch_stats_add": (C.c_int, [_P, _P, _P, _P]),
    "inet_gemm_bf3": (C.c_int, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _I, _I, _P, _I, _I, _P]),
    "inet_gemm_batched": (C.c_int, [_P, _L, _I, _P, _L, _I, _P, _L, _I, _I, _I, _I, _L, _L, _L, _P]),
    "inet_gru_step": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "inet_linear_fwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "inet_linear_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "inet_lstm_ws_bytes": (_L, [_I, _I, _I, _I]),
    "inet_lstm_fwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _P]),
    "inet_lstm_bwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    "inet_lstm2_ok": (C.c_int, [_I, _I, _I]),
    "inet_lstm2_fwd": (C.c_int, [_I, _I, _I] + [_P] * 7 + [_I] + [_P] * 5 + [_L, _I, _P]),
    "inet_lstm2_bwd": (C.c_int, [_I, _I, _I] + [_P] * 6 + [_I] + [_P] * 12 + [_L, _P]),
    "inet_embedding_fwd": (C.c_int, [_P, _P, _L, _I, _P, _P, _P]),
    "inet_embedding_bwd": (C.c_int, [_P, _P, _L, _I, _P, _P, _I, _P]),
    "inet_relu_bwd": (C.c_int, [_P, _P, _P, _L, _P]),
    "inet_argmax": (C.c_int, [_P, _L, _I, _I, _P, _L, _P]),
    "inet_tokens_to_i64": (C.c_int, [_P, _P, _L, _P]),
    "inet_split_score": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "inet_set_option": (C.c_int, [_I, _I]),
    "inet_side_join": (C.c_int, [_P]),
    "inet_arnn_generate_ws_floats": (C.c_int64, [_I, _I, _I, _I, _I, _I]),
    "inet_arnn_generate": (C.c_int, [_I] * 6 + [_P, _P, _L] + [_P] * 12 + [_P, _P, _P, _P, _L, _P]),
    "inet_side_wait": (C.c_int, [_P]),
    "inet_twin_stream": (C.c_int, [C.POINTER(C.c_void_p)]),
    "inet_debug_read": (C.c_int, [_P, _L]),
    "inet_chain_status": (C.c_int, [_I]),
    "inet_slow_waits": (C.c_int, [_P, _I, _I, C.POINTER(_L)]),
    "inet_decode_b1_plan": (C.c_int, [_I, _I, _I, C.POINTER(C.c_int)]),
    "inet_preload": (C.c_int, []),
    "inet_kernel_count": (C.c_int, []),
    "inet_prof_enable": (C.c_int, [_I]),
    "inet_prof_dump": (C.c_int, [C.c_char_p]),
    "inet_prof_read": (C.c_int, [_I, C.POINTER(_L), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

EXPORTS = tuple(_SIGNATURES)


def lib():
    """The loaded library; raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension is not built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and the tensors this library is handed live in
        # that runtime's context.  Loaded first, libinpaintnet_hip.so would pull in /opt/rocm's copy as a SECOND runtime, and
        # its first launch then fails with hipErrorNoDevice (build() followed by smoke() in one process did exactly that).
        # With torch imported first, the library's NEEDED libamdhip64 resolves to the runtime already in the process.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.inet_abi_version() != 1:
            raise ImportError("libinpaintnet_hip.so ABI version mismatch")
        _lib = L
    return _lib


class InetError(RuntimeError):
    pass


def check(rc, what):
    if rc == -1:
        raise ValueError(f"{what}: invalid argument (rc=-1)")
    if rc == -3:
        raise InetError(f"{what}: library options (inet_set_option keys 4, 7, 8, 9, 12) changed between a forward call and its "
                        "backward call on the same workspace (rc=-3)")
    if rc == -4:
        raise InetError(f"{what}: the workspace was sized under other library options than this call runs under (inet_set_option "
                        "keys 4, 7, 8, 9, 12 changed between *_ws_bytes and the call: a piece buffer it needs is missing) (rc=-4)")
    if rc != 0:
        raise InetError(f"{what}: HIP launch/runtime failure (rc={rc})")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
