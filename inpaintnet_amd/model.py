"""Model base: flat parameter arena + the reference's save/load contract.

Mirrors utils/model.py:16-53 of the reference (save / save_checkpoint / load
with torch.save(state_dict)), with the state_dict keys and shapes of the
reference's modules so checkpoints are interchangeable.
Parameters are views into ONE flat fp32 tensor (`self.flat`); gradients live in
`self.grad` of the same layout.
"""
import math
import os
from collections import OrderedDict

import torch


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("inpaintnet_amd needs a ROCm GPU (MI355X): torch.cuda.is_available() is False. "
                           "There is no CPU fallback; the CPU oracle lives in oracle/ and is test-only.")
    return torch.device("cuda", torch.cuda.current_device())


class Model(torch.nn.Module):
    """Abstract model class (utils/model.py:5-14)."""

    def __init__(self):
        super().__init__()
        self.filepath = None
        self._table = None          # [(key, offset, shape)]
        self.flat = None
        self.grad = None
        self.requires_grad_flag = True

    # ---- arena ---------------------------------------------------------------------
    def _alloc_arena(self, table, total, device):
        self._table = list(table)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        # the gradient arena carries a tail of _STATS floats: per-step accumulators of the loss kernels (KL sum, loss,
        # accuracy) live there, so that zero_grad()'s ONE fill also zeroes them (each used to cost a torch.zeros launch)
        # ... and a head of _HEAD floats in front of it (16 bytes: the arena keeps its alignment): word 0 is the STEP FLAG, this
        # rank's chain status of the step (ops.step_flag_export), summed over ranks by the same all-reduce that carries the first
        # gradient range (dp.allreduce_grads) and read by the optimizer kernel -- every rank skips a step any rank failed in
        self._grad_store = torch.zeros(self._HEAD + total + self._STATS, dtype=torch.float32, device=device)
        self.grad = self._grad_store[self._HEAD:self._HEAD + total]
        self.step_flag = self._grad_store[0:2]
        self._stats = self._grad_store[self._HEAD + total:]
        self._stats_used = 0
        self._views = OrderedDict()
        self._gviews = OrderedDict()
        for name, off, shape in self._table:
            n = 1
            for s in shape:
                n *= s
            self._views[name] = self.flat[off:off + n].view(shape)
            self._gviews[name] = self.grad[off:off + n].view(shape)

    def arena_named_parameters(self, prefix=""):
        for k, v in self._views.items():
            yield prefix + k, v

    def param(self, name):
        return self._views[name]

    def param_grad(self, name):
        return self._gviews[name]

    def num_parameters(self):
        n = 0
        for _, _, shape in self._table:
            k = 1
            for s in shape:
                k *= s
            n += k
        return n

    @torch.no_grad()
    def init_reference_style(self, generator=None):
        """xavier_normal_ on every tensor whose name contains 'weight' (encoder.py:71-78,
        decoder.py:47-54, latent_rnn.py:291-307); torch defaults elsewhere: GRU biases
        U(-1/sqrt(H), 1/sqrt(H)), Linear biases U(-1/sqrt(fan_in), ..), b_0 / x_0 zeros."""
        lin_fan = {}
        for name, _, shape in self._table:
            if name.endswith(".weight") and len(shape) == 2:
                lin_fan[name[:-len("weight")] + "bias"] = shape[1]
        for name, _, shape in self._table:
            v = self._views[name]
            if "weight" in name and len(shape) == 2:
                fan_out, fan_in = shape
                std = math.sqrt(2.0 / (fan_in + fan_out))
                v.copy_(torch.randn(shape, generator=generator) * std)
            elif "bias_ih" in name or "bias_hh" in name:
                b = 1.0 / math.sqrt(shape[0] // 3)
                v.copy_((torch.rand(shape, generator=generator) * 2 - 1) * b)
            elif name in lin_fan:
                b = 1.0 / math.sqrt(lin_fan[name])
                v.copy_((torch.rand(shape, generator=generator) * 2 - 1) * b)
            elif name == "x_0" and len(shape) == 3:
                v.copy_(torch.randn(shape, generator=generator))       # latent_rnn.py:74
            else:
                v.zero_()

    # ---- state_dict contract -------------------------------------------------------
    def state_dict(self, *args, **kwargs):
        out = OrderedDict()
        for k, v in self.arena_named_parameters():
            out[k] = v.detach().clone()
        return out

    def load_state_dict(self, sd, strict=True):
        own = dict(self.arena_named_parameters())
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing}, unexpected {unexpected}")
        with torch.no_grad():
            for k, v in own.items():
                if k in sd:
                    t = torch.as_tensor(sd[k])
                    if tuple(t.shape) != tuple(v.shape):
                        raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {tuple(v.shape)}")
                    # MeasureVAE/encoder.py:111-116, decoder.py:424-429 scan the weights for NaN in EVERY forward pass and raise
                    # ValueError; here the scan happens where weights can become NaN: at load time (once, host tensors) and in
                    # the optimizer kernel (inet_adam_step_ex's step report, Trainer.check_steps)
                    if t.is_floating_point() and bool(torch.isnan(t).any()):
                        print(f'{type(self).__name__} has become nan')
                        raise ValueError(f"{type(self).__name__} has become nan: {k} holds NaN")
                    v.copy_(t.to(dtype=torch.float32))
        return missing, unexpected

    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        for k, v in self.arena_named_parameters():
            yield (prefix + "." if prefix else "") + k, v

    def parameters(self, recurse=True):
        for _, v in self.named_parameters():
            yield v

    def cuda(self, device=None):
        return self          # the arena is allocated on the GPU at construction

    def zero_grad(self, set_to_none=False):
        self._grad_store.zero_()
        self._stats_used = 0

    _STATS = 64
    _HEAD = 4

    def take_stats(self, n):
        """n zeroed floats for a kernel to accumulate into: a fresh slice of the arena's tail (zeroed by the last zero_grad();
        every call hands out another slice, so nothing is ever accumulated twice), torch.zeros once the tail is used up."""
        lo = self._stats_used
        if lo + n > self._STATS or not torch.is_grad_enabled():
            return torch.zeros(n, dtype=torch.float32, device=self.flat.device)
        self._stats_used = lo + n
        return self._stats[lo:lo + n]

    # ---- utils/model.py:16-53 ------------------------------------------------------
    def save(self):
        save_dir = os.path.dirname(self.filepath)
        if not os.path.exists(save_dir):
            os.makedirs(save_dir, exist_ok=True)
        torch.save({k: v.cpu() for k, v in self.state_dict().items()}, self.filepath)
        print(f'Model {self.__repr__()} saved')

    def save_checkpoint(self, epoch_num):
        save_dir = os.path.dirname(self.filepath)
        os.makedirs(save_dir, exist_ok=True)
        torch.save({k: v.cpu() for k, v in self.state_dict().items()}, self.filepath + '_' + str(epoch_num))
        print(f'Model checkpoint {self.__repr__()} saved for epoch')

    def load(self, cpu=False):
        sd = torch.load(self.filepath, map_location="cpu")
        self.load_state_dict(sd)
        print(f'Model {self.__repr__()} loaded')
