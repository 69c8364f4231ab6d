"""LatentRNN with the reference's Python surface, computed by the HIP library.

Mirrors LatentRNN/latent_rnn.py:11-307 of the reference: frozen MeasureVAE encodes the past / future /
target measures into latent samples, two 2-layer bi-GRU context encoders summarise past and future, their
final hiddens initialise a 2-layer bi-GRU generator (hidden 2x) whose outputs are mapped to latents and
decoded by the frozen VAE decoder.  Differences that are the point of the build:
  * the 16 measures of every sequence are encoded in ONE encoder call (batch B*16), and the generated
    measures are decoded in ONE decoder call (batch B*n_target) -- the reference loops (latent_rnn.py:131-133,
    237-240); the arithmetic per measure is identical;
  * every GRU / Linear is a single autograd Function over the C-ABI (inet_bigru2_*, inet_linear_*);
    the decoder runs dgrad-only (frozen parameters, latent_rnn.py:42-43).
"""
import os
import random

import torch

from . import dp, layout, ops
from ._lib import LatentConfig
from .measure_vae import MeasureVAE, _DropState, _next_mask_offset
from .model import Model


class _BiGru2Fn(ops.TrackedFunction):
    """2-layer bidirectional nn.GRU(batch_first).  x (B,T,K) or the scalar parameter x_0 (K == 1)."""

    @staticmethod
    def forward(ctx, x, x_scalar, h0, flat, owner, off, H, B, T, K, mask):
        need = ops.outer_grad() and any(ctx.needs_input_grad[:4])
        weights = flat[off:]
        xin = x.contiguous() if x is not None else None
        out, hn, ws = ops.bigru2_fwd(xin, x_scalar, weights, H, B, T, K, h0=h0.contiguous() if h0 is not None else None,
                                     mask=mask, save=need)
        ctx.args = (owner, off, H, B, T, K, mask, ws, xin, x_scalar)
        if need:                                   # applications of this module awaiting their backward (data-parallel buckets)
            owner.__dict__.setdefault("_dp_open", {})[off] = owner.__dict__.setdefault("_dp_open", {}).get(off, 0) + 1
        return out, hn

    @staticmethod
    def backward(ctx, dout, dhn):
        owner, off, H, B, T, K, mask, ws, xin, x_scalar = ctx.args
        ctx.args = None
        # the scalar input is the parameter x_0 (latent_rnn.py:74): its gradient accumulates straight into the arena
        dxs = owner.param_grad("x_0").view(1) if x_scalar is not None else None
        dx, dh0 = ops.bigru2_bwd(xin, x_scalar, owner.flat[off:], owner.grad[off:], H, B, T, K, mask,
                                 dout.contiguous(), dhn.contiguous(), ws,
                                 want_dx=ctx.needs_input_grad[0], dx_scalar=dxs, want_dh0=ctx.needs_input_grad[2])
        opened = owner.__dict__.setdefault("_dp_open", {})
        opened[off] = opened.get(off, 1) - 1
        # a module applied several times in one forward (the auto-regressive generator) is final after its LAST backward
        if dp.world_size() > 1 and opened[off] <= 0:
            if off == getattr(owner, "dp_bucket_from", -1):
                # data parallel: everything from the generator GRU to the end of the arena (generation_rnn +
                # generation_linear, 103 of the 160 MB) is final now -> its all-reduce runs under the context GRUs' backward
                dp.start_bucket(owner.grad, off, owner.grad.numel(), join_side=True)
            elif off in getattr(owner, "dp_module_ranges", {}):
                # a context GRU: its own 28 MB are final -> reduced under the other context's backward
                dp.start_bucket(owner.grad, off, owner.dp_module_ranges[off], join_side=True)
        return dx, None, dh0, None, None, None, None, None, None, None, None


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat, owner, w_name, b_name):
        W, b = owner.param(w_name), owner.param(b_name)
        ctx.args = (owner, w_name, b_name)
        ctx.save_for_backward(x)
        return ops.linear_fwd(x.contiguous(), W, b)

    @staticmethod
    def backward(ctx, dy):
        owner, w_name, b_name = ctx.args
        (x,) = ctx.saved_tensors
        dx = ops.linear_bwd(dy.contiguous(), x.contiguous(), owner.param(w_name), owner.param_grad(w_name),
                            owner.param_grad(b_name), need_dx=ctx.needs_input_grad[0])
        return dx, None, None, None, None


class LatentRNN(Model):
    def __init__(self, dataset, vae_model: MeasureVAE, num_rnn_layers, rnn_hidden_size, dropout, rnn_class,
                 auto_reg=False, teacher_forcing=True):
        super().__init__()
        if num_rnn_layers != 2:
            raise NotImplementedError("as in the reference (latent_rnn.py:77,140) only num_rnn_layers == 2 works")
        self.dataset = dataset.__repr__()
        self.vae_model = vae_model
        self.auto_reg = auto_reg
        self.use_teacher_forcing = teacher_forcing if self.auto_reg else False
        self.teacher_forcing_prob = 0.5
        self.vae_model.freeze()
        print('Freeze the ', self.vae_model.__repr__(), ' model.')
        self.num_rnn_layers = num_rnn_layers
        self.rnn_hidden_size = rnn_hidden_size
        self.dropout = dropout
        self.z_dim = self.vae_model.latent_space_dim
        self.rnn_class = rnn_class
        self.bidirectional = True
        self.rnn_num_direction = 2
        self.gen_rnn_input_dim = self.z_dim if self.auto_reg else 1
        self.trainable = True
        self.lcfg = LatentConfig(self.z_dim, rnn_hidden_size, int(bool(auto_reg)))
        # which contexts initialise the generator: both (LatentRNN) or one of them (LatentRNNAblations)
        # The reference encodes the target measures in every forward pass (latent_rnn.py:133) but reads the result only when the
        # auto-regressive generator is teacher-forced (:148-149): for auto_reg=False, for the free-running side of the coin and in
        # eval mode a quarter of the frozen encoder's work (4 of 16 measures at 6/4/6) is computed and dropped.  Outputs, gradients
        # and updated weights do not depend on it, so by default it is not computed (the coin is drawn first; the `random` stream
        # sees the same draws).  True = do the reference's work measure for measure (what bench.py's LatentRNN lines time).
        self.encode_unused_target = os.environ.get("INET_LATENT_ENCODE_ALL", "0") == "1"
        self.context_mode = getattr(self, "context_mode", "both")
        self.gen_hidden = 2 * rnn_hidden_size if self.context_mode == "both" else rnn_hidden_size
        if self.context_mode == "both":
            table, total = ops.latent_param_table(self.lcfg)
        else:
            offs, total = layout.arena_offsets(layout.latent_param_shapes(self.z_dim, rnn_hidden_size, bool(auto_reg),
                                                                          gen_hidden=self.gen_hidden))
            table = [(k, off, shp) for k, (off, shp) in offs.items()]
        self._alloc_arena(table, total, self.vae_model.flat.device)
        self._off = {name: off for name, off, _ in table}
        # arena order: x_0 | context_rnn_past | context_rnn_future | generation_rnn | generation_linear; backward visits
        # generation_linear, generation_rnn, then the contexts -> the tail starting here completes first
        self.dp_bucket_from = self._off["generation_rnn.weight_ih_l0"]
        assert all(off >= self.dp_bucket_from for name, off, _ in table if name.startswith("generation_"))
        assert all(off < self.dp_bucket_from for name, off, _ in table if not name.startswith("generation_"))
        # the context GRUs: [first parameter, first parameter of the next module) -- contiguous by construction of the arena
        self.dp_module_ranges = {}
        for prefix in ("context_rnn_past.", "context_rnn_future."):
            lo = min((off for name, off, _ in table if name.startswith(prefix)), default=None)
            if lo is not None:
                hi = min(off for name, off, _ in table if off > lo and not name.startswith(prefix))
                assert all(lo <= off < hi for name, off, _ in table if name.startswith(prefix))
                self.dp_module_ranges[lo] = hi
        self._flat_leaf = None
        self.init_reference_style()
        cur_dir = os.path.dirname(os.path.realpath(__file__))
        self.filepath = os.path.join(cur_dir, 'models/', self.__repr__())

    def __repr__(self):
        filestr = f'LatentRNN(' \
                  f'{self.dataset}' \
                  f'{self.rnn_class},' \
                  f'{self.num_rnn_layers},' \
                  f'{self.rnn_hidden_size},' \
                  f'{self.dropout},' \
                  f')'
        if self.auto_reg:
            filestr += 'auto_reg'
        filestr += ',tf' if self.use_teacher_forcing else ',no_tf'
        return filestr

    # state_dict carries the frozen VAE under 'vae_model.' like the reference's submodule (latent_rnn.py:35)
    def arena_named_parameters(self, prefix=""):
        for k, v in self._views.items():
            yield prefix + k, v
        for k, v in self.vae_model.arena_named_parameters(prefix + "vae_model."):
            yield k, v

    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        for k, v in self._views.items():
            yield k, v

    def flat_for_autograd(self):
        if self._flat_leaf is None:
            self._flat_leaf = self.flat.detach().requires_grad_(True)
        return self._flat_leaf

    # ------------------------------------------------------------------------------------------------
    def _mask(self, T, B, H):
        if self.training and self.dropout > 0:
            return ops.dropout_mask((T, B, 2 * H), self.dropout, _DropState.seed, _next_mask_offset(T * B * 2 * H),
                                    self.flat.device)
        return None

    def _bigru(self, name, x, x_scalar, h0, H, K, T=None):
        if x is not None:
            B, T, _ = x.shape
        else:
            B = h0.shape[1]
        return _BiGru2Fn.call(x, x_scalar, h0, self.flat_for_autograd(), self, self._off[name + ".weight_ih_l0"], H, B,
                               T, K, self._mask(T, B, H))

    def get_z_seq(self, measures_tensor, eps=None):
        """(B,n,24) -> z samples (B,n,Z) from the frozen encoder (latent_rnn.py:161-174)."""
        batch_size, num_measures, measure_seq_len = measures_tensor.size()
        with torch.no_grad():
            z_dist = self.vae_model.encoder(measures_tensor.reshape(-1, measure_seq_len).contiguous())
            z = z_dist.rsample(eps=eps.reshape(-1, self.z_dim) if eps is not None else None)
        return z.view(batch_size, -1, self.z_dim)

    def forward_context(self, z, type):
        """h_n (4,B,H) of the past / future context bi-GRU (latent_rnn.py:176-193)."""
        if type not in ("past", "future"):
            raise ValueError
        _, hn = self._bigru("context_rnn_" + type, z, None, None, self.rnn_hidden_size, self.z_dim)
        return hn

    def hidden_init(self, batch_size):
        return torch.zeros(self.num_rnn_layers * self.rnn_num_direction, batch_size, self.rnn_hidden_size,
                           device=self.flat.device)

    def _decode(self, z2d):
        """frozen decoder, train=False (latent_rnn.py:238): dropout still follows module.training (the quirk)."""
        dummy = torch.zeros(z2d.shape[0], self.vae_model.num_ticks_per_measure, device=z2d.device)
        return self.vae_model.decoder(z2d, dummy, train=False)

    def forward(self, past_context, future_context, target, measures_to_generate, train=True, eps=None,
                teacher_forcing=None, eps_ar=None):
        """-> weights (B,nt,24,V), samples (B,1,24*nt), gen_z (B,nt,Z)   (latent_rnn.py:110-159).
        eps: optional (eps_past (B,np,Z), eps_future, eps_target) injection; eps_ar: list of (B,Z) for the
        free-running auto-regressive path."""
        batch_size, _, measure_seq_len = past_context.size()
        n_past, n_future = past_context.size(1), future_context.size(1)
        n_target = target.size(1) if target is not None else 0        # inference: no target (latent_rnn_tester.py:231-236)
        # the teacher-forcing coin first (the reference draws it behind the context GRUs, :142-145: nothing else reads `random` in
        # between, so the stream is the same): whether the target measures' latents are needed depends on it
        if teacher_forcing is None:
            if self.use_teacher_forcing and train:
                teacher_forcing = random.random() < self.teacher_forcing_prob
            else:
                teacher_forcing = False
        if teacher_forcing and target is None:
            raise ValueError("teacher forcing needs the target measures")
        encode_target = target is not None and (teacher_forcing or self.encode_unused_target)
        # one encoder call over all measures of the sequence that are read
        parts = (past_context, target, future_context) if encode_target else (past_context, future_context)
        allm = torch.cat(parts, 1)
        e = None
        if eps is not None:
            es = [eps[0].view(batch_size, n_past, -1)]
            if encode_target:
                es.append(eps[2].view(batch_size, n_target, -1))
            es.append(eps[1].view(batch_size, n_future, -1))
            e = torch.cat(es, 1)
        z_all = self.get_z_seq(allm, e)
        n_enc_t = n_target if encode_target else 0
        zp = z_all[:, :n_past].contiguous()
        zt = z_all[:, n_past:n_past + n_enc_t].contiguous()
        zf = z_all[:, n_past + n_enc_t:].contiguous()
        if self.context_mode == "both":
            comb_context = torch.cat((self.forward_context(zp, type="past"), self.forward_context(zf, type="future")), 2)
        elif self.context_mode == "past":                          # latent_rnn_ablations.py:143-146
            comb_context = self.forward_context(zp, type="past")
        else:
            comb_context = self.forward_context(zf, type="future")
        if teacher_forcing:
            seed = torch.cat((zp[:, -1, :].unsqueeze(1), zt[:, :-1, :]), 1)
        else:
            seed = zp[:, -1, :].unsqueeze(1)
        return self.forward_generation(comb_context, measures_to_generate, seed, measure_seq_len, teacher_forcing,
                                       eps_ar=eps_ar)

    def forward_generation(self, context_vector, measures_to_gen, seed, measure_seq_len, teacher_forcing=False,
                           eps_ar=None):
        """latent_rnn.py:211-263"""
        batch_size = context_vector.size(1)
        Hg = self.gen_hidden
        if teacher_forcing or not self.auto_reg:
            if self.auto_reg:
                out, _ = self._bigru("generation_rnn", seed.contiguous(), None, context_vector, Hg, self.z_dim)
            else:
                out, _ = self._bigru("generation_rnn", None, self.param("x_0").view(1), context_vector, Hg, 1,
                                     T=measures_to_gen)
            z2d = _LinearFn.apply(out.reshape(batch_size * measures_to_gen, -1), self.flat_for_autograd(), self,
                                  "generation_linear.weight", "generation_linear.bias")
            z_out = z2d.view(batch_size, measures_to_gen, -1)
            w, s = self._decode(z2d)                       # rows ordered (b, measure): all measures in one call
            weights = w.view(batch_size, measures_to_gen, measure_seq_len, -1)
            samples = s.view(batch_size, 1, measures_to_gen * measure_seq_len)
            return weights, samples, z_out
        hidden = context_vector
        gen_rnn_input = seed
        z_out, weights, samples = [], [], []
        for i in range(measures_to_gen):
            rnn_out, hidden = self._bigru("generation_rnn", gen_rnn_input.contiguous(), None, hidden, Hg, self.z_dim)
            gen_z = _LinearFn.apply(rnn_out.reshape(batch_size, -1), self.flat_for_autograd(), self,
                                    "generation_linear.weight", "generation_linear.bias")
            z_out.append(gen_z.view(batch_size, 1, -1))
            w, s = self._decode(gen_z)
            samples.append(s)
            weights.append(w.unsqueeze(1))
            # (the reference re-encodes the LAST generated measure too, latent_rnn.py:259, and drops the result)
            if i + 1 < measures_to_gen or self.encode_unused_target:
                gen_rnn_input = self.get_z_seq(s, eps_ar[i] if eps_ar is not None else None)
        return torch.cat(weights, 1), torch.cat(samples, 2), torch.cat(z_out, 1)

    def save(self):
        os.makedirs(os.path.dirname(self.filepath), exist_ok=True)
        torch.save({k: v.cpu() for k, v in self.state_dict().items()}, self.filepath)
        print(f'Model {self.__repr__()} saved')

    def xavier_initialization(self):
        self.init_reference_style()
