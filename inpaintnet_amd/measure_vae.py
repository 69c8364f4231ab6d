"""MeasureVAE with the reference's Python surface, computed by the HIP library.

Mirrors MeasureVAE/measure_vae.py:10-169, MeasureVAE/encoder.py:9-134 and
MeasureVAE/decoder.py:313-529 of the reference: same constructor arguments,
attributes, method names, return arities, shapes and dtypes.  Each module is one
torch.autograd.Function whose forward/backward are single calls into the C-ABI
(include/inpaintnet_hip.h); parameter gradients are accumulated straight into
the model's flat gradient arena (`model.grad`) by the backward kernels.
"""
import os
import random

import torch
from torch import distributions

from . import dp, ops
from .model import Model, default_device

_mask_counter = [0]


def _next_mask_offset(n):
    off = _mask_counter[0]
    _mask_counter[0] += int(n)
    return off


def set_dropout_seed(seed, rank=0):
    """Seed of the counter-based dropout stream (rank-offset for data parallel)."""
    _DropState.seed = (int(seed) * 1000003 + int(rank) * 7919) & (2 ** 63 - 1)
    _mask_counter[0] = 0


class _DropState:
    seed = 0x5eed


class _EncoderFn(ops.TrackedFunction):
    @staticmethod
    def forward(ctx, flat, enc, tokens, mask):
        need = ops.outer_grad() and ctx.needs_input_grad[0]     # grad mode is off inside forward(): ask the ctx
        mu, ls, ws = ops.encoder_fwd(enc.cfg, tokens, flat, mask=mask, save=need)
        ctx.enc, ctx.tokens, ctx.mask, ctx.ws = enc, tokens, mask, ws
        return mu, ls

    @staticmethod
    def backward(ctx, dmu, dls):
        enc = ctx.enc
        owner = enc.owner
        dmu, dls = dmu.contiguous(), dls.contiguous()
        buckets = getattr(owner, "encoder_early_buckets", None) if dp.world_size() > 1 else None
        if buckets:
            # data parallel: the heads and GRU layer 1 first -- 38 of the encoder's 44 MB of gradients are final about a
            # millisecond before the step ends -- their all-reduce runs under the layer-0 BPTT chain
            ops.encoder_bwd(enc.cfg, ctx.tokens, owner.flat, owner.grad, ctx.mask, dmu, dls, ctx.ws, stage=1)
            for lo, hi in buckets:
                dp.start_bucket(owner.grad, lo, hi, join_side=True)
            ops.encoder_bwd(enc.cfg, ctx.tokens, owner.flat, owner.grad, ctx.mask, dmu, dls, ctx.ws, stage=2)
        else:
            ops.encoder_bwd(enc.cfg, ctx.tokens, owner.flat, owner.grad, ctx.mask, dmu, dls, ctx.ws)
        ctx.ws = None
        return None, None, None, None


class _DecoderFn(ops.TrackedFunction):
    @staticmethod
    def forward(ctx, z, flat, dec, target, teacher_forced, mask_beat, mask_tick, multinomial_seed=0):
        need = ops.outer_grad() and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        weights, samples, ws = ops.decoder_fwd(dec.cfg, z.contiguous(), target, teacher_forced, flat, mask_beat,
                                               mask_tick, save=need, multinomial_seed=multinomial_seed)
        ctx.dec, ctx.ws, ctx.mb, ctx.mt = dec, ws, mask_beat, mask_tick
        if getattr(dec, "keep_ws", False):         # test hook: lets a parity test read intermediates (ops.ws_field)
            dec.last_ws = ws
        ctx.save_for_backward(weights, samples)
        ctx.mark_non_differentiable(samples)
        # (no zero tensor for the gradient of `samples`: autograd would launch a fill kernel for it in every backward pass -- 6 us of
        #  stream time for a tensor nobody reads)
        ctx.set_materialize_grads(False)
        return weights, samples

    @staticmethod
    def backward(ctx, dweights, _dsamples):
        dec = ctx.dec
        if dweights is None:                            # nothing downstream depends on the weights
            ctx.ws = None
            return None, None, None, None, None, None, None, None
        weights, samples = ctx.saved_tensors
        grads = dec.owner.grad if dec.owner.trainable else None
        dz = ops.decoder_bwd(dec.cfg, dweights.contiguous(), weights, samples, dec.owner.flat, grads, ctx.mb, ctx.mt,
                             ctx.ws, need_dz=ctx.needs_input_grad[0])
        ctx.ws = None
        if grads is not None:
            # data parallel: the decoder half of the arena is final now -> start its all-reduce under the encoder's backward
            if dp.world_size() > 1:
                # behind the decoder's leaf GEMMs on the side streams, without holding up the encoder's backward
                dp.start_bucket(grads, dec.owner.decoder_arena_start, grads.numel(), join_side=True)
        return dz, None, None, None, None, None, None, None


class _ReparamFn(torch.autograd.Function):
    """z = mu + eps * exp(logsigma) (measure_vae.py:119 rsample) and, from the same pass over (mu, logsigma), the sum of
    the KL terms against N(0, 1) that VAETrainer.compute_kld_loss needs (vae_trainer.py:128-139)."""

    @staticmethod
    def forward(ctx, mu, ls, eps, kl=None):
        if kl is None:
            kl = torch.zeros(1, dtype=torch.float32, device=mu.device)
        z, _ = ops.reparam_kl(mu, ls, eps, kl_sum=kl)
        ctx.save_for_backward(mu, ls, eps)
        ctx.set_materialize_grads(False)
        return z, kl[0]

    @staticmethod
    def backward(ctx, dz, dkl):
        mu, ls, eps = ctx.saved_tensors
        if dz is not None:
            dz = dz.contiguous()
        if dkl is not None:
            dkl = dkl.reshape(1).contiguous()
        dmu, dls = ops.latent_bwd(dz, mu, ls, eps, 1.0 if dkl is not None else 0.0, kscale_dev=dkl)
        return dmu, dls, None, None


class NormalLogScale(distributions.Normal):
    """torch Normal whose log-scale (the encoder's actual output) is kept alongside, so the KL kernel needs no log()
    round trip.  `scale` (= exp(log_scale), encoder.py:133) is computed the first time something reads it: the training
    step never does.  rsample()/sample() draw eps with torch's device generator and combine on the GPU with the
    reparameterisation kernel, which also leaves the KL sum in `kl_sum`."""

    def __init__(self, loc, log_scale, owner=None):
        self.loc = loc
        self.log_scale = log_scale
        self._scale = None
        self.kl_sum = None
        self._owner = owner                      # the model whose arena tail holds the per-step accumulators (or None)
        distributions.Distribution.__init__(self, loc.size(), validate_args=False)

    @property
    def scale(self):
        if self._scale is None:
            self._scale = _ExpFn.apply(self.log_scale)
        return self._scale

    @scale.setter
    def scale(self, value):
        self._scale = value

    def rsample(self, sample_shape=torch.Size(), eps=None):
        if len(sample_shape) != 0:
            return super().rsample(sample_shape)
        if eps is None:
            eps = torch.randn_like(self.loc)
        kl = self._owner.take_stats(1) if self._owner is not None else None
        z, self.kl_sum = _ReparamFn.apply(self.loc, self.log_scale, eps, kl)
        self.last_eps = eps
        return z


class Encoder(torch.nn.Module):
    """MeasureVAE/encoder.py:9-134."""

    def __init__(self, owner, prefix, note_embedding_dim, rnn_hidden_size, num_layers, num_notes, dropout,
                 bidirectional, z_dim, rnn_class):
        super().__init__()
        if num_layers != 2 or not bidirectional:
            raise NotImplementedError("the HIP encoder implements the reference configuration: 2 layers, bidirectional")
        object.__setattr__(self, "owner", owner)
        self.prefix = prefix
        self.bidirectional = bidirectional
        self.num_directions = 2
        self.note_embedding_dim = note_embedding_dim
        self.num_layers = num_layers
        self.rnn_hidden_size = rnn_hidden_size
        self.z_dim = z_dim
        self.dropout = dropout
        self.rnn_class = rnn_class
        self.num_notes = num_notes

    @property
    def cfg(self):
        return self.owner.cfg

    def __repr__(self):
        return f'Encoder(' \
               f'{self.note_embedding_dim},' \
               f'{self.rnn_class},' \
               f'{self.num_layers},' \
               f'{self.rnn_hidden_size},' \
               f'{self.dropout},' \
               f'{self.bidirectional},' \
               f'{self.z_dim},' \
               f')'

    def forward(self, score_tensor, mask=None):
        """score_tensor (B, 24) int64 -> Normal(loc, scale)   (encoder.py:104-134)"""
        batch_size, T = score_tensor.size()
        tokens = score_tensor.contiguous()
        if mask is None and self.training and self.dropout > 0:
            n = T * batch_size * 2 * self.rnn_hidden_size
            mask = ops.dropout_mask((T, batch_size, 2 * self.rnn_hidden_size), self.dropout, _DropState.seed,
                                    _next_mask_offset(n), tokens.device)
        mu, ls = _EncoderFn.call(self.owner.flat_for_autograd(), self, tokens, mask)
        return NormalLogScale(mu, ls, self.owner)


class _ExpFn(torch.autograd.Function):
    """sigma = exp(logsigma)  (encoder.py:133), by the reparameterisation kernel."""

    @staticmethod
    def forward(ctx, ls):
        _, sigma = ops.reparam_kl(ls, ls, None, kl_sum=None, want_sigma=True)
        ctx.save_for_backward(sigma)
        return sigma

    @staticmethod
    def backward(ctx, dsigma):
        (sigma,) = ctx.saved_tensors
        return dsigma * sigma


class HierarchicalDecoder(torch.nn.Module):
    """MeasureVAE/decoder.py:313-529."""

    def __init__(self, owner, prefix, note_embedding_dim, num_notes, z_dim, num_layers, rnn_hidden_size, dropout,
                 rnn_class):
        super().__init__()
        if num_layers != 2:
            raise NotImplementedError("the HIP decoder implements the reference configuration: 2 layers")
        object.__setattr__(self, "owner", owner)
        self.prefix = prefix
        self.name = 'HierarchicalDecoder'
        self.num_notes = num_notes
        self.note_embedding_dim = note_embedding_dim
        self.z_dim = z_dim
        self.rnn_class = rnn_class
        self.num_layers = num_layers
        self.rnn_hidden_size = rnn_hidden_size
        self.dropout = dropout
        self.use_teacher_forcing = True
        self.teacher_forcing_prob = 0.5
        self.sampling = 'argmax'

    @property
    def cfg(self):
        return self.owner.cfg

    def __repr__(self):
        return f'{self.name}' \
               f'{self.note_embedding_dim},' \
               f'{self.rnn_class},' \
               f'{self.num_layers},' \
               f'{self.rnn_hidden_size},' \
               f'{self.dropout},' \
               f')'

    def forward(self, z, score_tensor, train, masks=None, teacher_forced=None):
        """z (B,Z), score_tensor (B,24) -> weights (B,24,V), samples (B,1,24)   (decoder.py:412-453).
        One Bernoulli(0.5) teacher-forcing coin per call when train=True (decoder.py:431-434); it can be
        injected with `teacher_forced=`."""
        if teacher_forced is None:
            if self.use_teacher_forcing and train:
                teacher_forced = random.random() < self.teacher_forcing_prob
            else:
                teacher_forced = False
        if self.sampling not in ('argmax', 'multinomial'):
            raise ValueError(f"sampling must be 'argmax' or 'multinomial' (decoder.py:506-516), got {self.sampling!r}")
        batch_size_z, z_dim = z.size()
        assert z_dim == self.z_dim
        batch_size = score_tensor.size(0)
        assert batch_size == batch_size_z
        T = self.cfg.beats * self.cfg.ticks_per_beat
        target = None
        if teacher_forced:
            target = score_tensor.detach()
            if target.dtype != torch.int64:
                target = target.long()
            target = target.contiguous()
        mb = mt = None
        if masks is not None:
            mb, mt = masks
        elif self.training and self.dropout > 0:
            H = self.rnn_hidden_size
            dev = z.device
            mb = ops.dropout_mask((self.cfg.beats, batch_size, H), self.dropout, _DropState.seed,
                                  _next_mask_offset(self.cfg.beats * batch_size * H), dev)
            mt = ops.dropout_mask((T, batch_size, H), self.dropout, _DropState.seed,
                                  _next_mask_offset(T * batch_size * H), dev)
        seed = 0
        if self.sampling == 'multinomial' and not teacher_forced:
            # one counter-based stream per call, derived from the dropout seed (set_dropout_seed) and the call counter
            seed = ((_DropState.seed * 0x9E3779B97F4A7C15 + _next_mask_offset(T * batch_size) + 1) & (2 ** 64 - 1)) or 1
        weights, samples = _DecoderFn.call(z, self.owner.flat_for_autograd(), self, target, teacher_forced, mb, mt, seed)
        return weights, samples


class MeasureVAE(Model):
    """MeasureVAE/measure_vae.py:10-169."""

    def __init__(self, dataset, note_embedding_dim=10, metadata_embedding_dim=2, num_encoder_layers=2,
                 encoder_hidden_size=512, encoder_dropout_prob=0.5, latent_space_dim=256, num_decoder_layers=2,
                 decoder_hidden_size=512, decoder_dropout_prob=0.5, has_metadata=False, device=None):
        super().__init__()
        self.num_beats_per_measure = 4
        self.num_ticks_per_measure = 24
        self.num_ticks_per_beat = int(self.num_ticks_per_measure / self.num_beats_per_measure)
        self.dataset = dataset.__repr__()
        self.note_embedding_dim = note_embedding_dim
        self.metadata_embedding_dim = metadata_embedding_dim
        self.num_encoder_layers = num_encoder_layers
        self.encoder_hidden_size = encoder_hidden_size
        self.encoder_dropout_prob = encoder_dropout_prob
        self.latent_space_dim = latent_space_dim
        self.num_decoder_layers = num_decoder_layers
        self.decoder_hidden_size = decoder_hidden_size
        self.decoder_dropout_prob = decoder_dropout_prob
        self.has_metadata = has_metadata
        self.num_notes = len(dataset.note2index_dicts[0])
        self.trainable = True
        self.cfg = ops.vae_config(self.num_notes, note_embedding_dim, encoder_hidden_size, latent_space_dim,
                                  decoder_hidden_size, self.num_beats_per_measure, self.num_ticks_per_beat)
        table, total = ops.vae_param_table(self.cfg)
        self._alloc_arena(table, total, device or default_device())
        self.decoder_arena_start = min(off for name, off, _ in table if name.startswith("decoder."))
        # arena ranges whose gradients are final after stage 1 of the encoder's backward (layer 1; the Linear heads): the
        # note embedding sits between them and is final only at the very end
        offs = {name: off for name, off, _ in table}
        self.encoder_early_buckets = ((offs["encoder.lstm.weight_ih_l1"], offs["encoder.note_embedding_layer.weight"]),
                                      (offs["encoder.linear_mean.0.weight"], self.decoder_arena_start))
        self._flat_leaf = None
        self.encoder = Encoder(self, "encoder", note_embedding_dim, encoder_hidden_size, num_encoder_layers,
                               self.num_notes, encoder_dropout_prob, True, latent_space_dim, torch.nn.GRU)
        self.decoder = HierarchicalDecoder(self, "decoder", note_embedding_dim, self.num_notes, latent_space_dim,
                                           num_decoder_layers, decoder_hidden_size, decoder_dropout_prob,
                                           torch.nn.GRU)
        self.init_reference_style()
        cur_dir = os.path.dirname(os.path.realpath(__file__))
        self.filepath = os.path.join(cur_dir, 'models/', self.__repr__())

    def flat_for_autograd(self):
        """The arena as an autograd leaf, so that the module Functions are recorded whenever the model is
        trainable.  Its own .grad is never populated: the backward kernels write into self.grad."""
        if not self.trainable:
            return self.flat
        if self._flat_leaf is None:
            self._flat_leaf = self.flat.detach().requires_grad_(True)     # shares storage
        return self._flat_leaf

    def freeze(self):
        """for p in vae.parameters(): p.requires_grad = False   (latent_rnn.py:42-43)"""
        self.trainable = False

    def __repr__(self):
        return f'MeasureVAE(' \
               f'{self.dataset},' \
               f'{self.encoder.__repr__()},' \
               f'{self.decoder.__repr__()},' \
               f')'

    def forward(self, measure_score_tensor, train=True, eps=None, teacher_forced=None):
        """(B,24) int64 -> (weights, samples, z_dist, prior_dist, z_tilde, z_prior)   (measure_vae.py:97-134)"""
        seq_len = measure_score_tensor.size(1)
        assert seq_len == self.num_ticks_per_measure
        enc_mask = dec_masks = None
        pe, pd = self.encoder.dropout, self.decoder.dropout
        if self.training and pe > 0 and pe == pd:
            # the three dropout masks of a step (encoder layer 0 -> 1, decoder beat and tick layers) in ONE launch: the mask
            # stream is a pure function of (seed, offset + index), and the three would have taken consecutive offsets
            B, T, nb = measure_score_tensor.size(0), seq_len, self.num_beats_per_measure
            He, Hd = self.encoder_hidden_size, self.decoder_hidden_size
            sizes = (T * B * 2 * He, nb * B * Hd, T * B * Hd)
            buf = ops.dropout_mask((sum(sizes),), pe, _DropState.seed, _next_mask_offset(sum(sizes)),
                                   measure_score_tensor.device)
            m0, m1, m2 = torch.split(buf, sizes)
            enc_mask = m0.view(T, B, 2 * He)
            dec_masks = (m1.view(nb, B, Hd), m2.view(T, B, Hd))
        z_dist = self.encoder(measure_score_tensor, mask=enc_mask)
        # prior_dist.sample() (measure_vae.py:127) for N(0, 1) is a plain standard-normal draw (torch.normal(mean, std)
        # would also validate std >= 0 with a device->host read, i.e. stall the host once per forward pass): eps of the
        # reparameterisation and z_prior come out of one generator call
        if eps is None:
            draws = torch.randn((2,) + tuple(z_dist.loc.shape), dtype=z_dist.loc.dtype, device=z_dist.loc.device)
            eps, z_prior = draws[0], draws[1]
        else:
            z_prior = torch.randn_like(z_dist.loc)
        z_tilde = z_dist.rsample(eps=eps)
        prior_dist = self._standard_normal(z_dist.loc)
        weights, samples = self.decoder(z=z_tilde, score_tensor=measure_score_tensor, train=train,
                                        teacher_forced=teacher_forced, masks=dec_masks)
        return weights, samples, z_dist, prior_dist, z_tilde, z_prior

    def _standard_normal(self, like):
        """N(0, 1) of the latent's shape (measure_vae.py:122-125): the constant loc / scale tensors are built once."""
        key = (tuple(like.shape), like.device, like.dtype)
        cache = self.__dict__.setdefault("_prior_cache", {})
        if key not in cache:
            cache[key] = (torch.zeros_like(like), torch.ones_like(like))
        loc, scale = cache[key]
        return distributions.Normal(loc=loc, scale=scale, validate_args=False)

    def forward_test(self, measure_score_tensor):
        """(B,M,24) -> weights (B,M,24,V), samples (B,1,24M)   (measure_vae.py:136-169).  The M measures are
        independent, so they run as one batch of B*M."""
        batch_size, num_measures, seq_len = measure_score_tensor.size()
        assert seq_len == self.num_ticks_per_measure
        flat_in = measure_score_tensor.reshape(batch_size * num_measures, seq_len).contiguous()
        z = self.encoder(flat_in).rsample()
        w, s = self.decoder(z=z, score_tensor=flat_in, train=False)
        weights = w.view(batch_size, num_measures, seq_len, -1)
        samples = s.view(batch_size, 1, num_measures * seq_len)
        return weights, samples
