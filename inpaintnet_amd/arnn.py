"""AnticipationRNN (BASELINE.json config 5) with the reference's Python surface on the HIP kernels.

Mirrors AnticipationRNN/anticipation_rnn_gauss_reg_model.py (ConstraintModelGaussianReg: forward,
_forward_tf :348-404, _forward_no_tf :190-259, embed_*, output_lstm_constraints :459-476, mask_tensor_score
:512-532) and AnticipationRNN/anticipation_rnn_trainer.py (loss / accuracy / constraint sampling) for
single-voice datasets (FolkDataset: num_voices = 1 -- everything the reference trains on).  The LSTM cells run
in the fused step kernels of csrc/lstm.hip (same K-split geometry as the GRU steps); input projections,
the Linear+ReLU head and the embedding gathers are the batched kernels of the C-ABI.

Out of scope as in SURVEY.md section 2.1: generate()/generation() (music21 I/O), forward_inpaint, the unused
gaussian_regularization.
"""
import os
import random

import torch

from . import layout, ops
from .helpers import to_cuda_variable_long
from .latent_rnn_trainer import LatentRNNTrainer
from .measure_vae import _DropState, _next_mask_offset
from .model import Model, default_device
from .trainer import Trainer


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, owner, name, idx, row_scale):
        ctx.args = (owner, name, idx, row_scale)
        return ops.embedding_fwd(owner.param(name), idx, row_scale)

    @staticmethod
    def backward(ctx, dout):
        owner, name, idx, row_scale = ctx.args
        ops.embedding_bwd(dout.contiguous(), idx, owner.param_grad(name), row_scale)
        return None, None, None, None, None


class _LinearFn(torch.autograd.Function):
    """y = [ReLU](x W^T + b)"""

    @staticmethod
    def forward(ctx, x, flat, owner, w_name, b_name, relu):
        y = ops.linear_fwd(x.contiguous(), owner.param(w_name), owner.param(b_name), epi=2 if relu else 0)
        ctx.args = (owner, w_name, b_name, relu)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        owner, w_name, b_name, relu = ctx.args
        x, y = ctx.saved_tensors
        dy = dy.contiguous()
        if relu:
            dy = ops.relu_bwd(dy, y)
        dx = ops.linear_bwd(dy, x.contiguous(), owner.param(w_name), owner.param_grad(w_name),
                            owner.param_grad(b_name), need_dx=ctx.needs_input_grad[0])
        return dx, None, None, None, None, None


class _LstmLayerFn(ops.TrackedFunction):
    """One nn.LSTM(num_layers=1) over a time-major sequence x [T,B,K]; optional carried state (single steps of the
    free-running path)."""

    @staticmethod
    def forward(ctx, x, h0, c0, flat, owner, prefix, reverse):
        T, B, K = x.shape
        W_ih, W_hh = owner.param(prefix + ".weight_ih_l0"), owner.param(prefix + ".weight_hh_l0")
        H = W_hh.shape[1]
        need = ops.outer_grad() and any(ctx.needs_input_grad[:4])
        x2 = x.contiguous().view(T * B, K)
        gi = ops.linear_fwd(x2, W_ih, owner.param(prefix + ".bias_ih_l0"))
        out, hT, cT, ws = ops.lstm_fwd(gi.view(T, B, 4 * H), W_hh, owner.param(prefix + ".bias_hh_l0"), H,
                                       reverse=reverse, h0=h0, c0=c0, save=need, want_state=True)
        ctx.args = (owner, prefix, reverse, H, ws, h0)
        ctx.save_for_backward(x2, out)
        return out, hT, cT

    @staticmethod
    def backward(ctx, dout, dhT, dcT):
        owner, prefix, reverse, H, ws, h0 = ctx.args
        ctx.args = None
        x2, out = ctx.saved_tensors
        T, B, _ = out.shape
        g = owner.param_grad
        dgi, dh0, dc0 = ops.lstm_bwd(owner.param(prefix + ".weight_hh_l0"), out, dout.contiguous(), H, reverse, ws,
                                     dW_hh=g(prefix + ".weight_hh_l0"), db_ih=g(prefix + ".bias_ih_l0"),
                                     db_hh=g(prefix + ".bias_hh_l0"), h0=h0, dhT=dhT.contiguous(),
                                     dcT=dcT.contiguous(), want_dstate=ctx.needs_input_grad[1])
        dx = ops.linear_bwd(dgi.view(T * B, 4 * H), x2, owner.param(prefix + ".weight_ih_l0"),
                            g(prefix + ".weight_ih_l0"), None, need_dx=ctx.needs_input_grad[0])
        return (dx.view(T, B, -1) if dx is not None else None), dh0, dc0, None, None, None, None


class _Lstm2Fn(ops.TrackedFunction):
    """Two stacked nn.LSTM(num_layers=1) modules with zero initial states over x [T,B,K] (lstm_with_activations,
    anticipation_rnn_gauss_reg_model.py:14-39), the layers pipelined over chunks of time steps (ops.lstm2_fwd)."""

    @staticmethod
    def forward(ctx, x, flat, owner, prefix0, prefix1, reverse):
        T, B, K = x.shape
        pr = owner.param
        W_hh0 = pr(prefix0 + ".weight_hh_l0")
        H = W_hh0.shape[1]
        need = ops.outer_grad() and any(ctx.needs_input_grad[:2])
        x2 = x.contiguous().view(T * B, K)
        gi0 = ops.linear_fwd(x2, pr(prefix0 + ".weight_ih_l0"), pr(prefix0 + ".bias_ih_l0"))
        out0, out1, ws0, ws1 = ops.lstm2_fwd(gi0.view(T, B, 4 * H), W_hh0, pr(prefix0 + ".bias_hh_l0"),
                                             pr(prefix1 + ".weight_ih_l0"), pr(prefix1 + ".bias_ih_l0"),
                                             pr(prefix1 + ".weight_hh_l0"), pr(prefix1 + ".bias_hh_l0"), H,
                                             reverse=reverse, save=need)
        ctx.args = (owner, prefix0, prefix1, reverse, H, ws0, ws1)
        ctx.save_for_backward(x2, out0, out1)
        return out1

    @staticmethod
    def backward(ctx, dout1):
        owner, p0, p1, reverse, H, ws0, ws1 = ctx.args
        ctx.args = None
        x2, out0, out1 = ctx.saved_tensors
        T, B, _ = out1.shape
        pr, g = owner.param, owner.param_grad
        dgi0 = ops.lstm2_bwd(pr(p0 + ".weight_hh_l0"), pr(p1 + ".weight_ih_l0"), pr(p1 + ".weight_hh_l0"), out0, out1,
                             dout1.contiguous(), H, reverse, ws0, ws1,
                             grads=(g(p0 + ".weight_hh_l0"), g(p0 + ".bias_ih_l0"), g(p0 + ".bias_hh_l0"),
                                    g(p1 + ".weight_ih_l0"), g(p1 + ".weight_hh_l0"), g(p1 + ".bias_ih_l0"),
                                    g(p1 + ".bias_hh_l0")))
        dx = ops.linear_bwd(dgi0.view(T * B, 4 * H), x2, pr(p0 + ".weight_ih_l0"), g(p0 + ".weight_ih_l0"), None,
                            need_dx=ctx.needs_input_grad[0])
        return (dx.view(T, B, -1) if dx is not None else None), None, None, None, None, None


class ConstraintModelGaussianReg(Model):
    def __init__(self, dataset, note_embedding_dim=20, metadata_embedding_dim=30, num_lstm_constraints_units=256,
                 num_lstm_generation_units=256, linear_hidden_size=128, num_layers=1, dropout_input_prob=0.2,
                 dropout_prob=0.5, unary_constraint=False, teacher_forcing=True, device=None):
        super().__init__()
        if dataset.num_voices != 1 or not unary_constraint:
            raise NotImplementedError("single-voice datasets with unary_constraint=True (train_arnn_reg.py:100)")
        if num_lstm_constraints_units != num_lstm_generation_units:
            raise NotImplementedError("the reference sizes the generation LSTMs with the constraint units (:125-133)")
        self.dataset = dataset
        self.use_teacher_forcing = teacher_forcing
        self.teacher_forcing_prob = 0.5
        self.num_layers = num_layers
        self.num_units_linear = linear_hidden_size
        self.unary_constraint = unary_constraint
        self.note_embedding_dim = note_embedding_dim
        self.num_lstm_generation_units = num_lstm_generation_units
        self.num_lstm_constraints_units = num_lstm_constraints_units
        self.metadata_embedding_dim = metadata_embedding_dim
        self.num_notes_per_voice = [len(d) for d in dataset.note2index_dicts]
        self.num_elements_per_metadata = [m.num_values for m in dataset.metadatas] + [dataset.num_voices]
        self.dropout_input_prob = dropout_input_prob
        self.dropout_prob = dropout_prob
        self.trainable = True
        # The generation LSTMs run forward in time and only the unconstrained ticks' outputs are returned (:433-435): everything they
        # compute BEHIND the last unconstrained tick -- a third of the 384 ticks on average for the trainer's windows -- is read by
        # nobody, forward and backward (the gradient into those ticks is exactly zero), and the head's two products are read on the
        # unconstrained ticks only.  A caller that does not read the second return value (the trainers) asks forward() to stop
        # there (`trim=True`): identical weights and gradients.  False here: always all ticks, as the reference computes them
        # (what bench.py's AnticipationRNN line times); INET_ARNN_ALL_TICKS=1 sets it.
        self.skip_unread_ticks = os.environ.get("INET_ARNN_ALL_TICKS", "0") != "1"
        shapes = layout.arnn_param_shapes(self.num_notes_per_voice[0], note_embedding_dim, metadata_embedding_dim,
                                          num_lstm_constraints_units, linear_hidden_size, num_layers,
                                          tuple(self.num_elements_per_metadata))
        offs, total = layout.arena_offsets(shapes)
        self._alloc_arena([(k, off, shp) for k, (off, shp) in offs.items()], total, device or default_device())
        self._flat_leaf = None
        self.init_torch_defaults()
        cur_dir = os.path.dirname(os.path.realpath(__file__))
        self.filepath = os.path.join(cur_dir, 'models/', self.__repr__())

    @torch.no_grad()
    def init_torch_defaults(self):
        """torch defaults (the reference applies no custom init): Embedding N(0,1); LSTM and Linear U(+-1/sqrt(fan))."""
        for name, _, shape in self._table:
            v = self._views[name]
            if "embeddings" in name:
                v.copy_(torch.randn(shape))
            elif "lstm" in name:
                b = 1.0 / (self.num_lstm_constraints_units ** 0.5)
                v.copy_((torch.rand(shape) * 2 - 1) * b)
            else:
                fan = shape[1] if len(shape) == 2 else self._views[name.replace("bias", "weight")].shape[1]
                v.copy_((torch.rand(shape) * 2 - 1) / fan ** 0.5)

    def __repr__(self):
        filestr = f'AnticipationRNNReg(' \
                  f'{self.dataset.__repr__()},' \
                  f'{self.note_embedding_dim},' \
                  f'{self.metadata_embedding_dim},' \
                  f'{self.num_lstm_constraints_units},' \
                  f'{self.num_lstm_generation_units},' \
                  f'{self.num_units_linear},' \
                  f'{self.num_layers},' \
                  f'{self.dropout_input_prob},' \
                  f'{self.dropout_prob},' \
                  f'{self.unary_constraint},' \
                  f')'
        return filestr + (',tf' if self.use_teacher_forcing else ',no_tf')

    def flat_for_autograd(self):
        if self._flat_leaf is None:
            self._flat_leaf = self.flat.detach().requires_grad_(True)
        return self._flat_leaf

    # ---- building blocks (time-major [L,B,*] internally) ----------------------------------------------------
    def _embed(self, name, idx_tm, row_scale=None):
        L, B = idx_tm.shape
        e = _EmbeddingFn.apply(self.flat_for_autograd(), self, name, idx_tm.contiguous().view(-1), row_scale)
        return e.view(L, B, -1)

    def _lstm(self, prefix, x_tm, reverse, state=None):
        h0, c0 = state if state is not None else (None, None)
        return _LstmLayerFn.call(x_tm, h0, c0, self.flat_for_autograd(), self, prefix, reverse)

    def _lstm_stack(self, base, x_tm, reverse):
        """All layers of one nn.ModuleList of LSTMs with zero initial states (lstm_with_activations, :14-39)."""
        T, B, _ = x_tm.shape
        if self.num_layers == 2 and ops.lstm2_ok(B, T, self.num_lstm_generation_units):
            return _Lstm2Fn.call(x_tm, self.flat_for_autograd(), self, f"{base}.0", f"{base}.1", reverse)
        for l in range(self.num_layers):
            x_tm, _, _ = self._lstm(f"{base}.{l}", x_tm, reverse)
        return x_tm

    def _head(self, h2d):
        a = _LinearFn.apply(h2d, self.flat_for_autograd(), self, "linear_1.weight", "linear_1.bias", True)
        return _LinearFn.apply(a, self.flat_for_autograd(), self, "linear_ouput_notes.0.weight",
                               "linear_ouput_notes.0.bias", False)

    def mask_tensor_score(self, tensor_score, constraints_location=None):
        """tokens where constrained, the extra 'no constraint' symbol elsewhere (:512-532)"""
        p = random.random() * 0.5          # drawn on EVERY call, as the reference does (:517): keeps the `random` stream
        if constraints_location is None:   # -- which also feeds the teacher-forcing coin -- aligned with it
            constraints_location = (torch.rand(*tensor_score.size()) < p).long().to(tensor_score.device)
        no_constraint = self.num_notes_per_voice[0]
        return tensor_score * constraints_location + no_constraint * (1 - constraints_location)

    def _constraints(self, score_tensor, metadata_tensor, constraints_loc):
        """output_lstm_constraints(embed_metadata(...)) -> [L,B,H]"""
        md_tm = metadata_tensor[:, 0].permute(1, 0, 2)                         # [L,B,3]
        parts = [self._embed(f"metadata_embeddings.{i}.weight", md_tm[..., i]) for i in range(md_tm.shape[-1])]
        masked = self.mask_tensor_score(score_tensor, constraints_loc)[:, 0].t()  # [L,B]
        parts.append(self._embed("note_embeddings.0.weight", masked))
        oc = torch.cat(parts, 2)
        return self._lstm_stack("lstm_constraint", oc, True)

    def _forward_tf(self, score_tensor, metadata_tensor, constraints_loc, free=None, L_eff=None):
        """-> [weights (B,L,V)], None   (:348-404).  With `free` (the unconstrained ticks) and L_eff = last of them + 1: the generation
        LSTMs stop behind tick L_eff - 1 and the head runs on the free ticks only -> [weights (B, n_free, V)]."""
        B, _, L = score_tensor.shape
        oc = self._constraints(score_tensor, metadata_tensor, constraints_loc)
        tok_tm = score_tensor[:, 0].t()                                        # [L,B]
        shifted = torch.cat((torch.zeros_like(tok_tm[:1]), tok_tm[:-1]), 0)
        scale = torch.ones(L, B, dtype=torch.float32, device=tok_tm.device)
        if self.training and self.dropout_input_prob > 0:                      # Dropout2d drops whole time steps (:437-442)
            scale = ops.dropout_mask((L, B), self.dropout_input_prob, _DropState.seed, _next_mask_offset(L * B),
                                     tok_tm.device)
        scale[0] = 0.0                                                         # the sequence is offset by a zero vector
        if free is not None:                                                   # (the same mask stream, its first L_eff steps)
            shifted, scale, oc = shifted[:L_eff], scale[:L_eff].contiguous(), oc[:L_eff]
        off = self._embed("note_embeddings.0.weight", shifted, scale.view(-1))
        h = torch.cat((off, oc), 2)
        h = self._lstm_stack("lstm_generation", h, False)
        if free is not None:
            hf = h.index_select(0, free)                                       # [n_free, B, H]
            return [self._head(hf.reshape(free.numel() * B, -1)).view(free.numel(), B, -1).permute(1, 0, 2)], None
        w = self._head(h.view(L * B, -1)).view(L, B, -1).permute(1, 0, 2)
        return [w], None

    def _batched_free_run(self, oc):
        """Whether the free-running passes take the [token pass over batch element 0 + batched kernels] form (else: the per-tick loop)."""
        if not (_FREE_RUN_BATCHED and self.num_layers == 2 and oc.is_cuda):
            return False
        E = self.param("note_embeddings.0.weight").shape[1]
        U, V = self.param("linear_1.weight").shape[0], self.param("linear_ouput_notes.0.weight").shape[0]
        return ops.arnn_generate_ok(E, oc.shape[-1], self.num_lstm_generation_units, U, V)

    def _forward_no_tf(self, score_tensor, metadata_tensor, constraints_loc, free=None, L_eff=None):
        """-> [weights (B,L,V)], gen_chorale (B,1,L): the argmax of BATCH ELEMENT 0 is fed to the whole batch
        (:190-259, quirk at :253-256).  With `free` / L_eff (see _forward_tf): the ticks behind the last unconstrained one are not
        generated -> [weights (B, n_free, V)], gen_chorale (B,1,L_eff)."""
        B, _, L = score_tensor.shape
        oc = self._constraints(score_tensor, metadata_tensor, constraints_loc)
        dev = score_tensor.device
        if free is not None:
            oc, L = oc[:L_eff], L_eff
        if self._batched_free_run(oc):
            # Only the argmax of batch element 0 is fed back (:253-256): its L tokens come from one sequential pass over that one row
            # (ops.arnn_generate: 4 small launches per tick, no host round trip, no autograd), and with the tokens known the whole
            # batch goes through the batched kernels -- the same graph as the teacher-forced pass over the sequence [0, tok_0, ..,
            # tok_{L-2}] (the start symbol is TOKEN 0 here, not the zero vector, and there is no input dropout: :215-231).
            # 195 -> 14 ms per training step; INET_ARNN_FREE_RUN=loop: the per-tick loop below.
            # The tokens come from fp32 one-row kernels, the returned logits from the batched MFMA kernels (another summation
            # order): gen_chorale equals argmax(weights[0]) except possibly on rows whose top-2 logits agree to rounding (~1e-6
            # relative) -- the equivalence test compares them on rows with a clear margin.
            pr = self.param
            with torch.no_grad():
                toks = ops.arnn_generate(pr("note_embeddings.0.weight"), oc.detach()[:, 0, :],
                                         pr("lstm_generation.0.weight_ih_l0"), pr("lstm_generation.0.bias_ih_l0"),
                                         pr("lstm_generation.0.weight_hh_l0"), pr("lstm_generation.0.bias_hh_l0"),
                                         pr("lstm_generation.1.weight_ih_l0"), pr("lstm_generation.1.bias_ih_l0"),
                                         pr("lstm_generation.1.weight_hh_l0"), pr("lstm_generation.1.bias_hh_l0"),
                                         pr("linear_1.weight"), pr("linear_1.bias"),
                                         pr("linear_ouput_notes.0.weight"), pr("linear_ouput_notes.0.bias"))
                prev_tm = torch.cat((torch.zeros(1, dtype=torch.int64, device=dev), toks[:-1])).view(L, 1).expand(L, B).contiguous()
            h = torch.cat((self._embed("note_embeddings.0.weight", prev_tm), oc), 2)
            h = self._lstm_stack("lstm_generation", h, False)
            gen = toks.view(1, 1, L).expand(B, 1, L).contiguous()
            if free is not None:
                hf = h.index_select(0, free)
                return [self._head(hf.reshape(free.numel() * B, -1)).view(free.numel(), B, -1).permute(1, 0, 2)], gen
            w = self._head(h.view(L * B, -1)).view(L, B, -1).permute(1, 0, 2)
            return [w], gen
        prev = torch.zeros(1, B, dtype=torch.int64, device=dev)               # start symbol 0
        states = [None] * self.num_layers
        ws, gen = [], []
        for t in range(L):
            inp = torch.cat((self._embed("note_embeddings.0.weight", prev), oc[t:t + 1]), 2)
            for l in range(self.num_layers):
                inp, hT, cT = self._lstm(f"lstm_generation.{l}", inp, False, states[l])
                states[l] = (hT, cT)
            w = self._head(inp.view(B, -1))
            ws.append(w)
            tok = ops.argmax_rows(w.detach()[0:1])                             # batch element 0
            prev = tok.view(1, 1).expand(1, B).contiguous()
            gen.append(prev)
        wall = torch.stack(ws, 1)
        if free is not None:
            wall = wall[:, free, :]
        return [wall], torch.cat(gen, 0).t().unsqueeze(1).contiguous()

    def forward_inpaint(self, score_tensor, metadata_tensor, constraints_loc, start_tick, end_tick):
        """Inpainting as the testers use it (:261-346): the generation LSTMs read the ground truth up to start_tick
        (teacher-forced, one batched pass that leaves their state), then ticks start_tick .. end_tick-1 are generated
        one at a time -- the argmax of BATCH ELEMENT 0 is written to the whole batch, as in _forward_no_tf.
        -> [weights (B, end_tick - start_tick, V)], gen_chorale (B, 1, L) with the generated window filled in."""
        B, _, L = score_tensor.shape
        if not 0 <= start_tick < end_tick <= L:
            raise ValueError("need 0 <= start_tick < end_tick <= sequence length")
        oc = self._constraints(score_tensor, metadata_tensor, constraints_loc)
        tok_tm = score_tensor[:, 0].t()                                        # [L,B]
        gen = torch.zeros_like(score_tensor)
        gen[:, :, :start_tick] = score_tensor[:, :, :start_tick]
        gen[:, :, end_tick:] = score_tensor[:, :, end_tick:]
        # teacher-forced prefix: ticks 0 .. start_tick-1 (input = zero vector, then the embedded ground truth shifted by one)
        if start_tick > 0:
            shifted = torch.cat((torch.zeros_like(tok_tm[:1]), tok_tm[:start_tick - 1]), 0)
            scale = torch.ones(start_tick, B, dtype=torch.float32, device=tok_tm.device)
            if self.training and self.dropout_input_prob > 0:
                scale = ops.dropout_mask((start_tick, B), self.dropout_input_prob, _DropState.seed,
                                         _next_mask_offset(start_tick * B), tok_tm.device)
            scale[0] = 0.0
        states = [None] * self.num_layers
        if start_tick > 0:
            h = torch.cat((self._embed("note_embeddings.0.weight", shifted, scale.view(-1)), oc[:start_tick]), 2)
            for l in range(self.num_layers):
                h, hT, cT = self._lstm(f"lstm_generation.{l}", h, False)
                states[l] = (hT, cT)
        if self._batched_free_run(oc):
            # as in _forward_no_tf: the window's tokens depend on batch element 0 alone (its state behind the prefix, its token in
            # front of the window) -- one sequential pass over that row, then the window for the whole batch in one batched pass
            W, H, dev = end_tick - start_tick, self.num_lstm_generation_units, gen.device
            pr = self.param
            with torch.no_grad():
                hc = None
                if start_tick > 0:
                    hc = torch.stack([torch.stack((states[l][0].reshape(-1, H)[0], states[l][1].reshape(-1, H)[0])) for l in range(2)])
                first = gen[0, 0, start_tick - 1].reshape(1).contiguous() if start_tick > 0 else None
                toks = ops.arnn_generate(pr("note_embeddings.0.weight"), oc.detach()[start_tick:end_tick, 0, :],
                                         pr("lstm_generation.0.weight_ih_l0"), pr("lstm_generation.0.bias_ih_l0"),
                                         pr("lstm_generation.0.weight_hh_l0"), pr("lstm_generation.0.bias_hh_l0"),
                                         pr("lstm_generation.1.weight_ih_l0"), pr("lstm_generation.1.bias_ih_l0"),
                                         pr("lstm_generation.1.weight_hh_l0"), pr("lstm_generation.1.bias_hh_l0"),
                                         pr("linear_1.weight"), pr("linear_1.bias"),
                                         pr("linear_ouput_notes.0.weight"), pr("linear_ouput_notes.0.bias"),
                                         hc_init=hc.contiguous() if hc is not None else None, first_tok=first)
                gen[:, 0, start_tick:end_tick] = toks.view(1, W)
                # the token in front of the window is each row's own ground truth (or the start symbol), behind it the generated ones
                row0 = gen[:, 0, start_tick - 1].reshape(1, B) if start_tick > 0 else torch.zeros(1, B, dtype=torch.int64, device=dev)
                prev_tm = torch.cat((row0, toks[:-1].view(W - 1, 1).expand(W - 1, B)), 0).contiguous()
            inp = torch.cat((self._embed("note_embeddings.0.weight", prev_tm), oc[start_tick:end_tick]), 2)
            for l in range(self.num_layers):
                inp, _, _ = self._lstm(f"lstm_generation.{l}", inp, False, states[l])
            w = self._head(inp.view(W * B, -1)).view(W, B, -1).permute(1, 0, 2)
            return [w], gen
        ws = []
        for tick in range(start_tick - 1, end_tick - 1):
            # token at `tick` predicts tick + 1; before the first tick the start symbol 0 is embedded (:308-313)
            prev = gen[:, 0, tick].reshape(1, B).contiguous() if tick >= 0 else torch.zeros(1, B, dtype=torch.int64, device=gen.device)
            inp = torch.cat((self._embed("note_embeddings.0.weight", prev), oc[tick + 1:tick + 2]), 2)
            for l in range(self.num_layers):
                inp, hT, cT = self._lstm(f"lstm_generation.{l}", inp, False, states[l])
                states[l] = (hT, cT)
            w = self._head(inp.view(B, -1))
            ws.append(w)
            gen[:, 0, tick + 1] = ops.argmax_rows(w.detach()[0:1]).view(1)     # batch element 0 decides (:340-343)
        return [torch.stack(ws, 1)], gen

    def forward(self, score_tensor, metadata_tensor, constraints_loc, start_tick=None, end_tick=None, train=True,
                teacher_forcing=None, trim=False):
        """-> list (one per voice) of (B, n_unconstrained, V) weights, extra   (:406-435).  trim=True (a caller that does not read
        `extra` behind the last unconstrained tick: the trainers): the generation LSTMs and the head skip what nobody reads
        (`skip_unread_ticks`); the weights are the same."""
        if teacher_forcing is None:
            if self.use_teacher_forcing and train:
                teacher_forcing = random.random() <= self.teacher_forcing_prob
            else:
                teacher_forcing = False
        fwd = self._forward_tf if teacher_forcing else self._forward_no_tf
        free = free_positions(constraints_loc)
        if trim and self.skip_unread_ticks and free.numel() > 0:
            weights, add_args = fwd(score_tensor, metadata_tensor, constraints_loc, free, last_free_position(constraints_loc) + 1)
            return weights, add_args
        weights, add_args = fwd(score_tensor, metadata_tensor, constraints_loc)
        return [w[:, free, :] for w in weights], add_args


_FREE_RUN_BATCHED = os.environ.get("INET_ARNN_FREE_RUN", "batched") != "loop"


def free_positions(constraints_loc, host_copy=None):
    """Ticks to be generated = where voice 0 of batch element 0 is unconstrained (reference :433:
    `(constraints_loc[0, i, :] == 0).nonzero()`).  On the device nonzero() is a device -> host round trip: the launch queue
    drains and the ~40 small launches of the loss behind it run at host pace (0.9 ms of a 7.3 ms training step in the
    kernel trace).  So the answer is computed once per tensor and kept on it (dropped if the tensor is written in place:
    `_version`); the trainer, which builds the tensor on the host, fills it from the host copy without any round trip.

    The cache assumes that writes to `constraints_loc` go through torch in-place operations (they move `_version`).  Writes
    that bypass it -- `.data` edits, a numpy view of a CPU tensor, a kernel writing through `data_ptr()` -- are not seen:
    `del constraints_loc._inet_free` (or a fresh tensor) after such a write."""
    cached = getattr(constraints_loc, "_inet_free", None)
    if cached is not None and cached[0] == constraints_loc._version:
        return cached[1]
    src = constraints_loc if host_copy is None else host_copy
    free_src = (src[0, 0, :] == 0).nonzero().squeeze(-1)
    last = int(free_src[-1]) if free_src.numel() else -1          # (host copy: no round trip; device tensor: the nonzero() was one already)
    free = free_src.to(constraints_loc.device)
    constraints_loc._inet_free = (constraints_loc._version, free, last)
    return free


def last_free_position(constraints_loc):
    """The last unconstrained tick (-1: none), a host integer kept with the cached free positions."""
    free_positions(constraints_loc)
    return constraints_loc._inet_free[2]


class AnticipationRNNBaseline(ConstraintModelGaussianReg):
    """The unconstrained-position baseline of the reference (:682-726): the same network under another name (its
    checkpoints live in their own file); what differs is the trainer's constraint sampling."""

    def __repr__(self):
        return super().__repr__().replace('AnticipationRNNReg(', 'AnticipationRNNBaseline(', 1)


class AnticipationRNNGaussianRegTrainer(Trainer):
    """AnticipationRNN/anticipation_rnn_trainer.py:8-182"""

    def __init__(self, dataset, model, lr=1e-4, early_stopping=False):
        super().__init__(dataset, model, lr, early_stopping)
        self.min_num_measures_target = 2
        self.max_num_measure_target = 6
        self.measure_seq_len = self.dataset.subdivision * self.dataset.num_beats_per_bar

    def loss_and_acc_for_batch(self, batch, epoch_num=None, train=True):
        score_tensor, metadata_tensor, constraints_loc, start_tick, end_tick = batch
        weights, _ = self.model(score_tensor=score_tensor, metadata_tensor=metadata_tensor,
                                constraints_loc=constraints_loc, start_tick=start_tick, end_tick=end_tick, train=train, trim=True)
        free = free_positions(constraints_loc)
        targets = score_tensor[:, :, free].transpose(0, 1)                     # (voice, batch, n_free)
        return self.mean_crossentropy_loss_and_accuracy_voices(weights, targets)

    @staticmethod
    def mean_crossentropy_loss_and_accuracy_voices(weights, targets):
        """mean over voices of the per-voice mean CE / accuracy (:154-182)"""
        loss = acc = 0
        for i, w in enumerate(weights):
            l, a = Trainer.mean_crossentropy_loss_and_accuracy(w, targets[i])
            loss, acc = loss + l, acc + a
        return loss / len(weights), acc / len(weights)

    def process_batch_data(self, batch):
        score_tensor, metadata_tensor = batch
        constraint_loc, start_tick, end_tick = self.get_constraints_location(score_tensor)
        loc = to_cuda_variable_long(constraint_loc)
        free_positions(loc, host_copy=constraint_loc)                          # from the host copy: no device round trip later
        return (to_cuda_variable_long(score_tensor), to_cuda_variable_long(metadata_tensor), loc, start_tick, end_tick)

    def get_constraints_location(self, score_tensor, extra_outs=False, fix_num_target=None):
        """1 = constrained (given) tick, 0 = to be generated: a window of n_target measures (:93-128)"""
        num_measures = LatentRNNTrainer.split_to_measures(score_tensor, self.measure_seq_len).size(1)
        assert num_measures == self.dataset.n_bars
        if fix_num_target is None:
            num_target = int(torch.randint(low=self.min_num_measures_target, high=self.max_num_measure_target + 1,
                                           size=(1,)).item())
        else:
            num_target = fix_num_target
        num_past = int(torch.randint(low=1, high=num_measures - num_target - 1, size=(1,)).item())
        start_tick = (num_past + 1) * self.measure_seq_len
        end_tick = start_tick + num_target * self.measure_seq_len
        constraints_location = torch.zeros_like(score_tensor)
        if start_tick > 0:
            constraints_location[:, :, :start_tick] = 1
        if end_tick < constraints_location.size(2) - 1:
            constraints_location[:, :, end_tick:] = 1
        return constraints_location, start_tick, end_tick

    def update_scheduler(self, epoch_num):
        return


class AnticipationRNNBaselineTrainer(AnticipationRNNGaussianRegTrainer):
    """AnticipationRNN/anticipation_rnn_trainer.py:185-210: constraints are not a contiguous window but an i.i.d. Bernoulli
    mask over the ticks (rate p ~ U(0, 0.5) per batch, the same mask for every sequence of the batch).  The draws use
    Python's `random` (p) and torch's CPU generator (the mask), in the reference's order."""

    def __init__(self, dataset, model, lr=1e-4, early_stopping=False):
        super().__init__(dataset, model, lr, early_stopping)
        self.constraint_prod = 0.5

    def process_batch_data(self, batch):
        score_tensor, metadata_tensor = batch
        p = random.random() * 0.5
        loc = (torch.rand(*score_tensor[0, :, :].size()) < p).unsqueeze(0).repeat(score_tensor.size(0), 1, 1)
        dev_loc = to_cuda_variable_long(loc.to(torch.int32))
        free_positions(dev_loc, host_copy=loc)
        return (to_cuda_variable_long(score_tensor), to_cuda_variable_long(metadata_tensor), dev_loc, None, None)
