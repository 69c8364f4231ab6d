"""Trainer base class: the reference's optimizer loop on the HIP kernels.

Mirrors utils/trainer.py:15-376 of the reference: Trainer(dataset, model, lr,
early_stopping), train_model, loss_and_acc_on_epoch, zero_grad, step, the
abstract loss_and_acc_for_batch / process_batch_data / update_scheduler, and
the static loss helpers.  Differences that are the point of the build:
  * zero_grad() / step() act on the model's flat arenas: one memset, one fused
    Adam kernel (torch.optim.Adam semantics, lr=1e-4, betas (0.9,0.999), eps 1e-8);
  * with torch.distributed initialised (one process per GPU, backend "nccl" =
    RCCL), step() first sums the flat gradient arena over ranks in ONE all-reduce
    and folds the 1/world_size into the Adam kernel;
  * losses are the fused cross-entropy / KL kernels (autograd Functions);
  * loss/accuracy are accumulated on the device, read back once per epoch
    (the reference syncs every batch: utils/trainer.py:154).
Plotting / tensorboard plumbing of the reference is out of scope.
"""
import os
import time
from abc import ABC, abstractmethod

import torch

from . import dp, ops
from .feed import DeviceFeed


def _dist_ready():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


class _CrossEntropyFn(torch.autograd.Function):
    """mean CE over rows + accuracy, one wavefront per row (utils/trainer.py:271-306)."""

    @staticmethod
    def forward(ctx, weights2d, targets1d):
        rows, V = weights2d.shape
        out = torch.zeros(2, dtype=torch.float32, device=weights2d.device)
        need = ctx.needs_input_grad[0]
        dW = torch.empty_like(weights2d) if need else None
        ops.cross_entropy(weights2d, targets1d, out, dW=dW, scale=1.0 / rows, out_scale=1.0 / rows)
        ctx.dW = dW
        loss, acc = out[0], out[1]                  # the kernel accumulated the means: no follow-up launch
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, gloss, _gacc):
        dW = ctx.dW
        ctx.dW = None
        return dW * gloss, None


class _ElboFn(torch.autograd.Function):
    """loss = mean CE + kscale * kl_sum, accuracy -- VAETrainer.loss_and_acc_for_batch (vae_trainer.py:29-40) in ONE launch
    forward (the CE kernel also adds the KL term) and ONE launch backward (the same kernel writes dW already multiplied
    by the upstream gradient, read on the device, and hands kscale * gradient on to the KL sum's producer): the dozen
    one-element torch kernels of the unfused expression (zeros, mul, add, select-backward, dW * g) are gone."""

    @staticmethod
    def forward(ctx, weights2d, targets1d, kl_sum, kscale, out2):
        rows, V = weights2d.shape
        ops.cross_entropy_ex(weights2d, targets1d, loss_sum=out2[0:1], correct=out2[1:2], out_scale=1.0 / rows,
                             add_term=kl_sum.reshape(1), add_scale=kscale)
        ctx.save_for_backward(weights2d, targets1d)
        ctx.kscale = kscale
        loss, acc = out2[0], out2[1]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, gloss, _gacc):
        w, t = ctx.saved_tensors
        dW = torch.empty_like(w)
        gkl = torch.empty((), dtype=torch.float32, device=w.device)
        ops.cross_entropy_ex(w, t, dW=dW, scale=1.0 / w.shape[0], scale_dev=gloss.reshape(1).contiguous(),
                             fwd_out=gkl.reshape(1), fwd_scale=ctx.kscale)
        return dW, None, gkl, None, None


class _KLFn(torch.autograd.Function):
    """beta * mean_b sum_d KL(N(mu, sigma) || N(0,1))   (vae_trainer.py:128-139)."""

    @staticmethod
    def forward(ctx, mu, ls, beta):
        acc = torch.zeros(1, dtype=torch.float32, device=mu.device)
        ops.reparam_kl(mu, ls, None, kl_sum=acc)
        ctx.save_for_backward(mu, ls)
        ctx.k = beta / mu.shape[0]
        return acc[0] * ctx.k

    @staticmethod
    def backward(ctx, g):
        mu, ls = ctx.saved_tensors
        dmu, dls = ops.latent_bwd(None, mu, ls, None, ctx.k)
        return dmu * g, dls * g, None


class Trainer(ABC):
    """utils/trainer.py:15-39"""

    feed_fields = (0, 1)       # which members of a (score, metadata) batch the trainer needs on the device

    def __init__(self, dataset, model, lr=1e-4, early_stopping=False):
        self.dataset = dataset
        self.model = model
        self.lr = lr
        self.betas = (0.9, 0.999)
        self.eps = 1e-8
        self.adam_m = torch.zeros_like(model.flat)
        self.adam_v = torch.zeros_like(model.flat)
        self.adam_t = 0
        self.early_stopping = False
        if early_stopping:
            self.early_stopping = True
            self.early_stopper = EarlyStopping()
        self.last_epoch_seconds = None
        self.start_epoch = 0                         # load_training_state() moves it
        self.chain_fallback = True                   # on a chain-kernel timeout: per-step kernels + retry (else raise)
        self.chain_fallbacks = 0
        # deferred side-stream joins for zero_grad() -> forward -> backward -> step() sequences (see zero_grad)
        self.overlap_backward = False

    # ---- utils/trainer.py:41-124 (plot/log plumbing omitted) -----------------------
    def train_model(self, batch_size, num_epochs, plot=False, log=False, seed=0):
        """Epoch loop of the reference (train pass, validation pass, stats, save, optional early stopping).
        `batch_size` is the GLOBAL batch.  Under torch.distributed every rank iterates the same loader (same shuffle:
        the host generators are seeded identically), takes its contiguous shard of every batch, and sees the same
        epoch statistics (they are summed over ranks), so save / early-stopping decisions agree on all ranks."""
        if dp.world_size() > 1:
            dp.seed_shared(seed)
            dp.seed_rank(seed)
            dp.broadcast_params(self.model.flat)
        train_loader, val_loader, _ = self.dataset.data_loaders(batch_size=batch_size, split=(0.70, 0.20))
        print('Num Train Batches: ', len(train_loader))
        print('Num Valid Batches: ', len(val_loader))
        for epoch in range(self.start_epoch, num_epochs):
            self.update_scheduler(epoch)
            stats = []
            for loader, train in ((train_loader, True), (val_loader, False)):
                self.model.train(train)
                stats += self.loss_and_acc_on_epoch(data_loader=loader, epoch_num=epoch, train=train)
            self.print_epoch_stats(epoch, num_epochs, *stats)
            if dp.rank() == 0:
                self.model.save()
                if epoch > 0 and epoch % 10 == 0:
                    self.model.save_checkpoint(epoch)
                    self.save_training_state(self.model.filepath + f'_{epoch}.trainer', epoch + 1)
            if self.early_stopping and self.early_stopper(stats[2], self.model):
                print("Early Stopping")
                return

    # ---- utils/trainer.py:126-163 ---------------------------------------------------
    def loss_and_acc_on_epoch(self, data_loader, epoch_num=None, train=True):
        dev = self.model.flat.device
        sums = torch.zeros(3, dtype=torch.float32, device=dev)        # loss sum, accuracy sum, batches
        t0 = time.time()
        prev_overlap, self.overlap_backward = self.overlap_backward, bool(train)
        try:
            if not isinstance(data_loader, DeviceFeed):
                # this rank's shard of every batch, copied on a side stream two batches ahead (feed.py)
                data_loader = DeviceFeed(data_loader, fields=self.feed_fields, device=dev)
            for sample_id, batch in enumerate(data_loader):
                batch_data = self.process_batch_data(batch)
                for attempt in (0, 1):
                    try:
                        self.zero_grad()
                        if train:
                            loss, accuracy = self.loss_and_acc_for_batch(batch_data, epoch_num, train=True)
                            loss.backward()
                            self.step()
                        else:
                            with torch.no_grad():
                                loss, accuracy = self.loss_and_acc_for_batch(batch_data, epoch_num, train=False)
                        break
                    except ops.ChainTimeoutError:
                        # A persistent kernel of an earlier step (the check does not synchronise) gave up waiting for its
                        # group.  The optimizer kernel has left the weights alone since then (inet_adam_step reads the same
                        # flag on the device), so nothing is corrupted: switch to the per-step kernels for the rest of the
                        # process, and run this batch again.  Batches between the failure and its detection were not applied.
                        if attempt or not self.chain_fallback:
                            raise
                        self._fall_back_from_chains()
                self._accumulate_stats(sums, loss, accuracy)
        finally:
            self.overlap_backward = prev_overlap
            ops.side_defer(False)
        dp.allreduce_sum_(sums)                      # every rank reports (and early-stops on) the global means
        out = sums.tolist()                          # the one device->host sync of the epoch
        timeouts = ops.chain_status(reset=True)      # persistent kernels: bounded spins report here instead of hanging
        if timeouts > 0:                             # < 0: no chain kernel has run in this process
            raise RuntimeError(f"{timeouts} chain-kernel workgroups gave up waiting for their group during this epoch "
                               "(inet_chain_status); its results are not valid")
        n = max(out[2], 1.0)
        self.last_epoch_seconds = time.time() - t0
        return out[0] / n, out[1] / n

    @staticmethod
    def _accumulate_stats(sums, loss, accuracy):
        """sums[:3] += (loss, accuracy, 1) on the device in one launch (inet_epoch_stats_add); a batch whose chain kernels
        timed out -- its results are not valid and the optimizer kernel skipped it -- stays out of the means."""
        loss = loss.detach()
        if loss.dim() > 0:
            loss = loss.mean()
        loss = loss.reshape(1).to(sums.dtype).contiguous()
        acc = None if accuracy is None else accuracy.detach().reshape(1).to(sums.dtype).contiguous()
        if not sums.is_cuda:                         # (the world_size-2 gloo tests drive this loop with host tensors)
            sums[0] += loss[0]
            if acc is not None:
                sums[1] += acc[0]
            sums[2] += 1.0
            return
        ops.epoch_stats_add(sums, loss, acc)

    def zero_grad(self):
        """utils/trainer.py:165-170.  With `overlap_backward` set (the epoch loop and bench.py set it) the step that
        starts here runs with deferred side-stream joins: the gradient arena is only complete after step() (or
        ops.side_join()), not right after loss.backward()."""
        dp.reset_buckets(self.model.grad)            # nothing may survive from a step that never reached step()
        self.model.__dict__.pop("_dp_open", None)    # (nor counts of forward passes that never saw their backward)
        ops.side_defer(bool(self.overlap_backward))
        self.model.zero_grad()

    def step(self):
        """utils/trainer.py:172-177 (+ the data-parallel gradient exchange)."""
        ops.side_defer(False, release=False)         # joins; deferred mode only lives between zero_grad() and step()
        gscale = dp.allreduce_grads(self.model.grad)
        ops.release_held()                           # (after the exchange: the buckets' streams were ordered behind side work)
        self.adam_t += 1
        ops.adam_step(self.model.flat, self.model.grad, self.adam_m, self.adam_v, self.lr, self.adam_t,
                      self.betas[0], self.betas[1], self.eps, gscale)
        # persistent kernels: a bounded spin that ran out raises here, at the latest a few steps after it happened (the
        # host-mapped counter is read without synchronising); the Adam kernel itself skips its update while the flag is up
        ops.check_chains("Trainer.step")

    def _fall_back_from_chains(self):
        torch.cuda.synchronize()
        n = ops.chain_status(reset=True)
        ops.set_option(4, 0)                          # per-step kernels from here on (INET_CHAIN=0 semantics, in-process)
        self.chain_fallbacks += 1
        self.adam_t = max(self.adam_t - 1, 0)         # the step that raised did not update anything
        print(f"[inpaintnet_amd] {n} chain-kernel workgroups timed out; chain kernels are now OFF for this process and the "
              "batch is run again (is the GPU shared, partitioned or CU-masked?)")

    # ---- optimizer / epoch resume (SURVEY 8f1 add-on: the reference saves weights only, utils/model.py:16-53) ----
    def training_state(self, next_epoch=0):
        return {"adam_m": self.adam_m.cpu(), "adam_v": self.adam_v.cpu(), "adam_t": self.adam_t, "lr": self.lr,
                "betas": self.betas, "eps": self.eps, "next_epoch": int(next_epoch),
                "num_parameters": int(self.model.flat.numel())}

    def save_training_state(self, path, next_epoch=0):
        torch.save(self.training_state(next_epoch), path)

    def load_training_state(self, path_or_state):
        """Restore the Adam moments, step count and the epoch to resume from (train_model starts there)."""
        st = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, (str, bytes, os.PathLike)) \
            else path_or_state
        if int(st["num_parameters"]) != self.model.flat.numel():
            raise RuntimeError("training state belongs to a model with a different parameter arena")
        self.adam_m.copy_(st["adam_m"])
        self.adam_v.copy_(st["adam_v"])
        self.adam_t = int(st["adam_t"])
        self.lr, self.betas, self.eps = float(st["lr"]), tuple(st["betas"]), float(st["eps"])
        self.start_epoch = int(st["next_epoch"])
        return self.start_epoch

    @abstractmethod
    def loss_and_acc_for_batch(self, batch, epoch_num=None, train=True):
        pass

    @abstractmethod
    def process_batch_data(self, batch):
        pass

    @abstractmethod
    def update_scheduler(self, epoch_num):
        pass

    @staticmethod
    def print_epoch_stats(epoch_index, num_epochs, mean_loss_train, mean_accuracy_train, mean_loss_val,
                          mean_accuracy_val):
        print(f'Train Epoch: {epoch_index + 1}/{num_epochs}')
        print(f'\tTrain Loss: {mean_loss_train}\tTrain Accuracy: {mean_accuracy_train * 100} %')
        print(f'\tValid Loss: {mean_loss_val}\tValid Accuracy: {mean_accuracy_val * 100} %')

    # ---- static losses, utils/trainer.py:271-376 -------------------------------------
    @staticmethod
    def mean_crossentropy_loss_and_accuracy(weights, targets):
        V = weights.size(-1)
        w2 = weights.contiguous().view(-1, V)
        t1 = targets.contiguous().view(-1)
        if t1.dtype != torch.int64:
            t1 = t1.long()
        return _CrossEntropyFn.apply(w2, t1)

    @staticmethod
    def mean_crossentropy_loss(weights, targets):
        """weights (B,T,V), targets (B,T)   (utils/trainer.py:271-288)"""
        batch_size, seq_len, num_notes = weights.size()
        assert batch_size == targets.size(0)
        assert seq_len == targets.size(1)
        return Trainer.mean_crossentropy_loss_and_accuracy(weights, targets)[0]

    @staticmethod
    def mean_accuracy(weights, targets):
        """utils/trainer.py:290-306 (argmax = lowest index among maxima, as Tensor.max(1))"""
        with torch.no_grad():
            return Trainer.mean_crossentropy_loss_and_accuracy(weights.detach(), targets)[1]

    @staticmethod
    def mean_crossentropy_loss_alt(weights, targets):
        """weights (B,M,T,V), targets (B,M,T)   (utils/trainer.py:344-359)"""
        return Trainer.mean_crossentropy_loss_and_accuracy(weights, targets)[0]

    @staticmethod
    def mean_accuracy_alt(weights, targets):
        """utils/trainer.py:361-376"""
        with torch.no_grad():
            return Trainer.mean_crossentropy_loss_and_accuracy(weights.detach(), targets)[1]

    @staticmethod
    def mean_mse_loss_rnn(weights, targets):
        """utils/trainer.py:327-342 (diagnostic only: never part of a training loss in the reference)"""
        assert weights.size() == targets.size()
        return ((weights - targets) ** 2).mean()

    @staticmethod
    def mean_l1_loss_rnn(weights, targets):
        """utils/trainer.py:308-325"""
        assert weights.size() == targets.size()
        return (weights - targets).abs().mean()


class EarlyStopping:
    """Patience counter on the validation loss (behaviour of utils/trainer.py:379-413): an epoch counts as an
    improvement only if the loss drops by at least `min_delta` below the best seen; `patience` epochs in a row without
    one set `early_stop`.  Calling the object returns the flag."""

    def __init__(self, patience=5, verbose=False, min_delta=1e-5):
        self.patience = patience
        self.verbose = verbose
        self.min_delta = min_delta
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = float("inf")

    def __call__(self, val_loss, model=None):
        score = -float(val_loss)
        if self.best_score is None or score - self.best_score >= self.min_delta:
            if self.best_score is not None:
                self.counter = 0
                self.val_loss_min = float(val_loss)
            self.best_score = score
        else:
            self.counter += 1
            self.early_stop = self.early_stop or self.counter >= self.patience
        return self.early_stop
