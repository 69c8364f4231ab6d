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
import time
from abc import ABC, abstractmethod

import torch

from . import dp, ops


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
        ops.cross_entropy(weights2d, targets1d, out, dW=dW, scale=1.0 / rows)
        ctx.dW = dW
        res = out / rows
        loss, acc = res[0], res[1]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, gloss, _gacc):
        dW = ctx.dW
        ctx.dW = None
        return dW * gloss, None


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
        # deferred side-stream joins for zero_grad() -> forward -> backward -> step() sequences (see zero_grad)
        self.overlap_backward = False

    # ---- utils/trainer.py:41-124 (plot/log plumbing omitted) -----------------------
    def train_model(self, batch_size, num_epochs, plot=False, log=False):
        (generator_train, generator_val, _) = self.dataset.data_loaders(batch_size=batch_size, split=(0.70, 0.20))
        print('Num Train Batches: ', len(generator_train))
        print('Num Valid Batches: ', len(generator_val))
        for epoch_index in range(num_epochs):
            self.update_scheduler(epoch_index)
            self.model.train()
            mean_loss_train, mean_accuracy_train = self.loss_and_acc_on_epoch(
                data_loader=generator_train, epoch_num=epoch_index, train=True)
            self.model.eval()
            mean_loss_val, mean_accuracy_val = self.loss_and_acc_on_epoch(
                data_loader=generator_val, epoch_num=epoch_index, train=False)
            self.print_epoch_stats(epoch_index, num_epochs, mean_loss_train, mean_accuracy_train, mean_loss_val,
                                   mean_accuracy_val)
            if not _dist_ready() or torch.distributed.get_rank() == 0:
                self.model.save()
                if epoch_index > 0 and epoch_index % 10 == 0:
                    self.model.save_checkpoint(epoch_index)
            if self.early_stopping:
                self.early_stopper(mean_loss_val, self.model)
                if self.early_stopper.early_stop:
                    print("Early Stopping")
                    return

    # ---- utils/trainer.py:126-163 ---------------------------------------------------
    def loss_and_acc_on_epoch(self, data_loader, epoch_num=None, train=True):
        dev = self.model.flat.device
        sums = torch.zeros(2, dtype=torch.float32, device=dev)
        t0 = time.time()
        n = 0
        prev_overlap, self.overlap_backward = self.overlap_backward, bool(train)
        for sample_id, batch in enumerate(data_loader):
            batch_data = self.process_batch_data(batch)
            self.zero_grad()
            if train:
                loss, accuracy = self.loss_and_acc_for_batch(batch_data, epoch_num, train=True)
                loss.backward()
                self.step()
            else:
                with torch.no_grad():
                    loss, accuracy = self.loss_and_acc_for_batch(batch_data, epoch_num, train=False)
            sums[0] += loss.detach().mean()
            if accuracy is not None:
                sums[1] += accuracy.detach()
            n += 1
        self.overlap_backward = prev_overlap
        out = (sums / max(n, 1)).tolist()            # the one device->host sync of the epoch
        self.last_epoch_seconds = time.time() - t0
        return out[0], out[1]

    def zero_grad(self):
        """utils/trainer.py:165-170.  With `overlap_backward` set (the epoch loop and bench.py set it) the step that
        starts here runs with deferred side-stream joins: the gradient arena is only complete after step() (or
        ops.side_join()), not right after loss.backward()."""
        if self.overlap_backward:
            ops.side_defer(True)
        self.model.zero_grad()

    def step(self):
        """utils/trainer.py:172-177 (+ the data-parallel gradient exchange)."""
        ops.side_defer(False)                        # joins; deferred mode only lives between zero_grad() and step()
        gscale = dp.allreduce_grads(self.model.grad)
        self.adam_t += 1
        ops.adam_step(self.model.flat, self.model.grad, self.adam_m, self.adam_v, self.lr, self.adam_t,
                      self.betas[0], self.betas[1], self.eps, gscale)

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
    """utils/trainer.py:379-413 (np.Inf replaced by float('inf'): removed in NumPy 2)"""

    def __init__(self, patience=5, verbose=False):
        self.patience = patience
        self.verbose = verbose
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = float("inf")

    def __call__(self, val_loss, model):
        score = -val_loss
        if self.best_score is None:
            self.best_score = score
        elif score <= self.best_score:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            if score - self.best_score < 1e-5:
                self.counter += 1
                if self.counter >= self.patience:
                    self.early_stop = True
            else:
                self.best_score = score
                self.val_loss_min = val_loss
                self.counter = 0
