"""LatentRNNTrainer: LatentRNN/latent_rnn_trainer.py:8-176 of the reference on the HIP kernels.

Same public methods (process_batch_data, loss_and_acc_for_batch, split_score_stochastic, split_score,
split_to_measures) and the same consumption of torch's CPU generator (two randint draws per batch: n_target, then
n_past), so a seeded reference run and a seeded run here split every batch identically.  What differs is where the
work happens: the (B,1,384) int32 score crosses the bus once, as int32, and ONE kernel (inet_split_score) writes the
three contiguous int64 tensors; the reference slices on the host and moves three int64 tensors.
"""
import torch

from . import ops
from .model import default_device
from .trainer import Trainer


class LatentRNNTrainer(Trainer):
    feed_fields = (0,)                     # the metadata tensor of a batch is never read (latent_rnn_trainer.py:31-33)

    def __init__(self, dataset, model, lr=1e-4, early_stopping=False):
        super().__init__(dataset, model, lr, early_stopping)
        # window sizes of the stochastic split (latent_rnn_trainer.py:17-22)
        self.min_num_measures_target, self.max_num_measure_target = 2, 6
        n_bars = self.dataset.n_bars
        if not (self.min_num_measures_target <= self.max_num_measure_target < n_bars):
            raise AssertionError("target window does not fit the dataset's sequences")
        self.measure_seq_len = self.dataset.subdivision * self.dataset.num_beats_per_bar

    # ---- Trainer interface ----------------------------------------------------------------------------------
    def process_batch_data(self, batch):
        return self.split_score_stochastic(batch[0])

    def loss_and_acc_for_batch(self, batch, epoch_num=None, train=True):
        """mean CE over (B, n_target, 24) rows + accuracy   (latent_rnn_trainer.py:36-67)"""
        past, future, target = batch
        n_gen = self.dataset.n_bars - past.size(1) - future.size(1)
        weights, _samples, _gen_z = self.model(past_context=past, future_context=future, target=target,
                                               measures_to_generate=n_gen, train=train)
        return self.mean_crossentropy_loss_and_accuracy(weights, target)

    def update_scheduler(self, epoch_num):
        return

    # ---- splitting ------------------------------------------------------------------------------------------
    def draw_split(self, num_measures, fix_num_target=None):
        """(n_past, n_target) for one batch.  Draw order and ranges are the reference's (latent_rnn_trainer.py:99-111):
        n_target ~ U{min..max} first (skipped when fixed), then n_past ~ U{1..num_measures - n_target - 2}.  Both come
        from torch's global CPU generator: data-parallel ranks seed it identically (dp.seed_shared) and so agree."""
        if fix_num_target is None:
            n_target = int(torch.randint(self.min_num_measures_target, self.max_num_measure_target + 1, (1,)))
        else:
            n_target = int(fix_num_target)
        n_past = int(torch.randint(1, num_measures - n_target - 1, (1,)))
        return n_past, n_target

    def split_score_stochastic(self, score_tensor, extra_outs=False, fix_num_target=None):
        """latent_rnn_trainer.py:77-132"""
        num_measures = score_tensor.size(-1) // self.measure_seq_len
        if num_measures != self.dataset.n_bars or score_tensor.size(-1) % self.measure_seq_len:
            raise AssertionError("sequence length does not match dataset.n_bars")
        n_past, n_target = self.draw_split(num_measures, fix_num_target)
        parts = self.split_score(score_tensor, n_past, num_measures - n_past - n_target, n_target, self.measure_seq_len)
        return parts + (n_past, n_target) if extra_outs else parts

    @staticmethod
    def split_score(score_tensor, num_past, num_future, num_target, measure_seq_len):
        """(B,1,L) -> past (B,np,24), future (B,nf,24), target (B,nt,24), int64 on the device
        (latent_rnn_trainer.py:134-160)."""
        batch_size, _, seq_len = score_tensor.size()
        if seq_len % measure_seq_len != 0:
            raise ValueError
        assert seq_len // measure_seq_len == num_past + num_future + num_target
        dev = default_device()
        if score_tensor.dtype != torch.int32:
            score_tensor = score_tensor.to(torch.int32)
        score = score_tensor.to(dev, non_blocking=True).contiguous()
        return ops.split_score(score, num_past, num_target, measure_seq_len)

    @staticmethod
    def split_to_measures(score_tensor, measure_seq_len):
        """(B,1,L) -> (B, L/measure_seq_len, measure_seq_len) view   (latent_rnn_trainer.py:162-176)"""
        batch_size, _, seq_len = score_tensor.size()
        if seq_len % measure_seq_len != 0:
            raise ValueError
        return score_tensor.reshape(batch_size, seq_len // measure_seq_len, measure_seq_len)
