"""LatentRNNTrainer: LatentRNN/latent_rnn_trainer.py:8-176 of the reference on the HIP kernels."""
import torch

from .helpers import to_cuda_variable_long
from .trainer import Trainer


class LatentRNNTrainer(Trainer):
    def __init__(self, dataset, model, lr=1e-4, early_stopping=False):
        super().__init__(dataset, model, lr, early_stopping)
        self.min_num_measures_target = 2
        self.max_num_measure_target = 6
        assert self.max_num_measure_target >= self.min_num_measures_target
        assert self.dataset.n_bars > self.min_num_measures_target
        assert self.dataset.n_bars > self.max_num_measure_target
        self.measure_seq_len = self.dataset.subdivision * self.dataset.num_beats_per_bar

    def process_batch_data(self, batch):
        score_tensor, _ = batch
        return self.split_score_stochastic(score_tensor)

    def loss_and_acc_for_batch(self, batch, epoch_num=None, train=True):
        """mean CE over (B, n_target, 24) rows + accuracy   (latent_rnn_trainer.py:36-67)"""
        tensor_past, tensor_future, tensor_target = batch
        num_measures_past = tensor_past.size(1)
        num_measures_future = tensor_future.size(1)
        weights, pred, _ = self.model(past_context=tensor_past, future_context=tensor_future, target=tensor_target,
                                      measures_to_generate=self.dataset.n_bars - num_measures_past - num_measures_future,
                                      train=train)
        return self.mean_crossentropy_loss_and_accuracy(weights, tensor_target)

    def update_scheduler(self, epoch_num):
        return

    def split_score_stochastic(self, score_tensor, extra_outs=False, fix_num_target=None):
        """One (n_target, n_past) draw per batch from torch's CPU generator (latent_rnn_trainer.py:77-132):
        every data-parallel rank seeds that generator identically, so all ranks draw the same split."""
        measures_tensor = LatentRNNTrainer.split_to_measures(score_tensor, self.measure_seq_len)
        num_measures = measures_tensor.size(1)
        assert num_measures == self.dataset.n_bars
        if fix_num_target is None:
            num_target = int(torch.randint(low=self.min_num_measures_target, high=self.max_num_measure_target + 1,
                                           size=(1,)).item())
        else:
            num_target = fix_num_target
        num_past = int(torch.randint(low=1, high=num_measures - num_target - 1, size=(1,)).item())
        num_future = num_measures - num_past - num_target
        tensor_past, tensor_future, tensor_target = LatentRNNTrainer.split_score(
            score_tensor=score_tensor, num_past=num_past, num_future=num_future, num_target=num_target,
            measure_seq_len=self.measure_seq_len)
        if extra_outs:
            return tensor_past, tensor_future, tensor_target, num_past, num_target
        return tensor_past, tensor_future, tensor_target

    @staticmethod
    def split_score(score_tensor, num_past, num_future, num_target, measure_seq_len):
        """latent_rnn_trainer.py:134-160"""
        measures_tensor = LatentRNNTrainer.split_to_measures(score_tensor, measure_seq_len)
        num_measures = measures_tensor.size(1)
        assert num_measures == num_past + num_future + num_target
        tensor_past = to_cuda_variable_long(measures_tensor[:, 0:num_past, :])
        tensor_future = to_cuda_variable_long(measures_tensor[:, num_measures - num_future:, :])
        tensor_target = to_cuda_variable_long(measures_tensor[:, num_past:num_measures - num_future, :])
        return tensor_past, tensor_future, tensor_target

    @staticmethod
    def split_to_measures(score_tensor, measure_seq_len):
        """(B,1,L) -> (B, L/measure_seq_len, measure_seq_len)   (latent_rnn_trainer.py:162-176)"""
        batch_size, _, seq_len = score_tensor.size()
        if seq_len % measure_seq_len != 0:
            raise ValueError
        return score_tensor.reshape(batch_size, -1, measure_seq_len)
