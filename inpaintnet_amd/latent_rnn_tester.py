"""Inference surface of the LatentRNN: B = 1 inpainting (LatentRNN/latent_rnn_tester.py:13-300 of the reference).

`generate` fills `num_target_measures` measures between a past and a future context with the model's free-running path
(train=False: no teacher forcing, no dropout) and returns the full token tensor past | generated | future.  The
tensor -> music21 score conversion of the reference (dataset.tensor_to_score) is outside the hot path: it is called when
the dataset offers it, otherwise the score slots of the return tuple are None.  The reference's call omits the `target`
argument of LatentRNN.forward (latent_rnn_tester.py:231-236, a TypeError as written); here forward() accepts
target=None.
"""
import os

import torch

from . import ops
from .helpers import to_cuda_variable_long
from .trainer import Trainer


class LatentRNNTester(object):
    def __init__(self, dataset, model):
        self.dataset = dataset
        self.model = model
        self.model.eval()
        if self.model.flat.is_cuda:
            ops.preload()                            # no generation call pays a kernel's first launch (csrc/preload.hip)
        self.filepath = os.path.join('models/', self.model.__repr__())
        self.min_num_measures_target = 1
        self.max_num_measure_target = 4
        assert self.dataset.n_bars > self.max_num_measure_target >= self.min_num_measures_target
        self.measure_seq_len = self.dataset.subdivision * self.dataset.num_beats_per_bar
        self.batch_size = 1

    def _to_score(self, tensor):
        fn = getattr(self.dataset, "tensor_to_score", None)
        return fn(tensor.cpu()) if fn is not None else None

    def generate(self, tensor_past, tensor_future, tensor_target, num_target_measures, eval=False):
        """-> (gen_score | None, gen_score_tensor (B, n_past + n_target + n_future, 24), original_score | None)
        (latent_rnn_tester.py:197-266)"""
        if tensor_target is not None:
            if num_target_measures is not None:
                assert num_target_measures == tensor_target.size(1)
            num_target_measures = tensor_target.size(1)
        elif num_target_measures is None:
            raise ValueError
        if tensor_past is None:
            tensor_past = self.create_empty_context('start')
        if tensor_future is None:
            tensor_future = self.create_empty_context('end')
        with torch.no_grad():
            weights, gen_target, _ = self.model(past_context=tensor_past, future_context=tensor_future, target=None,
                                                measures_to_generate=num_target_measures, train=False)
        self.last_weights = weights
        torch.cuda.synchronize()
        ops.check_chains("LatentRNNTester.generate")                   # persistent kernels: never hand back results of a failed launch
        if tensor_target is not None and eval:
            loss, accuracy = Trainer.mean_crossentropy_loss_and_accuracy(weights, tensor_target)
            self.last_eval = (float(loss), float(accuracy))
            print('Accuracy for Test Case:')
            print(f'\tLoss: {self.last_eval[0]}\tAccuracy: {self.last_eval[1] * 100} %')
        batch_size = gen_target.size(0)
        gen_target = gen_target.view(batch_size, num_target_measures, self.measure_seq_len)
        gen_score_tensor = torch.cat((tensor_past, gen_target, tensor_future), 1)
        original = None
        if tensor_target is not None:
            original = self._to_score(torch.cat((tensor_past, tensor_target, tensor_future), 1))
        return self._to_score(gen_score_tensor), gen_score_tensor, original

    def create_empty_context(self, type):
        """(1, num_measures, 24) of one symbol: 3 START measures, 1 END measure or 1 rest measure (:268-296)."""
        notes = self.dataset.note2index_dicts[getattr(self.dataset, "NOTES", 0)]
        if type == 'start':
            num_measures, symbol = 3, notes[getattr(self.dataset, "START_SYMBOL", "START")]
        elif type == 'end':
            num_measures, symbol = 1, notes[getattr(self.dataset, "END_SYMBOL", "END")]
        elif type == 'rest':
            num_measures, symbol = 1, notes['rest']
        else:
            raise ValueError('Invalid argument "type"')
        return to_cuda_variable_long(torch.full((1, num_measures, self.measure_seq_len), int(symbol), dtype=torch.int32))

    def loss_and_acc_test(self, data_loader, fix_num_target=4):
        """Mean loss / accuracy of inpainting over a loader, fixed split as the reference's test loop (:298-340)."""
        from .latent_rnn_trainer import LatentRNNTrainer
        tot = torch.zeros(3)
        for score_tensor, _ in data_loader:
            n_meas = score_tensor.size(-1) // self.measure_seq_len
            n_past = (n_meas - fix_num_target) // 2
            past, future, target = LatentRNNTrainer.split_score(score_tensor, n_past, n_meas - n_past - fix_num_target,
                                                                fix_num_target, self.measure_seq_len)
            with torch.no_grad():
                w, _, _ = self.model(past, future, target, fix_num_target, train=False)
                loss, acc = Trainer.mean_crossentropy_loss_and_accuracy(w, target)
            tot += torch.tensor([float(loss), float(acc), 1.0])
            ops.check_chains("LatentRNNTester.loss_and_acc_test")
        n = max(float(tot[2]), 1.0)
        return float(tot[0]) / n, float(tot[1]) / n
