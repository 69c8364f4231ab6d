"""Inference surface of the MeasureVAE: latent interpolation (MeasureVAE/vae_tester.py:17-160 of the reference).

decode_mid_point decodes z1, n evenly spaced points between z1 and z2, and z2 with the free-running decoder (eval mode,
argmax) -- n + 2 small-batch decodes, here ONE decoder call of batch n + 2 (the rows are independent).  Score rendering
(music21) stays outside the hot path.
"""
import os

import torch

from . import ops
from .helpers import to_cuda_variable_long
from .trainer import Trainer


class VAETester(object):
    def __init__(self, dataset, model):
        self.dataset = dataset
        self.model = model
        self.model.eval()
        if self.model.flat.is_cuda:
            ops.preload()                            # no generation call pays a kernel's first launch (csrc/preload.hip)
        self.filepath = os.path.join('models/', self.model.__repr__())
        self.decoder = self.model.decoder
        self.train = False
        self.z_dim = self.decoder.z_dim
        self.batch_size = 1
        self.measure_seq_len = 24

    def decode_mid_point(self, z1, z2, n):
        """z1, z2 (1, z_dim) -> token tensor (1, (n + 2) * 24): z1 | n interpolated | z2   (vae_tester.py:72-93)"""
        assert n >= 1 and isinstance(n, int)
        # z1 and z2 themselves are decoded at the ends (vae_tester.py:82-91), not z1 + (z2 - z1) * 1.0, which is not
        # bit-equal to z2 in fp32 and could flip a near-tie argmax of the last measure
        ks = torch.arange(1, n + 1, device=z1.device, dtype=z1.dtype).view(-1, 1)
        z = torch.cat((z1, z1 + (z2 - z1) * ks / (n + 1), z2), 0)      # same operation order as the reference's loop
        dummy = torch.zeros(n + 2, self.measure_seq_len, device=z1.device)
        with torch.no_grad():
            _, samples = self.decoder(z.contiguous(), dummy, self.train)
        torch.cuda.synchronize()
        ops.check_chains("VAETester.decode_mid_point")             # never hand back tokens of a failed persistent launch
        return samples.reshape(1, -1)

    def test_interpolation(self, tensor_score1, tensor_score2, n=1):
        """(1, 24) x 2 -> interpolation through the encoder means (vae_tester.py:95-111); returns the token tensor
        (and the rendered score when the dataset can render)."""
        with torch.no_grad():
            z1 = self.model.encoder(tensor_score1).loc
            z2 = self.model.encoder(tensor_score2).loc
        tensor_score = self.decode_mid_point(z1, z2, n)
        fn = getattr(self.dataset, "tensor_to_score", None)
        return (fn(tensor_score.cpu()) if fn is not None else None), tensor_score

    def loss_and_acc_test(self, data_loader):
        """vae_tester.py:113-160: mean reconstruction CE / accuracy over a loader (eval mode, no teacher forcing)."""
        tot = torch.zeros(3)
        n_bars = getattr(self.dataset, "n_bars", None)
        for score_tensor, _ in data_loader:
            if n_bars is not None and score_tensor.dim() == 3:
                score_tensor = score_tensor.reshape(score_tensor.size(0) * n_bars, -1)
            score = to_cuda_variable_long(score_tensor)
            with torch.no_grad():
                weights = self.model(measure_score_tensor=score, train=False)[0]
                loss, acc = Trainer.mean_crossentropy_loss_and_accuracy(weights, score)
            tot += torch.tensor([float(loss), float(acc), 1.0])
            ops.check_chains("VAETester.loss_and_acc_test")
        n = max(float(tot[2]), 1.0)
        return float(tot[0]) / n, float(tot[1]) / n
