"""VAETrainer: MeasureVAE/vae_trainer.py:10-139 of the reference on the HIP kernels."""
import torch

from .helpers import to_cuda_variable_long
from .trainer import Trainer, _ElboFn, _KLFn


class VAETrainer(Trainer):
    feed_fields = (0,)                     # the metadata tensor is never read (vae_trainer.py:42-55)

    def __init__(self, dataset, model, lr=1e-4):
        super().__init__(dataset, model, lr)

    def loss_and_acc_for_batch(self, batch, epoch_num=None, train=True):
        """loss = CE_mean + 1e-3 * mean_b KL ; accuracy   (vae_trainer.py:16-40)"""
        score = batch
        weights, samples, z_dist, prior_dist, z_tilde, z_prior = self.model(measure_score_tensor=score, train=train)
        kl_sum = getattr(z_dist, "kl_sum", None)
        if kl_sum is not None and type(self).compute_kld_loss is VAETrainer.compute_kld_loss and \
                type(self).mean_crossentropy_loss_and_accuracy is Trainer.mean_crossentropy_loss_and_accuracy:
            # the same three lines (vae_trainer.py:29-40: recons + 1e-3 * mean KL, accuracy) as one kernel each way
            V = weights.size(-1)
            t1 = score.contiguous().view(-1)
            if t1.dtype != torch.int64:
                t1 = t1.long()
            return _ElboFn.apply(weights.contiguous().view(-1, V), t1, kl_sum, 0.001 / z_dist.loc.shape[0],
                                 self.model.take_stats(2))
        recons_loss, accuracy = self.mean_crossentropy_loss_and_accuracy(weights, score)
        dist_loss = self.compute_kld_loss(z_dist, prior_dist)
        loss = recons_loss + dist_loss
        return loss, accuracy

    def process_batch_data(self, batch):
        """(B,1,384) int32 -> (B*16, 24) int64 on the device   (vae_trainer.py:42-55)"""
        score_tensor = batch[0]
        n_bars = getattr(self.dataset, "n_bars", None)
        if n_bars is not None and score_tensor.dim() == 3:
            batch_size = score_tensor.size(0)
            score_tensor = score_tensor.reshape(batch_size, n_bars, -1)
            score_tensor = score_tensor.reshape(batch_size * n_bars, -1)
        return to_cuda_variable_long(score_tensor)

    def update_scheduler(self, epoch_num):
        return

    @staticmethod
    def compute_kld_loss(z_dist, prior_dist, beta=0.001):
        """vae_trainer.py:128-139; prior must be N(0,1) as built at measure_vae.py:122-125."""
        log_scale = getattr(z_dist, "log_scale", None)
        if log_scale is None:
            raise ValueError("compute_kld_loss expects the encoder's distribution (it carries log_scale)")
        kl_sum = getattr(z_dist, "kl_sum", None)
        if kl_sum is not None:
            # rsample() already summed the KL terms in the reparameterisation kernel; its backward handles dz and the KL
            # gradient in one pass (no second kernel, no gradient accumulation between two paths into mu / log sigma)
            return kl_sum * (beta / z_dist.loc.shape[0])
        return _KLFn.apply(z_dist.loc, log_scale, beta)
