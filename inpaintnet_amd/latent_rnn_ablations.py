"""Past-only / future-only ablations of the LatentRNN (LatentRNN/latent_rnn_ablations.py:11-313 of the reference).

Same modules and state_dict keys as the LatentRNN, with two differences: only ONE context (the past or the future
bi-GRU's final hidden states) initialises the generation GRU, whose hidden size is therefore rnn_hidden_size instead of
2 x rnn_hidden_size (latent_rnn_ablations.py:77-85,143-146).  Everything runs on the same HIP kernels as the LatentRNN;
the arena layout comes from layout.latent_param_shapes(gen_hidden=rnn_hidden_size).  (The reference still runs both
context GRUs and throws one result away; here the unused one is simply not evaluated -- its gradient is zero either way.)
"""
from .latent_rnn import LatentRNN


class LatentRNNAblations(LatentRNN):
    def __init__(self, dataset, vae_model, num_rnn_layers, rnn_hidden_size, dropout, rnn_class, auto_reg=False,
                 teacher_forcing=True, type='past'):
        if type not in ("past", "future"):
            raise ValueError("type must be 'past' or 'future'")
        self.context_mode = type
        self.type = type
        super().__init__(dataset, vae_model, num_rnn_layers, rnn_hidden_size, dropout, rnn_class, auto_reg=auto_reg,
                         teacher_forcing=teacher_forcing)

    def __repr__(self):
        filestr = f'LatentRNN(' \
                  f'{self.type}' \
                  f'{self.dataset}' \
                  f'{self.rnn_class},' \
                  f'{self.num_rnn_layers},' \
                  f'{self.rnn_hidden_size},' \
                  f'{self.dropout},' \
                  f')'
        if self.auto_reg:
            filestr += 'auto_reg'
        return filestr + (',tf' if self.use_teacher_forcing else ',no_tf')
