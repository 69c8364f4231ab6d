"""utils/helpers.py:5-50 of the reference; the device is always the ROCm GPU."""
import torch

from .model import default_device


def to_cuda_variable(tensor):
    return tensor.to(default_device())


def to_cuda_variable_long(tensor):
    return tensor.to(device=default_device(), dtype=torch.int64, non_blocking=True).contiguous()


def to_numpy(variable):
    return variable.detach().cpu().numpy()


def init_hidden_lstm(num_layers, batch_size, lstm_hidden_size):
    dev = default_device()
    return (torch.zeros(num_layers, batch_size, lstm_hidden_size, device=dev),
            torch.zeros(num_layers, batch_size, lstm_hidden_size, device=dev))
