"""utils/helpers.py:5-50 of the reference; the device is always the ROCm GPU."""
import torch

from .model import default_device


def to_cuda_variable(tensor):
    return tensor.to(default_device())


def to_cuda_variable_long(tensor):
    """utils/helpers.py:17-26.  int32 dataset tensors cross the bus as int32 and are widened by the HIP kernel
    (inet_tokens_to_i64); anything else takes torch's conversion."""
    dev = default_device()
    if tensor.dtype == torch.int32:
        from . import ops
        t = tensor if tensor.is_cuda else tensor.to(dev, non_blocking=True)
        return ops.tokens_to_long(t.contiguous())
    return tensor.to(device=dev, dtype=torch.int64, non_blocking=True).contiguous()


def to_numpy(variable):
    return variable.detach().cpu().numpy()


def init_hidden_lstm(num_layers, batch_size, lstm_hidden_size):
    dev = default_device()
    return (torch.zeros(num_layers, batch_size, lstm_hidden_size, device=dev),
            torch.zeros(num_layers, batch_size, lstm_hidden_size, device=dev))
