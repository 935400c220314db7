import torch


def normalize(tensor, mean, std, inplace=False):
    """(t - mean[c]) / std[c] over the channel dim (-3), the only torchvision op the
    reference's infer() path uses (unidepthv2.py:288-293)."""
    mean = torch.as_tensor(mean, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
    std = torch.as_tensor(std, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
    return (tensor - mean) / std
