"""Device selection semantics of rumpy/shared_framework/configuration/gpu_check.py:15-25."""
import torch


def device_selector(gpu, sp_device):
    """gpu: 'off' | 'single' | 'multi'; sp_device: GPU index.  -> the raw index when a GPU is usable, else cpu."""
    if gpu != 'off' and torch.cuda.is_available():
        return sp_device
    return torch.device('cpu')
