"""Closed-form, name-keyed parameter fill shared by make_golden.py (applied to the
reference's modules) and the tests (applied to this repo's modules), so no weight
file has to ship: the same formula on both sides gives the same weights.

value[i] = amp(name, shape) * sin(0.37 * i + phase(name, salt)) + offset(name)
"""
import zlib

import numpy as np
import torch


def _phase(name, salt):
    return (zlib.crc32(f'{salt}:{name}'.encode()) % 100003) / 100003.0 * 2.0 * np.pi


def closed_form_array(name, shape, salt):
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    s = np.sin(0.37 * i + _phase(name, salt))
    leaf = name.rsplit('.', 1)[-1]
    if leaf == 'weight' and len(shape) >= 2:            # conv / linear weight
        fan_in = int(np.prod(shape[1:]))
        v = s * (2.0 / np.sqrt(fan_in))
    elif leaf == 'weight':                              # norm gamma
        v = 0.75 + 0.25 * s
    elif leaf == 'bias':
        v = 0.1 * s
    elif leaf == 'running_mean':
        v = 0.1 * s
    elif leaf == 'running_var':
        v = 1.0 + 0.3 * s
    elif leaf == 'num_batches_tracked':
        return np.zeros(shape, dtype=np.int64)
    else:
        v = 0.1 * s
    return v.reshape(shape).astype(np.float32)


def fill_named_tensors(named, salt):
    """named: iterable of (name, tensor); fills in place."""
    with torch.no_grad():
        for name, t in named:
            a = closed_form_array(name, tuple(t.shape), salt)
            t.copy_(torch.from_numpy(a).to(t.dtype))


def fill_module_closed_form(module, salt):
    fill_named_tensors(list(module.named_parameters()) + list(module.named_buffers()), salt)


def volume_like(shape):
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    v = np.sin(0.05 * i) + 0.3 * np.cos(0.0137 * i + 1.0) + 0.2 * np.sin(1.7 * i)
    return torch.from_numpy(v.reshape(shape).astype(np.float32))


def cotangent_like(shape, salt):
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    return torch.from_numpy(np.cos(0.11 * i + salt).reshape(tuple(shape)).astype(np.float32))
