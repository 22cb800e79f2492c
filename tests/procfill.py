"""Key-seeded procedural weights for the full-size aligner (47.9 M parameters, 192 MB): the G7 ``encoder_full``
fixture stores inputs and expected outputs only; both the generator (tools/gen_golden.py, which loads these
weights into the REFERENCE's Feature_Aligner) and the tests (which load them into the mirror) rebuild the
state dict from the key names, so no weight file travels.

Each tensor is filled from its own numpy ``RandomState`` (legacy MT19937 stream: bit-stable across numpy
versions) seeded by SHA-256 of the key, uniform in +-1/sqrt(fan_in) like torch's default Linear / Conv init;
normalisation gains sit around 1, biases around 0, so that every LayerNorm / GroupNorm affine term and every
bias is exercised with a non-trivial value.
"""
import hashlib

import numpy as np


def _seed(key: str) -> int:
    return int.from_bytes(hashlib.sha256(key.encode()).digest()[:4], "little")


def procedural_tensor(key: str, shape) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    rs = np.random.RandomState(_seed(key))
    n = int(np.prod(shape)) if shape else 1
    u = rs.random_sample(n).astype(np.float64) * 2.0 - 1.0
    leaf = key.rsplit(".", 1)[-1]
    if "num_batches_tracked" in key:
        return np.zeros(shape, dtype=np.int64)
    if "running_var" in key:
        return np.ones(shape, dtype=np.float32)
    if "running_mean" in key:
        return np.zeros(shape, dtype=np.float32)
    is_norm = ".norm" in key or key.startswith("att.norm") or "bn_down" in key
    if is_norm and leaf == "weight":
        v = 1.0 + 0.25 * u
    elif leaf == "bias":
        v = 0.1 * u
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        v = u / np.sqrt(float(fan_in))
    return v.astype(np.float32).reshape(shape)


def procedural_state_dict(reference_state_dict) -> dict:
    """{key: torch tensor} for every key of ``reference_state_dict`` (shapes and dtypes taken from it)."""
    import torch
    out = {}
    for k, t in reference_state_dict.items():
        a = procedural_tensor(k, t.shape)
        out[k] = torch.from_numpy(a).to(t.dtype)
    return out
