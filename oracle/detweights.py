"""Key-addressed deterministic weights and inputs.

TEST INFRASTRUCTURE (see oracle/README.md).  Every tensor is generated from
(seed, state_dict key) alone, so the upstream reference modules (in the build
container), the CPU oracle and the HIP product modules all receive bit-identical
parameters without shipping checkpoints and without depending on the order in which
any framework's init code consumes its RNG (`_init_weights` is
re-applied at three nesting levels upstream).

numpy's PCG64 bit stream is stable across platforms/versions, which is what makes the
committed golden vectors reproducible on the GPU box.
"""
import zlib

import numpy as np
import torch


def _rng(seed, key):
    return np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, zlib.crc32(key.encode())]))


def det_tensor(key, shape, seed=0, dtype=torch.float32):
    """Deterministic value for one state_dict entry, chosen by the key's role."""
    shape = tuple(int(s) for s in shape)
    g = _rng(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    n = int(np.prod(shape)) if len(shape) else 1
    z = g.standard_normal(n).astype(np.float64)
    if leaf == "running_var":
        v = 0.5 + g.random(n)  # (0.5, 1.5)
    elif leaf == "running_mean":
        v = 0.1 * z
    elif leaf == "bias":
        v = 0.05 * z
    elif len(shape) >= 2:
        # conv / linear weight: variance-preserving with a gain > 1 so that deep stacks
        # of ReLU/GELU layers keep non-trivial activations.
        fan_in = int(np.prod(shape[1:]))
        v = z * (1.3 / np.sqrt(max(fan_in, 1)))
    elif shape == (1,):
        v = np.full(n, 0.2) + 0.01 * z  # the shared scalar PReLU slope
    else:
        v = 1.0 + 0.1 * z  # LayerNorm / BatchNorm gains
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def det_state_dict(shapes, seed=0, dtype=torch.float32):
    """shapes: mapping key -> shape (or tensors / anything with .shape)."""
    out = {}
    for k, s in shapes.items():
        shp = tuple(s.shape) if hasattr(s, "shape") else tuple(s)
        out[k] = det_tensor(k, shp, seed=seed, dtype=dtype)
    return out


def load_det_weights(module, seed=0):
    """Fill an nn.Module (reference, or product) in place; returns the state dict used."""
    sd = module.state_dict()
    new = det_state_dict(sd, seed=seed)
    new = {k: v.to(sd[k].dtype) for k, v in new.items()}
    module.load_state_dict(new, strict=True)
    return new


def det_input(name, shape, seed=1, lo=0.0, hi=1.0):
    """U[lo,hi) input tensor addressed by name (BASELINE.md §3 'Inputs')."""
    g = _rng(seed, "input:" + name)
    n = int(np.prod(shape))
    return torch.from_numpy((lo + (hi - lo) * g.random(n)).astype(np.float32).reshape(shape))


def det_labels(name, shape, num_classes=9, seed=1):
    g = _rng(seed, "labels:" + name)
    return torch.from_numpy(g.integers(0, num_classes, size=shape, dtype=np.int64))
