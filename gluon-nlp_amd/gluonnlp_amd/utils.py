"""Parameter files: read a LOCAL checkpoint into a gluonnlp_amd module.

The modules keep the reference's structural parameter names (``encoder.all_layers.0.attn_qkv.weight``
...), so a file written by Gluon's ``Block.save_parameters`` loads by name.  MXNet 2.x writes
numpy-mode parameters with ``npx.savez`` (mxnet/gluon/block.py ``save_parameters`` ->
``_mx_npx.savez(filename, **arg_dict)``): a zip of ``.npy`` members, i.e. the ``.npz`` container,
whatever the file's suffix (the model zoo names them ``model-xxxxxxxx.params``,
models/bert.py:141-206).  Also accepted: a torch ``state_dict`` file.  Both are read with loaders
that execute nothing from the file (``numpy.load(allow_pickle=False)``,
``torch.load(weights_only=True)``); the legacy MXNet 1.x NDArray binary format is not supported.
Nothing is downloaded."""
import zipfile

import numpy as np
import torch

__all__ = ['load_parameters', 'save_parameters', 'read_parameter_file']


def read_parameter_file(filename):
    """-> {name: torch.Tensor (CPU)} from an npz-container ``.params`` / ``.npz`` file or a torch
    ``state_dict`` file."""
    if zipfile.is_zipfile(filename):
        with zipfile.ZipFile(filename) as z:
            is_npz = all(n.endswith('.npy') for n in z.namelist())
        if is_npz:
            out = {}
            with np.load(filename, allow_pickle=False) as f:
                for k in f.files:
                    a = f[k]
                    if a.dtype == np.float16:
                        a = a.astype(np.float32)
                    out[k] = torch.from_numpy(np.ascontiguousarray(a))
            return out
    try:
        sd = torch.load(filename, map_location='cpu', weights_only=True)
    except Exception as e:   # noqa: BLE001
        raise ValueError('{} is neither an npz-container parameter file (MXNet 2.x '
                         'save_parameters) nor a torch state_dict readable with '
                         'weights_only=True: {}'.format(filename, e))
    if not isinstance(sd, dict):
        raise ValueError('{} does not hold a name -> tensor mapping'.format(filename))
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}


def load_parameters(module, filename, allow_missing=False, ignore_extra=False):
    """Gluon ``Block.load_parameters(filename, allow_missing, ignore_extra)``: copy the file's
    arrays into ``module``'s parameters by name, keeping each parameter's device and dtype
    (so a float32 checkpoint fills a bfloat16 model).  Shapes must match."""
    loaded = read_parameter_file(filename)
    own = dict(module.named_parameters())
    missing = [k for k in own if k not in loaded]
    extra = [k for k in loaded if k not in own]
    if missing and not allow_missing:
        raise AssertionError('Parameter(s) {} missing in file {} (set allow_missing=True to '
                             'ignore)'.format(missing[:5], filename))
    if extra and not ignore_extra:
        raise ValueError('Parameter(s) {} in file {} are not present in the module (set '
                         'ignore_extra=True to ignore)'.format(extra[:5], filename))
    with torch.no_grad():
        for k, p in own.items():
            if k not in loaded:
                continue
            v = loaded[k]
            if tuple(v.shape) != tuple(p.shape):
                raise ValueError('shape of {} in file is {}, the module expects {}'.format(
                    k, tuple(v.shape), tuple(p.shape)))
            p.data.copy_(v.to(p.dtype))
    return module


def save_parameters(module, filename):
    """Write ``module``'s parameters as MXNet 2.x ``save_parameters`` does: an npz container keyed
    by the structural names (bfloat16 is stored as float32, which is exact)."""
    arrays = {k: p.detach().to(torch.float32).cpu().numpy() for k, p in module.named_parameters()}
    with open(filename, 'wb') as f:    # np.savez would append '.npz' to a bare file name
        np.savez(f, **arrays)
