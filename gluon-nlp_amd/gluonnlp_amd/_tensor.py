"""Tensor plumbing for the host side: DLPack zero-copy in, caller-framework tensors out.

PyTorch is used only for device memory, streams and ``torch.distributed``; every computation is a
call into the C ABI.  Any object exposing ``__dlpack__`` (``mx.np.ndarray``, ``torch.Tensor``,
cupy ...) is accepted: it is viewed -- never copied -- as a torch tensor on the same device, and
the device pointer handed to the kernels is ``DLTensor.data + byte_offset``.
"""
import torch

from . import _lib

_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
_NAMES = {'float32': torch.float32, 'bfloat16': torch.bfloat16, 'bf16': torch.bfloat16,
          'fp32': torch.float32, torch.float32: torch.float32, torch.bfloat16: torch.bfloat16}


def torch_dtype(dtype):
    try:
        import numpy as np
        if dtype is np.float32 or dtype == np.dtype('float32'):
            return torch.float32
    except Exception:
        pass
    if dtype in _NAMES:
        return _NAMES[dtype]
    raise NotImplementedError('dtype={} is not supported (float32 and bfloat16 are)'.format(dtype))


def as_tensor(x):
    """View ``x`` as a torch tensor without copying (DLPack)."""
    if x is None or isinstance(x, torch.Tensor):
        return x
    if hasattr(x, '__dlpack__'):
        return torch.from_dlpack(x)
    if hasattr(x, 'to_dlpack_for_read'):          # MXNet 1.x style
        return torch.utils.dlpack.from_dlpack(x.to_dlpack_for_read())
    raise TypeError('expected a tensor exposing __dlpack__, got {}'.format(type(x)))


def like_caller(out, ref):
    """Return ``out`` (a torch tensor) in the framework of ``ref``.

    Contract: a ``torch.Tensor`` caller gets a ``torch.Tensor``; a caller whose top-level module
    (``mxnet`` -> ``mx.np``, ``cupy``, ``jax.numpy`` ...) exposes ``from_dlpack`` gets the output
    through that function (zero-copy; its errors propagate -- nothing is swallowed); a caller type
    with no such module function (a bare ``__dlpack__`` provider) gets the ``torch.Tensor``, which
    itself implements ``__dlpack__`` / ``__dlpack_device__``."""
    if ref is None or isinstance(ref, torch.Tensor):
        return out
    import sys
    mod = sys.modules.get(type(ref).__module__.split('.')[0])
    conv = getattr(mod, 'from_dlpack', None)
    if conv is None:
        conv = getattr(getattr(mod, 'np', None), 'from_dlpack', None)
    if conv is None:
        return out
    return conv(out)


def require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('gluonnlp_amd runs on MI355X device tensors only; got a {} tensor '
                               '(there is no CPU fallback)'.format(t.device))
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError('tensors live on different devices: {} vs {}'.format(dev, t.device))
    return dev


def code(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise NotImplementedError('dtype {} is not supported (float32 / bfloat16)'.format(t.dtype))


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream(dev):
    """Handle of torch's CURRENT stream on ``dev`` (what every launch is ordered on).  The raw
    accessor skips the Stream object ``torch.cuda.current_stream`` builds per call (4 us of a
    launch-bound decoding step's 10 us per op)."""
    if _raw_stream is not None and isinstance(dev, torch.device) and dev.index is not None:
        return _raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


def valid_length_arg(vl, dev=None):
    """-> (tensor kept alive, pointer, int dtype code)."""
    if vl is None:
        return None, None, 0
    vl = as_tensor(vl)
    if vl.dtype not in (torch.int32, torch.int64):
        vl = vl.to(torch.int64) if not vl.dtype.is_floating_point else vl.long()
    vl = vl.contiguous()
    return vl, vl.data_ptr(), (_lib.I64 if vl.dtype == torch.int64 else _lib.I32)


_WS = {}


def workspace(dev, nbytes):
    """Grow-only scratch buffer per (device, stream): every C call enqueues on the current stream,
    so two streams never share scratch, and a buffer that grows is released to the caching
    allocator under the stream it was allocated (and only ever used) on."""
    nbytes = int(nbytes)
    if nbytes == 0:
        return None
    key = (dev.type, dev.index, stream(dev))
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _WS[key] = buf
    return buf
