"""ctypes binding of libgluonnlp_amd.so (the C ABI declared in include/gluonnlp_amd.h).

The product path has NO CPU fallback: if the shared library is missing or a call is made with
host tensors, an error is raised.  Build the library with ``python __graft_entry__.py`` /
``make -C gluon-nlp_amd``.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_uint8, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GNLP_LIB_PATH') or os.path.join(_HERE, 'libgluonnlp_amd.so')

# enums (include/gluonnlp_amd.h)
F32, BF16 = 0, 1
I32, I64 = 0, 1
ATTN_FULL, ATTN_CAUSAL = 0, 1
ACT = {None: 0, 'identity': 0, 'relu': 1, 'gelu': 2, 'gelu(tanh)': 3, 'gelu(sigmoid)': 4,
       'sigmoid': 5, 'tanh': 6, 'leaky': 7, 'elu': 8, 'softrelu': 9, 'softsign': 10}
NORM = {'layer_norm': 0, 'rms_norm': 1, 'no_norm': 2}
COMBINE_ADD, COMBINE_MUL = 0, 1
NOALLOC_SCRATCH_BYTES = 64 << 20
ATTN_KERNEL_NONE, ATTN_KERNEL_FUSED_BF16, ATTN_KERNEL_GENERAL_MFMA, ATTN_KERNEL_UNFUSED = 0, 1, 2, 3


class AttnDesc(Structure):
    _fields_ = [('B', c_int64), ('H', c_int64), ('Lq', c_int64), ('Lm', c_int64),
                ('dk', c_int64), ('dv', c_int64),
                ('q_stride', c_int64 * 3), ('k_stride', c_int64 * 3),
                ('v_stride', c_int64 * 3), ('o_stride', c_int64 * 3), ('causal', c_int64),
                ('mem_lengths', c_int64)]


class AttnGradDesc(Structure):
    _fields_ = [('g_stride', c_int64 * 3), ('dq_stride', c_int64 * 3),
                ('dk_stride', c_int64 * 3), ('dv_stride', c_int64 * 3),
                ('fwd_ctx', c_void_p), ('fwd_ctx_lo', c_void_p), ('fwd_lse', c_void_p)]


class FfnParams(Structure):
    _fields_ = [('ffn1_w', c_void_p), ('ffn1_b', c_void_p), ('gate_w', c_void_p),
                ('gate_b', c_void_p), ('ffn2_w', c_void_p), ('ffn2_b', c_void_p),
                ('ln_g', c_void_p), ('ln_b', c_void_p)]


class FfnCfg(Structure):
    _fields_ = [('units', c_int64), ('hidden', c_int64), ('act', c_int), ('gated', c_int),
                ('norm_kind', c_int), ('pre_norm', c_int), ('dtype', c_int), ('ln_eps', c_float)]


class EncoderLayerParams(Structure):
    _fields_ = [('qkv_w', c_void_p), ('qkv_b', c_void_p), ('proj_w', c_void_p),
                ('proj_b', c_void_p), ('ln_g', c_void_p), ('ln_b', c_void_p), ('ffn', FfnParams)]


class EncoderLayerFolded(Structure):
    _fields_ = [('qkv_wf', c_void_p), ('qkv_c', c_void_p), ('qkv_bf', c_void_p),
                ('ffn1_wf', c_void_p), ('ffn1_c', c_void_p), ('ffn1_bf', c_void_p)]


class EncoderCfg(Structure):
    _fields_ = [('units', c_int64), ('hidden', c_int64), ('heads', c_int64), ('act', c_int),
                ('pre_norm', c_int), ('dtype', c_int), ('time_major', c_int), ('ln_eps', c_float)]


class T5BlockParams(Structure):
    _fields_ = [('attn_ln_g', c_void_p), ('attn_ln_b', c_void_p), ('q_w', c_void_p),
                ('k_w', c_void_p), ('v_w', c_void_p), ('proj_w', c_void_p), ('ffn', FfnParams),
                ('qkv_w', c_void_p)]


class T5BlockFolded(Structure):
    _fields_ = [('qkv_wf', c_void_p), ('qkv_c', c_void_p), ('qkv_bf', c_void_p),
                ('ffn1_wf', c_void_p), ('ffn1_c', c_void_p), ('ffn1_bf', c_void_p)]


class T5Cfg(Structure):
    _fields_ = [('d_model', c_int64), ('d_kv', c_int64), ('d_ff', c_int64), ('heads', c_int64),
                ('gated', c_int), ('dtype', c_int), ('time_major', c_int), ('ln_eps', c_float),
                ('bias_toeplitz', c_int)]


I64x3 = c_int64 * 3
I64x4 = c_int64 * 4

# name -> (restype, argtypes); every symbol declared in include/gluonnlp_amd.h
PROTOTYPES = {
    'gnlp_version': (c_int, []),
    'gnlp_last_error': (c_char_p, []),
    'gnlp_device_count': (c_int, []),
    'gnlp_reload_knobs': (None, []),
    'gnlp_attn_last_kernel': (c_int, []),
    'gnlp_stack_last_packed': (c_int, []),
    'gnlp_gen_self_attn_mask': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int,
                                        c_void_p]),
    'gnlp_gen_mem_attn_mask': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64,
                                       c_int64, c_void_p]),
    'gnlp_sequence_mask': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64,
                                   c_int64, c_int, c_void_p]),
    'gnlp_masked_softmax': (c_int, [c_void_p, c_void_p, c_int, c_void_p, POINTER(c_int64),
                                    POINTER(c_int64), c_float, c_int, c_void_p]),
    'gnlp_l2_normalize': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_float, c_void_p]),
    'gnlp_activation': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    'gnlp_norm': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                          c_int64, c_int64, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p]),
    'gnlp_embedding_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int, c_float, c_int64,
                                       c_int64, c_int64, c_int64, c_int64, c_int64, c_int,
                                       c_void_p]),
    'gnlp_dense_epilogue': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int,
                                    c_void_p]),
    'gnlp_dense_epilogue_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int, c_int64, c_int, c_int, c_void_p]),
    'gnlp_dense': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                           c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    'gnlp_attn_workspace_bytes': (c_size_t, [POINTER(AttnDesc), c_int, c_int, c_int]),
    'gnlp_multi_head_dot_attn': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, POINTER(AttnDesc), c_int, c_void_p,
                                         POINTER(c_int64), c_void_p, c_int, c_void_p,
                                         POINTER(c_int64), c_float, c_void_p, c_size_t, c_void_p]),
    'gnlp_multi_head_dot_attn_lse': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, POINTER(AttnDesc), c_int, c_void_p, POINTER(c_int64),
                                             c_void_p, c_int, c_void_p, POINTER(c_int64), c_float,
                                             c_void_p]),
    'gnlp_attn_backward_workspace_bytes': (c_size_t, [POINTER(AttnDesc)]),
    'gnlp_multi_head_dot_attn_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                                  POINTER(c_int64), c_void_p, c_void_p, c_void_p,
                                                  c_void_p, POINTER(AttnDesc),
                                                  POINTER(AttnGradDesc), c_int, c_void_p,
                                                  POINTER(c_int64), c_void_p, c_int, c_void_p,
                                                  POINTER(c_int64), c_float, c_void_p, c_size_t,
                                                  c_void_p]),
    'gnlp_norm_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int, c_int, c_float, c_int64, c_int64, c_void_p]),
    'gnlp_l2_normalize_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64,
                                           c_float, c_void_p]),
    'gnlp_cast_from_f32': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    'gnlp_rel_attn_score_backward': (c_int, [c_void_p, POINTER(c_int64), c_void_p, c_void_p,
                                             c_void_p, POINTER(c_int64), c_void_p, c_void_p, c_int,
                                             c_int64, c_int64, c_int64, c_int64, c_int64, c_int64,
                                             c_int32, c_int32, c_int32, c_int, c_void_p]),
    'gnlp_bucket_embedding_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                               c_int64, c_int, c_int, c_int, c_void_p]),
    'gnlp_colsum': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p]),
    'gnlp_gen_rel_position': (c_int, [c_void_p, c_int64, c_void_p]),
    'gnlp_relative_position_bucket': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                              c_void_p]),
    'gnlp_bucket_embedding': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64,
                                      c_int, c_int, c_int, c_void_p]),
    'gnlp_t5_rel_bias': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int,
                                 c_int, c_int, c_void_p]),
    'gnlp_t5_rel_bias_table': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int,
                                       c_int, c_int, c_void_p]),
    'gnlp_sinusoidal_embedding': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64,
                                          c_void_p]),
    'gnlp_rel_attn_score': (c_int, [c_void_p, c_void_p, POINTER(c_int64), c_void_p, c_void_p, c_int,
                                    c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int32,
                                    c_int32, c_int32, c_int, c_void_p]),
    'gnlp_profile_dense_arm': (c_int, [c_int64, c_int64, c_int64, c_int]),
    'gnlp_profile_dense_read': (c_int, [c_void_p, c_int]),
    'gnlp_profile_dense_disarm': (c_int, []),
    'gnlp_ffn_workspace_bytes': (c_size_t, [POINTER(FfnCfg), c_int64]),
    'gnlp_positionwise_ffn': (c_int, [c_void_p, c_void_p, POINTER(FfnParams), POINTER(FfnCfg),
                                      c_int64, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p,
                                      c_size_t, c_void_p]),
    'gnlp_encoder_workspace_bytes': (c_size_t, [POINTER(EncoderCfg), c_int64, c_int64, c_int]),
    'gnlp_encoder_layer_forward': (c_int, [c_void_p, c_void_p, POINTER(EncoderLayerParams),
                                           POINTER(EncoderCfg), c_int64, c_int64, c_void_p,
                                           c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t,
                                           c_void_p]),
    'gnlp_encoder_layer_forward_noalloc': (c_int, [c_void_p, c_void_p,
                                                   POINTER(EncoderLayerParams),
                                                   POINTER(EncoderCfg), c_int64, c_int64, c_void_p,
                                                   c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'gnlp_encoder_stack_forward': (c_int, [c_void_p, c_void_p, POINTER(EncoderLayerParams), c_int,
                                           POINTER(EncoderCfg), c_int64, c_int64, c_void_p, c_int,
                                           c_int, POINTER(c_void_p), c_void_p, c_size_t,
                                           c_void_p]),
    'gnlp_encoder_stack_forward_folded': (c_int, [c_void_p, c_void_p, POINTER(EncoderLayerParams),
                                                  POINTER(EncoderLayerFolded), c_int,
                                                  POINTER(EncoderCfg), c_int64, c_int64, c_void_p,
                                                  c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'gnlp_stack_last_folded': (c_int, []),
    'gnlp_fold_norm_dense': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int, c_int64, c_int64, c_int64, c_float,
                                     c_void_p]),
    'gnlp_row_stats': (c_int, [c_void_p, c_void_p, c_int, c_float, c_int64, c_int64, c_void_p]),
    'gnlp_t5_workspace_bytes': (c_size_t, [POINTER(T5Cfg), c_int64, c_int64]),
    'gnlp_t5_encoder_forward': (c_int, [c_void_p, c_void_p, POINTER(T5BlockParams), c_int,
                                        c_void_p, c_void_p, POINTER(T5Cfg), c_int64, c_int64,
                                        c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'gnlp_t5_encoder_forward_folded': (c_int, [c_void_p, c_void_p, POINTER(T5BlockParams),
                                               POINTER(T5BlockFolded), c_int, c_void_p, c_void_p,
                                               POINTER(T5Cfg), c_int64, c_int64, c_void_p, c_int,
                                               c_void_p, c_void_p, c_size_t, c_void_p]),
}

_lib = None


class GnlpError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


def load():
    """Load the shared library (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libgluonnlp_amd.so is not built ({}). Run `python __graft_entry__.py` or '
            '`make -C gluon-nlp_amd`; there is no CPU fallback.'.format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().gnlp_last_error()
        raise GnlpError('gluonnlp_amd error {}: {}'.format(rc, (msg or b'').decode()))


def set_knob(name, value):
    """Developer helper: set (or, with None, unset) a GNLP_* environment knob and make the library
    re-read its knobs (they are read once per process, never on a launch path)."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    load().gnlp_reload_knobs()
