"""CPU-side tests: the C-ABI library loads and exports every symbol include/gluonnlp_amd.h
declares, the ctypes mirror of the structs matches the C layout, and the host logic (argument
checking, error types, parameter names, sharding arithmetic) behaves like the reference API.
No compute call is made (there is no GPU here and no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'gluonnlp_amd.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(gnlp_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from gluonnlp_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), 'symbol {} declared in the header is not exported'.format(n)
        assert n in _lib.PROTOTYPES, 'no ctypes prototype for {}'.format(n)
    assert lib.gnlp_version() >= 100
    assert isinstance(lib.gnlp_last_error(), bytes)


def test_ctypes_structs_match_c_layout(tmp_path):
    """sizeof / offsetof of every struct as gcc sees the header == the ctypes mirror."""
    from gluonnlp_amd import _lib
    structs = {'gnlp_attn_desc': _lib.AttnDesc, 'gnlp_ffn_params': _lib.FfnParams,
               'gnlp_ffn_cfg': _lib.FfnCfg, 'gnlp_encoder_layer_params': _lib.EncoderLayerParams,
               'gnlp_encoder_cfg': _lib.EncoderCfg, 'gnlp_t5_block_params': _lib.T5BlockParams,
               'gnlp_t5_cfg': _lib.T5Cfg, 'gnlp_encoder_layer_folded': _lib.EncoderLayerFolded}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gluonnlp_amd.h"',
             'int main(void) {']
    for cname, cls in structs.items():
        lines.append('printf("%s %zu\\n", "{0}", sizeof({0}));'.format(cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %zu\\n", "{0}", "{1}", offsetof({0}, {1}));'.format(
                cname, fname))
    lines += ['return 0; }']
    src = tmp_path / 'abi.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'abi'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split('\n')
    got = dict(l.split() for l in out if l.strip())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got['{}.{}'.format(cname, fname)]) == getattr(cls, fname).offset, \
                '{}.{}'.format(cname, fname)


def test_missing_library_fails_loudly(monkeypatch):
    from gluonnlp_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libgluonnlp_amd.so')
    with pytest.raises(ImportError, match='no CPU fallback'):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'gluon-nlp_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(d, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f


def test_no_cpu_fallback_and_error_types():
    import gluonnlp_amd as g
    ac, layers = g.attention_cell, g.layers
    x = torch.zeros(2, 4, 8)
    vl = torch.tensor([4, 2])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ac.gen_self_attn_mask(x, vl)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        layers.LayerNorm(in_channels=8)(x)
    # reference error behaviour is raised by the shim before reaching C
    with pytest.raises(NotImplementedError):
        ac.gen_self_attn_mask(x, vl, layout='NC')
    with pytest.raises(NotImplementedError):
        ac.gen_mem_attn_mask(x, vl, x, vl, layout='XX')
    q = torch.zeros(2, 2, 4, 8)
    with pytest.raises(NotImplementedError, match='query_head_units'):
        ac.multi_head_dot_attn(q, q, q, scaled=True)
    with pytest.raises(NotImplementedError, match='layout'):
        ac.multi_head_dot_attn(q, q, q, scaled=False, layout='TKN')
    with pytest.raises(AssertionError):
        ac.MultiHeadAttentionCell(query_units=7, num_heads=2)
    with pytest.raises(AssertionError):
        ac.MultiHeadAttentionCell(query_units=8, num_heads=2, layout='XYZ')
    with pytest.raises(ValueError):
        layers.GELU('exact')
    with pytest.raises(NotImplementedError):
        layers.get_activation('mish')
    with pytest.raises(NotImplementedError):
        layers.get_norm_layer('batch_norm2')
    with pytest.raises(NotImplementedError):
        layers.get_norm_layer(3)
    with pytest.raises(AssertionError):
        g.models.TransformerEncoderLayer(units=10, num_heads=3)
    with pytest.raises(AssertionError):
        g.models.TransformerEncoderLayer(layout='NCT')
    with pytest.raises(AssertionError):
        g.models.BertTransformer(units=10, num_heads=4)
    with pytest.raises(AssertionError):
        g.models.T5Block(32, 8, 64, False, activation='swiglu')
    with pytest.raises(AssertionError):
        g.op.relative_position_bucket(torch.zeros(3, dtype=torch.int32), True, num_buckets=31)
    assert layers.get_activation(None)(5) == 5
    assert ac.MultiHeadAttentionCell(query_units=8, num_heads=2).layout == 'NTK'


def test_parameter_names_match_reference():
    """State-dict keys equal the MXNet parameter names (convert_tf_hub_model.py:150-179,449-494,
    convert_t5.py:49-59) so reference checkpoints load unchanged."""
    from gluonnlp_amd.models import (BertTransformer, ElectraEncoder, RobertaEncoder, T5Encoder,
                                     TransformerEncoder)
    keys = set(BertTransformer(units=8, hidden_size=16, num_layers=1, num_heads=2).state_dict())
    assert keys == {'all_layers.0.' + k for k in [
        'attn_qkv.weight', 'attn_qkv.bias', 'attention_proj.weight', 'attention_proj.bias',
        'layer_norm.gamma', 'layer_norm.beta', 'ffn.ffn_1.weight', 'ffn.ffn_1.bias',
        'ffn.ffn_2.weight', 'ffn.ffn_2.bias', 'ffn.layer_norm.gamma', 'ffn.layer_norm.beta']}
    assert all(k.startswith('all_encoder_layers.0.') for k in
               ElectraEncoder(units=8, hidden_size=16, num_layers=1, num_heads=2).state_dict())
    assert all(k.startswith('all_layers.0.') for k in
               RobertaEncoder(units=8, hidden_size=16, num_layers=1, num_heads=2).state_dict())
    te = TransformerEncoder(num_layers=2, units=8, hidden_size=16, num_heads=2, pre_norm=True,
                            data_norm=True)
    assert {'ln_final.gamma', 'ln_data.beta', 'layers.1.attn_qkv.weight'} <= set(te.state_dict())
    t5 = set(T5Encoder(16, 4, 32, num_layers=1, num_heads=4, activation='gated-gelu').state_dict())
    assert {'relative_position_encoder._rel_pos_embed.weight', 'final_layer_norm.gamma',
            'layers.0.self_attn_layer_norm.gamma', 'layers.0.self_attn_q.weight',
            'layers.0.self_attn_k.weight', 'layers.0.self_attn_v.weight',
            'layers.0.self_attn_proj.weight', 'layers.0.ffn.ffn_1.weight',
            'layers.0.ffn.gated_ffn_1.weight', 'layers.0.ffn.ffn_2.weight',
            'layers.0.ffn.layer_norm.gamma'} <= t5
    assert 'layers.0.self_attn_q.bias' not in t5 and 'layers.0.ffn.ffn_1.bias' not in t5
    # backbones (models/bert.py:376-400, roberta.py:273-313, electra.py:366-397)
    from gluonnlp_amd.models import BertModel, ElectraModel, RobertaModel
    kw = dict(vocab_size=20, units=8, hidden_size=16, num_layers=1, num_heads=2, max_length=6)
    bm = {k for k in BertModel(**kw).state_dict() if not k.startswith('encoder.')}
    assert bm == {'word_embed.weight', 'token_type_embed.weight', 'token_pos_embed._embed.weight',
                  'embed_layer_norm.gamma', 'embed_layer_norm.beta', 'pooler.weight',
                  'pooler.bias'}
    assert 'encoder.all_layers.0.attn_qkv.weight' in BertModel(**kw).state_dict()
    rm = {k for k in RobertaModel(classifier_activation=True, **kw).state_dict()
          if not k.startswith('encoder.')}
    assert rm == {'word_embed.weight', 'pos_embed._embed.weight', 'embed_ln.gamma',
                  'embed_ln.beta', 'pooler.weight', 'pooler.bias'}
    assert not any(k.startswith(('pooler', 'embed_ln')) for k in
                   RobertaModel(encoder_normalize_before=False, **kw).state_dict())
    em = {k for k in ElectraModel(embed_size=4, **kw).state_dict() if not k.startswith('encoder.')}
    assert em == {'word_embed.weight', 'token_type_embed.weight', 'token_pos_embed._embed.weight',
                  'embed_layer_norm.gamma', 'embed_layer_norm.beta',
                  'embed_factorized_proj.weight', 'embed_factorized_proj.bias'}
    assert 'embed_factorized_proj.weight' not in ElectraModel(embed_size=8, **kw).state_dict()
    sin = BertModel(pos_embed_type='sinusoidal', **kw).state_dict()       # layers.py:378-380
    assert 'token_pos_embed._embed.base_mult' in sin and 'token_pos_embed._embed.weight' not in sin
    with pytest.raises(NotImplementedError):
        BertModel(pos_embed_type='rotary', **kw)
    m = BertTransformer(units=8, hidden_size=16, num_layers=1, num_heads=2)
    assert m.all_layers[0].attn_qkv.weight.shape == (24, 8)      # (3U, U): Q | K | V thirds
    assert m.all_layers[0].ffn.ffn_1.weight.shape == (16, 8)


def test_trunc_norm_matches_reference_distribution():
    """initializer.py:52-81: clipped at +-2 sigma."""
    from gluonnlp_amd.layers import trunc_norm_
    t = trunc_norm_(torch.empty(20000), 0.02)
    assert float(t.abs().max()) <= 0.04 + 1e-7
    assert abs(float(t.std()) - 0.02 * 0.88) < 0.002


def test_shard_bounds():
    from gluonnlp_amd.parallel import shard_bounds
    for n in [0, 1, 7, 64, 65]:
        for w in [1, 2, 3, 8]:
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_one_wave_per_simd_gemm_static_checks(tmp_path):
    """gemm_bf16_w4.hip relies on what the compiler makes of it: the 256 accumulators must live in
    AGPRs and nothing may spill (scratch traffic counts in vmcnt, which the kernel waits on by
    hand), and every K-tile body must stay 128 MFMAs + 32 ds_read_b128 + 16 buffer_load-to-LDS with
    no accumulator copies, no other vector-memory instruction and exactly one vmcnt wait,
    `vmcnt(11)`: the 8 x-pieces and 3 weight pieces the K-tile has requested by then may stay in
    flight (256x192 tiles: 192 accumulators, 96 + 28 + 14, the same wait).  Checked on the
    gfx950 assembly of every instantiation (hipcc cross-compiles)."""
    hipcc = '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, 'gluon-nlp_amd', 'csrc', 'gemm_bf16_w4.hip')
    asm = str(tmp_path / 'w4.s')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S',
                    '--cuda-device-only', '-o', asm, src], check=True, capture_output=True)
    text = open(asm).read()
    names = re.findall(r'\.name:\s+(_ZN4gnlp20dense_bf16_w4_kernel\w+)', text)
    # 4 activations x {residual, none} x {256, 192 wide} + the instantiations with a LayerNorm
    # folded in: 3 activations (not the tanh GELU) x 2 widths (FOLD 1) and the identity + residual
    # x 2 widths (FOLD 2)
    assert len(names) == 24
    narrow = [re.search(r'Li6ELi\dEEEvNS_11DenseW4ArgsE$', n) is not None for n in names]   # NJ = 6
    assert sum(narrow) == 12
    # (scalar registers may be parked in lanes of a VGPR -- v_writelane / v_readlane, no memory
    # traffic; the FOLD 2 instantiations do that for a few per-tile values -- but nothing may go
    # to scratch memory)
    for key, most in [('vgpr_spill_count', 0), ('sgpr_spill_count', 8),
                      ('private_segment_fixed_size', 0)]:
        vals = [int(v) for v in re.findall(r'\.%s:\s+(\d+)' % key, text)]
        assert len(vals) == len(names) and all(v <= most for v in vals), (key, vals)
    agprs = [int(v) for v in re.findall(r'\.agpr_count:\s+(\d+)', text)]
    # (192-wide tiles leave 64 AGPRs free and the allocator parks drain values there; the K-tile
    # bodies are checked for accumulator copies below)
    assert all((192 <= g <= 256) if n else g == 256 for g, n in zip(agprs, narrow)), agprs
    bodies = 0
    for name, nar in zip(names, narrow):
        n_mfma, n_rd, n_dma = (96, 28, 14) if nar else (128, 32, 16)
        body = text[text.index('\n' + name + ':'):]
        body = body[:body.index('s_endpgm')]
        lines = [l.split(';')[0].strip() for l in body.splitlines()]
        lines = [l for l in lines if l and not l.startswith('.') or l.startswith('.LBB')]
        # split into straight-line runs and keep those that hold a whole K-tile (128 MFMAs)
        run = []
        for l in lines + ['.LBB_end:']:
            if l.startswith('.LBB') or l.startswith('s_cbranch') or l.startswith('s_branch'):
                mf = [x for x in run if x.startswith('v_mfma')]
                if len(mf) >= n_mfma:
                    assert len(mf) == n_mfma, (name, len(mf))
                    ops = [x.split()[0] for x in run]
                    assert ops.count('ds_read_b128') == n_rd, (name, ops.count('ds_read_b128'))
                    dma = [x for x in run if x.startswith('buffer_load_dwordx4') and 'lds' in x]
                    assert len(dma) == n_dma and all('offen' in x for x in dma), (name, len(dma))
                    assert ops.count('s_barrier') == 3, name
                    vm = [x for x in run if x.startswith('s_waitcnt') and 'vmcnt' in x]
                    assert vm == ['s_waitcnt vmcnt(11)'], (name, vm)
                    assert not any(o.startswith('scratch_') or o.startswith('global_') or
                                   o.startswith('ds_write') for o in ops), name
                    # (no accumulator copies; the 192-wide tanh-GELU instantiation fetches two
                    # loop invariants it parked in spare AGPRs)
                    assert sum(o.startswith('v_accvgpr') for o in ops) <= 2, name
                    valu = [o for o in ops if o.startswith('v_') and not o.startswith('v_mfma')]
                    assert len(valu) <= 8, (name, valu)   # fragment-read bases of the two buffers
                    bodies += 1
                run = []
            else:
                run.append(l)
    # the tile's first K-tile (C = 0) and the loop body; the instantiations with a folded LayerNorm
    # branch to their statistics loads ahead of the last K-tile and the compiler may peel that
    # K-tile into a third body (every body found has passed the checks above)
    n_fold = sum(re.search(r'ELi[12]EEEvNS_11DenseW4ArgsE$', n) is not None for n in names)
    assert n_fold == 8 and 2 * len(names) <= bodies <= 2 * len(names) + n_fold, bodies
    # the residual ring's inline-asm global loads: no register they write is named before a vmcnt
    # wait (the compiler believes an asm output is valid at once)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_async_regs.py'), asm],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert ', 0 problems' in r.stdout and 'checked 0 ' not in r.stdout


def test_parameter_files_round_trip(tmp_path):
    """a Gluon-style ``.params`` file (npz container keyed by the structural
    parameter names, what MXNet 2.x save_parameters writes) and a torch state_dict both load by
    name into a model of another dtype; missing / extra / mis-shaped entries raise like Gluon's
    load_parameters."""
    from gluonnlp_amd.models import BertModel
    from gluonnlp_amd.utils import load_parameters, read_parameter_file, save_parameters
    kw = dict(vocab_size=20, units=8, hidden_size=16, num_layers=1, num_heads=2, max_length=6)
    torch.manual_seed(3)
    src = BertModel(**kw)
    for p in src.parameters():
        p.data.normal_(0, 0.1)
    f = str(tmp_path / 'model-0000.params')
    save_parameters(src, f)
    got = read_parameter_file(f)
    assert set(got) == set(dict(src.named_parameters()))
    dst = load_parameters(BertModel(dtype='bfloat16', **kw), f)
    for (k, a), (_, b) in zip(src.named_parameters(), dst.named_parameters()):
        assert b.dtype == torch.bfloat16 and torch.equal(a.data.to(torch.bfloat16), b.data), k
    # a file written with numpy directly (float16 members are widened)
    f2 = str(tmp_path / 'half.params')
    with open(f2, 'wb') as fh:
        np.savez(fh, **{k: v.numpy().astype(np.float16) for k, v in got.items()})
    half = load_parameters(BertModel(**kw), f2)
    assert torch.allclose(half.pooler.weight.data, src.pooler.weight.data, atol=1e-3)
    # torch state_dict
    f3 = str(tmp_path / 'model.pt')
    torch.save(src.state_dict(), f3)
    dst3 = load_parameters(BertModel(**kw), f3)
    assert torch.equal(dst3.word_embed.weight.data, src.word_embed.weight.data)
    # error behaviour of Block.load_parameters
    part = {k: v for k, v in src.state_dict().items() if not k.startswith('pooler')}
    f4 = str(tmp_path / 'part.pt')
    torch.save(part, f4)
    with pytest.raises(AssertionError):
        load_parameters(BertModel(**kw), f4)
    load_parameters(BertModel(**kw), f4, allow_missing=True)
    with pytest.raises(ValueError):
        load_parameters(BertModel(use_pooler=False, **kw), f3)
    load_parameters(BertModel(use_pooler=False, **kw), f3, ignore_extra=True)
    with pytest.raises(ValueError):
        load_parameters(BertModel(**dict(kw, units=16, hidden_size=16)), f3)
    bad = str(tmp_path / 'bad.params')
    open(bad, 'wb').write(b'\x12\x01\x00\x00 not a parameter file')
    with pytest.raises(ValueError):
        read_parameter_file(bad)


def test_reference_layout_parameter_fixture_loads_by_name():
    """tests/golden/bert_hf_h4.params (oracle/gen_params_fixture.py): a parameter file in the
    reference's on-disk layout -- MXNet 2.x save_parameters' npz container under a .params suffix,
    float32, keyed by BertModel's structural names spelled out from the reference's class
    definitions -- and the same arrays as a torch state_dict.  Both load into BertModel by name
    with nothing missing and nothing extra, into float32 and bfloat16 models alike, and hold the
    arrays of the HuggingFace golden they were made from (the GPU test checks the outputs)."""
    import zipfile
    from gluonnlp_amd.models import BertModel
    from gluonnlp_amd.utils import load_parameters, read_parameter_file
    G = os.path.join(ROOT, 'tests', 'golden')
    g = np.load(os.path.join(G, 'bert_hf_h4.npz'))
    want = {k[2:]: g[k].astype(np.float32) for k in g.files if k.startswith('p:')}
    V, U = want['word_embed.weight'].shape
    kw = dict(vocab_size=V, units=U, hidden_size=want['encoder.all_layers.0.ffn.ffn_1.weight'].shape[0],
              num_layers=int(g['num_layers']), num_heads=int(g['num_heads']),
              max_length=want['token_pos_embed._embed.weight'].shape[0])
    f = os.path.join(G, 'bert_hf_h4.params')
    with zipfile.ZipFile(f) as z:     # the container itself: .npy members, stored uncompressed
        assert all(n.endswith('.npy') for n in z.namelist())
        assert all(i.compress_type == zipfile.ZIP_STORED for i in z.infolist())
    for path in (f, os.path.join(G, 'bert_hf_h4.state_dict.pt')):
        loaded = read_parameter_file(path)
        assert set(loaded) == set(want)
        for dn, dt in (('float32', torch.float32), ('bfloat16', torch.bfloat16)):
            m = load_parameters(BertModel(dtype=dn, **kw), path)    # strict: no missing, no extra
            own = dict(m.named_parameters())
            assert set(own) == set(want)
            for k, v in want.items():
                assert own[k].dtype == dt
                assert torch.equal(own[k].data, torch.from_numpy(v).to(dt)), k
