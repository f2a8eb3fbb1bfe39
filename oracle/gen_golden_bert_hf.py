"""Golden vectors for the whole BERT backbone (token ids -> contextual embeddings, pooled output)
from HuggingFace's BertModel (transformers, CPU, float64, random init -- nothing is downloaded).

TEST INFRASTRUCTURE ONLY.  HF's BertModel and the reference's BertModel are two implementations of
the same published TF checkpoint layout: the reference's converter maps the TF variables to its
names (scripts/conversion_toolkits/convert_tf_hub_model.py:150-179,223-226,449-494: query / key /
value kernels concatenated into ``attn_qkv``, ``attention/output/dense`` -> ``attention_proj``,
``intermediate/dense`` -> ``ffn.ffn_1``, ``output/dense`` -> ``ffn.ffn_2``, the two LayerNorms,
``embeddings/*`` and ``pooler/dense``); HF keeps the TF structure under its own spelling.  This
script composes the two renamings, runs HF, and stores ids, renamed weights and outputs;
tests check oracle/ref.py (and, on the GPU, BertModel) against them on the valid positions.

Run here (build container) only:  python oracle/gen_golden_bert_hf.py
"""
import os

import numpy as np
import torch
from transformers import BertConfig, BertModel

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def rename(sd, num_layers):
    n = lambda k: sd[k].detach().numpy().astype(np.float64)
    out = {
        'word_embed.weight': n('embeddings.word_embeddings.weight'),
        'token_type_embed.weight': n('embeddings.token_type_embeddings.weight'),
        'token_pos_embed._embed.weight': n('embeddings.position_embeddings.weight'),
        'embed_layer_norm.gamma': n('embeddings.LayerNorm.weight'),
        'embed_layer_norm.beta': n('embeddings.LayerNorm.bias'),
        'pooler.weight': n('pooler.dense.weight'),
        'pooler.bias': n('pooler.dense.bias'),
    }
    for i in range(num_layers):
        hf, gl = 'encoder.layer.{}.'.format(i), 'encoder.all_layers.{}.'.format(i)
        at = hf + 'attention.self.'
        out[gl + 'attn_qkv.weight'] = np.concatenate(
            [n(at + 'query.weight'), n(at + 'key.weight'), n(at + 'value.weight')], axis=0)
        out[gl + 'attn_qkv.bias'] = np.concatenate(
            [n(at + 'query.bias'), n(at + 'key.bias'), n(at + 'value.bias')], axis=0)
        for a, b in [('attention.output.dense', 'attention_proj'), ('intermediate.dense', 'ffn.ffn_1'),
                     ('output.dense', 'ffn.ffn_2')]:
            out[gl + b + '.weight'] = n(hf + a + '.weight')
            out[gl + b + '.bias'] = n(hf + a + '.bias')
        for a, b in [('attention.output.LayerNorm', 'layer_norm'),
                     ('output.LayerNorm', 'ffn.layer_norm')]:
            out[gl + b + '.gamma'] = n(hf + a + '.weight')
            out[gl + b + '.beta'] = n(hf + a + '.bias')
    return out


def run(name, seed, vocab=97, units=64, hidden=128, layers=2, heads=1, max_len=48, B=4, L=40):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    cfg = BertConfig(vocab_size=vocab, hidden_size=units, num_hidden_layers=layers,
                     num_attention_heads=heads, intermediate_size=hidden, hidden_act='gelu',
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                     max_position_embeddings=max_len, type_vocab_size=2, layer_norm_eps=1e-12)
    m = BertModel(cfg).double().eval()
    with torch.no_grad():
        for k, v in m.state_dict().items():     # HF zero-inits biases: make them non-trivial
            if k.endswith('.bias'):
                v.copy_(0.1 * torch.randn_like(v))
            if k.endswith('LayerNorm.weight'):
                v.copy_(1.0 + 0.1 * torch.randn_like(v))
    ids = rng.integers(0, vocab, (B, L))
    tt = rng.integers(0, 2, (B, L))
    vl = rng.integers(2, L, (B,)); vl[0] = L; vl[1] = 1
    mask = np.arange(L)[None, :] < vl[:, None]
    with torch.no_grad():
        o = m(input_ids=torch.from_numpy(ids), token_type_ids=torch.from_numpy(tt),
              attention_mask=torch.from_numpy(mask))
    np.savez_compressed(
        os.path.join(OUT, name + '.npz'), ids=ids, token_types=tt, valid_length=vl,
        sequence=o['last_hidden_state'].numpy(), pooled=o['pooler_output'].numpy(),
        num_layers=layers, num_heads=heads,
        **{'p:' + k: v for k, v in rename(m.state_dict(), layers).items()})
    print(name, 'written')


if __name__ == '__main__':
    run('bert_hf_d64', 21)                       # one 64-wide head: the fused attention kernel in bf16
    run('bert_hf_h4', 22, units=48, hidden=96, heads=4, layers=3)
