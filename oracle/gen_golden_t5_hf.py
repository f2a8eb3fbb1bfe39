"""Golden vectors for the T5 encoder-decoder from HuggingFace's T5Model (transformers, CPU,
float64, random init -- nothing is downloaded).

TEST INFRASTRUCTURE ONLY.  Why HF is a valid pin for the reference here: the reference's own
conversion script loads HF weights into its T5Model by a pure renaming (PARAM_MAP,
scripts/conversion_toolkits/convert_t5.py:45-60,121-186) and its ``test_conversion`` asserts that
the two forwards agree on the valid target positions (:203-232).  This script re-states that
renaming, runs HF with the masks that test builds, and stores inputs, renamed weights and HF's
outputs; tests/test_oracle_golden.py checks oracle/ref.py's t5_model against them at 5e-6
(float64 everywhere except HF's float32 mean of squares in T5LayerNorm), far inside the
reference's own 1e-3.

Run here (build container) only:  python oracle/gen_golden_t5_hf.py
"""
import os

import numpy as np
import torch
from transformers import T5Config, T5Model

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def rename(sd, num_layers, gated):
    """HF state_dict -> the reference's parameter names (convert_t5.py:45-60)."""
    out = {'input_embedding_layer.weight': sd['shared.weight']}
    for st in ('encoder', 'decoder'):
        out[st + '.relative_position_encoder._rel_pos_embed.weight'] = sd[
            st + '.block.0.layer.0.SelfAttention.relative_attention_bias.weight']
        out[st + '.final_layer_norm.gamma'] = sd[st + '.final_layer_norm.weight']
        norms = (['self_attn_layer_norm', 'ffn.layer_norm'] if st == 'encoder' else
                 ['self_attn_layer_norm', 'cross_attn_layer_norm', 'ffn.layer_norm'])
        ffn_at = 1 if st == 'encoder' else 2
        for i in range(num_layers):
            hf, gl = '{}.block.{}.layer.'.format(st, i), '{}.layers.{}.'.format(st, i)
            for j, name in enumerate(norms):
                out[gl + name + '.gamma'] = sd['{}{}.layer_norm.weight'.format(hf, j)]
            for x in 'qkv':
                out[gl + 'self_attn_{}.weight'.format(x)] = sd[hf + '0.SelfAttention.{}.weight'.format(x)]
                if st == 'decoder':
                    out[gl + 'cross_attn_{}.weight'.format(x)] = sd[
                        hf + '1.EncDecAttention.{}.weight'.format(x)]
            out[gl + 'self_attn_proj.weight'] = sd[hf + '0.SelfAttention.o.weight']
            if st == 'decoder':
                out[gl + 'cross_attn_proj.weight'] = sd[hf + '1.EncDecAttention.o.weight']
            pairs = ([('wi_0', 'gated_ffn_1'), ('wi_1', 'ffn_1'), ('wo', 'ffn_2')] if gated
                     else [('wi', 'ffn_1'), ('wo', 'ffn_2')])
            for a, b in pairs:
                out[gl + 'ffn.{}.weight'.format(b)] = sd['{}{}.DenseReluDense.{}.weight'.format(
                    hf, ffn_at, a)]
    return {k: v.detach().numpy().astype(np.float64) for k, v in out.items()}


def run(name, gated, seed, vocab=61, d_model=32, d_kv=8, d_ff=64, layers=2, heads=4, B=5,
        Ls=40, Lt=12):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    cfg = T5Config(vocab_size=vocab, d_model=d_model, d_kv=d_kv, d_ff=d_ff, num_layers=layers,
                   num_decoder_layers=layers, num_heads=heads, dropout_rate=0.0,
                   feed_forward_proj='gated-gelu' if gated else 'relu',
                   relative_attention_num_buckets=32, relative_attention_max_distance=128,
                   layer_norm_epsilon=1e-6)
    m = T5Model(cfg).double().eval()
    with torch.no_grad():
        for k, v in m.state_dict().items():     # HF zero-mean inits are fine; make the gains non-trivial
            if k.endswith('layer_norm.weight'):
                v.copy_(1.0 + 0.1 * torch.randn_like(v))
            if k.endswith('relative_attention_bias.weight'):
                v.copy_(0.5 * torch.randn_like(v))
    src = rng.integers(1, vocab, (B, Ls))
    tgt = rng.integers(1, vocab, (B, Lt))
    # Ls > 32 reaches the logarithmic buckets; lengths as in convert_t5.py:210-213 plus the extremes
    svl = rng.integers(Ls // 2, Ls, (B,)); svl[0] = Ls; svl[1] = 1
    tvl = rng.integers(Lt // 2, Lt, (B,)); tvl[0] = Lt; tvl[2] = 1
    enc_mask = np.arange(Ls)[None, :] < svl[:, None]
    dec_mask = np.arange(Lt)[None, :] < tvl[:, None]
    with torch.no_grad():
        o = m(input_ids=torch.from_numpy(src), attention_mask=torch.from_numpy(enc_mask),
              decoder_input_ids=torch.from_numpy(tgt),
              decoder_attention_mask=torch.from_numpy(dec_mask))
    params = rename(m.state_dict(), layers, gated)
    np.savez_compressed(
        os.path.join(OUT, name + '.npz'), src=src, tgt=tgt, src_valid_length=svl,
        tgt_valid_length=tvl, enc_out=o['encoder_last_hidden_state'].numpy(),
        dec_out=o['last_hidden_state'].numpy(), num_layers=layers, num_heads=heads,
        gated=int(gated), **{'p:' + k: v for k, v in params.items()})
    print(name, 'written')


if __name__ == '__main__':
    run('t5_hf_relu', False, 11)
    run('t5_hf_gated', True, 12)
