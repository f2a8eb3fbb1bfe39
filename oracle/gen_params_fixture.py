"""Parameter-file fixtures in the reference's own on-disk layout.

TEST INFRASTRUCTURE ONLY.  What the reference's converter leaves on disk for a BERT backbone
(scripts/conversion_toolkits/convert_tf_hub_model.py:449-494 builds the arrays, :589-591 calls
``model.backbone_model.save_parameters(.../model.params)``) is a Gluon 2.x parameter file: MXNet
2.x ``Block.save_parameters`` writes numpy-mode parameters with ``npx.savez(filename, **arg_dict)``
-- an uncompressed zip of ``.npy`` members (the ``.npz`` container under a ``.params`` suffix),
float32, keyed by the STRUCTURAL names of ``BertModel`` (models/bert.py:318-402:
``word_embed.weight``, ``token_type_embed.weight``, ``token_pos_embed._embed.weight``,
``embed_layer_norm.{gamma,beta}``, ``encoder.all_layers.{i}.{attn_qkv,attention_proj}.{weight,bias}``,
``encoder.all_layers.{i}.layer_norm.{gamma,beta}``, ``encoder.all_layers.{i}.ffn.{ffn_1,ffn_2}.
{weight,bias}``, ``encoder.all_layers.{i}.ffn.layer_norm.{gamma,beta}``, ``pooler.{weight,bias}``).
``mxnet`` is not installable here, so the container is written with numpy; the names are spelled
out below from the reference's class definitions -- NOT taken from this repository's modules --
and the arrays are the HuggingFace-initialised weights of tests/golden/bert_hf_*.npz (renamed as
the reference's converter renames the TF layout, oracle/gen_golden_bert_hf.py), so the outputs a
model must produce after loading the file are HuggingFace's (``sequence`` / ``pooled`` there).

Also written: the same arrays as a torch ``state_dict`` file (``torch.save`` of name -> tensor),
the layout of the reference's torch sibling checkpoints.

Run here (build container) only:  python oracle/gen_params_fixture.py
"""
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def structural_names(num_layers):
    """models/bert.py:318-402 (BertModel.__init__) + models/transformer.py:193-226 +
    layers.py:590-617, in definition order."""
    names = ['word_embed.weight', 'embed_layer_norm.gamma', 'embed_layer_norm.beta',
             'token_type_embed.weight', 'token_pos_embed._embed.weight']
    for i in range(num_layers):
        p = 'encoder.all_layers.{}.'.format(i)
        names += [p + 'attn_qkv.weight', p + 'attn_qkv.bias',
                  p + 'attention_proj.weight', p + 'attention_proj.bias',
                  p + 'layer_norm.gamma', p + 'layer_norm.beta',
                  p + 'ffn.ffn_1.weight', p + 'ffn.ffn_1.bias',
                  p + 'ffn.ffn_2.weight', p + 'ffn.ffn_2.bias',
                  p + 'ffn.layer_norm.gamma', p + 'ffn.layer_norm.beta']
    names += ['pooler.weight', 'pooler.bias']
    return names


def run(name):
    g = np.load(os.path.join(G, name + '.npz'))
    nl = int(g['num_layers'])
    arrays = {}
    for k in structural_names(nl):
        arrays[k] = np.ascontiguousarray(g['p:' + k].astype(np.float32))   # zoo files are float32
    # MXNet 2.x npx.savez: uncompressed npz container, whatever the suffix
    with open(os.path.join(G, name + '.params'), 'wb') as f:
        np.savez(f, **arrays)
    torch.save({k: torch.from_numpy(v) for k, v in arrays.items()},
               os.path.join(G, name + '.state_dict.pt'))
    print(name, 'written:', len(arrays), 'parameters')


if __name__ == '__main__':
    run('bert_hf_h4')
