"""The N>1 path on CPU: two gloo ranks shard a global batch, run the encoder on their shard and
all-gather the CLS rows.  The compute is injected (the numpy oracle) because the product has no
CPU path; what is tested is the sharding, the collective and the global ordering."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, ret):
    for p in (ROOT, os.path.join(ROOT, 'gluon-nlp_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gluonnlp_amd.parallel import BatchShardedEncoder
    from oracle import ref
    rng = np.random.default_rng(5)
    U, F, H, L = 16, 32, 2, 6
    layers = [ref.make_encoder_layer_params(rng, U, F, np.float64) for _ in range(2)]
    x = rng.standard_normal((B, L, U))
    vl = rng.integers(1, L + 1, size=B)

    def compute(data, valid_length):
        out, _ = ref.bert_transformer(data.numpy(), valid_length.numpy(), layers, H)
        return torch.from_numpy(out)

    enc = BatchShardedEncoder(world_size=world, rank=rank, layout='NT', compute=compute)
    out_local, cls_all = enc.forward(torch.from_numpy(x), torch.from_numpy(vl))
    full, _ = ref.bert_transformer(x, vl, layers, H)
    lo, hi = __import__('gluonnlp_amd.parallel', fromlist=['shard_bounds']).shard_bounds(B, world, rank)
    ok = np.allclose(out_local.numpy(), full[lo:hi], rtol=1e-12, atol=1e-12) and \
        np.allclose(cls_all.numpy(), full[:, 0, :], rtol=1e-12, atol=1e-12) and \
        cls_all.shape == (B, U)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('B', [8, 7])
def test_batch_sharded_allgather_world2(B):
    world = 2
    port = 29600 + (os.getpid() % 200) + B
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, B, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
