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


# ---- bench.py --gpus N means N (the flag used to be parsed and ignored) --------------
def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def test_bench_resolve_launch_argument_logic():
    b = _bench()
    assert b.resolve_launch(1, {}, 0) == ('rank', 1)
    assert b.resolve_launch(1, {'WORLD_SIZE': '1'}, 1) == ('rank', 1)
    assert b.resolve_launch(4, {'WORLD_SIZE': '4'}, 8) == ('rank', 4)
    assert b.resolve_launch(8, {}, 8) == ('spawn', 8)
    with pytest.raises(SystemExit) as e:       # launcher and flag disagree
        b.resolve_launch(8, {'WORLD_SIZE': '1'}, 8)
    assert 'WORLD_SIZE=1' in str(e.value)
    with pytest.raises(SystemExit) as e:       # 1-GPU box asked for 2: loud, never n_gpus = 1
        b.resolve_launch(2, {}, 1)
    assert 'only 1 GPU' in str(e.value)
    with pytest.raises(SystemExit):
        b.resolve_launch(0, {}, 1)


def test_bench_gpus_2_without_gpus_fails_loudly():
    """`python bench.py --gpus 2` on a box with fewer than 2 GPUs exits non-zero with a message and
    prints no JSON line (this container has none; a 1-GPU box takes the same branch)."""
    import subprocess
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a box with fewer than 2 GPUs')
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps',
                        '1', '--warmup', '0'], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode != 0
    assert 'GPU(s) are visible' in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


def _count_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ret[rank] = _bench().count_ranks(dist, torch.device('cpu'))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_count_ranks_world2():
    world = 2
    port = 29850 + (os.getpid() % 100)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_count_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 2, 1: 2}
