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


# ---- bench.py's control flow after the build, on two gloo ranks with a stub step -------------------
def _protocol_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    b = _bench()
    calls = {'step': 0, 'alone': 0, 'gather_waits': 0, 'report': 0}
    dev = torch.device('cpu')

    def step():                      # like the real step: one collective per call
        calls['step'] += 1
        t = torch.full((1,), float(rank))
        got = [torch.empty(1) for _ in range(world)]
        dist.all_gather(got, t)
        return (torch.cat(got),)

    def step_alone():                # like encode_local: no collective
        calls['alone'] += 1

    def wait_gather():
        calls['gather_waits'] += 1

    def rank0_block(elapsed, harness_ms, median_ms, out):
        # what rank 0 does alone in bench.py: repeat the forward WITHOUT its collective, a
        # different number of times than anything the other rank does, and take its time over it
        calls['report'] += 1
        import time
        for _ in range(7):
            step_alone()
        time.sleep(0.5)
        assert elapsed > 0 and harness_ms > 0 and median_ms > 0
        assert out[0].tolist() == [0.0, 1.0]

    elapsed = b.run_protocol(step, wait_gather, lambda: None, dist, dev, 4, 2, rank, rank0_block)
    # the MAX over the ranks: both ranks hold the same figure
    t = torch.tensor([elapsed], dtype=torch.float64)
    lo = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    ret[rank] = dict(calls, same_elapsed=bool(t.item() == lo.item()))
    dist.destroy_process_group()


def test_bench_protocol_world2_every_rank_makes_the_same_collective_calls():
    """the post-timing part of bench.py (every rank in the harness-protocol
    repeats, rank 0 alone in the roofline block, final barrier) on two gloo ranks.  A rank that
    skipped the repeats, or a collective inside rank 0's block, would leave the other rank in a
    collective nobody answers: the workers are joined under a time limit."""
    world = 2
    port = 29950 + (os.getpid() % 40)
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.spawn(_protocol_worker, args=(world, port, ret), nprocs=world, join=False)
    import time
    deadline = time.time() + 120
    while not ctx.join(timeout=1.0):
        if time.time() > deadline:
            for p in ctx.processes:          # the exact processes this test started
                p.kill()
            pytest.fail('bench.run_protocol hung on two ranks')
    r = dict(ret)
    n_steps = 2 + 4 + 3 * 3 + 20                 # warm-up, timed, harness protocol, median of 20
    assert r[0]['step'] == r[1]['step'] == n_steps
    assert r[0]['report'] == 1 and r[1]['report'] == 0
    assert r[0]['alone'] == 7 and r[1]['alone'] == 0
    assert r[0]['gather_waits'] == r[1]['gather_waits'] == 2
    assert r[0]['same_elapsed'] and r[1]['same_elapsed']
