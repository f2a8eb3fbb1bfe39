#!/usr/bin/env python
"""bench.py -- BERT-large encoder forward (L=512, bf16) sequences/s on N MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W``.  ``--gpus N`` MEANS N: under torchrun
(WORLD_SIZE set) the world size must equal N or the run exits non-zero; started plainly with
N > 1 the parent -- before it touches a GPU -- starts N ranks through ``torch.distributed.run``
(one per GPU, 127.0.0.1 rendezvous) and exits with their status; fewer than N visible GPUs is an
error, never a silent 1-GPU run.  The printed ``n_gpus`` is the live process group's size and
``rccl_ranks`` an all-reduce of ones over it.  A step = one forward of the 24-layer encoder stack (the hot path of
BASELINE.json) over one synthetic batch already resident in HBM; with N>1 the batch is sharded
(weak scaling, B=64 per GPU) and the CLS rows are all-gathered over RCCL each step.
Rank 0 prints ONE JSON line with the metric, the roofline of the dominant kernel (the FFN-1
GEMM+GELU, timed live with HIP events on the launch stream) and the CPU baseline (the numpy
oracle restatement timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'gluon-nlp_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {
    # name: (kind, units, heads, hidden, layers, L, per-GPU batch, eps)
    'bert_large': ('bert', 1024, 16, 4096, 24, 512, 64, 1e-12),
    'bert_base': ('bert', 768, 12, 3072, 12, 128, 256, 1e-12),
    'electra_large': ('bert', 1024, 16, 4096, 24, 512, 32, 1e-12),
    'roberta_large': ('roberta', 1024, 16, 4096, 24, 512, 32, 1e-5),
    't5_base': ('t5', 768, 12, 3072, 12, 1024, 16, 1e-6),
}
PEAK = {'bf16': 2500.0, 'f32': 157.3}      # dense MFMA TFLOP/s, MI355X_MICROARCH.md


def flops_per_seq(U, F, layers, L):
    # 2 x MACs of the dense contractions, dense over the padded length
    return layers * L * (8 * U * U + 4 * U * F + 4 * L * U)


def build_model(kind, U, H, F, layers, eps, dtype_name, dev):
    from gluonnlp_amd.models import BertTransformer, RobertaEncoder, T5Encoder
    torch.manual_seed(0)
    if kind == 't5':
        m = T5Encoder(U, 64, F, num_layers=layers, num_heads=H, dtype=dtype_name)
    elif kind == 'roberta':
        m = RobertaEncoder(units=U, hidden_size=F, num_layers=layers, num_heads=H,
                           layer_norm_eps=eps, dtype=dtype_name)
    else:
        m = BertTransformer(units=U, hidden_size=F, num_layers=layers, num_heads=H,
                            layer_norm_eps=eps, dtype=dtype_name)
    return m.to(dev)


def time_dense(M, N, K, act, dtype, dev, iters=20):
    """Average duration (ms) of the dominant kernel, HIP events on the launch stream."""
    from gluonnlp_amd.layers import Dense
    d = Dense(N, K, dtype='bfloat16' if dtype == torch.bfloat16 else 'float32').to(dev)
    d.bias.data.normal_(0, 0.02)
    x = torch.randn(M, K, device=dev).to(dtype)
    out = torch.empty(M, N, device=dev, dtype=dtype)
    for _ in range(3):
        d(x, act=act, out=out)
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        d(x, act=act, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize(dev)
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]
    return float(np.mean(ts))


def time_dense_in_situ(step, M, N, K, launches, dev):
    """Average duration (ms) of the dominant kernel where it runs: the library brackets every
    dense launch of this shape with HIP events on the launch stream (gnlp_profile_dense_*) while
    the same forward steps as the timed region are repeated."""
    import ctypes
    from gluonnlp_amd import _lib
    lib = _lib.load()
    _lib.check(lib.gnlp_profile_dense_arm(M, N, K, launches))
    try:
        done = 0
        while done < launches:
            step()
            torch.cuda.synchronize(dev)
            buf = (ctypes.c_float * launches)()
            n = lib.gnlp_profile_dense_read(buf, launches)
            if n <= done:      # this forward has no launch of that shape
                break
            done = n
        ts = [buf[i] for i in range(done)] if done else []
    finally:
        lib.gnlp_profile_dense_disarm()
    return (float(np.mean(ts)), len(ts)) if ts else (None, 0)


def cpu_baseline(kind, U, H, F, layers, L, eps, sample_b):
    """The oracle (numpy restatement of the reference's unfused op sequence) in float32 on the
    host cores, one forward over a bounded sample of the same workload."""
    from oracle import ref
    rng = np.random.default_rng(0)
    lp = [ref.make_encoder_layer_params(rng, U, F, np.float32, random_bias=False)
          for _ in range(2)]
    x = rng.standard_normal((sample_b, L, U)).astype(np.float32)
    vl = np.full(sample_b, L)
    plist = [lp[i % 2] for i in range(layers)]     # weights alternate: memory stays small
    t0 = time.perf_counter()
    ref.bert_transformer(x, vl, plist, H, eps, 'gelu', 'NT',
                         final_sequence_mask=(kind != 'roberta'))
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get('num_threads', 1) for i in threadpool_info()] or [1])
    except Exception:
        cores = int(os.environ.get('OMP_NUM_THREADS', '1'))
    return {'value': sample_b / dt, 'unit': 'sequences/s', 'cores': int(cores),
            'host_cpu_count': os.cpu_count(), 'kind': 'port',
            'sample': '1 forward of B={} sequences (L={}, {} layers, fp32 numpy restatement of '
                      'the reference op sequence; not MXNet), {:.1f} s'.format(sample_b, L, layers,
                                                                               dt)}


def resolve_launch(gpus, env, visible_gpus):
    """Decide how this process runs.  -> ('rank', world) when it is one rank of a launched job (or
    the single-GPU case), ('spawn', gpus) when it must start ``gpus`` ranks itself.  Raises
    SystemExit with a message when --gpus disagrees with the launcher or with the machine."""
    if gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1 (got {})'.format(gpus))
    ws = env.get('WORLD_SIZE')
    if ws is not None:
        if int(ws) != gpus:
            raise SystemExit('bench.py: --gpus {} but the launcher started WORLD_SIZE={} ranks; '
                             'refusing to report a {}-GPU number'.format(gpus, ws, gpus))
        return 'rank', gpus
    if gpus == 1:
        return 'rank', 1
    if visible_gpus < gpus:
        raise SystemExit('bench.py: --gpus {} but only {} GPU(s) are visible; refusing to run '
                         'fewer ranks than asked'.format(gpus, visible_gpus))
    return 'spawn', gpus


def spawn_ranks(gpus, argv):
    """Start ``gpus`` ranks of this script through torch.distributed.run as a CHILD process (this
    process has not initialised the GPU and never replaces itself) and return its exit status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
           str(gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd)


def count_ranks(dist, device):
    """All-reduce of ones over the live process group: what the collective backend really spans."""
    t = torch.ones(1, device=device, dtype=torch.float32)
    dist.all_reduce(t)
    return int(round(float(t.item())))


def run_protocol(step, wait_gather, sync, dist, dev, steps, warmup, rank, rank0_block):
    """Everything between "model built" and "process group torn down", with the device work
    injected (tests/test_dist_gloo.py runs it on two gloo ranks with a stub step):
      W untimed steps; barrier; K timed steps + wait_gather(); barrier; MAX of the elapsed time over
      the ranks; then the reference harness's own protocol (benchmark_utils.py:843-844: min over 3
      repeats of 3 forwards / 3) and the median of 20 single forwards ON EVERY RANK -- step()
      issues a collective when the job has more than one rank, so every rank makes the same calls;
      then rank 0 ALONE in rank0_block (nothing in it may issue a collective); a final barrier holds
      the other ranks until rank 0 has printed.
    ``dist`` is torch.distributed or None (single process); ``sync`` waits for the device."""
    for _ in range(warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
    wait_gather()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    def timed(n):
        sync()
        t = time.perf_counter()
        for _ in range(n):
            step()
        sync()
        return (time.perf_counter() - t) / n * 1e3
    harness_ms = min(timed(3) for _ in range(3))
    median_ms = float(np.median([timed(1) for _ in range(20)]))
    wait_gather()
    if rank == 0:
        rank0_block(elapsed, harness_ms, median_ms, out)
    if dist is not None:
        dist.barrier()
    return elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='bert_large', choices=sorted(CONFIGS))
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default: config)')
    ap.add_argument('--ragged', action='store_true', help='ragged valid_length (reported apart)')
    ap.add_argument('--from-ids', action='store_true',
                    help='BERT backbone from token ids: embedding front-end + stack + pooler '
                         '(SURVEY 8f rank 1; reported apart from the headline)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=12)   # ~13 s of host work
    args = ap.parse_args()

    # decided before anything touches a GPU (device_count() does not initialise it)
    mode, world = resolve_launch(args.gpus, os.environ, torch.cuda.device_count())
    if mode == 'spawn':
        sys.exit(spawn_ranks(world, sys.argv[1:]))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # GNLP_BENCH_FORCE_DIST=1: rehearse the multi-rank control flow (process group, barriers,
    # all-gather on the side stream, max over ranks) with a one-rank RCCL group on a 1-GPU box
    force_dist = os.environ.get('GNLP_BENCH_FORCE_DIST') == '1'
    dist_on = world > 1 or force_dist
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)
        if dist.get_world_size() != world:
            raise SystemExit('bench.py: process group has {} ranks, --gpus {}'.format(
                dist.get_world_size(), world))
        rccl_ranks = count_ranks(dist, dev)
    else:
        rccl_ranks = 1

    kind, U, H, F, layers, L, B, eps = CONFIGS[args.config]
    if args.batch:
        B = args.batch
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    dtype_name = 'bfloat16' if args.dtype == 'bf16' else 'float32'
    model = build_model(kind, U, H, F, layers, eps, dtype_name, dev)

    from gluonnlp_amd.parallel import BatchShardedEncoder
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, L, U, device=dev, generator=g).to(dtype)
    if args.ragged:
        vl = torch.randint(1, L + 1, (B,), device=dev, generator=g)
        vl[0], vl[-1] = L, 1
    else:
        vl = torch.full((B,), L, device=dev, dtype=torch.int64)   # reference harness: full
    # overlap=True: the all-gather of step i runs beside the forward of step i + 1 (its consumer
    # would call wait_gather(); the timed region ends with one, so every gather is inside it)
    runner = BatchShardedEncoder(model, world_size=world, rank=rank, force_collective=force_dist,
                                 overlap=True)

    def step():
        return runner.forward_local(x, vl)

    def step_no_collective():   # the same forward without the all-gather: what rank 0 may run alone
        return runner.encode_local(x, vl)

    if args.from_ids:
        if kind == 't5':
            raise SystemExit('--from-ids is for the BERT-style backbones')
        from gluonnlp_amd.models import BertModel
        torch.manual_seed(0)
        bm = BertModel(vocab_size=30522, units=U, hidden_size=F, num_layers=layers, num_heads=H,
                       max_length=max(512, L), layer_norm_eps=eps, dtype=dtype_name).to(dev)
        ids = torch.randint(0, 30522, (B, L), device=dev, generator=g)
        tts = torch.zeros_like(ids)

        def step():   # noqa: F811
            ctx, pooled = bm(ids, tts, vl)
            return ctx, pooled
        step_no_collective = step   # noqa: F811

    def rank0_report(elapsed, harness_ms, median_ms):
        seqs = world * B * args.steps / elapsed
        fps = flops_per_seq(U, F, layers, L)
        vls = vl.to(torch.float64)
        useful_fps = float((layers * (vls * (8 * U * U + 4 * U * F) + 4 * vls * vls * U)).mean())
        peak = PEAK[args.dtype]
        # dominant kernel: FFN-1 GEMM + bias + erf-GELU epilogue (M x F x U), one launch per layer
        M = B * L
        act = 'gelu' if kind != 't5' else 'relu'
        # in situ: the launches of the forward itself (same steps as the timed region, repeated
        # with the library's event hook armed); back-to-back launches of the kernel alone run
        # ~20 % slower (sustained matrix load -> lower clock) and are reported beside it
        ms_alone = time_dense(M, F, U, act, dtype, dev)
        # (rank 0 alone is in here: nothing in this block may issue a collective)
        ms, n_meas = time_dense_in_situ(step_no_collective, M, F, U,
                                        layers * min(args.steps, 5), dev)
        if ms is None:
            ms, n_meas = ms_alone, 20
        kflops = 2.0 * M * F * U
        achieved = kflops / (ms * 1e-3) / 1e12
        traffic = None   # HBM bytes per launch of the dominant kernel, from the separate PMC passes
        try:
            # (the instantiation the forward launches: <GELU, no residual, 256 wide, FOLD 1> with
            # the LayerNorm folded in -- the default -- or <..., FOLD 0> with GNLP_LN_FOLD=0)
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'r03_hbm_traffic_pmc.json')))
            want = '<2, false, 8, 0>' if os.environ.get('GNLP_LN_FOLD') == '0' else '<2, false, 8, 1>'
            for k, v in tj.items():
                if 'dense_bf16_w4' in k and want in k and args.config == 'bert_large' and act == 'gelu':
                    traffic = (v['fetch_MB_corrected_x2'] + v['write_MB']) * 1048576.0
        except Exception:
            traffic = None
        line = {
            'metric': ('BERT-large encoder fwd seqs/s L=512' if args.config == 'bert_large'
                       else '{} encoder fwd seqs/s L={}'.format(args.config, L)) +
                      (' (token ids -> pooled output)' if args.from_ids else ''),
            'value': seqs, 'unit': 'sequences/s',
            'n_gpus': dist.get_world_size() if dist_on else 1, 'rccl_ranks': rccl_ranks,
            'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '{} encoder forward: {} layers, units={}, heads={}, hidden={}, '
                                   'L={}, B={} per GPU, valid_length={}'.format(
                                       args.config, layers, U, H, F, L, B,
                                       'ragged' if args.ragged else 'full'),
                       'global_batch': world * B, 'seq_len': L,
                       'parallelism': 'batch-sharded x{} (+ RCCL all_gather of CLS rows)'.format(
                           world) if dist_on else 'single GPU'},
            'reference_harness_protocol_ms': harness_ms, 'median_of_20_ms': median_ms,
            'path_tflops': seqs * fps / 1e12,
            'path_frac_of_mfma_peak': seqs * fps / 1e12 / (peak * world),
            # ragged configs are priced both ways -- dense over the padded length
            # (above) and "useful": only the valid tokens' contractions, attention over vl x vl
            'path_tflops_useful': seqs * useful_fps / 1e12,
            'path_frac_of_mfma_peak_useful': seqs * useful_fps / 1e12 / (peak * world),
            'roofline': {'bound': 'mfma',
                         'kernel': '{} (FFN-1 GEMM + bias + {}{}), M={} N={} K={}'.format(
                             'dense_bf16_w4_kernel' if args.dtype == 'bf16' else 'dense_f32_kernel',
                             act, ', LayerNorm folded in' if (args.dtype == 'bf16' and kind != 't5' and
                                                              os.environ.get('GNLP_LN_FOLD') != '0')
                             else '', M, F, U),
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic,
                         'avg_launch_ms': ms, 'launches_measured': n_meas,
                         'measured': 'HIP events around every launch of this shape inside the '
                                     'forward (library hook), on the launch stream',
                         'avg_launch_ms_back_to_back': ms_alone, 'flops_per_launch': kflops},
        }
        if world == 1 and not args.no_cpu_baseline and kind != 't5':
            line['cpu_baseline'] = cpu_baseline(kind, U, H, F, layers, L, eps, args.cpu_sample)
        print(json.dumps(line), flush=True)

    def sync():
        torch.cuda.synchronize(dev)

    def report(elapsed, harness_ms, median_ms, out):
        assert torch.isfinite(out[0].float()).all()
        rank0_report(elapsed, harness_ms, median_ms)

    run_protocol(step, runner.wait_gather, sync, dist if dist_on else None, dev, args.steps,
                 args.warmup, rank, report)
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
