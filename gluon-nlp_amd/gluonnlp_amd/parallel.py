"""Multi-GPU harness: the encoder forward shards by batch (sequences never interact -- the mask is
per sequence and LayerNorm per token), one process per GPU, weights replicated.  The only exchange
is an all-gather of the CLS rows ``(B_local, units)`` (what ``BertModel.apply_pooling`` consumes,
reference src/gluonnlp/models/bert.py:506-529) over RCCL/xGMI -- <= 256 KiB per rank, latency
bound, issued on a side stream.  By default the main stream waits for it (the gathered rows are
safe to read at once); with ``overlap=True`` it does not: the next batch's forward starts while
the gather runs and the consumer calls ``wait_gather()`` before reading the gathered rows.  No
data-path collective otherwise.
(Reference analogue: hvd.allgather of eval outputs, scripts/machine_translation/
train_transformer.py:790-801.)"""
import torch

__all__ = ['shard_bounds', 'BatchShardedEncoder']


def shard_bounds(n, world_size, rank):
    """Contiguous [lo, hi) slice of ``n`` items for ``rank``; the first n % world ranks get one
    extra item."""
    q, r = divmod(n, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class BatchShardedEncoder:
    """Wraps an encoder stack (``model(data, valid_length)``) for batch-sharded inference.

    ``compute`` may replace the model call (the gloo/CPU tests inject the oracle there); ``group``
    is the torch.distributed process group (None = default); ``force_collective`` issues the
    all-gather even when ``world_size == 1`` (a one-rank RCCL group exercises the GPU path)."""

    def __init__(self, model=None, world_size=1, rank=0, layout='NT', compute=None, group=None,
                 force_collective=False, overlap=False):
        self.model = model
        self.world_size = world_size
        self.rank = rank
        self.layout = getattr(model, 'layout', layout) if model is not None else layout
        self._compute = compute
        self._group = group
        self._side = None
        self._force = force_collective   # run the gather even with one rank (tests)
        self._overlap = overlap          # do not make the main stream wait for the gather
        self._pending = None             # event of the last gather not yet waited for

    # ---- sharding ----------------------------------------------------------------------------
    def shard(self, data, valid_length):
        bax = 0 if self.layout == 'NT' else 1
        lo, hi = shard_bounds(data.shape[bax], self.world_size, self.rank)
        idx = [slice(None)] * data.dim()
        idx[bax] = slice(lo, hi)
        return data[tuple(idx)].contiguous(), valid_length[lo:hi].contiguous()

    def _run(self, data, valid_length):
        if self._compute is not None:
            return self._compute(data, valid_length)
        res = self.model(data, valid_length)
        return res[0] if isinstance(res, tuple) else res

    def encode_local(self, data, valid_length):
        """This rank's shard through the encoder, no collective (safe to call on one rank only)."""
        return self._run(data, valid_length)

    def cls_rows(self, out):
        return out[:, 0, :] if self.layout == 'NT' else out[0]

    # ---- forward -----------------------------------------------------------------------------
    def forward_local(self, data, valid_length, max_local=None):
        """Encode this rank's shard; all-gather the CLS rows.  -> (out_local, cls_all or cls)."""
        out = self._run(data, valid_length)
        cls = self.cls_rows(out).contiguous()
        if self.world_size == 1 and not self._force:
            return out, cls
        import torch.distributed as dist
        n = cls.shape[0]
        cap = n if max_local is None else max_local
        if cap != n:   # uneven shards: pad to the largest shard so the gather is regular
            pad = torch.zeros((cap - n,) + tuple(cls.shape[1:]), dtype=cls.dtype,
                              device=cls.device)
            cls = torch.cat([cls, pad], 0)
        gathered = torch.empty((self.world_size * cap,) + tuple(cls.shape[1:]), dtype=cls.dtype,
                               device=cls.device)
        if cls.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=cls.device)
            self._side.wait_stream(torch.cuda.current_stream(cls.device))
            with torch.cuda.stream(self._side):
                dist.all_gather_into_tensor(gathered, cls, group=self._group)
            cls.record_stream(self._side)
            gathered.record_stream(self._side)
            if self._overlap:
                # the next forward may start now; whoever reads `gathered` calls wait_gather()
                self._pending = self._side.record_event()
            else:
                torch.cuda.current_stream(cls.device).wait_stream(self._side)
        else:
            parts = [torch.empty_like(cls) for _ in range(self.world_size)]
            dist.all_gather(parts, cls, group=self._group)
            gathered = torch.cat(parts, 0)
        return out, gathered

    def wait_gather(self):
        """Make the current stream wait for the last all-gather issued with ``overlap=True``."""
        if self._pending is not None:
            torch.cuda.current_stream().wait_event(self._pending)
            self._pending = None

    def forward(self, data, valid_length):
        """Global batch in, (out_local, cls of the WHOLE batch in global order) out."""
        bax = 0 if self.layout == 'NT' else 1
        B = data.shape[bax]
        x, vl = self.shard(data, valid_length)
        cap = shard_bounds(B, self.world_size, 0)[1]
        out, g = self.forward_local(x, vl, max_local=cap)
        self.wait_gather()
        if self.world_size == 1:
            return out, g
        rows = []
        for r in range(self.world_size):
            lo, hi = shard_bounds(B, self.world_size, r)
            rows.append(g[r * cap:r * cap + (hi - lo)])
        return out, torch.cat(rows, 0)
