"""One short run of the fused attention kernel at the BERT-large shape (for rocprofv3 --pmc)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import attn_bench  # noqa: E402

ms, tf = attn_bench.run(64, 512, 16, iters=int(os.environ.get('ATTN_ONE_ITERS', '3')))
print('%.1f us %.0f TF' % (ms * 1e3, tf))
