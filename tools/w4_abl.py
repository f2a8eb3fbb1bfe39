"""Developer tool: timing-only ablations of the one-wave-per-SIMD GEMM on the FFN-2 / proj shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
os.environ['GNLP_GEMM_W4'] = '1'
from gemm_bench import run
names = {0: 'full', 1: 'no DMA', 2: 'no barriers 1/2/3', 3: 'no DMA, no barriers', 4: 'no ds_read',
         5: 'no DMA, no ds_read', 7: 'MFMA only', 10: 'sched 1 (DMA every 8)', 20: 'sched 2 (DMA every 2)', 30: 'sched 3 (VGPR staging)', 40: 'sched 4 (VGPR staging, linear src)', 50: 'sched 5 (DMA skewed by wave)', 60: 'sched 6 (linear LDS image)', 12: 'no ds_read, LINEAR source', 4: 'no ds_read'}
for rnd in range(2):
    for abl in (0, 60, 50, 30, 40, 4, 12, 1, 7):
        os.environ['GNLP_W4_ABL'] = str(abl)
        ms2, tf2 = run(32768, 1024, 4096, 'identity', True)
        ms1, tf1 = run(32768, 1024, 1024, 'identity', True)
        print('ABL=%d %-22s ffn2 %.0f us (%.2f us/K-tile) %.0f TF | proj %.0f us' %
              (abl, names[abl], ms2 * 1e3, ms2 * 1e3 / 128, tf2, ms1 * 1e3), flush=True)
