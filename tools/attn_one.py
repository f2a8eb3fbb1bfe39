import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.argv = ['x']
import importlib.util
spec = importlib.util.spec_from_file_location('ab', os.path.join(ROOT, 'tools', 'attn_bench.py'))
src = open(os.path.join(ROOT, 'tools', 'attn_bench.py')).read().split("for name, args in")[0]
exec(src)
ms, tf = run(64, 512, 16, iters=3)
print('%.1f us %.0f TF' % (ms * 1e3, tf))
