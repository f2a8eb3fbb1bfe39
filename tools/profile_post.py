"""Condense the rocprofv3 output of tools/profile_bench.sh into the small files kept in profiles/."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def find(pat):
    fs = glob.glob(os.path.join(out, pat), recursive=True)
    return fs[0] if fs else None


def short(name):
    name = name.replace('gnlp::', '')
    i = name.find('(')
    return name[:i] if i > 0 else name


# kernel stats: top kernels by total time
ks = find('stats/**/*kernel_stats.csv')
if ks:
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(out, 'kernel_stats_top.csv'), 'w') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
        for r in rows[:25]:
            w.writerow([short(r['Name']), r['Calls'], r['TotalDurationNs'], r['AverageNs'],
                        r['Percentage']])
        print(open(os.path.join(out, 'kernel_stats_top.csv')).read())

# PMC passes: per kernel name, average counter value per launch
traffic = collections.defaultdict(dict)
for kind in ('fetch', 'write'):
    cc = find('%s/**/*counter_collection.csv' % kind)
    if not cc:
        continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(cc)):
        a = agg[short(r['Kernel_Name'])]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
    for k, (v, n) in agg.items():
        kb = v / n          # FETCH_SIZE / WRITE_SIZE are reported in KiB
        if kind == 'fetch':
            traffic[k]['fetch_MB_raw'] = kb / 1024.0
            traffic[k]['fetch_MB_corrected_x2'] = 2.0 * kb / 1024.0   # MI355X_MICROARCH.md, HBM
        else:
            traffic[k]['write_MB'] = kb / 1024.0
        traffic[k]['launches_seen'] = n
big = {k: v for k, v in traffic.items()
       if v.get('fetch_MB_raw', 0) + v.get('write_MB', 0) > 8.0}
json.dump(big, open(os.path.join(out, 'hbm_traffic_pmc.json'), 'w'), indent=1, sort_keys=True)
print(json.dumps(big, indent=1, sort_keys=True))
