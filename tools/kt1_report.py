"""Per-kernel breakdown of one batch from the newest single-stream kernel trace under gpurun_out/kt_one/."""
import csv, glob, collections, os
f = max(glob.glob('gpurun_out/kt_one/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_gen_eye' in r['Kernel_Name']]
a, b = idx[3], idx[4]
t0 = int(rows[a]['Start_Timestamp'])
tot = collections.defaultdict(float)
for r in rows[a:b]:
    n = r['Kernel_Name']; n = n[n.find('k_'):n.find('(bdpt')] if 'k_' in n else n[:30]
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    tot[n] += (e - s) / 1e3
    if 'k_traverse<false' in n: print('%7.1f' % ((e - s) / 1e3), end=' ')
    if 'k_shade' in n: print('[%5.1f]' % ((e - s) / 1e3), end=' ')
print(); print({k: round(v) for k, v in tot.items()})
print('batch total', (int(rows[b]['Start_Timestamp']) - t0) / 1e3, 'us')
