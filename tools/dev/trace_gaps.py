#!/usr/bin/env python3
"""dev tool: gaps between consecutive kernels of the network in a rocprofv3 kernel trace, and what ran inside the largest ones."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
net = ('conv_mfma', 'sepconv_ws', 'dwconv', 'final_kernel', 'maxpool_add', 'pool_fix_add', 'stem_even')
R = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60], r['Queue_Id']) for r in rows)
N = [r for r in R if any(k in r[2] for k in net)]
gaps, cnt = collections.Counter(), collections.Counter()
big = []
for a, b in zip(N, N[1:]):
    key = (a[2], b[2]); gaps[key] += b[0] - a[1]; cnt[key] += 1
    if 'final_kernel' in a[2]: big.append((a[1], b[0]))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:6]:
    print(f"{v / cnt[k] / 1e3:9.1f} us avg x{cnt[k]:4d}  {k}")
starts = [r[0] for r in N if 'stem_even' in r[2]]
per = [(b - a) / 1e6 for a, b in zip(starts, starts[1:])]
print("pass period ms: median", sorted(per)[len(per) // 2], "min", min(per), "n", len(per))
g0, g1 = big[len(big) // 2]
print(f"inside one final -> stem gap ({(g1 - g0) / 1e3:.1f} us):")
for r in R:
    if r[1] > g0 and r[0] < g1 and not any(k in r[2] for k in net):
        print(f"   {(r[0] - g0) / 1e3:8.1f} .. {(r[1] - g0) / 1e3:8.1f} us  queue {r[3]}  {r[2]}")
