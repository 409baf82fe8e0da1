#!/usr/bin/env python3
"""dev tool: which kernels of the side streams run beside each layer of the network in a rocprofv3 kernel trace of bench.py -- per network
kernel (by position in the pass) its mean duration in the steady passes and the side kernels that overlap it (summed overlap, launches).
   python tools/dev/trace_overlap.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
net = ('conv_mfma', 'sepconv_ws', 'dwconv', 'final_kernel', 'maxpool_add', 'pool_fix_add', 'stem_even')
def short(n):
    return n.replace('void ', '').replace('tmat::', '').split('(')[0][:48]
R = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
N = [r for r in R if any(k in r[2] for k in net)]
S = [r for r in R if not any(k in r[2] for k in net)]
# passes: from one stem_even to the next
idx = [i for i, r in enumerate(N) if 'stem_even' in r[2]]
passes = [N[a:b] for a, b in zip(idx, idx[1:])]
L = collections.Counter(len(p) for p in passes).most_common(1)[0][0]
passes = [p for p in passes if len(p) == L][4:-4]          # steady state
print(f"{len(passes)} steady passes of {L} network kernels")
import bisect
starts = [s[0] for s in S]
for k in range(L):
    dur = sum(p[k][1] - p[k][0] for p in passes) / len(passes) / 1e6
    ov, cnt = collections.Counter(), collections.Counter()
    for p in passes:
        a, b = p[k][0], p[k][1]
        j = bisect.bisect_left(starts, a - 50_000_000)
        while j < len(S) and S[j][0] < b:
            o = min(b, S[j][1]) - max(a, S[j][0])
            if o > 0:
                ov[S[j][2]] += o; cnt[S[j][2]] += 1
            j += 1
    top = ", ".join(f"{n} {v / len(passes) / 1e6:.2f} ms x{cnt[n] / len(passes):.1f}" for n, v in ov.most_common(4))
    print(f"{k:2d} {dur:8.3f} ms  {passes[0][k][2]:<48s} | {top}")
