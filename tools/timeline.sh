#!/bin/bash
# tools/timeline.sh -- run on the GPU box (gpurun): rocprofv3 kernel + memory-copy trace of the headline bench, then the
# timeline of ONE filter step (start, duration, gap before every launch) and the per-step means over 50 steps:
# where a step's time goes BETWEEN the kernels (round 3: two fill kernels per memset, the beam-table copy, the four
# empty hand-over launches).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/bench.py --map mesh --steps 20 --warmup 3 --only-main > /dev/null 2>&1
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
ev = []
for p in glob.glob(O + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:48]))
for p in glob.glob(O + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
ev.sort()
# find the steps: k_predict_pose starts
idx = [i for i, e in enumerate(ev) if 'k_predict_pose' in e[2]]
i0, i1 = idx[-60], idx[-59]
print('one step (ns): start-rel, dur, gap-before, name')
t0 = ev[i0][0]
prev_end = ev[i0 - 1][1]
for e in ev[i0:i1]:
    print('%8d %8d %7d  %s' % (e[0] - t0, e[1] - e[0], e[0] - prev_end, e[2]))
    prev_end = max(prev_end, e[1])
print('step period', ev[i1][0] - ev[i0][0])
# averages over 50 steps
import collections
gaps = collections.defaultdict(list); durs = collections.defaultdict(list)
for k in range(-60, -10):
    a, b = idx[k], idx[k + 1]
    pe = ev[a - 1][1]
    for e in ev[a:b]:
        gaps[e[2]].append(e[0] - pe); durs[e[2]].append(e[1] - e[0]); pe = max(pe, e[1])
tot_gap = 0
for n in durs:
    g = sum(gaps[n]) / 50.0; d = sum(durs[n]) / 50.0; tot_gap += g
    print('%-50s dur/step %8.0f  gap/step %7.0f  count/step %.1f' % (n, d, g, len(durs[n]) / 50.0))
print('total gap per step', tot_gap)
PY
