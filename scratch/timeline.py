"""Analyse a rocprofv3 --kernel-trace CSV of bench.py: per-step timeline (kernel union busy time, per-kernel totals,
concurrency)."""
import csv, sys, glob, collections, os
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]) for r in rows])
# find the Adam kernels as step boundaries
mark = os.environ.get('STEP_MARK', 'k_opt_adam')
adam = [i for i, e in enumerate(ev) if mark in e[2]]
print('steps seen', len(adam))
import os
k_ = int(os.environ.get('STEP', -4))
a, b = adam[k_], adam[k_ + 1]          # one steady-state step (STEP: index of the Adam kernel that precedes it; the last 5 steps
                                       # of a bench run are the event-bracketed roofline leg: use e.g. STEP=5 for a timed step)
step = ev[a + 1:b + 1]
t0, t1 = step[0][0], step[-1][1]
print(f'step wall {(t1-t0)/1e6:.3f} ms, {len(step)} kernels')
# union busy time
busy = 0; cur_s, cur_e = step[0][0], step[0][1]
for s, e, _ in step[1:]:
  if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
  else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f'union busy {busy/1e6:.3f} ms ({100*busy/(t1-t0):.1f} %), sum of kernel times {sum(e-s for s,e,_ in step)/1e6:.3f} ms')
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in step: tot[n] += e - s; cnt[n] += 1
for n, t in tot.most_common(14): print(f'  {t/1e3:9.1f} us  x{cnt[n]:3d}  {n}')
# time with exactly 1 vs 2+ kernels active
pts = sorted([(s, 1) for s, e, _ in step] + [(e, -1) for s, e, _ in step])
act = 0; last = pts[0][0]; hist = collections.Counter()
for t, d in pts:
  hist[act] += t - last; last = t; act += d
print('time by #active kernels (ms):', {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
# sequence print (compact) for the nerf forward part
if len(sys.argv) > 2:
  for s, e, n in step: print(f'{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f}  {n}')
