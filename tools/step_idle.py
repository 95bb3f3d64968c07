"""Idle GPU time inside the steps of a bench run, from a rocprofv3 kernel trace (csv):   python tools/step_idle.py <kernel_trace.csv> [max_steps]
A step = the kernels from one adamw_dev_kernel to the next (training step's optimiser -> operator re-preparation -> prediction forward ->
next training step).  Prints, per step: wall time, idle time (sum of the gaps between consecutive kernels), launches, the largest gaps --
and for the first quiet step the timeline from the optimiser to the prediction forward's first conv.  tools/collect_profiles.sh keeps the
trace of its profiled bench run (bench_c3_kernel_trace.csv); the steps with a sync inside (warm-up, the leg-timing steps) show as such."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 40
S = lambda r: int(r['Start_Timestamp'])
E = lambda r: int(r['End_Timestamp'])
short = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
idx = [i for i, r in enumerate(rows) if 'adamw_dev_kernel' in r['Kernel_Name']]
shown = False
print(f'{len(rows)} kernels, {len(idx)} optimiser steps')
for a, b in list(zip(idx[:-1], idx[1:]))[:limit]:
    seg = rows[a:b]
    wall = (E(seg[-1]) - S(seg[0])) / 1e3
    if wall > 40000:
        continue
    gaps = [(S(seg[i + 1]) - max(E(r) for r in seg[:i + 1][-4:])) / 1e3 for i in range(len(seg) - 1)]
    idle = sum(g for g in gaps if g > 0)
    big = sorted(((g, short(seg[i]), short(seg[i + 1])) for i, g in enumerate(gaps) if g > 15), reverse=True)[:3]
    kind = 'x2m predict' if any('x2m' in r['Kernel_Name'] for r in seg) else '16-bit predict'
    print(f'step ({kind:14s}): wall {wall:8.1f} us, idle {idle:7.1f} us, {len(seg)} launches' +
          ''.join(f'; {g:.0f} us between {p} and {n}' for g, p, n in big))
    if not shown and idle < 20 and kind == 'x2m predict':
        shown = True
        t0 = S(seg[0])
        print('    optimiser -> prediction forward (us from the optimiser kernel''s start, duration):')
        for r in seg[:16]:
            print(f'      {(S(r) - t0) / 1e3:8.1f} {(E(r) - S(r)) / 1e3:7.1f}  {short(r)}')
            if 'first_conv' in r['Kernel_Name']:
                break
