"""One step of a rocprofv3 kernel trace as a timeline: launch order, duration and the idle gap in front of every kernel.
python tools/step_timeline.py <kernel_trace.csv> <launches per step> [which step from the end, default 2]
(launches per step: from tools/step_profile.py's launch counts; the trace is cut into steps by counting launches from the end)"""
import csv, re, sys
path, per = sys.argv[1], int(sys.argv[2])
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', n)
    if m:
        k = int(m.group(1)); base = n[m.end():m.end() + k]; rest = n[m.end() + k:]
        args = re.findall(r'DF16_|DF16b|L[ib]\d+E', rest.split('Ev')[0])
        n = base + '<' + ','.join({'DF16_': 'f16', 'DF16b': 'bf16'}.get(a, a[2:-1]) for a in args) + '>'
    return n[:70]
seg = rows[len(rows) - back * per:len(rows) - (back - 1) * per]
t0 = int(seg[0]['Start_Timestamp'])
prev_end = t0
gaps = 0.0
busy = 0.0
for i, r in enumerate(seg):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3
    gaps += max(gap, 0.0)
    busy += (e - s) / 1e3
    print(f'{i:4d} {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:7.1f}  grid {r.get("Grid_Size", "?"):>9} wg {r.get("Workgroup_Size", "?"):>5}  {short(r["Kernel_Name"])}')
    prev_end = max(prev_end, e)
print(f'step span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy:.1f} us, idle gaps {gaps:.1f} us over {len(seg)} launches')
