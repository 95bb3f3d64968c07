"""Per-kernel time of one bench step from a rocprofv3 kernel trace: rows between the first and the last launch of the timed region are
not separable, so the whole trace is summed and divided by `steps`.   python tools/step_profile.py <kernel_trace.csv> <steps> [top]"""
import collections, csv, re, sys
path, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot, cnt = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(path)):
    n = r['Kernel_Name']
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', n)
    if m:
        k = int(m.group(1)); base = n[m.end():m.end() + k]; rest = n[m.end() + k:]
        args = re.findall(r'DF16_|DF16b|L[ib]\d+E', rest.split('Ev')[0])
        n = base + '<' + ','.join({'DF16_': 'f16', 'DF16b': 'bf16'}.get(a, a[2:-1]) for a in args) + '>'
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot[n] += d; cnt[n] += 1
all_ = sum(tot.values())
print(f'total {all_ / steps / 1e3:.3f} ms per step over {steps:g} steps')
for n, t in tot.most_common(top):
    print(f'{t / steps:9.1f} us/step {100 * t / all_:5.1f} %  {cnt[n] / steps:6.1f} launches/step  avg {t / cnt[n]:8.1f} us  {n[:110]}')
