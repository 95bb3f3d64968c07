"""Register / scratch report of one HIP source: compiles it for gfx950 with -Rpass-analysis=kernel-resource-usage and prints one line
per kernel (demangled name, VGPRs, AGPRs, spilled VGPRs, scratch bytes per lane, LDS bytes, occupancy).  `python tools/regreport.py
interactive-unet_amd/csrc/conv3_v4.hip [filter]`; a kernel with scratch > 0 spills."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result', '-Wno-int-to-pointer-cast',
       '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/dev/null']
log = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in log.splitlines():
    m = re.search(r'remark: (?:.*?:\d+:\d+: )?\s*(Function Name|Name): (\S+)', line) or re.search(r'(Function Name|Name): (\S+)', line)
    if m:
        cur = {'name': m.group(2)}
        rows.append(cur)
        continue
    m = re.search(r'\s+(VGPRs|AGPRs|VGPR Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]|SGPRs): (\d+)', line)
    if m and cur is not None:
        cur[m.group(1).split(' [')[0]] = int(m.group(2))
if not rows:
    sys.stderr.write(log[-3000:])
    sys.exit(1)
def pretty(n):
    m = re.match(r'_ZN?(?:12_GLOBAL__N_1)?(\d+)', n)
    if not m:
        return n
    k = int(m.group(1))
    base, rest = n[m.end():m.end() + k], n[m.end() + k:]
    args = re.findall(r'DF16_|DF16b|L[ib]\d+E|f', rest.split('Ev')[0]) if rest.startswith('I') else []
    dec = {'DF16_': 'f16', 'DF16b': 'bf16', 'f': 'float'}
    out = [dec.get(a, a[2:-1] if a[0] == 'L' else a) for a in args]
    return base + ('<' + ','.join(out) + '>' if out else '')


names = [pretty(r['name']) for r in rows]
print(f'{"VGPR":>5} {"AGPR":>5} {"spill":>5} {"scratch":>7} {"occ":>3}  kernel')
for r, n in zip(rows, names):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*\)$', '', n)
    if flt and flt not in n:
        continue
    print(f'{r.get("VGPRs", 0):5d} {r.get("AGPRs", 0):5d} {r.get("VGPR Spill", 0):5d} {r.get("ScratchSize", 0):7d} {r.get("Occupancy", 0):3d}  {n}')
