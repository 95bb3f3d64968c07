"""Merge gpurun_out/levels_<tag>/* (tools/level_report.sh) into profiles/<round>_conv_levels_<tag>.md.
    python tools/level_report.py [tag=3d] [round=r03]        (a tag beginning with x2: the split-precision conv, tools/level_report.sh ... "--x2 2")"""
import csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag_ = sys.argv[1] if len(sys.argv) > 1 else '3d'
rnd = sys.argv[2] if len(sys.argv) > 2 else 'r03'
x2m = tag_.startswith('x2m')      # the split conv with its cross terms on the fp8 matrix cores (tools/level_report.sh ... "--x2m 2")
x2 = tag_.startswith('x2') and not x2m
f8 = tag_.startswith('f8')          # C5: the K = 128 fp8 kernel on e4m3 planes (tools/level_report.sh ... "--base 64 --levels 5 --f8 2")
wg = tag_.startswith('wgrad')       # the weight gradient (WGRAD=1 bash tools/level_report.sh ... wgrad_3d)
L = os.path.join(ROOT, 'gpurun_out', f'levels_{tag_}')
rows, head = [], None
for f in sorted(glob.glob(os.path.join(L, 'time_*.txt'))):
    tag = os.path.basename(f)[5:-4]
    line = open(f).readline()
    if x2m:     # "... | x2m 388.6 us 596.9 TF/s algorithmic = 1193.8 TF/s of matrix work ..."
        m = re.search(r'L(\d)\s+(\d+)->\s*(\d+) @(\d+)\^(\d) N=(\d+) layout (\d):.*x2m\s+([\d.]+) us\s+[\d.]+ TF/s algorithmic =\s+([\d.]+) TF/s', line)
    elif x2:    # "... | fp16x2 553.7 us 418.9 TF/s algorithmic = 1256.6 TF/s of MFMA work ..."
        m = re.search(r'L(\d)\s+(\d+)->\s*(\d+) @(\d+)\^(\d) N=(\d+) layout (\d):.*fp16x2\s+([\d.]+) us\s+[\d.]+ TF/s algorithmic =\s+([\d.]+) TF/s', line)
    elif f8:    # "... | fp8 (e4m3 planes) 358.0 us 2591.3 TF/s (1.91x)"
        m = re.search(r'L(\d)\s+(\d+)->\s*(\d+) @(\d+)\^(\d) N=(\d+) layout (\d):.*fp8[^|]*?\s([\d.]+) us\s+([\d.]+) TF/s', line)
    elif wg:    # "... | wgrad 229.2 us 1011.9 TF/s"
        m = re.search(r'L(\d)\s+(\d+)->\s*(\d+) @(\d+)\^(\d) N=(\d+) layout (\d):.*wgrad\s+([\d.]+) us\s+([\d.]+) TF/s', line)
    else:
        m = re.search(r'L(\d)\s+(\d+)->\s*(\d+) @(\d+)\^(\d) N=(\d+) layout (\d): fwd\s+([\d.]+) us\s+([\d.]+) TF/s', line)
    if not m:
        continue
    lvl, cin, cout, S, nd, n, lay, us, tf = m.groups()
    if x2:
        lay = 'split (3 x K16)'
    if x2m:
        lay = 'x2m (K16 + K128 fp8)'
    if f8:
        lay = 'K128 fp8' + (', split-K' if 'split-K' in line else '')
    if wg:
        lay = 'dy-reuse form'
    def ctr(kind, name):
        g = glob.glob(os.path.join(L, f'{kind}_{tag}', '**', '*_counter_collection.csv'), recursive=True)
        if not g:
            return None
        vals = [float(r['Counter_Value']) for r in csv.DictReader(open(g[0]))
                if r['Counter_Name'] == name and ('wgrad_v2_kernel' if wg else 'conv3_f8k' if f8 else '_x2m_kernel' if x2m else 'conv3') in r['Kernel_Name']
                and 'pack' not in r['Kernel_Name'] and (wg or 'wgrad' not in r['Kernel_Name'])]
        return sum(vals) / len(vals) if vals else None
    busy, gui = ctr('mfma', 'SQ_VALU_MFMA_BUSY_CYCLES'), ctr('mfma', 'GRBM_GUI_ACTIVE')
    fetch, write = ctr('fetch', 'FETCH_SIZE'), ctr('write', 'WRITE_SIZE')
    util = busy / (gui / 8 * 1024) if busy and gui else None       # 1024 SIMDs, GUI_ACTIVE summed over 8 XCDs
    nd, n = int(nd), int(n)
    vox = int(S) ** nd * n
    alg = (int(cin) + int(cout)) * 2 * vox + int(cin) * int(cout) * 3 ** nd * (4 if wg else 2)      # activations once in, once out + the filter once (weight gradient: input + output gradient once in, dW fp32 once out)
    if x2:
        alg *= 2                                                                        # hi + lo words of everything
    if x2m:
        alg = alg * 3 // 2                                                              # hi words + lo8 bytes: 3 bytes per element
    if f8:
        alg //= 2                                                                       # one byte per activation and per weight
    traffic = (2 * fetch + write) * 1024 if fetch is not None and write is not None else None
    head = (nd, n)
    rows.append((lvl, cin, cout, S, lay, float(us), float(tf), util, alg, traffic, nd))
out = os.path.join(ROOT, 'profiles', f'{rnd}_conv_levels_{tag_}.md')
with open(out, 'w') as o:
    nd, n = head
    o.write(f'# 3^{nd} conv {"WEIGHT GRADIENT" if wg else "forward"} per resolution level ({n} x level-0 tile per launch){" -- SPLIT PRECISION (fp16x2: TFLOP/s = MFMA work, 3 MFMAs per product; algorithmic = a third)" if x2 else " -- SPLIT PRECISION WITH THE CROSS TERMS ON THE fp8 MATRIX CORES (x2m: TFLOP/s = matrix work in 16-bit equivalents, one 16-bit + two double-rate fp8 products per multiply-add; algorithmic = half; MFMA pipe busy counts both instruction kinds)" if x2m else " -- C5 on the K = 128 fp8 instruction, e4m3 planes in and out (MFMA pipe busy counts both instruction forms)" if f8 else ""} -- rocprofv3 PMC, MI355X\n\n')
    o.write('time / TFLOP/s: HIP events over 30 back-to-back launches (tools/bench_conv.py); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / '
            '(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE correction), separate '
            'passes; algorithmic MB = activations once in + once out + the filter once; % of peak vs ' + ('5 PFLOP/s dense fp8' if f8 else '2.5 PFLOP/s dense 16-bit') + ' MFMA and 8 TB/s.  '
            'Produced by tools/level_report.sh + level_report.py.\n\n')
    o.write('| level | layer | kernel layout | time (us) | TFLOP/s | % MFMA peak | MFMA pipe busy | algorithmic MB | HBM MB (PMC) | traffic / algorithmic | HBM GB/s | % HBM peak |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n')
    for lvl, cin, cout, S, lay, us, tf, util, alg, traffic, nd in rows:
        gbs = traffic / (us * 1e-6) / 1e9 if traffic else None
        o.write(f'| {lvl} ({S}^{nd}) | {cin}->{cout} | {lay} | {us:.1f} | {tf:.0f} | {tf / (50 if f8 else 25):.1f} % | '
                f'{"%.0f %%" % (100 * util) if util else "n/a"} | {alg / 1e6:.0f} | {"%.0f" % (traffic / 1e6) if traffic else "n/a"} | '
                f'{"%.2f x" % (traffic / alg) if traffic else "n/a"} | {"%.0f" % gbs if gbs else "n/a"} | {"%.1f %%" % (gbs / 80) if gbs else "n/a"} |\n')
print(open(out).read())
