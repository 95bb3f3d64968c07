"""List the launches of the roofline layer (dec0.conv1 forward) inside a profiled bench.py run: rows of a rocprofv3 kernel trace
whose kernel, grid and duration match that layer.   python tools/roofline_launches.py <kernel_trace.csv> <kernel substring> <min us> <max us> [grid_x grid_y]"""
import csv, sys
path, kern, lo, hi = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])
grid = (sys.argv[5], sys.argv[6]) if len(sys.argv) > 6 else None
rows = []
for r in csv.DictReader(open(path)):
    if kern in r['Kernel_Name'] and (grid is None or (r['Grid_Size_X'], r['Grid_Size_Y']) == grid):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if lo <= d <= hi:
            rows.append((int(r['Start_Timestamp']), d, r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'], r['LDS_Block_Size']))
rows.sort()
print(f'# {len(rows)} launches of *{kern}* with {lo} <= duration <= {hi} us in {path.split("/")[-1]} (rocprofv3 --kernel-trace)')
print('# index  duration_us  grid_x  grid_y  workgroup  lds_bytes')
for i, (t, d, gx, gy, wg, lds) in enumerate(rows):
    print(f'{i:4d}  {d:9.1f}  {gx}  {gy}  {wg}  {lds}')
if rows:
    ds = [r[1] for r in rows]
    print(f'# mean {sum(ds) / len(ds):.1f} us, min {min(ds):.1f}, max {max(ds):.1f}')
