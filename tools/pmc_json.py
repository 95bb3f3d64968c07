"""Reduce the two rocprofv3 --pmc passes of a roofline kernel (FETCH_SIZE, WRITE_SIZE: tools/collect_profiles.sh) to the JSON that
bench.py's `roofline.traffic` reads:   python tools/pmc_json.py <dir> <round> <workload> <kernel filter> <layer text> <algorithmic bytes> <tiles per launch>
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B); both counters are in KB."""
import csv, json, sys

d, rnd, wl, flt, layer, alg, tiles = sys.argv[1:8]
def mean(counter):
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f'{d}/{rnd}_pmc_{counter}_{wl}.csv'))
            if flt in r['Kernel_Name'] and r['Counter_Name'] == counter]
    vals = vals[len(vals) // 2:] if len(vals) > 3 else vals          # the later launches (warm caches, as the timed ones)
    return sum(vals) / len(vals), len(vals)
f, nf = mean('FETCH_SIZE')
w, nw = mean('WRITE_SIZE')
traffic = 2 * f * 1024 + w * 1024
out = {'kernel': flt, 'layer': layer, 'FETCH_SIZE_KB_mean': f, 'WRITE_SIZE_KB_mean': w, 'launches_averaged': [nf, nw],
       'fetch_bytes_corrected_x2': 2 * f * 1024, 'write_bytes': w * 1024, 'traffic_bytes_per_launch': int(traffic),
       'algorithmic_activation_bytes_per_launch': int(alg), 'traffic_over_algorithmic': round(traffic / float(alg), 3),
       'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/collect_profiles.sh); FETCH_SIZE doubled per '
               'MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); counter unit KB', 'tiles_per_launch': int(tiles)}
json.dump(out, open(f'{d}/{rnd}_pmc_{wl}_dec0conv1.json', 'w'), indent=1)
print(json.dumps(out))
