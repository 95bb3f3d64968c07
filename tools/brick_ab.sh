# A/B of the per-XCD brick shape (IUNET_BRICK3="z y x" tiles, product 32) on the one-Cout-tile 3-D layers: time and HBM traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in "2 4 4" "4 4 2" "4 2 4" "2 8 2" "1 8 4" "1 4 8" "2 2 8"; do
  for s in 0:64:32 0:32:32; do
    echo "brick $b: $(IUNET_BRICK3="$b" python3 $R/tools/bench_conv.py --only $s --wgrad 0 --iters 30 --n 2 2>&1 | grep '^L0')"
  done
done
for b in "2 2 8" "1 4 8"; do
  for c in FETCH_SIZE; do
    rm -rf /tmp/pmc_ab
    IUNET_BRICK3="$b" rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_ab -- python3 $R/tools/bench_conv.py --only 0:64:32 --wgrad 0 --iters 2 > /dev/null 2>&1
    python3 - <<PY
import csv,glob
f=glob.glob('/tmp/pmc_ab/*/*counter_collection.csv')[0]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'conv3_v4' in r['Kernel_Name'] and r['Counter_Name']=='$c']
print('brick $b $c KB:', sum(v[-3:])/3)
PY
  done
done
