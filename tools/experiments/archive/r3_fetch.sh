# Round 3: fabric fetches of one config-D filter launch by XCD sub-group size and item size (FETCH_SIZE only, one pass each).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in "0 0" "16 0" "8 0" "32 0" "16 16" "8 24"; do
  set -- $o
  O=gpurun_out/fetch_$1_$2; mkdir -p $O
  timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE -d $O -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 1 --no-cpu --opt xcd_cells=$1 --opt tile=$2 > $O/log.txt 2>&1
  python3 - $O $1 $2 <<'PY'
import csv, glob, sys
tot = {}
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        if 'filter_kernel' in r['Kernel_Name']:
            tot.setdefault(r['Dispatch_Id'], [0.0, 0])
            tot[r['Dispatch_Id']][0] += float(r['Counter_Value']); tot[r['Dispatch_Id']][1] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
for k, (v, ms) in sorted(tot.items()): print('xcd_cells', sys.argv[2], 'tile', sys.argv[3], 'FETCH_SIZE %.4g KiB -> %.1f GB (x 2: the gfx950 correction, as summarise_pmc.py)' % (v, v * 2048 / 1e9), 'ms %.2f' % ms)
PY
done
