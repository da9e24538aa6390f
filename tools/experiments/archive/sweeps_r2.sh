# Round-2 tuning sweeps of the filter kernel (run on the GPU box; every arm of a sweep in ONE call: the pool's
# devices clock differently, numbers from different calls differ by a few per cent).
#   bash tools/experiments/sweeps_r2.sh tune | order | level | lds | traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 6 --warmup 2 --no-cpu"
g() { grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*\|\"seed_ms\": [0-9.]*\|\"visited_groups\": [0-9]*\|\"bound_dropped\": [0-9]*" | tr "\n" " "; echo; }
case "${1:-tune}" in
tune)      # candidates per item x cells per XCD sub-group, seeds, waves per SIMD
  for c in D E8; do for t in 32 48 64; do for xc in 32 64 128; do
    echo "== $c tile=$t xcd_cells=$xc"; $B --config $c --opt tile=$t --opt xcd_cells=$xc 2>/dev/null | g; done; done; done
  for s in 10 14 22 30; do echo "== D seeds=$s"; $B --opt seeds=$s 2>/dev/null | g; done
  echo "== D xcd_aware=0"; $B --opt xcd_aware=0 2>/dev/null | g
  for w in 4 6; do tools/build_variant.sh w$w -DMM_FILTER_WAVES=$w >/dev/null 2>&1 && { echo "== waves $w"; MOSAIC_HIP_LIB=gpurun_out/w$w.so $B 2>/dev/null | g; rm -f gpurun_out/w$w.so; }; done ;;
order)     # how many of the largest-bound groups are visited first
  for c in D E8 C; do for m in 0 1 2 3 4 6 8 16; do echo "== $c sort_groups=$m"; $B --config $c --opt sort_groups=$m 2>/dev/null | g; done; done ;;
level)     # the bound's stage 1 (several candidates per wave pass) against the per-candidate form, the interleaved image copies
  for c in D E8 C; do for o in 1 0; do echo "== $c two_level=$o"; $B --config $c --opt two_level=$o 2>/dev/null | g; done; done
  echo "== D interleave=0"; $B --opt interleave=0 2>/dev/null | g ;;
lds)       # the LDS-resident cell tile
  for t in 32 64 128; do echo "== D lds_filter=1 tile=$t"; $B --opt lds_filter=1 --opt tile=$t 2>/dev/null | g; done
  for c in C E8; do echo "== $c lds_filter=1"; $B --config $c --opt lds_filter=1 2>/dev/null | g; done ;;
traffic)   # fabric traffic of one launch (FETCH_SIZE x 2) against item size
  O=gpurun_out/traffic; mkdir -p $O
  for o in "" "--opt lds_filter=1 --opt tile=128" "--opt sort_groups=0"; do for xc in 16 32 64; do for t in 16 32; do
    d=$O/f_${xc}_${t}_$(echo $o | tr -d ' =-'); rocprofv3 --pmc FETCH_SIZE -d $d -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu --opt xcd_cells=$xc --opt tile=$t $o > $d.log 2>&1
  done; done; done
  python3 - $O <<'PY'
import csv, glob, sys
for path in sorted(glob.glob(sys.argv[1] + '/f_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(path)):
        if 'filter_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            print(path.split('/')[2], 'FETCH GB x2 = %.1f' % (float(r['Counter_Value']) * 2 * 1024 / 1e9),
                  'kernel ms %.2f' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY
  ;;
esac
