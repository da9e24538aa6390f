cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
g() { grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*\|\"visited_groups\": [0-9]*" | tr "\n" " "; echo; }
for o in "sort_groups=1" "sort_groups=0"; do
  echo "== lds tile=128 $o"; python3 bench.py --steps 6 --warmup 2 --no-cpu --opt lds_filter=1 --opt tile=128 --opt $o 2>/dev/null | g
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$o -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu --opt $o > $O/fetch_$o.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $O/tcc_$o -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu --opt $o > $O/tcc_$o.log 2>&1
done
python3 - $O <<'PY'
import csv,glob,sys
for path in sorted(glob.glob(sys.argv[1]+'/*/**/*counter_collection.csv',recursive=True)):
    for r in csv.DictReader(open(path)):
        if 'filter_kernel' in r['Kernel_Name']: print(path.split('/')[2], r['Counter_Name'], r['Counter_Value'])
PY
