cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q; mkdir -p $O
for xc in 16 32 64; do for t in 16 32; do
  rocprofv3 --pmc FETCH_SIZE -d $O/f_${xc}_$t -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu --opt xcd_cells=$xc --opt tile=$t > $O/f_${xc}_$t.log 2>&1
done; done
python3 - $O <<'PY'
import csv,glob,sys
for path in sorted(glob.glob(sys.argv[1]+'/f_*/**/*counter_collection.csv',recursive=True)):
    for r in csv.DictReader(open(path)):
        if 'filter_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
            ms=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
            print(path.split('/')[2], 'FETCH GB x2 = %.1f'%(float(r['Counter_Value'])*2*1024/1e9), 'kernel ms %.2f'%ms)
PY
