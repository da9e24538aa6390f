set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2e}
mkdir -p $O
python -m pytest tests/test_gpu_filter.py tests/test_gpu_host_cpp.py tests/test_gpu_generator.py -m gpu -q > $O/tests.log 2>&1 || tail -30 $O/tests.log
tail -3 $O/tests.log
B="python3 bench.py --steps 6 --warmup 2 --no-cpu"
g() { grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*\|\"visited_groups\": [0-9]*\|\"bound_dropped\": [0-9]*" | tr '\n' ' '; echo; }
echo "== default"; $B 2>$O/err.log | g
echo "== lds_filter=1"; $B --opt lds_filter=1 2>>$O/err.log | g
echo "== lds_filter=1 tile=128"; $B --opt lds_filter=1 --opt tile=128 2>>$O/err.log | g
echo "== lds_filter=1 tile=32"; $B --opt lds_filter=1 --opt tile=32 2>>$O/err.log | g
echo "== lds_filter=1 xcd_cells=32"; $B --opt lds_filter=1 --opt xcd_cells=32 2>>$O/err.log | g
echo "== lds C"; $B --config C --opt lds_filter=1 2>>$O/err.log | g
echo "== lds E8"; $B --config E8 --opt lds_filter=1 2>>$O/err.log | g
rocprofv3 --pmc FETCH_SIZE -d $O/fetch_lds -o p --output-format csv -- python3 bench.py --config D --steps 1 --warmup 0 --no-cpu --opt lds_filter=1 > $O/fetch_lds.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys
for path in glob.glob(sys.argv[1]+'/fetch_lds/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(path)):
        if 'filter_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': print('FETCH_SIZE KiB',r['Counter_Value'],'-> GB x2:',float(r['Counter_Value'])*2*1024/1e9, r['Kernel_Name'][:80], 'LDS',r['LDS_Block_Size'],'VGPR',r['VGPR_Count'],'scratch',r['Scratch_Size'])
PY
