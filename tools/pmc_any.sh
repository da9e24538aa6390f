# SQ / LDS / traffic counters of one filter launch, any configuration: bash tools/pmc_any.sh [CONFIG] [--opt ...]
# (a pass that asks for more counters than the hardware collects at once aborts and can hang: keep the sets small)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=${1:-E8}; shift
O=gpurun_out/pmc_$C; mkdir -p $O
CMD="python3 bench.py --config $C --steps 1 --warmup 1 --no-cpu $*"
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/sq -o p --output-format csv -- $CMD > $O/sq.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD -d $O/lds -o p --output-format csv -- $CMD > $O/lds.log 2>&1
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $CMD > $O/fetch.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
for path in sorted(glob.glob(sys.argv[1] + '/*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(path)):
        if 'filter_kernel' in r['Kernel_Name'] or 'filter_lds_kernel' in r['Kernel_Name']:
            print(path.split('/')[2], r['Counter_Name'], '%.4g' % float(r['Counter_Value']), 'ms %.2f' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY
