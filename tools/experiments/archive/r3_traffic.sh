# Round 3, after the filter form lost 9 % of its instructions: the fabric traffic runs at the read-stream rate, so the knobs that trade
# traffic against idle slots are measured again on config D (item size, XCD sub-group, how many groups leave plane order, long-first split).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_traffic.txt; : > $O
bash tools/build_variant.sh ls0 -DMM_LONG_SPLIT=0.0f > /dev/null 2>&1
run() { # label, lib, opts...
  label=$1; lib=$2; shift 2
  if [ $lib = default ]; then unset MOSAIC_HIP_LIB; else export MOSAIC_HIP_LIB=$PWD/gpurun_out/$lib.so; fi
  python3 bench.py --config ${CFG:-D} --steps 5 --warmup 2 --no-cpu "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print('%-28s %.0f cells/s %.3f ms/step filter %.3f tile %d' % (sys.argv[1], d['value'], d['ms_per_step'], t['score_ms'], t['filter_tile']), d.get('result_digest','')[:10])" "$label" >> $O
  unset MOSAIC_HIP_LIB
}
for round in 1 2; do
run default default
run long_split_0 ls0
run tile16 default --opt tile=16
run tile24 default --opt tile=24
run tile48 default --opt tile=48
run tile64 default --opt tile=64
run xcd32 default --opt xcd_cells=32
run xcd128 default --opt xcd_cells=128
run sort8 default --opt sort_groups=8
run sort4 default --opt sort_groups=4
run sort0 default --opt sort_groups=0
run tile16_ls0 ls0 --opt tile=16
done
cat $O
