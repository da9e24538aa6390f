# Round 3: stage 2 of filter_kernel hands out the likely-long candidates first (MM_LONG_SPLIT: share of the limit below which a
# candidate's block bound puts it in the first class; 0 = one class, the order of round 2).  Variants built on the box.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_long_first.txt; : > $O
for v in 0.0f 0.55f 0.8f 0.9f; do bash tools/build_variant.sh ls_$v -DMM_LONG_SPLIT=$v > /dev/null 2>&1; done
for round in 1 2; do
for v in default 0.0f 0.55f 0.8f 0.9f; do
  if [ $v = default ]; then unset MOSAIC_HIP_LIB; else export MOSAIC_HIP_LIB=$PWD/gpurun_out/ls_$v.so; fi
  for c in D E8 C; do
    k=6; [ $c = C ] && k=60
    python3 bench.py --config $c --steps $k --warmup 3 --no-cpu --opt phase_timing=1 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print(sys.argv[1], sys.argv[2], '%.0f cells/s %.3f ms/step filter %.3f' % (d['value'], d['ms_per_step'], t['score_ms']), d.get('result_digest','')[:10])" $c $v >> $O
  done
done
done
unset MOSAIC_HIP_LIB
cat $O
