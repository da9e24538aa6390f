cd $GRAFT_REPO_ROOT
bash tools/build_variant.sh eu_pf1 -DMM_EUCLID_PREFETCH=1 > /dev/null
bash tools/build_variant.sh eu_w6 -DMM_EUCLID_WAVES=6 -DMM_EUCLID_PREFETCH=1 > /dev/null
bash tools/build_variant.sh eu_w6_nopf -DMM_EUCLID_WAVES=6 > /dev/null
for v in default eu_pf1 eu_w6 eu_w6_nopf default; do
  if [ $v = default ]; then unset MOSAIC_HIP_LIB; else export MOSAIC_HIP_LIB=$PWD/gpurun_out/$v.so; fi
  python bench.py --config B --steps 100 --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); t=d[\"last_step_device_ms\"]; print(\"B\", sys.argv[1], d[\"value\"], d[\"ms_per_step\"], \"filter\", t[\"score_ms\"], \"tile\", t[\"filter_tile\"])" $v
done
unset MOSAIC_HIP_LIB
python bench.py --config A --steps 500 --warmup 5 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A', d['value'], d['ms_per_step'])"
python -m pytest tests/test_gpu_small_jobs.py tests/test_gpu_generator.py -m gpu -q -x 2>&1 | tail -2
