# Round 3: the whole GPU suite, then every measurement DESIGN.md quotes, in one call on one box.
cd $GRAFT_REPO_ROOT
[ "$SKIP_SUITE" = 1 ] || { python -m pytest tests -q -x -m gpu > gpurun_out/r03_gpu_suite.txt 2>&1; tail -3 gpurun_out/r03_gpu_suite.txt; }
bash tools/profile.sh r03_v34 > gpurun_out/r03_v34_profile.log 2>&1
python3 tools/filter_accuracy.py > gpurun_out/r03_filter_accuracy.txt 2>&1
bash tools/pmc_any.sh C > gpurun_out/r03_pmc_C.txt 2>&1
bash tools/pmc_any.sh B > gpurun_out/r03_pmc_B.txt 2>&1
python3 bench.py --sweep > gpurun_out/r03_sweep.json 2> gpurun_out/r03_sweep.err
python3 bench.py --stage build > gpurun_out/r03_stage_build.json 2>/dev/null
python3 bench.py --stage preprocess > gpurun_out/r03_stage_pre.json 2>/dev/null
python3 tools/smooth_bench.py > gpurun_out/r03_smooth.txt 2>&1
bash tools/experiments/r3_small_jobs.sh r03_small5 > gpurun_out/r03_small5.txt 2>&1
echo done
