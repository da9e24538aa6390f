# Round 4: the audit as a kernel of its own after the filter -- its tests, the goldens with every dropped candidate audited, and
# the step against the binary before any audit (tmp_ab/noaudit.so: commit b7418ad built in the container), same box, alternating.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_audit_ab; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_filter.py -x -q -s -k "audit" > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -a "^broken\|^\.broken\|passed\|failed\|Error" $O/pytest.log | cut -c1-250
timeout -k 10 600 python3 -m pytest tests/test_golden_grids.py tests/test_gpu_fullsize.py -x -q -s -m gpu > $O/pytest2.log 2>&1; echo "pytest2 rc $?"; grep -a "audited\|passed\|failed" $O/pytest2.log | cut -c1-200
for rep in 1 2 3; do
  for c in D E8 G C; do
    MOSAIC_HIP_LIB=tmp_ab/noaudit.so python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu > $O/x.json 2>/dev/null; python3 tools/experiments/r4_line.py $O/x.json "$c before the audit"
    for a in 0 1024 256; do
      python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu --opt audit=$a > $O/x.json 2>/dev/null; python3 tools/experiments/r4_line.py $O/x.json "$c audit=$a"
    done
  done
done
