# Round 4, final pass A: the whole suite on the final sources, then the PMC passes of ten configurations.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log | cut -c1-300
CONFIGS="D C B E8 A F G H I P" ONLY=pmc bash tools/profile.sh r04_v48
