# Round 5, final pass A: the whole suite on the final sources, then the PMC passes of ten configurations.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log | cut -c1-300
CONFIGS="D C B E8 A F G H I P" ONLY=pmc bash tools/profile.sh r05_v52 > $O/profile_a.log 2>&1; tail -3 $O/profile_a.log
ls gpurun_out/r05_v52
