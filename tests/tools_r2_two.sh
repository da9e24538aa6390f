cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
g() { grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*\|\"visited_groups\": [0-9]*\|\"bound_dropped\": [0-9]*" | tr "\n" " "; echo; }
python -m pytest tests/test_gpu_filter.py tests/test_gpu_generator.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_detail_steps.py -m gpu -q -x 2>&1 | tail -3
for c in D E8 C; do for o in 1 0; do echo "== $c two_level=$o"; python3 bench.py --config $c --steps 6 --warmup 2 --no-cpu --opt two_level=$o 2>/dev/null | g; done; done
