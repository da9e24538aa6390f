cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
g() { grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*\|\"visited_groups\": [0-9]*" | tr "\n" " "; echo; }
python -m pytest tests/test_gpu_filter.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -2
for m in 16 0 1 2 3 4 6 8 16; do echo "== sort_groups=$m"; python3 bench.py --steps 6 --warmup 2 --no-cpu --opt sort_groups=$m 2>/dev/null | g; done
for m in 0 2 4 16; do echo "== E8 sort_groups=$m"; python3 bench.py --config E8 --steps 6 --warmup 2 --no-cpu --opt sort_groups=$m 2>/dev/null | g; done
