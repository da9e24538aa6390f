for o in "spec_k=3" "spec_k=4" "spec_k=5" "spec_k=5 --opt seeds=14" "spec_k=6 --opt seeds=14 --opt tile=64" "spec_k=4 --opt seeds=10 --opt tile=64"; do
  echo "== $o"; python bench.py --steps 3 --warmup 1 --no-cpu --opt $o 2>&1 | grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*\|\"seed_ms\": [0-9.]*\|\"rescore_ms\": [0-9.]*\|\"deepened_cells\": [0-9]*\|\"spec_rounds\": [0-9]*" | tr '\n' ' '; echo
done
