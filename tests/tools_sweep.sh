for o in "seeds=6" "seeds=22" "seeds=54" "seeds=118" "seeds=246" "spec_k=6" "spec_k=16" "spec_k=6 --opt seeds=58"; do
  echo "== $o"; python bench.py --steps 3 --warmup 1 --no-cpu --opt $o 2>&1 | grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*\|\"seed_ms\": [0-9.]*\|\"rescore_ms\": [0-9.]*\|\"deepened_cells\": [0-9]*" | tr '\n' ' '; echo
done
