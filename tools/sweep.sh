for o in "seeds=6" "seeds=10" "seeds=14" "seeds=22" "seeds=14 --opt tile=64" "seeds=22 --opt tile=64" "seeds=22 --opt tile=48"; do
  echo "== $o"; python bench.py --steps 4 --warmup 2 --no-cpu --opt $o 2>&1 | grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*\|\"seed_ms\": [0-9.]*\|\"rescore_ms\": [0-9.]*\|\"deepened_cells\": [0-9]*" | tr '\n' ' '; echo
done
