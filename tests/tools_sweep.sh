for o in "lds_pad_kb=0" "lds_pad_kb=12" "lds_pad_kb=36" "lds_pad_kb=100"; do
  echo "== $o"; python bench.py --steps 3 --warmup 1 --no-cpu --opt $o 2>&1 | grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*" | tr '\n' ' '; echo
done
