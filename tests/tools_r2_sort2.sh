cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
g() { grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*" | tr "\n" " "; echo; }
for xc in 8 16 32 64; do for t in 16 32; do
  echo "== xcd_cells=$xc tile=$t"; python3 bench.py --steps 6 --warmup 2 --no-cpu --opt xcd_cells=$xc --opt tile=$t 2>/dev/null | g
done; done
