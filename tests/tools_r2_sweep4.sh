cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
g() { grep -o "\"value\": [0-9.]*\|\"score_ms\": [0-9.]*" | tr "\n" " "; echo; }
for c in D E8; do for t in 32 48 64; do for xc in 32 64 128; do
  echo "== $c tile=$t xcd_cells=$xc"; python3 bench.py --config $c --steps 6 --warmup 2 --no-cpu --opt tile=$t --opt xcd_cells=$xc 2>/dev/null | g
done; done; done
for s in 10 14 22 30; do echo "== D seeds=$s"; python3 bench.py --steps 6 --warmup 2 --no-cpu --opt seeds=$s 2>/dev/null | g; done
