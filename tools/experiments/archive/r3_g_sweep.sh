cd $GRAFT_REPO_ROOT
for o in "0 0" "8 0" "4 0" "64 0" "0 16" "0 24" "0 0"; do
  set -- $o
  python3 bench.py --config G --steps 5 --warmup 2 --no-cpu --opt xcd_cells=$1 --opt tile=$2 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print('G xcd_cells', sys.argv[1], 'tile', sys.argv[2], '%.0f cells/s %.3f ms/step filter %.3f tile %d' % (d['value'], d['ms_per_step'], t['score_ms'], t['filter_tile']), d['result_digest'][:10])" $1 $2
done
