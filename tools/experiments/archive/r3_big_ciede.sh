# Round 3: 128 x 128 CIEDE2000 cells (config G): how many of the 64 groups are visited by descending bound (sort_groups; the
# rest in plane order) and how often the exit test runs (check_every), in one call.
cd $GRAFT_REPO_ROOT
for o in "16 1" "32 1" "64 1" "24 1" "0 1" "16 1"; do
  set -- $o
  python3 bench.py --config G --steps 6 --warmup 2 --no-cpu --opt sort_groups=$1 --opt check_every=$2 --opt phase_timing=1 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['last_step_device_ms']; print('G sort_groups', sys.argv[1], 'check_every', sys.argv[2], '%.0f cells/s %.3f ms/step filter %.3f visited %d of %d' % (d['value'], d['ms_per_step'], t['score_ms'], t['visited_groups'], t['total_groups']), d.get('result_digest','')[:10])" $1 $2
done
