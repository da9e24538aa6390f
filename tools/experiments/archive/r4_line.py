import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["avg_launch_ms"], "visited %.4f" % (r["visited_fraction_of_groups"] or 0), d["result_digest"][:10])
except Exception as e:
    print(sys.argv[2], "unreadable", e)
