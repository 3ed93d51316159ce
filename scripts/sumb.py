import json,sys
for f in sys.argv[1:]:
    l=json.loads(open(f).read().strip().splitlines()[-1])
    k=l["kernel_ms_per_step"]
    print(f, "ring", round(l["ms_per_step"],2), "resident", round(l["resident_in_hbm"]["ms_per_step"],2), "host", l.get("from_pinned_host",{}).get("ms_per_step"), "prof", l["roofline"]["ms_per_step"], k["conv_dgrad"], k["conv_fwd"])
