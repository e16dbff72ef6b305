import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1], round(d["value"]/1e6,1), "Msamples/s", round(d["ms_per_step"],2), "ms; sx avg launch ms", round(r["avg_launch_ms"],4), "stages", {k:round(v,2) for k,v in r.get("stages",{}).items() if k.endswith("_ms")})
