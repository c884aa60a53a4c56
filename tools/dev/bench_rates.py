import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], {k:round(v/1e6,2) for k,v in d["config"].items() if "per_s" in k})
