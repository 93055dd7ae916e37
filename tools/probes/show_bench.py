import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d.get("infer",{}).get("ms_per_step"), d["roofline"]["kernel"][:50], d["roofline"].get("exclusive",{}).get("frac"))
