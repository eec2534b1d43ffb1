import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
if "frames_per_s" in d:
    print("default:", round(d["value"],1), {k:round(v,1) for k,v in d["frames_per_s"].items() if k.startswith(("front","with"))})
else:
    c=d["config"]["rank0_clip"]; print("video:", round(d["value"],1), round(c["frontend_seconds"],2), round(c["backend_seconds"],2))
