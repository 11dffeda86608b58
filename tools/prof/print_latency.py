import json,sys
for ln in sys.stdin.read().strip().splitlines():
    if ln.startswith("{"):
        d=json.loads(ln); print(d["value"], json.dumps(d["single_batch"]), json.dumps(d["end_to_end"]))
