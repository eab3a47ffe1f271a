import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    try: d=json.loads(line)
    except Exception: print(line[:200]); continue
    if "error" in d: print("ERROR", str(d)[:600]); continue
    print(d["ranks"], d["exchange"], "pieces", d["pieces"], "ms", d["ms_per_step"], d["bit_identical_to_whole_table"], d["env"])
