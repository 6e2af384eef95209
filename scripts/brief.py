"""Print the headline fields of a bench.py JSON line (last line of the given file)."""
import json, sys
line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d.get("roofline"))
for k in ("kernels", "parity", "cpu_baseline"):
    if k in d: print(k, d[k])
