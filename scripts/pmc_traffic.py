"""Build profiles/pmc_traffic.json from two rocprofv3 counter-collection CSVs (FETCH_SIZE pass, WRITE_SIZE pass)
taken over `python bench.py --steps 3 --warmup 1 --no-cpu-baseline` (each counter in its own run, with
--kernel-trace only).  usage: pmc_traffic.py <fetch.csv> <write.csv> <out.json> [kernel-substring]"""
import collections, csv, json, sys

def per_shape(path, counter, kern):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or kern not in r["Kernel_Name"]:
            continue
        agg[r["Kernel_Name"].split("(")[0] + "|grid=" + r["Grid_Size"]].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean_kb": sum(v) / len(v)} for k, v in agg.items()}

fetch_csv, write_csv, out = sys.argv[1:4]
kern = sys.argv[4] if len(sys.argv) > 4 else "k1_kernel"
f = per_shape(fetch_csv, "FETCH_SIZE", kern); w = per_shape(write_csv, "WRITE_SIZE", kern)
def mean_over_launches(d):   # every shape weighted by its launch count = mean over the K1 launches of a step
    n = sum(v["launches"] for v in d.values())
    return sum(v["mean_kb"] * v["launches"] for v in d.values()) / max(1, n)
fk, wk = mean_over_launches(f), mean_over_launches(w)
json.dump({
    "config": "amazon-670k", "scale": 1.0, "n_gpus": 1, "kernel": "k1_sparse",
    "fetch_kb_per_launch_raw": fk, "write_kb_per_launch_raw": wk,
    "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) over `python bench.py --steps 3 "
              "--warmup 1 --no-cpu-baseline`; mean over the 5 per-layer K1 launches of a step; FETCH_SIZE doubled per "
              "MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B); counts fabric requests incl. Infinity-Cache hits; "
              "not calibrated for 8-byte gathers",
    "per_shape": {"fetch": f, "write": w},
}, open(out, "w"), indent=1)
print("hbm bytes per K1 launch: %.3e" % ((2.0 * fk + wk) * 1024.0))
