"""Build profiles/pmc_traffic.json from rocprofv3 counter-collection CSVs taken over
`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats` (each counter set in its own run, --kernel-trace only).

usage: pmc_traffic.py <fetch.csv> <write.csv> <l2req.csv|-> <out.json> [fetch_factor]

fetch_factor: what FETCH_SIZE has to be multiplied with for THIS access pattern; taken from the calibration run
(scripts/calib_fetch.hip, profiles/r02_calib_fetch.md): gfx950's FETCH_SIZE counts 64 B per fabric read request, a coalesced 16 B/lane
stream issues 128-byte requests (factor 2), the 8..64-byte gathers of K1 / K1Q issue 64-byte requests (factor 1)."""
import collections, csv, json, sys

FAMILIES = (("k1q_kernel", "k1q_dense"), ("k1t_kernel", "k1t_sparse"), ("k1_kernel", "k1_sparse"), ("k2_topk", "k2_topk"), ("k0_prolongate", "k0_prolongate"))


def family(kernel_name):
    for sub, fam in FAMILIES:
        if sub in kernel_name:
            return fam
    return None


def per_family(path, counter):
    agg = collections.defaultdict(list)
    if path == "-":
        return {}
    for r in csv.DictReader(open(path)):
        fam = family(r["Kernel_Name"])
        if r["Counter_Name"] != counter or fam is None:
            continue
        agg[fam].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in agg.items()}


fetch_csv, write_csv, l2_csv, out = sys.argv[1:5]
factor = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
f = per_family(fetch_csv, "FETCH_SIZE"); w = per_family(write_csv, "WRITE_SIZE"); l2 = per_family(l2_csv, "TCP_TCC_READ_REQ_sum")
kernels = {}
for fam in f:
    fk = f[fam]["mean"]; wk = w.get(fam, {"mean": 0.0})["mean"]
    kernels[fam] = {"launches_sampled": f[fam]["launches"], "fetch_kb_per_launch_raw": fk, "write_kb_per_launch_raw": wk,
                    "hbm_bytes_per_launch": (factor * fk + wk) * 1024.0,
                    "l2_read_req_per_launch": l2.get(fam, {}).get("mean")}
json.dump({
    "config": "amazon-670k", "scale": 1.0, "n_gpus": 1, "fetch_factor": factor,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCP_TCC_READ_REQ_sum (separate passes, --kernel-trace only) over "
              "`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats`; mean over the launches of the family in a step; "
              "bytes = (FETCH_SIZE x fetch_factor + WRITE_SIZE) KiB, fetch_factor from the calibration run on 8..64-byte gathers "
              "(profiles/r02_calib_fetch.md); counts fabric requests, Infinity-Cache hits included",
    "kernels": kernels,
}, open(out, "w"), indent=1)
for fam, e in kernels.items():
    print(f"{fam}: hbm bytes per launch {e['hbm_bytes_per_launch']:.3e}  l2 read req {e['l2_read_req_per_launch']}")
