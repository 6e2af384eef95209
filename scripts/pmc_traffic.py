"""Build profiles/pmc_traffic.json from the per-kernel counter CSVs that `scripts/gpu_round.sh <tag> pmc` leaves under
gpurun_out/<tag>/ (each counter set in its OWN rocprofv3 --pmc pass, --kernel-trace only, over
`python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats`).

usage: pmc_traffic.py <dir with pmc_fetch.csv pmc_write.csv pmc_l2.csv pmc_sq.csv> <out.json> [fetch_factor] [config] [scale]

fetch_factor: what FETCH_SIZE has to be multiplied with for THIS access pattern; from the calibration run
(scripts/calib_fetch.hip, profiles/r02_calib_fetch.md): gfx950's FETCH_SIZE counts 64 B per fabric read request, a coalesced 16 B/lane
stream issues 128-byte requests (factor 2), the 8..64-byte gathers of K1 / K1Q issue 64-byte requests (factor 1)."""
import collections
import csv
import json
import os
import sys

FAMILIES = (("k1q_kernel", "k1q_dense"), ("k1_kernel", "k1_sparse"), ("k1g_kernel", "k1g_dense_x"),
            ("k2_topk", "k2_topk"), ("k0_prolongate", "k0_prolongate"), ("sort_", "k1_sort_items"))


def family(kernel_name):
    for sub, fam in FAMILIES:
        if sub in kernel_name:
            return fam
    return None


def per_family(path, counter, per_step_sum=False):
    """mean counter value per launch of every kernel family (sort_*: the four sort kernels of a step are one 'launch')"""
    if not os.path.exists(path):
        return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        fam = family(r["Kernel_Name"])
        if r["Counter_Name"] != counter or fam is None:
            continue
        agg[fam].append(float(r["Counter_Value"]))
    out = {}
    for k, v in agg.items():
        n = len(v) / 4.0 if k == "k1_sort_items" else len(v)
        out[k] = {"launches": n, "mean": sum(v) / max(n, 1)}
    return out


d, out = sys.argv[1:3]
factor = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
config = sys.argv[4] if len(sys.argv) > 4 else "amazon-670k"
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
f = per_family(os.path.join(d, "pmc_fetch.csv"), "FETCH_SIZE")
w = per_family(os.path.join(d, "pmc_write.csv"), "WRITE_SIZE")
l2 = per_family(os.path.join(d, "pmc_l2.csv"), "TCP_TCC_READ_REQ_sum")
miss = per_family(os.path.join(d, "pmc_l2.csv"), "TCC_MISS_sum")
hit = per_family(os.path.join(d, "pmc_l2.csv"), "TCC_HIT_sum")
valu = per_family(os.path.join(d, "pmc_sq.csv"), "SQ_INSTS_VALU")
salu = per_family(os.path.join(d, "pmc_sq.csv"), "SQ_INSTS_SALU")
vact = per_family(os.path.join(d, "pmc_sq.csv"), "SQ_ACTIVE_INST_VALU")
kernels = {}
for fam in f:
    fk = f[fam]["mean"]; wk = w.get(fam, {"mean": 0.0})["mean"]
    g = lambda t: t.get(fam, {}).get("mean")
    kernels[fam] = {"launches_sampled": f[fam]["launches"], "fetch_kb_per_launch_raw": fk, "write_kb_per_launch_raw": wk,
                    "hbm_bytes_per_launch": (factor * fk + wk) * 1024.0,
                    "l2_read_req_per_launch": g(l2), "fabric_read_req_per_launch": g(miss), "l2_hit_per_launch": g(hit),
                    "valu_insts_per_launch": g(valu), "salu_insts_per_launch": g(salu), "valu_active_quad_cycles_per_launch": g(vact)}
json.dump({
    "config": config, "scale": scale, "n_gpus": 1, "fetch_factor": factor,
    "source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum | SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU ...; "
              "separate runs, --kernel-trace only) over `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats`; mean over the "
              "launches of the family; bytes = (FETCH_SIZE x fetch_factor + WRITE_SIZE) KiB, fetch_factor from the calibration run on 8..64-byte gathers "
              "(profiles/r02_calib_fetch.md); counts fabric requests, Infinity-Cache hits included",
    "kernels": kernels,
}, open(out, "w"), indent=1)
for fam, e in kernels.items():
    print(f"{fam}: hbm bytes per launch {e['hbm_bytes_per_launch']:.3e}  l2 read req {e['l2_read_req_per_launch']}  fabric req {e['fabric_read_req_per_launch']}  valu {e['valu_insts_per_launch']}")
