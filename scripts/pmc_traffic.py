"""Build profiles/pmc_traffic.json from the per-kernel counter CSVs that `scripts/gpu_round.sh <tag> pmc` leaves under
gpurun_out/<tag>/ (each counter set in its OWN rocprofv3 --pmc pass, --kernel-trace only, over
`python bench.py --steps 4 --warmup 6 --no-cpu-baseline --no-host-abi --no-stats`).

usage: pmc_traffic.py <dir with pmc_fetch.csv pmc_write.csv pmc_l2.csv pmc_sq.csv> <out.json> [fetch_factor] [config] [scale] [opt,opt,...]
       pmc_traffic.py --merge <entry.json> [<entry.json> ...]      add / replace entries of profiles/pmc_traffic.json

The output is ONE entry {key, config, scale, opts, n_gpus, csrc_sha16, kernels{...}}: csrc_sha16 (bench.csrc_sha16) names the sources the
counters were taken on -- bench.py quotes an entry only for the same key on the same sources.

fetch_factor: what FETCH_SIZE has to be multiplied with for THIS access pattern; from the calibration run
(scripts/calib_fetch.hip, profiles/r02_calib_fetch.md): gfx950's FETCH_SIZE counts 64 B per fabric read request, a coalesced 16 B/lane
stream issues 128-byte requests (factor 2), the 8..64-byte gathers of K1 / K1Q issue 64-byte requests (factor 1)."""
import collections
import csv
import json
import os
import sys

FAMILIES = (("k1q_kernel", "k1q_dense"), ("k1_kernel", "k1_sparse"), ("k1t_kernel", "k1_sparse"), ("k1g_kernel", "k1g_dense_x"),
            ("k2_topk", "k2_topk"), ("k0_prolongate", "k0_prolongate"), ("sort_", "k1_sort_items"))


def family(kernel_name):
    for sub, fam in FAMILIES:
        if sub in kernel_name:
            return fam
    return None


N_STEPS = 4.0   # the profiled command runs --steps 4 --warmup 6; only its LAST four steps are counted: the pruning feedback (xrl_set_option
                # "adaptive") may switch layers to their unstaged kernels during the first steps, and the timed steps of bench.py run in the settled state


def last_steps_start(rows):
    """Start_Timestamp of the first dispatch of the last N_STEPS steps.  A step begins with a k1q_kernel / k0_prolongate / xguard launch that does
    not directly follow another beam-search kernel of the same step's opening run (K1Q launches may come in pairs: fused levels, then a wide layer)."""
    disp = sorted({(int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in rows if family(r["Kernel_Name"]) is not None or "xguard" in r["Kernel_Name"]})
    if not disp:
        return 0
    markers = sorted(int(r["Start_Timestamp"]) for r in rows if "step_marker_kernel" in r["Kernel_Name"])
    if markers:            # round 5: the library marks every predict (XRL_STEP_MARKER=1, set by scripts/gpu_round.sh pmc)
        markers = sorted(set(markers))
        return markers[-int(N_STEPS)] if len(markers) >= int(N_STEPS) else markers[0]
    opener = disp[0][1].split("<")[0]
    starts, prev_open = [], False
    for t, name in disp:
        is_open = name.split("<")[0] == opener
        if is_open and not prev_open:
            starts.append(t)
        prev_open = is_open
    return starts[-int(N_STEPS)] if len(starts) >= int(N_STEPS) else starts[0]


def per_family(path, counter):
    """counter value per STEP of every kernel family over the last N_STEPS steps: the sum over all launches of the family (both phases of a
    bound-pruned layer, every layer the family serves) divided by the number of steps"""
    if not os.path.exists(path):
        return {}
    rows = list(csv.DictReader(open(path)))
    t0 = last_steps_start(rows)
    agg = collections.defaultdict(list)
    for r in rows:
        fam = family(r["Kernel_Name"])
        if r["Counter_Name"] != counter or fam is None or int(r["Start_Timestamp"]) < t0:
            continue
        agg[fam].append(float(r["Counter_Value"]))
    return {k: {"launches": len(v) / N_STEPS, "mean": sum(v) / N_STEPS} for k, v in agg.items()}


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (csrc_sha16 / pmc_key only; its heavy imports live in main())

if sys.argv[1] == "--merge":
    tfile = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    cur = json.load(open(tfile)) if os.path.exists(tfile) else {}
    if "entries" not in cur:
        cur = {"version": 2, "entries": {}}
    for path in sys.argv[2:]:
        e = json.load(open(path))
        cur["entries"][e["key"]] = e
        print("merged", e["key"], e["csrc_sha16"], sorted(e["kernels"]))
    json.dump(cur, open(tfile, "w"), indent=1)
    sys.exit(0)

d, out = sys.argv[1:3]
factor = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
config = sys.argv[4] if len(sys.argv) > 4 else "amazon-670k"
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
opts = [o for o in sys.argv[6].split(",") if o] if len(sys.argv) > 6 else []
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
    kernels[fam] = {"launches_per_step": f[fam]["launches"], "fetch_kb_per_step_raw": fk, "write_kb_per_step_raw": wk,
                    "hbm_bytes_per_step": (factor * fk + wk) * 1024.0,
                    "l2_read_req_per_step": g(l2), "fabric_read_req_per_step": g(miss), "l2_hit_per_step": g(hit),
                    "valu_insts_per_step": g(valu), "salu_insts_per_step": g(salu), "valu_active_quad_cycles_per_step": g(vact)}
json.dump({
    "key": bench.pmc_key(config, scale, opts), "config": config, "scale": scale, "opts": opts, "n_gpus": 1, "fetch_factor": factor, "csrc_sha16": bench.csrc_sha16(),
    "source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum | SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU ...; "
              "separate runs, --kernel-trace only) over `python bench.py --steps 4 --warmup 6 --no-cpu-baseline --no-host-abi --no-stats` (the last 4 steps are counted: the pruning feedback has settled); SUM over all "
              "launches of the kernel family in a step (every layer it serves, both phases of a bound-pruned layer), averaged over the 4 steps; bytes = (FETCH_SIZE x fetch_factor + WRITE_SIZE) KiB, fetch_factor from the calibration run on 8..64-byte gathers "
              "(profiles/r02_calib_fetch.md); counts fabric requests, Infinity-Cache hits included",
    "kernels": kernels,
}, open(out, "w"), indent=1)
for fam, e in kernels.items():
    print(f"{fam}: per step: hbm bytes {e['hbm_bytes_per_step']:.3e}  l2 read req {e['l2_read_req_per_step']}  fabric req {e['fabric_read_req_per_step']}  valu {e['valu_insts_per_step']}  launches {e['launches_per_step']}")
