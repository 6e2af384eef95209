#!/bin/bash
# run bench.py (no CPU baseline) once per kernel-tuning variant under pecos_amd/lib/variants
mkdir -p gpurun_out
for so in pecos_amd/lib/variants/*.so; do
  tag=$(basename $so .so)
  PECOS_XRL_AMD_SO=$PWD/$so timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > gpurun_out/var_$tag.json 2> gpurun_out/var_$tag.err
  python - "$tag" gpurun_out/var_$tag.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    r = d["roofline"]
    import os; print(sys.argv[1], os.environ.get("BENCH_ARGS", ""), "q/s %.2fM" % (d["value"] / 1e6), "ms %.2f" % d["ms_per_step"], "k1/layer", [round(x["ms"], 2) for x in r["per_layer"]], "k2 %.2f" % r["per_kernel_ms_per_step"]["k2_topk"], d.get("parity", {}).get("indices_identical"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
