cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out/r05u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for hs in 2 1; do
  XRL_HOST_STREAMS=$hs XRL_HOST_TIMING=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt$hs -- python $GRAFT_REPO_ROOT/scripts/host_abi_probe.py --calls 8 --reuse-alloc > $O/run$hs.log 2>&1
  f=$(find $O/kt$hs -name "*kernel_trace.csv" | head -1); python - "$f" $hs <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "xrl::" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id") or r.get("Stream_Id") or "") for r in rows)
# the last call: kernels after the last long gap
gaps = [(ev[i + 1][0] - ev[i][1], i) for i in range(len(ev) - 1)]
cut = max(i for g, i in gaps if g > 2_000_000) + 1
last = ev[cut:]
span = (last[-1][1] - last[0][0]) / 1e6
busy = 0; cur_s, cur_e = last[0][0], last[0][1]; overlap = 0
for s, e, n, q in last[1:]:
    if s < cur_e:
        overlap += min(e, cur_e) - s; cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
busy += cur_e - cur_s
print(f"XRL_HOST_STREAMS={sys.argv[2]}: last call: {len(last)} kernels over {span:.2f} ms, GPU busy (union of kernel intervals) {busy / 1e6:.2f} ms, sum of kernel durations {sum(e - s for s, e, n, q in last) / 1e6:.2f} ms, time with two kernels running {overlap / 1e6:.2f} ms; queues {sorted(set(q for *_, q in last))}")
PY
  tail -1 $O/run$hs.log | cut -c1-200; rm -rf $O/kt$hs
done
