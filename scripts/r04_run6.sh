cd $GRAFT_REPO_ROOT; export XRL_SKIP_HUGE=1
bash scripts/gpu_round.sh r04f tests bench:--steps,20,--no-cpu-baseline,--no-host-abi,--no-stats,--parity-rows,0 bench:--config,wiki10-31k,--steps,50,--no-cpu-baseline,--no-host-abi bench:--config,wiki10-31k,--steps,50,--no-cpu-baseline,--no-host-abi,--opt,prune_mid=0 bench:--config,eurlex-4k,--steps,50,--no-cpu-baseline,--no-host-abi bench:--config,amazon-670k-hard,--steps,20,--no-cpu-baseline,--no-host-abi,--no-stats,--parity-rows,0
echo "== default, previous build (w8 without the scalar-offset rows)"; PECOS_XRL_AMD_SO=$PWD/pecos_amd/lib/libxrl_amd_w8.so python bench.py --steps 20 --no-cpu-baseline --no-host-abi --no-stats --parity-rows 0 2>&1 | grep -E "per-launch" | cut -c1-120
python scripts/host_abi_probe.py --calls 12 > gpurun_out/r04f/probe.json 2> gpurun_out/r04f/probe.err; grep -E "probe|xrl host" gpurun_out/r04f/probe.err | cut -c1-330
python scripts/host_abi_probe.py --calls 12 --reuse-alloc > gpurun_out/r04f/probe_reuse.json 2> gpurun_out/r04f/probe_reuse.err; grep -E "probe\]" gpurun_out/r04f/probe_reuse.err | cut -c1-100
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04f/trace -- python $GRAFT_REPO_ROOT/scripts/host_abi_probe.py --calls 9 > $GRAFT_REPO_ROOT/gpurun_out/r04f/trace.log 2>&1
cd $GRAFT_REPO_ROOT; grep -E "probe\]" gpurun_out/r04f/trace.log | cut -c1-100; du -sh gpurun_out/r04f/trace; find gpurun_out/r04f/trace -name "*.csv" | head; for f in $(find gpurun_out/r04f/trace -name "*hip_api_trace.csv"); do python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
long=[r for r in rows if int(r['End_Timestamp'])-int(r['Start_Timestamp'])>1_000_000]
print('hip api calls',len(rows),'longer than 1 ms:',len(long))
t0=int(rows[0]['Start_Timestamp'])
for r in long[:80]: print(r['Function'], round((int(r['Start_Timestamp'])-t0)/1e6,2),'ms  dur',round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,3))
PY
done
find gpurun_out/r04f/trace -name "*hip_api_trace.csv" -size +20M -delete
