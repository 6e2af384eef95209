cd $GRAFT_REPO_ROOT; export XRL_BENCH_TIMEOUT=260 XRL_PMC_TIMEOUT=150
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
A="--config,dense-768,--scale,0.25,--rows,250000"
el gen; timeout 400 python bench.py --config dense-768 --scale 0.25 --rows 250000 --steps 2 --warmup 1 --no-cpu-baseline --no-host-abi --no-stats --parity-rows 0 2>&1 | grep -E "per-launch|workload|Error" | cut -c1-300
el pmc; bash scripts/gpu_round.sh r04x "pmc:$A" 2>&1 | grep -E "per step" | cut -c1-300
python scripts/pmc_traffic.py --merge gpurun_out/r04x/pmc_entry_1.json 2>&1 | tail -1; cp profiles/pmc_traffic.json gpurun_out/r04x_pmc_traffic.json
el bench; bash scripts/gpu_round.sh r04x "bench:$A,--steps,30" 2>&1 | grep -E "per-launch|host ABI|cpu reference|value" | cut -c1-420
el done
