# round 4, the measurement call on a metered GPU (20 minutes left): every stage under its own short timeout, most valuable first.
# Per workload: counter passes first (scripts/gpu_round.sh pmc -> pmc_entry_N.json), merged into profiles/pmc_traffic.json ON THE BOX, then the bench line
# (which therefore quotes counters taken on exactly these sources); the merged counter file travels back under gpurun_out/.
cd $GRAFT_REPO_ROOT; export XRL_SKIP_HUGE=1 XRL_SKIP_FULLSIZE=1 XRL_TESTS_TIMEOUT=420 XRL_BENCH_TIMEOUT=280 XRL_PMC_TIMEOUT=170
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
(python scripts/gen_workload.py amazon-670k > gpurun_out/gen_a.log 2>&1; python scripts/gen_workload.py amazon-670k-hard > gpurun_out/gen_h.log 2>&1) &
GEN=$!
el smoke; timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
el tests; bash scripts/gpu_round.sh r04z tests 2>&1 | tail -4
one() { tag=$1; shift; cfgargs="$1"; shift; benchargs="$1"
  el "pmc $tag"; bash scripts/gpu_round.sh $tag "pmc:$cfgargs" 2>&1 | grep -E "per step" | cut -c1-300
  python scripts/pmc_traffic.py --merge gpurun_out/$tag/pmc_entry_1.json 2>&1 | tail -1; cp profiles/pmc_traffic.json gpurun_out/r04z_pmc_traffic.json
  el "bench $tag"; bash scripts/gpu_round.sh $tag "bench:$benchargs" 2>&1 | grep -E "per-launch|host ABI|cpu reference|value" | cut -c1-420; }
el "waiting for the workloads"; wait $GEN
one r04z_amazon "" "--steps,50"
one r04z_hard "--config,amazon-670k-hard" "--config,amazon-670k-hard,--steps,30"
one r04z_eurlex "--config,eurlex-4k" "--config,eurlex-4k,--steps,100"
one r04z_wiki "--config,wiki10-31k" "--config,wiki10-31k,--steps,100"
el shard; timeout 200 python bench.py --steps 30 --rows 61250 --no-cpu-baseline --no-host-abi --no-stats --parity-rows 0 2>&1 | grep -E "per-launch" | cut -c1-300
el done
