cd $GRAFT_REPO_ROOT; export XRL_SKIP_HUGE=1
O=gpurun_out/r05p; mkdir -p $O
for hs in 2 1; do
  echo "== XRL_HOST_STREAMS=$hs"
  XRL_HOST_STREAMS=$hs XRL_HOST_TIMING=0 timeout 300 python scripts/host_abi_probe.py --calls 12 --reuse-alloc > $O/probe_hs$hs.log 2>&1; tail -4 $O/probe_hs$hs.log | cut -c1-300
done
XRL_HOST_STREAMS=2 XRL_HOST_TIMING=0 timeout 300 python scripts/host_abi_probe.py --config amazon-670k-hard --calls 8 --reuse-alloc > $O/probe_hard_hs2.log 2>&1; tail -2 $O/probe_hard_hs2.log | cut -c1-300
XRL_HOST_STREAMS=1 XRL_HOST_TIMING=0 timeout 300 python scripts/host_abi_probe.py --config amazon-670k-hard --calls 8 --reuse-alloc > $O/probe_hard_hs1.log 2>&1; tail -2 $O/probe_hard_hs1.log | cut -c1-300
timeout 900 python -m pytest tests -m gpu -q -x -k "headline or full_size_vs_reference or multi_device or device_inputs or reference_binding or dense_input" 2>&1 | tail -5
