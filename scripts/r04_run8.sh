cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04h
for v in "" _nofb; do
  for cfg in amazon-670k eurlex-4k; do
  echo "== variant '$v' $cfg"; PECOS_XRL_AMD_SO=$PWD/pecos_amd/lib/libxrl_amd$v.so python bench.py --config $cfg --steps 20 --no-cpu-baseline --no-host-abi --no-stats --parity-rows 0 2>&1 | grep -E "per-launch" | cut -c1-200
  done
done
