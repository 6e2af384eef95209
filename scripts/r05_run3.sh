cd $GRAFT_REPO_ROOT; export XRL_SKIP_HUGE=1
O=gpurun_out/r05d; mkdir -p $O
for v in "" _u1_32 _u2_16 _u01_16 _rl2 _rl2_u _allu; do
  [ -f pecos_amd/lib/libxrl_amd$v.so ] || continue
  echo "== variant '$v'"
  PECOS_XRL_AMD_SO=$PWD/pecos_amd/lib/libxrl_amd$v.so timeout 300 python scripts/ab.py amazon-670k 1.0 20 "" "qsort=0" > $O/ab_default$v.log 2>&1; grep "ms/step" $O/ab_default$v.log | cut -c1-330
done
for v in "" _rl2 _rl2_u _allu; do
  [ -f pecos_amd/lib/libxrl_amd$v.so ] || continue
  echo "== hard, variant '$v'"
  PECOS_XRL_AMD_SO=$PWD/pecos_amd/lib/libxrl_amd$v.so timeout 400 python scripts/ab.py amazon-670k-hard 1.0 20 "" > $O/ab_hard$v.log 2>&1; grep "ms/step" $O/ab_hard$v.log | cut -c1-330
done
# small launches: a 61 250-row shard (what one of eight GPUs gets)
for v in "" _rl2_u _allu; do
  [ -f pecos_amd/lib/libxrl_amd$v.so ] || continue
  echo "== 61250 rows, variant '$v'"
  AB_ROWS=61250 PECOS_XRL_AMD_SO=$PWD/pecos_amd/lib/libxrl_amd$v.so timeout 300 python scripts/ab.py amazon-670k 1.0 50 "" > $O/ab_shard$v.log 2>&1; grep "ms/step" $O/ab_shard$v.log | cut -c1-330
done
