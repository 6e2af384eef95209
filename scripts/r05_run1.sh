cd $GRAFT_REPO_ROOT; export XRL_SKIP_HUGE=1
O=gpurun_out/r05a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -k "goldens or toy or fuzz or ties or guard or edge or scaled" 2>&1 | tail -40 > $O/pytest_quick.log; cat $O/pytest_quick.log
timeout 900 python scripts/ab.py amazon-670k 1.0 20 "" "qsort=0" "qsort_min_parents=16" "qsort=0,prune=0" "prune=0" > $O/ab_default.log 2>&1; grep -v "^\s*$" $O/ab_default.log | cut -c1-900 | tail -8
timeout 900 python scripts/ab.py amazon-670k-hard 1.0 20 "" "qsort=0" "qsort_min_parents=16" "presence=0" > $O/ab_hard.log 2>&1; cut -c1-900 $O/ab_hard.log | tail -8
