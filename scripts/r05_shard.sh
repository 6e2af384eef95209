cd $GRAFT_REPO_ROOT
AB_ROWS=61250 timeout 300 python scripts/ab.py amazon-670k 1.0 100 "" "overlap_min_rows=1" "overlap_min_rows=1,sort_rest=0" "sort_rest=0" 2>&1 | grep "ms/step" | cut -c1-420
AB_ROWS=122500 timeout 300 python scripts/ab.py amazon-670k 1.0 60 "" "overlap_min_rows=1" 2>&1 | grep "ms/step" | cut -c1-200
