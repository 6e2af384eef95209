cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out/r05m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/scripts/mfma_rescore_cost.py > $O/run.log 2>&1
tail -8 $O/run.log
for f in $(find $O/kt -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; head -12 $f | cut -c1-200; done
rm -rf $O/kt
