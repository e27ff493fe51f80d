cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 600 python tools/dual_stream_probe.py mixed > $O/dual.txt 2> $O/dual.err
cat $O/dual.txt; tail -5 $O/dual.err
