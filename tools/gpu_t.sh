cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s -k "retrieval or train_transform or dist" > $O/pytest_new.txt 2>&1
grep -E "rel |fused train|passed|failed" $O/pytest_new.txt | tail -12
