cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "conv_layer or det500m or c3_det or r50" > gpurun_out/talltail_test.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/talltail_test.log
python scripts/layer_times.py det 128 2>&1 | grep "merged\|fix-up\|total" | cut -c1-100
