cd $GRAFT_REPO_ROOT
bash tools/knob_sweep.sh 8 13 13 2>&1 | tail -4
