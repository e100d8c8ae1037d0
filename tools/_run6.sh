cd $GRAFT_REPO_ROOT
bash tools/knob_sweep.sh 8 0 12 > gpurun_out/s6_knobs_a.log 2>&1
grep -c passed gpurun_out/s6_knobs_a.log
grep -n "failed\|error" gpurun_out/s6_knobs_a.log | head
