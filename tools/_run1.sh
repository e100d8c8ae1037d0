set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_logdist.py -x -q > gpurun_out/s1_logdist.log 2>&1; echo "logdist rc=$?"
tail -3 gpurun_out/s1_logdist.log
RM_LIBRARY=radio-sim_amd/csrc/libradiomedium_hip_stamps.so timeout -k 10 200 python tools/scan_stamps.py 20 2>&1 | tail -13
export TMPDIR=/tmp
rm -rf gpurun_out/s1_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s1_prof -o c5 --output-format csv -- python3 bench.py --workload c5 --steps 100 --warmup 20 --no-cpu-baseline --no-scale-probe --no-host-transfer > gpurun_out/s1_c5_prof.json 2> gpurun_out/s1_c5_prof.err; echo "prof rc=$?"
find gpurun_out/s1_prof -name "*kernel_stats.csv" -exec head -6 {} \;
