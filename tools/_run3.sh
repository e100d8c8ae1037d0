set -o pipefail
cd $GRAFT_REPO_ROOT
RM_LIBRARY=radio-sim_amd/csrc/libradiomedium_hip_stamps.so timeout -k 10 200 python tools/scan_stamps.py 20 2>&1 | tail -13
timeout -k 10 200 python bench.py --workload c5 --steps 200 --warmup 20 --no-cpu-baseline --no-scale-probe --no-host-transfer > gpurun_out/s3_c5.json 2> gpurun_out/s3_c5.err; echo "c5 rc=$?"
python - <<P
import json
d=json.loads(open("gpurun_out/s3_c5.json").read().strip().splitlines()[-1])
print("c5 us/tick %.2f value %.3e" % (d["ms_per_step"]*1e3, d["value"]))
P
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/s3_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -5 gpurun_out/s3_gpu.log
