set -o pipefail
cd $GRAFT_REPO_ROOT
for k in 1 20; do
RM_LIBRARY=radio-sim_amd/csrc/libradiomedium_hip_stamps.so timeout -k 10 200 python tools/scan_stamps.py $k 2>&1 | tail -12
done
