set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 1500 python tools/accuracy_ablation.py --scans 1000 --out gpurun_out/r04/accuracy2.json 2>&1 | tail -22
timeout 1200 python tools/match_floor.py --out gpurun_out/r04/match_floor.json 2>&1 | tail -16
