set -x
df -h /tmp /dev/shm | head; nproc; free -g | head -2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04/gputests.log; cat gpurun_out/r04/gputests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench1.json 2> gpurun_out/r04/bench1.err; tail -5 gpurun_out/r04/bench1.err; head -c 1500 gpurun_out/r04/bench1.json
timeout 1200 python tools/accuracy_ablation.py --scans 1000 --out gpurun_out/r04/accuracy.json 2>&1 | tail -30
