set -x
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "loop_controls or batches_give_the_bits" 2>&1 | tail -4
timeout 600 python tools/fuzz_batch.py 60 101 2>&1 | tail -4
timeout 600 python tools/fuzz_align.py 150 102 2>&1 | tail -4
timeout 600 python tools/fuzz_odometry.py 12 103 2>&1 | tail -4
timeout 600 python tools/stress_multi_seq.py 2>&1 | tail -4
timeout 600 python tools/fuzz_bound.py 60 104 2>&1 | tail -3
timeout 600 python tools/stress_threads.py 2>&1 | tail -3
