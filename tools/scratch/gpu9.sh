set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_host_layer.py -m gpu -q -k "skipping_plane_paired or ndt_pipeline_align_matches_oracle" 2>&1 | tail -5
MH_CHAIN_R=1 timeout 300 python tools/phase_probe.py 2>&1 | tail -12
timeout 300 python tools/phase_probe.py 2>&1 | tail -12
