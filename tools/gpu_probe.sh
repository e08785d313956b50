#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests/test_odometry.py -m gpu -x -q 2>&1 | tail -8
