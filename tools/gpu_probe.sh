#!/bin/bash
# scratch: A/B runs on the GPU box
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-140; done
