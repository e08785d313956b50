#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py 2>&1 | tail -1 | cut -c1-330
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
