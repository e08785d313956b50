#!/bin/bash
cd /root/repo
FIT_ITERS=5,10,5,1,5 python tools/align_fit.py 2>&1 | head -12
