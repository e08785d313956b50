#!/bin/bash
# scratch: A/B runs on the GPU box (edit freely; the default re-checks the GPU suite, the smoke test and the bench)
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>&1 | tail -1 | cut -c1-300
