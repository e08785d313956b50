"""-m gpu: a short run of every randomized parity tool under tools/ (each compares the HIP path with the CPU oracle -- or a
batch with its jobs run one by one -- on seeded random inputs; the long runs are quoted in DESIGN.md section 5).  The seeds
differ from the ones used while developing, so every round-end run also covers cases nobody has looked at."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seed", [("fuzz_nn.py", 25, 101), ("fuzz_bound.py", 14, 102), ("fuzz_align.py", 14, 103),
                                             ("fuzz_batch.py", 6, 104), ("fuzz_map_insert.py", 12, 105),
                                             ("fuzz_preprocess.py", 25, 106), ("fuzz_odometry.py", 3, 107),
                                             ("fuzz_hook_replay.py", 20, 108), ("stress_cli_sequences.py", 1, 109)])
def test_randomized_parity_tool(tool, cases, seed):
    env = dict(os.environ)
    for k in ("MH_MATCH", "MH_NO_PREV_BOUND", "MH_NO_FUSE16", "MH_NO_STEP_CHAIN", "MH_NO_LOCKSTEP", "MH_NO_GRAPH"):
        env.pop(k, None)  # (the tools choose their own switches)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)], capture_output=True, text=True,
                       timeout=500, env=env, cwd=ROOT)
    tail = "\n".join(r.stdout.strip().splitlines()[-6:])
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, tail + "\n" + r.stderr[-600:]
