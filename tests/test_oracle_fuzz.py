"""CPU: a short seeded run of tools/fuzz_oracles.py -- the C oracle against the independent numpy restatement on random small
alignments (the long runs are quoted in DESIGN.md section 5)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_against_numpy_oracle_on_random_alignments():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_oracles.py"), "5", "2025"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]
