"""Short runs of the two randomised parity tools (tools/fuzz_parity.py: renders against the oracle;
tools/fuzz_trace.py: nearest-hit queries on random geometry, both tree builders, against brute
force). Each stops with a non-zero status at the first disagreement."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seed", [("fuzz_parity.py", 25, 11), ("fuzz_trace.py", 20, 12)])
def test_fuzz_tool(tool, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)], cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"all {cases} cases" in r.stdout
