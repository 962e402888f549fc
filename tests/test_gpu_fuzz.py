"""Short run of the randomised differential test (tests/fuzz_gpu.py): random workloads, random
scoring / band / mode, libgba through the C ABI against the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fuzz_short():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu.py"), "--seconds", "10", "--seed", "12345"],
                       capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fails 0" in r.stdout
