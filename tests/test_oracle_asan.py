"""The CPU oracles under AddressSanitizer and UndefinedBehaviorSanitizer: `make -C oracle asan` builds the same
sources with -fsanitize=address,undefined, and the oracle's own CPU tests (golden fixtures, hand cases, properties,
the CIGAR restatement) run against that build in a child process.  A bad read or write, or undefined arithmetic,
in any oracle aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_pass_on_the_sanitizer_build():
    if os.environ.get("GBA_ORACLE_SO"):
        pytest.skip("already inside the sanitizer run")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan runtime beside gcc")
    env = dict(os.environ, GBA_ORACLE_SO=so, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    pre = env.get("LD_PRELOAD", "")
    env["LD_PRELOAD"] = asan + (":" + pre if pre else "")
    tests = ["tests/test_oracle.py", "tests/test_kband_oracle.py", "tests/test_sdp_oracle.py", "tests/test_emit.py"]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert "passed" in p.stdout and "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr
