"""bench.py's command line: how --gpus N behaves with and without a launcher around it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GBA_BENCH_BACKEND")}
    env.update(kw)
    return env


def test_gpus_flag_without_enough_gpus_fails_loudly():
    """Started by hand, --gpus N must run N ranks or say why not; it never quietly runs one."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this host has the GPUs")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 2" in p.stderr and "visible" in p.stderr
    p = subprocess.run([sys.executable, BENCH, "--gpus", "0"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0


def test_gpus_flag_must_agree_with_the_launcher():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "must agree" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_plain_gpus_2_starts_two_ranks(scaling):
    """`python bench.py --gpus 2` with no launcher: two ranks (here sharing the one GPU, gloo for the timing barrier), each
    with its reads of one common set, one JSON line with n_gpus 2 and every pair of the set aligned."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "1", "--pairs", "8192", "--scaling", scaling],
                       env=_env(GBA_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    total = 8192 * 2 if scaling == "weak" else 8192
    assert line["n_gpus"] == 2 and line["scaling"] == scaling
    assert line["config"]["pairs_total"] == total and line["config"]["pairs_ok"] == total
    assert line["config"]["pairs_per_gpu"] == total // 2 and line["value"] > 0
