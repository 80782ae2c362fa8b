"""CPU: what bench.py does before it touches torch or HIP - the refusal of result-breaking diagnostics and the start of
its own ranks for `--gpus N` without a launcher. The ranks themselves need a GPU (tests/test_gpu_dist.py
runs the same command lines on one); here they must fail loudly and the parent must pass that on instead of hanging."""
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def test_bench_refuses_result_breaking_diagnostics():
    for var in ("PLAAC_DEBUG_SKIP", "PLAAC_VIT_STOP", "PLAAC_DEBUG_COUNTER"):
        r = subprocess.run([sys.executable, BENCH, "--steps", "1"], capture_output=True, text=True, env=_clean_env(**{var: "1"}), timeout=120)
        assert r.returncode != 0 and var in r.stderr and "--allow-diagnostics" in r.stderr


def test_bench_starts_its_own_ranks_and_reports_their_failure():
    """no GPU here: both ranks exit with the 'needs a GPU' message; the parent (which never imported torch) returns non-zero"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: tests/test_gpu_dist.py runs the same command for real")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--config", "2", "--steps", "1", "--no-e2e", "--backend", "gloo",
                        "--one-device"], capture_output=True, text=True, env=_clean_env(), timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") == 2  # both ranks were started and got as far as the device check


def test_the_launching_parent_imports_neither_torch_nor_the_library():
    """the parent process of a self-launched job must not initialise the GPU (a process that has may not be replaced or
    forked on this pool): everything before launch_ranks() is argument parsing"""
    src = open(BENCH).read()
    head = src[:src.index("sys.exit(launch_ranks(")]
    body = head[head.index("def main():"):]
    assert "import torch" not in body and "from plaac_amd" not in body and "native.load" not in body
    top = src[src.index('"""', 10):src.index("def usable_cores")]  # module level, behind the docstring
    assert "import torch" not in top and "import plaac_amd" not in top and "from plaac_amd" not in top
