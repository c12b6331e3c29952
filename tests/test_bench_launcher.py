"""`python bench.py --gpus N` must start N ranks by itself (the reference's mp.spawn, time_tuning.py:714-717), before any
GPU call.  Dry mode (TT_BENCH_DRY=1, gloo): no GPU, no kernels - the ranks only join the process group and report."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra_env=None, timeout=240):
    env = dict(os.environ, TT_BENCH_DRY="1", TT_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                          env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(300)
def test_gpus_n_spawns_n_distinct_ranks():
    r = _run(3)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["ranks"] == [0, 1, 2]
    assert len(set(out["pids"])) == 3 and os.getpid() not in out["pids"]
    assert out["rccl"] == {"world_size": 3, "backend": "gloo"}
    assert out["config"]["global_batch"] == 96 and out["config"]["parallelism"] == "dp3"


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_run_without_hanging():
    r = _run(2, {"TT_BENCH_DRY_FAIL_RANK": "1"}, timeout=200)
    assert r.returncode != 0
    assert "rank process 1 exited with 3" in r.stderr


def test_the_launcher_parent_never_touches_the_gpu():
    """Static check: what the parent executes (main() up to the launch_ranks call, launch_ranks, _free_port) contains no
    torch.cuda attribute access and no import of the HIP front end."""
    import ast

    src = open(os.path.join(REPO, "bench.py")).read()
    tree = ast.parse(src)
    fn = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}

    def gpu_touching(nodes):
        for top in nodes:
            for n in ast.walk(top):
                if isinstance(n, ast.Attribute) and n.attr == "cuda":
                    return True
                if isinstance(n, (ast.Import, ast.ImportFrom)) and "timetuning_amd" in ast.dump(n):
                    return True
        return False

    assert not gpu_touching([fn["launch_ranks"], fn["_free_port"]])
    head = []
    for stmt in fn["main"].body:
        head.append(stmt)
        if "launch_ranks" in ast.dump(stmt):
            break
    else:
        raise AssertionError("main() no longer calls launch_ranks")
    assert not gpu_touching(head)
