"""`python bench.py --gpus N` must start its N ranks itself (the reference's
run_dist_train.sh:16 does it with torchrun).  CPU self-test of that launcher: 2 gloo ranks run
the --stub step (a CPU stand-in model through the real TrainStep exchange) and rank 0 prints the
one JSON line with n_gpus = 2.  The parent must not need a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "2"
    if env_extra:
        env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "2", "--warmup", "1",
                        "--batch", "8", "--points", "64", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0]), p.stderr


def test_bench_gpus2_launches_two_ranks_itself():
    line, err = _run(["--gpus", "2"])
    assert "launching" in err and "torch.distributed.run" in err
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16
    assert line["config"]["parallelism"].startswith("dp2")
    assert line["scaling"] == "weak" and line["value"] > 0 and line["steps"] == 2 and line["warmup"] == 1
    assert line["metric"].startswith("STUB")          # a self-test line can never pass for a measurement


def test_bench_single_rank_is_a_plain_process():
    line, err = _run([])
    assert "launching" not in err
    assert line["n_gpus"] == 1 and line["config"]["parallelism"] == "dp1"
