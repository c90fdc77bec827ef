"""Two ranks push the PRODUCT LineRefineNet (HIP kernels, fused Adam, chunked decoder) through
TrainStep (train_dist.py:102-147,173-189): flat-gradient all-reduce = mean of the per-rank
gradients, identical weights after the step, rank-0 BatchNorm buffers broadcast before the
forward.  The GPU box has one GPU, so both ranks sit on cuda:0 and the group runs over gloo
(RCCL refuses two ranks on one device); the 8-GPU RCCL run is the driver's."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_product_model_through_train_step(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "res.json")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_product_worker.py"), out],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    res = json.load(open(out))
    assert res["n_params"] == 9695954
    # chunked vs monolithic decoder: different GEMM tilings, i.e. fp32 re-association only - 1e-4 on every
    # GEMM mode, PRH_GEMM=fp32 included.  What read 4.2e-4 there in round 2 is ONE ReLU unit whose
    # pre-activation sits inside fp32 noise of zero and takes different sides in the two evaluations
    # (profiles/r03_fp32_chunk_root_cause.txt); the worker counts the ReLU masks that differ, and only a
    # counted flip widens the gate (~5e-4 per unit).
    assert res["grad_rel_l2"] < (1e-4 if res["relu_flips"] == 0 else 2e-3 * res["relu_flips"]), res
    assert res["weights_equal"], res
    # per-rank statistics, no SyncBN: the ranks' running means differ by their local batches
    # only, not by the +5 rank 1 started with (overwritten by the rank-0 broadcast)
    assert res["bn_gap"] < 1.0, res
