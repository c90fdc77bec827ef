"""The opt-in kernel variants the library carries for A/B runs are compiled into the shipped .so, so they
are held to the same tests as the defaults: each switch is read when the library loads, hence a child
interpreter per switch (one at a time, on top of this process's GPU context: two processes on the card).
  PRH_H2_PP=1     split-fp16 NT GEMMs on the phase-split ("ping-pong") k-loop (csrc/prh_gemm_h2.hpp)
  PRH_B16_DMA=0   bf16-mode plain-operand NT GEMMs on the register-staged core instead of gemm_nt_b16d_kernel
  PRH_STAGGER=1   late start of half the first generation of workgroups of a large bf16 NT launch"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env,files", [
    ({"PRH_H2_PP": "1"}, ["tests/test_gemm_gpu.py", "tests/test_encoder_gpu.py", "tests/test_oracle_fp64_gpu.py"]),
    ({"PRH_B16_DMA": "0"}, ["tests/test_bf16_gpu.py"]),
    ({"PRH_STAGGER": "1"}, ["tests/test_bf16_gpu.py"]),
], ids=["h2_phase_split", "b16_register_staged", "b16_stagger"])
def test_switch_passes_the_tests_of_the_default(env, files):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-1500:])
    assert " passed" in r.stdout
