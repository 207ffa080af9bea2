"""`python bench.py --gpus N` (N > 1) must start its own rank processes: that plain form is what the driver's scaling run and the GPU
suite's two-rank self-test (tests/conftest.py) call.  No GPU here, so the ranks get as far as bench.py's "needs a GPU" assertion --
which is behind the launch, the rendezvous environment and the argument hand-over this test is about.  (Reference: the path's multi-GPU
mechanism is nn.DataParallel, trainers/classification/coop.py:268-272; one process per GPU replaces it.)"""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() > 0, reason="CPU-box test: with a GPU the ranks would run the bench itself (tests/test_gpu_multirank.py)")
def test_plain_invocation_launches_its_own_ranks():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-roofline",
                        "--no-cpu-baseline"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    out = p.stdout
    assert "launch N>1 with" not in out                      # the round-3 refusal is gone
    # What cannot race: torch.distributed.run SIGTERMs the surviving rank as soon as the first one has failed, so "both ranks printed"
    # depends on how far the slower one got.  At least one rank of a TWO-rank job must have started (RANK / WORLD_SIZE handed over) and
    # have reached the GPU assertion, and the launcher's failure must be the parent's return code.
    started = set(re.findall(r"\[bench rank (\d) of 2\] started", out))
    refused = set(re.findall(r"needs a GPU \(no CPU fallback\) \[rank (\d) of 2\]", out))
    assert started and started <= {"0", "1"} and refused and refused <= started, out[-3000:]
    assert p.returncode != 0


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in p.stdout
