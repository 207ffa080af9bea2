"""The N > 1 code path on real hardware (reference: nn.DataParallel wrapping, trainers/classification/
coop.py:266-272).  The GPU box has ONE MI355X, so the two ranks share cuda:0 and the embedding exchange runs over
torch.distributed/gloo (RCCL refuses two ranks on one device); everything else -- sharded batch, fp16 embeddings, one
all-gather per step, fused tail on the gathered batch, ECE bins -- is the code an 8-GPU run executes.  The RCCL entry
points themselves are exercised on a one-rank communicator."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import MULTIRANK

pytestmark = pytest.mark.gpu


def _finish(key, timeout=900):
    p = MULTIRANK.get(key)
    if p is None:
        pytest.skip("multi-rank children were not started (session was not run with -m gpu on a GPU box)")
    rc = p.wait(timeout=timeout)
    log = open(os.path.join(MULTIRANK["tmp"], f"{key}.out")).read()
    assert rc == 0, f"{key}-process bench failed (rc {rc}):\n{log[-3000:]}"
    line = [ln for ln in log.splitlines() if ln.startswith("{")][-1]
    return json.loads(line), np.load(os.path.join(MULTIRANK["tmp"], f"{key}.npz"))


def test_two_ranks_equal_one_process_bitwise():
    """bench.py --gpus 2 (two processes, batch sharded 24 + 24, fp16 embeddings gathered once per step) against ONE process
    that runs both shards itself: rank 0's logits, confidences, predictions and ECE bins are bit-identical -- gather, not
    reduce, so the result cannot depend on the rank count."""
    j2, d2 = _finish("two")
    j1, d1 = _finish("one")
    assert j2["n_gpus"] == 2 and j2["exchange"]["world_size"] == 2 and j2["exchange"]["backend"] == "torch"
    assert j2["config"]["global_batch"] == 48 and j1["config"]["batch_per_gpu"] == 24
    assert d2["logits"].shape == (48, 200) and np.array_equal(d2["logits"], d1["logits"])
    assert np.array_equal(d2["pred"], d1["pred"]) and np.array_equal(d2["conf"], d1["conf"])
    b2, b1 = d2["bins"].reshape(3, -1), d1["bins"].reshape(3, -1)
    assert np.array_equal(b2[0], b1[0]) and np.array_equal(b2[2], b1[2])          # counts and hits: exact
    np.testing.assert_allclose(b2[1], b1[1], rtol=1e-13)                           # sums of confidences: atomics order
    assert j2["ece_percent"] == pytest.approx(j1["ece_percent"], abs=1e-9)
    assert b2[0].sum() == 2 * 48                                                   # 2 timed steps x the gathered batch


def test_rccl_communicator_single_rank():
    """clipmi_comm_* / clipmi_allgather (RCCL ncclAllGather on the caller's stream) on a one-rank communicator: the
    library opens librccl, builds the communicator from a unique id, reports what RCCL itself says about it, and the
    gather of one shard returns the shard."""
    from clip_calibration_amd.parallel import EmbeddingExchange
    ex = EmbeddingExchange(torch.device("cuda", torch.cuda.current_device()), backend="rccl")
    try:
        assert ex.backend == "rccl" and ex.rccl_ranks == 1
        x = torch.randn(256, 512, device="cuda").half()
        y = ex.all_gather(x)
        torch.cuda.synchronize()
        assert y.data_ptr() != x.data_ptr() and torch.equal(y, x)
    finally:
        ex.close()
