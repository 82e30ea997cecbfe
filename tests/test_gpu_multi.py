"""Two processes, one gradient all-reduce (SURVEY.md §8e parity): post-all-reduce gradients equal the 1-GPU gradients of the whole
batch to 5e-6, post-step weights to 1e-5.  Three transports:
  * "torch" / "cabi": two real GPUs over RCCL (torch.distributed "nccl", or the C ABI's ocrl_comm_*) — skipped on boxes with fewer than
    two GPUs; the driver's multi-GPU node runs them;
  * "gloo-one-gpu": taken when exactly one GPU is visible — both ranks run the real SLATE.update() (HIP kernels, per-rank slices,
    allreduce_grads_, the 1/world mean folded into clip_adam_kernel) on cuda:0 and reduce over gloo.  RCCL itself is not exercised by
    this variant (it refuses two ranks per device); everything around it is."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("transport", ["torch", "cabi", "gloo-one-gpu"])
def test_two_rank_update_equals_single_gpu_global_batch(tmp_path, transport, rank_launcher):
    ngpu = torch.cuda.device_count()          # counting devices does not initialise the GPU in this process: the ranks are spawned first
    one_gpu = transport == "gloo-one-gpu"
    if one_gpu and ngpu != 1:
        pytest.skip("the one-GPU rehearsal of the data-parallel step runs only where a single GPU is visible")
    if not one_gpu and ngpu < 2:
        pytest.skip("needs two GPUs")
    # fresh rank processes, started by the helper that conftest.py launched before this process touched the GPU (tests/rank_launcher.py)
    rcs = rank_launcher.run([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path)], world=2,
                            env=dict(OCRL_COMM="cabi" if transport == "cabi" else "", OCRL_DP_ONE_GPU="1" if one_gpu else ""), timeout=600)
    assert rcs == [0, 0], rcs
    from tests import dp_worker as W
    cfg, P, obs, noise = W.batch()
    ref = W.run_update(cfg, P, obs, noise, torch.device("cuda", 0))
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=True) for r in range(2))
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["p"], r1["p"]), "ranks disagree after the all-reduce"
    eg = ((r0["g"].double() - ref["g"].double()).abs().max() / ref["g"].double().abs().max()).item()
    ep = ((r0["p"].double() - ref["p"].double()).abs().max() / ref["p"].double().abs().max()).item()
    en = abs(r0["norm"] - ref["norm"]) / ref["norm"]
    from tests.gpu_util import log
    log(f"[dp2 {transport}] gradients {eg:.2e} weights {ep:.2e} norm {en:.2e}")
    assert eg < 5e-6 and ep < 1e-5 and en < 1e-5      # ~1e-6 (fp32 summation order: two B/2 partial sums vs one B sum), weights 1e-5
