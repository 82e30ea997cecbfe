"""N > 1 path on CPU: two gloo ranks each take half of a batch, compute their local gradients (with the CPU
oracle standing in for the per-GPU step), run the product's all-reduce helper, and must end with the
global-batch gradient on both ranks (SURVEY.md §8e parity check)."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TINY = dict(obs_size=16, vocab_size=128, d_model=64, slot_size=64, mlp_hidden=64, num_slots=3, num_iterations=2, num_dec_blocks=1)


def _flat_grad(tr):
    return torch.cat([tr.P[n].grad.reshape(-1) for n, _, _, t in tr.spec if t])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import slate_oracle as O
    from ocrl_amd.dist_utils import allreduce_grads_
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.default_cfg(**TINY)
    B = 4
    obs = torch.rand(B, 3, 16, 16, generator=torch.Generator().manual_seed(1))
    noise = O.make_noise(cfg, B, 2)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    tr = O.OracleTrainer(cfg, O.formula_params(cfg))
    tr.loss_and_grads(obs[sl], {k: v[sl] for k, v in noise.items()}, 0, None)
    flat = _flat_grad(tr).clone()
    scale = allreduce_grads_(flat)
    flat *= scale
    total, _ = O.grad_clip_inf([flat], cfg.clip)
    q.put((rank, flat.numpy().copy(), float(total)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_global_batch_gradient():
    from oracle import slate_oracle as O
    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, flat, total = q.get(timeout=300)
        got[r] = (torch.from_numpy(flat), total)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg = O.default_cfg(**TINY)
    obs = torch.rand(4, 3, 16, 16, generator=torch.Generator().manual_seed(1))
    tr = O.OracleTrainer(cfg, O.formula_params(cfg))
    tr.loss_and_grads(obs, O.make_noise(cfg, 4, 2), 0, None)
    ref = _flat_grad(tr)
    assert torch.equal(got[0][0], got[1][0]), "ranks disagree after the all-reduce"
    err = (got[0][0].double() - ref.double()).abs().max() / ref.double().abs().max()
    assert err < 1e-5, err
    assert got[0][1] == got[1][1]


def test_allreduce_helper_is_identity_without_process_group():
    from ocrl_amd.dist_utils import allreduce_grads_
    g = torch.arange(8, dtype=torch.float32)
    assert allreduce_grads_(g) == 1.0
    assert torch.equal(g, torch.arange(8, dtype=torch.float32))
