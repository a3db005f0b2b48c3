"""world_size-2 gloo tests (CPU) of the data-parallel pieces: the ITC all-gather with slice-only
backward (xvlm.py:140-160), the flat broadcast, and the gradient-bucket bookkeeping of
accelerator.GradientBuckets (arenas reduced early only for single-use layers, leftovers at finish)."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = globals()[fn_name](rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn_name, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _allgather_case(rank, world):
    xvlm = importlib.import_module("x2-vlm_amd.xvlm")
    torch.manual_seed(rank)
    x = torch.randn(3, 4, requires_grad=True)
    y = xvlm.allgather(x)
    w = torch.arange(world * 3 * 4, dtype=torch.float32).view(world * 3, 4)
    (y * w).sum().backward()
    return y.detach().clone(), x.grad.clone(), x.detach().clone()


def test_allgather_forward_concat_backward_local_slice():
    out = _run("_allgather_case")
    full = torch.cat([o[2] for o in out])
    w = torch.arange(2 * 3 * 4, dtype=torch.float32).view(6, 4)
    for r, (y, g, _x) in enumerate(out):
        assert torch.equal(y, full)                       # every rank sees the rank-ordered concatenation
        assert torch.equal(g, w[3 * r: 3 * r + 3])        # gradient = local rows only, no cross-rank reduction


def _bucket_case(rank, world):
    acc = importlib.import_module("x2-vlm_amd.accelerator")
    eng = importlib.import_module("x2-vlm_amd.engine")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2))
    a = acc.RocmDDPAccelerator(dict(RNG_SEED=1), None)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(rank)                                   # ranks start different; broadcast must fix it
    a.world_size = world
    a.broadcast(model)
    after_bcast = [p.detach().clone() for p in model.parameters()]
    gb = acc.GradientBuckets(model, world)
    # layer 0: gradients live in one arena, used once -> reduced at publish time
    p0 = list(model[0].parameters())
    arena = eng.Grads("cpu", [("w", p0[0].shape, False), ("b", p0[1].shape, True)], key=("t", 0))
    eng._count_call(("t", 0))
    arena["w"].fill_(rank + 1.0); arena["b"].fill_(10.0 * (rank + 1))
    p0[0].grad, p0[1].grad = arena["w"], arena["b"]
    arena.publish()
    early = arena["w"].clone()
    # layer 1: used twice in the step -> must NOT be reduced early, only as a leftover
    p1 = list(model[1].parameters())
    arena2 = eng.Grads("cpu", [("w", p1[0].shape, False), ("b", p1[1].shape, True)], key=("t", 1))
    eng._count_call(("t", 1)); eng._count_call(("t", 1))
    arena2["w"].fill_(rank + 1.0); arena2["b"].fill_(rank + 1.0)
    p1[0].grad, p1[1].grad = arena2["w"].clone(), arena2["b"].clone()
    arena2.publish()
    not_early = arena2["w"].clone()
    gb.finish()
    gb.close()
    return after_bcast, early, not_early, [p.grad.clone() for p in model.parameters()]


def test_gradient_buckets_average_once_and_broadcast():
    out = _run("_bucket_case")
    for r, (bc, early, not_early, grads) in enumerate(out):
        for a, b in zip(bc, out[0][0]):
            assert torch.equal(a, b)                                   # identical parameters after the flat broadcast
        assert torch.allclose(early, torch.full_like(early, 1.5))      # (1+2)/2 already at publish time
        assert torch.allclose(not_early, torch.full_like(not_early, r + 1.0))
        assert torch.allclose(grads[0], torch.full_like(grads[0], 1.5)) and torch.allclose(grads[1], torch.full_like(grads[1], 15.0))
        assert torch.allclose(grads[2], torch.full_like(grads[2], 1.5)) and torch.allclose(grads[3], torch.full_like(grads[3], 1.5))


def _itc_grad_case(rank, world):
    """ITC over the all-gathered features: rank-local gradient == the rank's rows of d(global loss)/d(features)
    (SURVEY.md 8e).  Pure-CPU maths through the oracle's contrastive_loss with the package's allgather."""
    from oracle import x2vlm_oracle as O
    xvlm = importlib.import_module("x2-vlm_amd.xvlm")
    g = torch.Generator().manual_seed(100 + rank)
    fi = torch.nn.functional.normalize(torch.randn(4, 8, generator=g), dim=-1).requires_grad_(True)
    ft = torch.nn.functional.normalize(torch.randn(4, 8, generator=g), dim=-1).requires_grad_(True)
    sd = {"temp": torch.tensor(0.07)}
    loss, _ = O.contrastive_loss(sd, fi, ft, xvlm.allgather(fi), xvlm.allgather(ft))
    loss.backward()
    return fi.detach().clone(), ft.detach().clone(), fi.grad.clone(), ft.grad.clone(), float(loss)


def test_itc_loss_gradient_is_local_slice_of_global():
    from oracle import x2vlm_oracle as O
    out = _run("_itc_grad_case")
    FI = torch.cat([o[0] for o in out]).requires_grad_(True)
    FT = torch.cat([o[1] for o in out]).requires_grad_(True)
    loss, _ = O.contrastive_loss({"temp": torch.tensor(0.07)}, FI, FT)
    loss.backward()
    for r, o in enumerate(out):
        assert abs(o[4] - float(loss)) < 1e-6
        assert torch.allclose(o[2], FI.grad[4 * r: 4 * r + 4], atol=1e-6)
        assert torch.allclose(o[3], FT.grad[4 * r: 4 * r + 4], atol=1e-6)
