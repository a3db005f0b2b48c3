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


class _ArenaLinear(torch.autograd.Function):
    """y = x W^T + b whose backward behaves like an engine stage: parameter gradients are views of one flat arena,
    published (engine.Grads.publish) before autograd sees them."""

    @staticmethod
    def forward(ctx, x, w, b, key):
        eng = importlib.import_module("x2-vlm_amd.engine")
        eng._count_call(key)
        ctx.save_for_backward(x, w)
        ctx.key, ctx.params = key, (w, b)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        eng = importlib.import_module("x2-vlm_amd.engine")
        x, w = ctx.saved_tensors
        G = eng.Grads(dy.device, [("w", w.shape, False), ("b", (w.shape[0],), True)], key=ctx.key, params=list(ctx.params))
        G["w"].copy_(dy.t() @ x)
        G["b"].copy_(dy.sum(0))
        G.publish()
        dw, db = G.take(["w", "b"])
        return dy @ w, dw, db, None


def _bucket_case(rank, world):
    """Two arena layers + a plain head.  Iteration 1: one backward_step.  Iteration 2 (no zero_grad in between, as
    run_mixed_iter's video + image backward, Pretrain.py:197, 247): gradients must accumulate and stay averaged."""
    acc = importlib.import_module("x2-vlm_amd.accelerator")
    torch.manual_seed(0)
    l0, l1, head = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 2)
    model = torch.nn.Sequential(l0, l1, head)
    a = acc.RocmDDPAccelerator(dict(RNG_SEED=1), None)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(rank)                                   # ranks start different; broadcast must fix it
    a.world_size = world
    a.broadcast(model)
    after_bcast = [p.detach().clone() for p in model.parameters()]
    a.buckets = gb = acc.GradientBuckets(model, world)

    def run(x, shared_l1=False):
        h = _ArenaLinear.apply(x, l0.weight, l0.bias, ("t", 0))
        h = _ArenaLinear.apply(torch.tanh(h), l1.weight, l1.bias, ("t", 1))
        if shared_l1:                                      # a layer used twice in one forward: never reduced early
            h = _ArenaLinear.apply(torch.tanh(h), l1.weight, l1.bias, ("t", 1))
        return head(torch.tanh(h)).square().sum()

    g = torch.Generator().manual_seed(50 + rank)
    xa, xb, xc = (torch.randn(5, 4, generator=g) for _ in range(3))
    out = {"bcast": after_bcast, "x": (xa, xb, xc)}
    a.backward_step(run(xa), None)
    out["g1"] = [p.grad.clone() for p in model.parameters()]
    out["msgs1"] = gb.messages                             # 2 early arenas + 1 leftover bucket
    a.backward_step(run(xb), None)                         # second backward_step of the iteration: accumulate
    out["g2"] = [p.grad.clone() for p in model.parameters()]
    out["msgs2"] = gb.messages - out["msgs1"]              # everything through the late path: 1 message
    for p in model.parameters():
        p.grad = None
    a.backward_step(run(xc, shared_l1=True), None)
    out["g3"] = [p.grad.clone() for p in model.parameters()]
    out["msgs3"] = gb.messages - out["msgs1"] - out["msgs2"]   # arena 0 early, shared layer + head late
    gb.close()
    return out


def _hooked_case(rank, world):
    """A gradient hook makes autograd store hook(view) instead of adopting the arena view: the early-reduced arena no
    longer is .grad, and finish() must notice and send that parameter with the leftovers."""
    acc = importlib.import_module("x2-vlm_amd.accelerator")
    torch.manual_seed(0)
    l0, head = torch.nn.Linear(4, 4), torch.nn.Linear(4, 2)
    model = torch.nn.Sequential(l0, head)
    a = acc.RocmDDPAccelerator(dict(RNG_SEED=1), None)
    a.world_size = world
    a.buckets = gb = acc.GradientBuckets(model, world)
    l0.weight.register_hook(lambda g: g * 1.0)                 # returns a NEW tensor
    g = torch.Generator().manual_seed(70 + rank)
    x = torch.randn(5, 4, generator=g)
    h = _ArenaLinear.apply(x, l0.weight, l0.bias, ("t", 0))
    a.backward_step(head(torch.tanh(h)).square().sum(), None)
    out = dict(g=[p.grad.clone() for p in model.parameters()], demoted=gb.demoted, x=x,
               p=[p.detach().clone() for p in model.parameters()])
    gb.close()
    return out


def test_early_reduced_arena_that_autograd_did_not_adopt_is_demoted():
    out = _run("_hooked_case")
    P = out[0]["p"]

    def local(x):
        w0, b0, wh, bh = [t.clone().requires_grad_(True) for t in P]
        (torch.tanh(x @ w0.t() + b0) @ wh.t() + bh).square().sum().backward()
        return [t.grad for t in (w0, b0, wh, bh)]
    want = [sum(gs) / len(out) for gs in zip(*[local(o["x"]) for o in out])]
    for o in out:
        assert o["demoted"] == 1
        for got, w in zip(o["g"], want):
            assert torch.allclose(got, w, atol=1e-5)


def _local_grads(params0, x, shared_l1=False):
    """Plain-autograd gradients of the same little network on one rank's input."""
    w0, b0, w1, b1, wh, bh = [t.clone().requires_grad_(True) for t in params0]
    h = torch.tanh(x @ w0.t() + b0) @ w1.t() + b1
    if shared_l1:
        h = torch.tanh(h) @ w1.t() + b1
    (torch.tanh(h) @ wh.t() + bh).square().sum().backward()
    return [t.grad for t in (w0, b0, w1, b1, wh, bh)]


def test_gradient_buckets_average_once_and_broadcast():
    out = _run("_bucket_case")
    for o in out:
        for a, b in zip(o["bcast"], out[0]["bcast"]):
            assert torch.equal(a, b)                                   # identical parameters after the flat broadcast
    P = out[0]["bcast"]
    mean = lambda k, **kw: [sum(gs) / len(out) for gs in zip(*[_local_grads(P, o["x"][k], **kw) for o in out])]
    ga, gb_, gc = mean(0), mean(1), mean(2, shared_l1=True)
    for o in out:
        assert (o["msgs1"], o["msgs2"], o["msgs3"]) == (3, 1, 2)
        for got, want in zip(o["g1"], ga):
            assert torch.allclose(got, want, atol=1e-5)
        for got, wa, wb in zip(o["g2"], ga, gb_):                      # avg(g_a) + avg(g_b), averaged exactly once each
            assert torch.allclose(got, wa + wb, atol=1e-5)
        for got, want in zip(o["g3"], gc):
            assert torch.allclose(got, want, atol=1e-5)


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


def test_file_rendezvous_ignores_leftovers_of_a_crashed_launch(tmp_path):
    """comm.X2Comm._from_file (the path form of from_store): a launch that crashed before rank 0's clean-up leaves the id file
    (and announcements) of the SAME name behind; the next launch must not pick them up - every rank ends with the id rank 0
    published in THIS launch, files are gone afterwards, and a rank 0 that never shows up is a time-out error, not a hang."""
    import threading
    import time
    comm = importlib.import_module("x2-vlm_amd.comm")
    lib_mod = importlib.import_module("x2-vlm_amd._lib")

    class Fake(comm.X2Comm):
        made = []

        def __init__(self, ident, rank, world):          # no device: only the rendezvous is under test
            self.ident, self.rank, self.world, self.h = ident, rank, world, None

        @staticmethod
        def unique_id():
            b = os.urandom(128)
            Fake.made.append(b)
            return b

    path = str(tmp_path / "x2id.29500.x2_comm_id.1")
    stale = b"\x01" * 128
    with open(path, "wb") as f:
        f.write(stale)
    open(path + ".hello1", "w").close()
    open(path + ".ack2", "w").close()
    open(path + ".hello0", "w").close()                  # rank 0's own leftovers are removed as well
    open(path + ".ack0", "w").close()
    # timestamps must not matter (coarse mtime granularity / clock skew on shared file systems): the stale id "from the future"
    future = time.time() + 3600
    os.utime(path, (future, future))
    world, got, errs = 3, {}, []

    def run(rank, delay):
        try:
            time.sleep(delay)
            got[rank] = Fake._from_file(path, rank, world, timeout=20.0).ident
        except Exception as e:      # noqa: BLE001
            errs.append((rank, e))
    th = [threading.Thread(target=run, args=(r, d)) for r, d in ((1, 0.0), (0, 0.3), (2, 0.6))]     # a reader polls before rank 0 starts
    for t in th:
        t.start()
    for t in th:
        t.join(40)
    assert not errs, errs
    assert len(Fake.made) == 1 and all(got[r] == Fake.made[0] for r in range(world)) and got[0] != stale
    assert not [f for f in os.listdir(tmp_path) if f.startswith("x2id")], os.listdir(tmp_path)
    with pytest.raises(lib_mod.X2HipError):
        Fake._from_file(str(tmp_path / "never"), 1, 2, timeout=0.3)
