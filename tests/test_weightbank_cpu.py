"""Host logic of engine.WeightBank without a GPU (the cast kernels are replaced by recorders).

Regression guard for a stale-cache bug: the bank used to key on id(parameter); when a model is dropped and another
built, a new Parameter can reuse the address (and storage pointer and version) of a dead one, and the new model was then
served the previous model's bf16 weights."""
import gc
import importlib

import pytest
import torch

eng = importlib.import_module("x2-vlm_amd.engine")


@pytest.fixture
def bank(monkeypatch):
    calls = {"multi": 0, "single": 0, "vec": 0}
    monkeypatch.setattr(eng.K, "cast_transpose_multi", lambda desc: calls.__setitem__("multi", calls["multi"] + 1))
    monkeypatch.setattr(eng.K, "copy_f32_multi", lambda desc: calls.__setitem__("vec", calls["vec"] + 1))

    def single(src):
        calls["single"] += 1
        return src.to(torch.bfloat16), src.t().contiguous().to(torch.bfloat16)
    monkeypatch.setattr(eng.K, "cast_transpose_bf16", single)
    b = eng.WeightBank()
    b.calls = calls
    return b


def test_prepare_builds_once_per_version(bank):
    w = [torch.nn.Parameter(torch.randn(8, 4)) for _ in range(3)]
    groups = [(w[0], w[1]), (w[2],)]
    bank.prepare(groups)
    assert bank.calls["multi"] == 1 and len(bank._c) == 2
    plain, tr = bank.linear(w[0], w[1])                      # hit: no single-weight cast
    assert plain.shape == (16, 4) and tr.shape == (4, 16) and bank.calls["single"] == 0
    bank.prepare(groups)
    assert bank.calls["multi"] == 1                          # nothing stale
    with torch.no_grad():
        w[2].add_(1.0)                                       # optimizer step: version moves
    bank.prepare(groups)
    assert bank.calls["multi"] == 2
    odd = torch.nn.Parameter(torch.randn(6, 5))              # not 4-aligned: left to the one-at-a-time path
    bank.prepare([(odd,)])
    assert bank.calls["multi"] == 2
    bank.linear(odd)
    assert bank.calls["single"] == 1


def test_keys_are_object_serials_not_addresses(bank):
    seen = set()
    for _ in range(50):                                      # CPython reuses the freed object's address most of the time
        p = torch.nn.Parameter(torch.zeros(4, 4))
        tok = eng._tok(p)
        assert tok not in seen and eng._tok(p) == tok        # stable per object, never reused
        seen.add(tok)
        del p
    a = torch.nn.Parameter(torch.ones(4, 4))
    bank.linear(a)
    key_a = next(iter(bank._c))
    del a
    gc.collect()
    b = torch.nn.Parameter(torch.full((4, 4), 2.0))          # may well sit where `a` was, same version, same storage slot
    plain, _ = bank.linear(b)
    assert float(plain.float().mean()) == 2.0 and bank.calls["single"] == 2
    bank._purge()
    assert key_a not in bank._c and len(bank._c) == 1        # the dead parameter's copies are dropped


def test_vector_cache(bank):
    q, v = torch.nn.Parameter(torch.arange(4.0)), torch.nn.Parameter(torch.arange(4.0) + 10)
    got = bank.vector(q, 4, v)                               # lazy path: torch.cat with a zero segment (BEiT has no k bias)
    assert torch.equal(got, torch.tensor([0., 1, 2, 3, 0, 0, 0, 0, 10, 11, 12, 13]))
    assert bank.vector(q, 4, v) is got
    bank.prepare_vectors([(q, 4, v)])
    assert bank.calls["vec"] == 0                            # already current
    with torch.no_grad():
        q.mul_(2.0)
    bank.prepare_vectors([(q, 4, v)])
    assert bank.calls["vec"] == 1


def test_copies_expire_after_a_backward_even_without_version_bump(bank):
    """Optimizers writing through p.data leave _version alone (transformers 4.12.5 AdamW, apex FusedAdam)."""
    w = torch.nn.Parameter(torch.ones(8, 4))
    bank.prepare([(w,)])
    assert bank.calls["multi"] == 1
    v0 = w._version
    w.data.add_(1.0)
    assert w._version == v0                                  # the blind spot of a version-keyed cache
    bank.prepare([(w,)])
    assert bank.calls["multi"] == 1                          # (inference between optimizer steps keeps its copies)
    bank.note_backward()                                     # a stage's backward ran: an optimizer step may follow
    bank.prepare([(w,)])
    assert bank.calls["multi"] == 2
    bank.prepare([(w,)])
    assert bank.calls["multi"] == 2                          # rebuilt once, not on every call

    class Probe(torch.autograd.Function):                    # inside autograd's backward nothing expires
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            bank.note_backward()
            bank.linear(w)
            Probe.calls_in_backward = dict(bank.calls)
            return g
    n_single = bank.calls["single"]
    Probe.apply(torch.ones(1, requires_grad=True)).sum().backward()
    assert Probe.calls_in_backward["single"] == n_single     # served from the cache while the backward is running
    bank.linear(w)
    assert bank.calls["single"] == n_single + 1              # first use after it: rebuilt


def test_scaled_transposed_copies_are_their_own_entries(bank, monkeypatch):
    """linear(w, tscale=gamma) / prepare([(w, TScale(gamma))]): W^T carries gamma (the layer scale folded into the weight of the
    input-gradient GEMM); keyed apart from the plain pair, rebuilt when either the weight or gamma moved, descriptor rows carry
    the scale pointer in slot 7."""
    descs = []
    monkeypatch.setattr(eng.K, "cast_transpose_multi", lambda desc: (descs.append(list(desc)), bank.calls.__setitem__("multi", bank.calls["multi"] + 1)))
    w, g = torch.nn.Parameter(torch.randn(8, 4)), torch.nn.Parameter(torch.rand(8) + 0.5)
    bank.prepare([(w,), (w, eng.TScale(g))])
    assert bank.calls["multi"] == 1 and len(bank._c) == 2 and len(descs[0]) == 2
    plain_row, scaled_row = descs[0]
    assert len(plain_row) == 8 and plain_row[7] == 0 and scaled_row[7] == g.data_ptr() and scaled_row[0] == w.data_ptr()
    bank.linear(w, tscale=g)
    bank.linear(w)
    assert bank.calls["single"] == 0                          # both were built by prepare()
    with torch.no_grad():
        g.mul_(2.0)                                          # only gamma moved: the scaled entry is stale, the plain one is not
    bank.prepare([(w,), (w, eng.TScale(g))])
    assert bank.calls["multi"] == 2 and len(descs[1]) == 1 and descs[1][0][7] == g.data_ptr()
    # one-at-a-time path (no multi-tensor launch possible): same values
    odd, go = torch.nn.Parameter(torch.randn(6, 5)), torch.nn.Parameter(torch.rand(6) + 0.5)
    plain, tr = bank.linear(odd, tscale=go)
    assert torch.equal(plain, odd.detach().to(torch.bfloat16))
    assert torch.equal(tr, (odd.detach() * go.detach()[:, None]).t().contiguous().to(torch.bfloat16))


def test_vector_of_a_non_contiguous_or_half_precision_bias_is_packed_by_value(bank):
    """The multi-tensor vector build copies numel() floats from the raw address; an item that is not fp32 + contiguous must take the
    reshape + cat build instead (advisor finding, round 4): a strided slice or a bf16 bias packed by address would be silently wrong."""
    base = torch.nn.Parameter(torch.arange(8.0))
    strided = base[::2]                                       # 4 elements, stride 2
    got = bank.vector(strided, 2)
    assert torch.equal(got, torch.tensor([0., 2, 4, 6, 0, 0])) and bank.calls["vec"] == 0
    half = torch.nn.Parameter(torch.arange(4.0).to(torch.bfloat16))
    got = bank.vector(half, 1)
    assert got.dtype == torch.float32 and torch.equal(got, torch.tensor([0., 1, 2, 3, 0]))
    bank.prepare_vectors([(strided, 2), (half, 1)])            # nothing for the raw-address launch
    assert bank.calls["vec"] == 0
