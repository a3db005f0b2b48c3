"""The algebra behind the layer-scale backward (csrc/rowwise.hip, x2_layerscale_finish), checked on the CPU in double precision against
autograd: x_out = x_in + r[m] * gamma * (A . W^T + b) (models/beit2.py:206-207 with DropPath).  With dX' = r * dX:
    dA = dX' . (diag(gamma) W),   G = dX'^T . A,   dW = diag(gamma) G,   db = gamma * colsum(dX'),   dgamma = rowdot(G, W) + b * colsum(dX')
No GPU: this pins the identities the HIP path relies on; tests/test_kernels_gpu.py::test_layerscale_backward_without_activation checks the
kernels that implement them."""
import pytest
import torch


@pytest.mark.parametrize("M,D,F,droppath", [(37, 16, 24, False), (50, 8, 40, True), (5, 32, 8, True)])
def test_layerscale_backward_identities(M, D, F, droppath):
    g = torch.Generator().manual_seed(M * 1000 + D)
    A, W, b, gamma, x_in, dX = (torch.randn(*s, generator=g, dtype=torch.float64) for s in ((M, F), (D, F), (D,), (D,), (M, D), (M, D)))
    r = ((torch.rand(M, generator=g) > 0.3).double() / 0.7) if droppath else torch.ones(M, dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (A, W, b, gamma, x_in)]
    Al, Wl, bl, gl, xl = leaves
    out = xl + r[:, None] * gl * (Al @ Wl.t() + bl)
    out.backward(dX)
    dXp = r[:, None] * dX
    cs = dXp.sum(0)
    G = dXp.t() @ A
    tol = dict(rtol=1e-10, atol=1e-10)
    assert torch.allclose(Al.grad, dXp @ (gamma[:, None] * W), **tol)            # input gradient through the gamma-folded weight
    assert torch.allclose(Wl.grad, gamma[:, None] * G, **tol)                    # weight gradient = scaled GEMM result
    assert torch.allclose(bl.grad, gamma * cs, **tol)
    assert torch.allclose(gl.grad, (G * W).sum(1) + b * cs, **tol)               # no pass over the branch output u
    assert torch.allclose(xl.grad, dX, **tol)
