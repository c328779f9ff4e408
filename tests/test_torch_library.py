"""The functional operators of the path are registered with torch.library (namespace ``hrviton``, CUDA key only):
schemas on CPU, dispatcher behaviour (schema / fake-tensor checks, autograd through torch.ops) on the GPU."""
import pytest
import torch

import hr_viton_amd  # noqa: F401
from hr_viton_amd import functional as HF


def test_ops_are_registered_with_schemas():
    want = {
        "grid_sample": "hrviton::grid_sample(Tensor inp, Tensor grid) -> Tensor",
        "softmax2d": "hrviton::softmax2d(Tensor x) -> Tensor",
        "cross_entropy2d": "hrviton::cross_entropy2d(Tensor x, Tensor target, bool with_grad) -> Tensor[]",
        "tv_loss": "hrviton::tv_loss(Tensor flow, bool with_grad) -> Tensor[]",
    }
    for name in HF.REGISTERED_OPS:
        op = getattr(torch.ops.hrviton, name).default
        if name in want:
            assert str(op._schema) == want[name]
    # CUDA kernels only: a CPU tensor is refused by the host mirror before dispatch, and by the dispatcher itself
    with pytest.raises(Exception):
        HF.softmax(torch.zeros(1, 3, 4, 4))
    with pytest.raises(NotImplementedError):
        torch.ops.hrviton.softmax2d(torch.zeros(1, 3, 4, 4))


@pytest.mark.gpu
def test_dispatcher_ops_opcheck_and_autograd():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 12, 10, generator=g).cuda().requires_grad_()
    grid = (torch.rand(2, 9, 7, 2, generator=g) * 2.4 - 1.2).cuda().requires_grad_()
    utils = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.hrviton.softmax2d.default, (x,), test_utils=utils)
    torch.library.opcheck(torch.ops.hrviton.grid_sample.default, (x, grid), test_utils=utils)
    torch.library.opcheck(torch.ops.hrviton.interpolate_bilinear.default, (x, 24, 20, 0.5, 0.5), test_utils=utils)
    # autograd through the registered ops == torch's own functions
    y = torch.ops.hrviton.grid_sample(x, grid)
    w = torch.randn(y.shape, generator=g).cuda()
    (y * w).sum().backward()
    xr, gr = x.detach().clone().requires_grad_(), grid.detach().clone().requires_grad_()
    yr = torch.nn.functional.grid_sample(xr, gr, mode="bilinear", padding_mode="border", align_corners=False)
    (yr * w).sum().backward()
    assert torch.allclose(y, yr, atol=1e-5) and torch.allclose(x.grad, xr.grad, atol=1e-4)
    assert torch.allclose(grid.grad, gr.grad, atol=1e-3)
    tgt = torch.randint(0, 5, (2, 12, 10), generator=g).cuda()
    xl = x.detach().clone().requires_grad_()
    loss = HF.cross_entropy2d(xl, tgt) + HF.tv_loss(grid.detach().clone().requires_grad_())
    loss.backward()
    xl2 = x.detach().clone().requires_grad_()
    torch.nn.functional.cross_entropy(xl2, tgt).backward()
    assert torch.allclose(xl.grad, xl2.grad, atol=1e-6)


def test_module_level_ops_are_registered_with_schemas():
    """SURVEY 8(b)'s operator list under torch.ops.hrviton.* (library_ops.py): schemas on CPU."""
    from hr_viton_amd import library_ops as L
    for name in L.REGISTERED_OPS:
        assert hasattr(torch.ops.hrviton, name), name
    s = str(torch.ops.hrviton.conv2d_nhwc_fwd.default._schema)
    assert s == ("hrviton::conv2d_nhwc_fwd(Tensor x, Tensor weight, Tensor? bias, SymInt stride, SymInt pad, SymInt act, float slope) "
                 "-> Tensor"), s
    assert "Tensor(a0!) w" in str(torch.ops.hrviton.fused_adam.default._schema)          # in-place updates are declared
    assert "Tensor(a1!) u" in str(torch.ops.hrviton.spectral_sigma.default._schema)
    with pytest.raises(NotImplementedError):      # CUDA kernels only
        torch.ops.hrviton.instnorm_stats(torch.zeros(1, 4, 4, 8), 1e-5)


@pytest.mark.gpu
def test_module_level_ops_match_torch_and_pass_opcheck():
    import torch.nn.functional as F
    from hr_viton_amd import library_ops as L  # noqa: F401
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 24, 20, 8, generator=g).cuda()                 # NHWC
    w = (torch.randn(12, 8, 3, 3, generator=g) * 0.2).cuda()
    b = torch.randn(12, generator=g).cuda()
    utils = ("test_schema", "test_faketensor")
    torch.library.opcheck(torch.ops.hrviton.conv2d_nhwc_fwd.default, (x, w, b, 1, 1, 0, 0.2), test_utils=utils)
    y = torch.ops.hrviton.conv2d_nhwc_fwd(x, w, b, 1, 1, 0, 0.2)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1)
    assert torch.allclose(y[..., :12], ref, atol=1e-4)
    dy = torch.randn(2, 24, 20, 12, generator=g).cuda()
    dx = torch.ops.hrviton.conv2d_nhwc_dgrad(dy, w, 24, 20, 1, 1)
    dw, db = torch.ops.hrviton.conv2d_nhwc_wgrad(dy, x, 3, 3, 1, 1)
    xr, wr, br = x.permute(0, 3, 1, 2).clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    (F.conv2d(xr, wr, br, padding=1) * dy.permute(0, 3, 1, 2)).sum().backward()
    assert torch.allclose(dx[..., :8], xr.grad.permute(0, 2, 3, 1), atol=1e-3)
    assert torch.allclose(dw, wr.grad, atol=2e-3) and torch.allclose(db, br.grad, atol=2e-3)
    mean, rstd = torch.ops.hrviton.instnorm_stats(x, 1e-5)
    xm = x.permute(0, 3, 1, 2).flatten(2)
    assert torch.allclose(mean, xm.mean(2), atol=1e-5) and torch.allclose(rstd, (xm.var(2, unbiased=False) + 1e-5).rsqrt(), atol=1e-4)
    # fused Adam == torch.optim.Adam on a flat buffer
    p = torch.randn(1000, generator=g).cuda()
    gr = torch.randn(1000, generator=g).cuda()
    pt = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([pt], lr=1e-3, betas=(0.5, 0.999))
    pt.grad = gr.clone()
    opt.step()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    torch.ops.hrviton.fused_adam(p, gr, m, v, 1e-3, 0.5, 0.999, 1e-8, 0.0, 1, 1.0)
    assert torch.allclose(p, pt.detach(), atol=1e-6)
    lo, gl = torch.ops.hrviton.loss_reduce(x, torch.zeros_like(x), 0, 1.0 / x.numel(), 1.0 / x.numel(), True)
    assert torch.allclose(lo[0], x.abs().mean(), atol=1e-5) and torch.allclose(gl, torch.sign(x) / x.numel(), atol=1e-8)
