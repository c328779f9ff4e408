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
