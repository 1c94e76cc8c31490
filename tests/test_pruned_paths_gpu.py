"""Pruned checkpoints on the library's own kernels only (VERDICT r04 missing #3): their channel counts are not multiples
of 8 (d_model 55, 114, 477 ...; x_proj rows 58, 60), which used to send the Mamba projections to F.linear (hipBLASLt)
and the LayerNorms to ATen.  Reference: load_pruned_state_dict, src/network/CleanUMamba.py:492-550;
src/examples/loading_pretrained_models.py:7-19."""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_ckpt, record, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(37, 114, 16), (50, 8, 114), (129, 48, 58), (64, 55, 60), (1, 114, 96)])
def test_projection_of_any_width_forward_and_gradients(cuda, shape, dtype):
    """_proj (cum_gemm_nt / cum_gemm_tn on padded operands) against an f64 matmul: y, dx, dW."""
    from cleanumamba_amd.mamba_ssm.modules.mamba_simple import _proj
    M, K, N = shape
    g = torch.Generator(device=cuda).manual_seed(M * 7 + K)
    x = torch.randn(3, M, K, generator=g, device=cuda).to(dtype).requires_grad_(True)
    w = (torch.randn(N, K, generator=g, device=cuda) / K ** 0.5).requires_grad_(True)
    dy = torch.randn(3, M, N, generator=g, device=cuda)
    with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
        y = _proj(x, w)
    assert y.shape == (3, M, N)
    y.backward(dy.to(y.dtype))
    wd = w.detach().to(dtype).double()
    want = x.detach().double() @ wd.t()
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    assert record(f"proj_any[{shape}-{dtype}].y", rel_l2(y, want)) < tol
    dyd = dy.to(y.dtype).double()
    assert rel_l2(x.grad, dyd @ wd) < tol
    assert rel_l2(w.grad, (dyd.reshape(-1, N).t() @ x.detach().double().reshape(-1, K))) < tol


@pytest.mark.parametrize("dim", [55, 114, 477, 130])
def test_add_layernorm_of_any_width(cuda, dim):
    from cleanumamba_amd.mamba_ssm.ops import layernorm as ln
    g = torch.Generator(device=cuda).manual_seed(dim)
    h = torch.randn(2, 33, dim, generator=g, device=cuda, requires_grad=True)
    r = torch.randn(2, 33, dim, generator=g, device=cuda, requires_grad=True)
    norm = torch.nn.LayerNorm(dim).to(cuda)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.1 * torch.randn(dim, generator=g, device=cuda))
        norm.bias.copy_(0.1 * torch.randn(dim, generator=g, device=cuda))
    assert ln.supported(h, norm)
    y, res = ln.add_layer_norm(h, r, norm)
    gy, gr = torch.randn_like(y), torch.randn_like(res)
    (y * gy).sum().add((res * gr).sum()).backward()
    got = [h.grad.clone(), r.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()]
    for t in (h, r, norm.weight, norm.bias):
        t.grad = None
    hd, rd = h.detach().double().requires_grad_(True), r.detach().double().requires_grad_(True)
    wd, bd = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    res_ref = hd + rd
    y_ref = F.layer_norm(res_ref, (dim,), wd, bd, norm.eps)
    (y_ref * gy.double()).sum().add((res_ref * gr.double()).sum()).backward()
    assert rel_l2(y, y_ref) < 1e-5 and rel_l2(res, res_ref) < 1e-6
    for a, b in zip(got, (hd.grad, rd.grad, wd.grad, bd.grad)):
        assert rel_l2(a, b) < 1e-4


@pytest.mark.parametrize("name", ["pruned500k", "e6_pruned2m", "e8_pruned200k"])
def test_pruned_forward_backward_and_stream_take_no_vendor_gemm_and_no_aten_layernorm(cuda, name, monkeypatch):
    """Forward + backward + a stream of hops of a pruned checkpoint with F.linear and F.layer_norm / nn.LayerNorm.forward
    booby-trapped: every projection and every LayerNorm runs on the library's kernels."""
    from cleanumamba_amd.network import CleanUMamba
    sd, cfg = load_ckpt(name)
    net = CleanUMamba(**cfg)
    net.load_pruned_state_dict(sd)
    net = net.to(cuda).train()
    x = (0.1 * torch.randn(2, 1, 9000, generator=torch.Generator().manual_seed(1))).to(cuda)

    def trap(*a, **k):
        raise AssertionError("vendor / ATen path taken")
    monkeypatch.setattr(F, "linear", trap)
    monkeypatch.setattr(F, "layer_norm", trap)
    monkeypatch.setattr(torch.nn.LayerNorm, "forward", trap)
    y = net(x)
    y.square().mean().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    net.eval()
    with torch.no_grad():
        out = torch.cat([net.feed_batch(x[:, 0]), net.flush_batch()], 1)
    assert out.shape == (2, 9000) and bool(torch.isfinite(out).all())
