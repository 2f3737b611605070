"""ldm_rmsnorm_bwd_noalloc (the entry point that needs no workspace and allocates nothing) against
ldm_rmsnorm_bwd, through the C ABI, on two calls in a row -- the second with a smaller input, so a
stale tail of the shared scratch array from the first call would show."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, N, x, w, rstd, dy, dres, T, d, code):
    dev = x.device
    st = N.stream()
    out = []
    for noalloc in (False, True):
        dx = torch.empty_like(x)
        dw = torch.full((d,), float("nan"), dtype=torch.float32, device=dev)
        if noalloc:
            rc = lib.ldm_rmsnorm_bwd_noalloc(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), N.ptr(dres),
                                             T, d, code, dx.data_ptr(), dw.data_ptr(), st)
        else:
            ws = torch.empty(lib.ldm_rmsnorm_bwd_ws_floats(T, d), dtype=torch.float32, device=dev)
            rc = lib.ldm_rmsnorm_bwd(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), N.ptr(dres), T, d,
                                     code, dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        out.append((dx, dw))
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_noalloc_rmsnorm_backward_equals_workspace_form(dtype):
    from src import _native as N
    lib = N.lib()
    dev = torch.device("cuda:0")
    code = N.dtype_code(dtype)
    g = torch.Generator(device="cpu").manual_seed(0)
    for T, d, with_res in ((8192, 768, True), (1000, 768, False), (4096, 2048, True), (37, 256, True),
                           (2048, 4096, True), (515, 4096, False)):
        x = torch.randn(T, d, generator=g).to(dev, dtype)
        dy = torch.randn(T, d, generator=g).to(dev, dtype)
        dres = torch.randn(T, d, generator=g).to(dev, dtype) if with_res else None
        w = (1.0 + 0.1 * torch.randn(d, generator=g)).to(dev)
        y = torch.empty_like(x)
        rstd = torch.empty(T, dtype=torch.float32, device=dev)
        assert lib.ldm_rmsnorm_fwd(x.data_ptr(), w.data_ptr(), T, d, 1e-12, code, y.data_ptr(), rstd.data_ptr(),
                                   N.stream()) == 0
        (dx_ws, dw_ws), (dx_na, dw_na) = _both(lib, N, x, w, rstd, dy, dres, T, d, code)
        # the entry point allocates nothing: free device memory is the same before and after a call
        dx2 = torch.empty_like(x)
        dw2 = torch.empty(d, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        assert lib.ldm_rmsnorm_bwd_noalloc(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), N.ptr(dres),
                                           T, d, code, dx2.data_ptr(), dw2.data_ptr(), N.stream()) == 0
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0
        assert torch.equal(dx_ws, dx_na), (T, d)
        if d <= 1024:      # same number of workgroups in both forms: the same sums in the same order
            assert torch.equal(dw_ws, dw_na), (T, d)
        else:              # fewer workgroups share the fixed array: another summation order
            assert torch.allclose(dw_ws, dw_na, rtol=1e-5, atol=1e-5 * float(dw_ws.abs().max())), (T, d)
        # and against plain fp64 arithmetic
        xd, dyd, wd = x.double(), dy.double(), w.double()
        r = rstd.double()[:, None]
        gw = dyd * wd
        ref_dx = r * gw - xd * r ** 3 * (gw * xd).mean(-1, keepdim=True)
        if dres is not None:
            ref_dx = ref_dx + dres.double()
        ref_dw = (dyd * xd * r).sum(0)
        tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
        assert float((dx_na.double() - ref_dx).norm() / ref_dx.norm()) < tol
        assert float((dw_na.double() - ref_dw).norm() / ref_dw.norm()) < 1e-4
    # sizes beyond the fixed array are refused with the workspace error, not run
    assert lib.ldm_rmsnorm_bwd_noalloc_max_d() == 4096
    d_big = 4096 + 8
    z = torch.zeros(4, d_big, dtype=dtype, device=dev)
    zf = torch.zeros(d_big, dtype=torch.float32, device=dev)
    rc = lib.ldm_rmsnorm_bwd_noalloc(z.data_ptr(), zf.data_ptr(), zf.data_ptr(), z.data_ptr(), None, 4, d_big, code,
                                     z.data_ptr(), zf.data_ptr(), N.stream())
    assert rc == -5
