"""Error behaviour of the C ABI (include/ttk.h): a rejected call returns a negative code and
launches nothing; sizes reported by the *_ws_bytes queries are what the calls accept."""
import ctypes

import pytest
import torch

gpu = pytest.mark.gpu
OK, EINVAL, EWORKSPACE = 0, -1, -2


def _setup(S=300, C=1, rl=8, p=3, rr=8):
    from tensornetworksfork_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    L = torch.rand(S, C, rl, generator=g, dtype=torch.float64).to(dev)
    X = torch.rand(S, p, generator=g, dtype=torch.float64).to(dev)
    R = torch.rand(S, rr, generator=g, dtype=torch.float64).to(dev)
    H = torch.ones(S, C, C, dtype=torch.float64, device=dev)
    return _lib.lib(), (S, C, rl, p, rr), (L, X, R, H)


@gpu
def test_workspace_queries_and_rejections():
    lib, (S, C, rl, p, rr), (L, X, R, H) = _setup()
    assert lib.ttk_gram_ws_bytes(0, C, rl, p, rr, 1) == EINVAL
    assert lib.ttk_gram_ws_bytes(S, C, rl, 60, rr, 1) == EINVAL      # packed sample row does not fit an LDS chunk
    assert lib.ttk_rhs_ws_bytes(S, 0, rl, p, rr) == EINVAL
    assert lib.ttk_chol_ws_bytes(0) == EINVAL
    assert lib.ttk_qr_lds_bytes(4000, 64) > 150 * 1024               # too large for one workgroup
    nbytes = lib.ttk_gram_ws_bytes(S, C, rl, p, rr, 1)
    assert nbytes > 0
    ws = torch.full(((nbytes + 7) // 8,), 7.0, dtype=torch.float64, device=L.device)
    args = (L.data_ptr(), H.data_ptr(), X.data_ptr(), R.data_ptr(), S, C, rl, p, rr, 1)
    # one byte short: refused, workspace untouched
    assert lib.ttk_gram_f64(*args, None, ws.data_ptr(), nbytes - 1, None) == EWORKSPACE
    # null operand, bad extent
    assert lib.ttk_gram_f64(None, H.data_ptr(), X.data_ptr(), R.data_ptr(), S, C, rl, p, rr, 1, None, ws.data_ptr(), nbytes,
                            None) == EINVAL
    assert lib.ttk_gram_f64(L.data_ptr(), H.data_ptr(), X.data_ptr(), R.data_ptr(), S, C, rl, 0, rr, 1, None, ws.data_ptr(),
                            nbytes, None) == EINVAL
    # feature-pair classes must name class representatives (rep[b] <= b and rep[rep[b]] == rep[b])
    nx = p * (p + 1) // 2
    bad = (ctypes.c_int * nx)(*([1] + list(range(1, nx))))
    assert lib.ttk_gram_f64(*args, ctypes.cast(bad, ctypes.c_void_p), ws.data_ptr(), nbytes, None) == EINVAL
    torch.cuda.synchronize()
    assert bool((ws == 7.0).all())
    good = (ctypes.c_int * nx)(*range(nx))
    assert lib.ttk_gram_f64(*args, ctypes.cast(good, ctypes.c_void_p), ws.data_ptr(), nbytes, None) == OK
    torch.cuda.synchronize()
    assert not bool((ws == 7.0).all())


@gpu
def test_ops_refuse_host_tensors_and_wrong_shapes():
    from tensornetworksfork_amd import _lib, ops
    lib, (S, C, rl, p, rr), (L, X, R, H) = _setup()
    with pytest.raises(_lib.TTKError):
        ops.env_step_left(L.cpu(), X, torch.zeros(rl, p, rr, dtype=torch.float64, device=L.device))
    with pytest.raises(_lib.TTKError):
        ops.env_step_left(L, X[:, :2].contiguous(), torch.zeros(rl, p, rr, dtype=torch.float64, device=L.device))
    with pytest.raises(_lib.TTKError):
        ops.rhs(L, X, R, torch.zeros(S, C, dtype=torch.float32, device=L.device))
    with pytest.raises(_lib.TTKError):
        ops.gram(L, H, X, R, pair_rep=[0, 0])      # one entry per feature pair expected
