"""ttk_rhs_static_f64 (the entry point that allocates nothing: its slab partials live in a file-scope
device array) against ttk_rhs_f64 through the raw C ABI: identical bits, on two calls in a row, the
second with a smaller input; sizes beyond the array are refused."""
import pytest
import torch

gpu = pytest.mark.gpu
OK, EWORKSPACE = 0, -2


def _case(S, C, rl, p, rr, seed, dev):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *shape: (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1).to(dev)   # noqa: E731
    return mk(S, C, rl), mk(S, p), mk(S, rr), mk(S, C)


@gpu
def test_rhs_static_matches_rhs_twice():
    from tensornetworksfork_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    assert lib.ttk_rhs_static_max_bytes() == 32 * 1024 * 1024
    for (S, C, rl, p, rr), seed in (((54000, 1, 32, 6, 32), 1), ((3001, 3, 16, 5, 24), 2), ((5000, 9, 10, 50, 10), 5),
                                   ((4000, 1, 16, 40, 16), 6), ((130, 2, 6, 4, 6), 3)):
        L, X, R, g = _case(S, C, rl, p, rr, seed, dev)
        nbytes = lib.ttk_rhs_ws_bytes(S, C, rl, p, rr)
        assert 0 < nbytes <= lib.ttk_rhs_static_max_bytes()
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dev)
        ref = torch.zeros(rl, p, rr, dtype=torch.float64, device=dev)
        got = torch.full((rl, p, rr), 3.0, dtype=torch.float64, device=dev)
        args = (L.data_ptr(), X.data_ptr(), R.data_ptr(), g.data_ptr())
        assert lib.ttk_rhs_f64(*args, ref.data_ptr(), S, C, rl, p, rr, 0, ws.data_ptr(), nbytes, st) == OK
        before = torch.cuda.memory_allocated(dev)
        assert lib.ttk_rhs_static_f64(*args, got.data_ptr(), S, C, rl, p, rr, 0, st) == OK
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(dev) == before
        assert torch.equal(got, ref), (S, C, rl, p, rr)
        # accumulating form on top of the previous result; the device's free memory as the driver reports it
        # (which a hipMalloc inside the library would lower, unseen by torch's allocator) stays where it was
        free_before = torch.cuda.mem_get_info(dev)[0]
        assert lib.ttk_rhs_static_f64(*args, got.data_ptr(), S, C, rl, p, rr, 1, st) == OK
        torch.cuda.synchronize()
        assert abs(torch.cuda.mem_get_info(dev)[0] - free_before) < (1 << 20)    # (the smallest device allocation is 2 MiB)
        assert lib.ttk_rhs_f64(*args, ref.data_ptr(), S, C, rl, p, rr, 1, ws.data_ptr(), nbytes, st) == OK
        torch.cuda.synchronize()
        assert torch.equal(got, ref)


@gpu
def test_rhs_static_refuses_what_does_not_fit():
    from tensornetworksfork_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    S, C, rl, p, rr = 40000, 1, 48, 8, 48          # 512 slabs x 18432 doubles = 72 MiB of partials
    assert lib.ttk_rhs_ws_bytes(S, C, rl, p, rr) > lib.ttk_rhs_static_max_bytes()
    L, X, R, g = _case(S, C, rl, p, rr, 4, dev)
    b = torch.full((rl, p, rr), 5.0, dtype=torch.float64, device=dev)
    rc = lib.ttk_rhs_static_f64(L.data_ptr(), X.data_ptr(), R.data_ptr(), g.data_ptr(), b.data_ptr(), S, C, rl, p, rr, 0,
                                torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    assert rc == EWORKSPACE and bool((b == 5.0).all())
