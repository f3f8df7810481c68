"""Right-environment step against the left-environment kernel on the end-transposed core:
python tools/time_env_steps.py S rl p rr [S rl p rr ...]"""
import sys
import torch
sys.path.insert(0, ".")
from tensornetworksfork_amd import ops


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


a = [int(v) for v in sys.argv[1:]]
dev = torch.device("cuda:0")
for k in range(0, len(a), 4):
    S, rl, p, rr = a[k:k + 4]
    E = torch.rand(S, 1, rl, dtype=torch.float64, device=dev)
    X = torch.rand(S, p, dtype=torch.float64, device=dev)
    R = torch.rand(S, rr, dtype=torch.float64, device=dev)
    core = torch.rand(rl, p, rr, dtype=torch.float64, device=dev)
    t_left = timed(lambda: ops.env_step_left(E, X, core))
    t_right = timed(lambda: ops.env_step_right(R, X, core))
    R1 = R.unsqueeze(1)

    def via_left():
        return ops.env_step_left(R1, X, core.permute(2, 1, 0).contiguous()).reshape(S, rl)
    t_via = timed(via_left)
    err = ((via_left() - ops.env_step_right(R, X, core)).norm() / ops.env_step_right(R, X, core).norm()).item()
    print((S, rl, p, rr), "left %.1f us, right %.1f us, right through the left kernel %.1f us (rel diff %.1e)" % (t_left, t_right, t_via, err))
