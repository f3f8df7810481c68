"""Fused CG iterations queued eagerly against the same iterations replayed from a captured HIP graph
(blocks of 20 iterations; the headline middle-site problem: 54 000 samples, bonds 32, 6 features)."""
import sys

import torch

sys.path.insert(0, ".")
from tensornetworksfork_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
S, rl, p, rr = 54000, 32, 6, 32
g = torch.Generator(device=dev).manual_seed(3)
L = torch.rand(S, 1, rl, generator=g, dtype=torch.float64, device=dev) - 0.5
X = torch.rand(S, p, generator=g, dtype=torch.float64, device=dev)
R = torch.rand(S, rr, generator=g, dtype=torch.float64, device=dev) - 0.5
H = torch.ones(S, 1, 1, dtype=torch.float64, device=dev)
D = rl * p * rr
b = torch.rand(D, generator=g, dtype=torch.float64, device=dev) - 0.5
fused = ops.CGFused(L, X, R, H)


def fresh():
    xk = torch.zeros(D, dtype=torch.float64, device=dev)
    r = b.clone()
    pbuf = torch.zeros(2 * D, dtype=torch.float64, device=dev)
    bb = ops.dot(b, b)
    state = torch.stack((torch.zeros_like(bb), ops.dot(r, r), 1e-30 * bb.sqrt(), torch.zeros_like(bb), torch.zeros_like(bb))).contiguous()
    return xk, r, pbuf, state


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


N = 200
xk, r, pbuf, state = fresh()


def eager():
    for it in range(2, 2 + N):
        fused.iterate(xk, r, pbuf, state, it)


fused.iterate(xk, r, pbuf, state, 0)
fused.iterate(xk, r, pbuf, state, 1)
t_eager = timed(eager, reps=1)
print("state after eager run", state.tolist(), "x norm", xk.norm().item())
x_eager = xk.clone()

xk, r, pbuf, state = fresh()
fused.iterate(xk, r, pbuf, state, 0)
fused.iterate(xk, r, pbuf, state, 1)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for it in range(2, 22):          # warm-up on the side stream, as torch's capture wants
        fused.iterate(xk, r, pbuf, state, it)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for it in range(2, 22):
        fused.iterate(xk, r, pbuf, state, it)


def replay():
    for _ in range(N // 20):
        graph.replay()


t_graph = timed(replay, reps=1)
print("state after replay", state.tolist(), "x norm", xk.norm().item(), "same x", torch.equal(xk, x_eager))
print(f"{N} fused CG iterations: eager {t_eager * 1e3 / N:.2f} us per iteration, graph replay (blocks of 20) {t_graph * 1e3 / N:.2f} us per iteration")
