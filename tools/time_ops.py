import torch, time
torch.set_default_dtype(torch.float64)
dev = torch.device("cuda:0")
S, C, rl, p, rr = 60000, 9, 24, 50, 24
L = torch.rand(S, C, rl, device=dev); X = torch.rand(S, p, device=dev); R = torch.rand(S, rr, device=dev)
v = torch.rand(rl, p, rr, device=dev); g = torch.rand(S, C, device=dev)
def t(name, f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms", flush=True)
vm = v.permute(1, 0, 2).reshape(p, rl * rr).contiguous()
t("X @ v (S,p)x(p,576)", lambda: X @ vm)
T = X @ vm
t("z = (T*R).sum", lambda: (T.reshape(S, rl, rr) * R.unsqueeze(1)).sum(dim=2))
z = (T.reshape(S, rl, rr) * R.unsqueeze(1)).sum(dim=2)
t("bmm L z", lambda: torch.bmm(L, z.unsqueeze(2)))
t("(L*z).sum", lambda: (L * z.unsqueeze(1)).sum(-1))
t("einsum scl,sl->sc", lambda: torch.einsum('scl,sl->sc', L, z))
t("bmm g L", lambda: torch.bmm(g.unsqueeze(1), L))
t("(g*L).sum(1)", lambda: (g.unsqueeze(-1) * L).sum(1))
GL = (g.unsqueeze(-1) * L).sum(1)
t("W outer", lambda: (GL.unsqueeze(2) * R.unsqueeze(1)).reshape(S, rl * rr))
W = (GL.unsqueeze(2) * R.unsqueeze(1)).reshape(S, rl * rr)
t("X.T @ W", lambda: X.T @ W)
Xt = X.T.contiguous()
t("Xt @ W (contig)", lambda: Xt @ W)
t("(W.T @ X)", lambda: W.T @ X)
# split-K by hand: 64 chunks batched
nk = 60
Xc = X.reshape(nk, S // nk, p); Wc = W.reshape(nk, S // nk, rl * rr)
t("bmm splitK 60 + sum", lambda: torch.bmm(Xc.transpose(1, 2), Wc).sum(0))
