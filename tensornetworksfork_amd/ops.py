"""torch.Tensor front-ends of the C ABI: argument checks, output allocation,
current-stream plumbing.  No arithmetic happens here."""
import ctypes

import torch

from . import _lib

F64 = torch.float64


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk(t, name, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.TTKError(f"{name}: expected a CUDA/HIP tensor (the core-update path has no CPU fallback)")
    if t.dtype != F64:
        raise _lib.TTKError(f"{name}: expected float64, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.TTKError(f"{name}: expected a contiguous tensor")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.TTKError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.data_ptr()


def prep(t):
    """Contiguous float64 view/copy on the same device (plumbing only)."""
    if t.dtype != F64:
        t = t.to(F64)
    return t.contiguous()


def env_step_left(E, X, core, out=None):
    """E (S,C,rl), X (S,p), core (rl,p,rr) -> (S,C,rr); with `out` given the result is added to it."""
    S, C, rl = E.shape
    p, rr = core.shape[1], core.shape[2]
    acc = out is not None
    if out is None:
        out = torch.empty((S, C, rr), dtype=F64, device=E.device)
    if S == 0:
        return out
    rc = _lib.lib().ttk_env_step_left_f64(_chk(E, "E"), _chk(X, "X", (S, p)), _chk(core, "core", (rl, p, rr)),
                                          _chk(out, "out", (S, C, rr)), S, C, rl, p, rr, int(acc), _stream(E))
    if rc == -1 and p > FEATURE_BLOCK:      # rows of rl + p doubles beyond the LDS image: sum over feature blocks
        for k, (a, b) in enumerate(_feature_blocks(p)):
            Xb, cb = X[:, a:b].contiguous(), core[:, a:b, :].contiguous()
            rc = _lib.lib().ttk_env_step_left_f64(E.data_ptr(), Xb.data_ptr(), cb.data_ptr(), out.data_ptr(), S, C, rl, b - a, rr,
                                                  int(acc or k > 0), _stream(E))
            _lib.check(rc, "ttk_env_step_left_f64")
        return out
    _lib.check(rc, "ttk_env_step_left_f64")
    return out


DENSE_MIN_P = 16    # from this many features on a site contracts its features first (ttk_dense kernels)
DENSE_MAX_P = 192   # ... up to what the row GEMM keeps in LDS (beyond: the streaming Khatri-Rao kernels)
FEATURE_BLOCK = 128  # a site whose sample rows are too wide for any kernel's LDS image goes feature block by feature block


def _feature_blocks(p):
    return [(a, min(a + FEATURE_BLOCK, p)) for a in range(0, p, FEATURE_BLOCK)]


def env_step_right(R, X, core, out=None, mm32=False):
    """R (S,rr), X (S,p), core (rl,p,rr) -> (S,rl); with `out` given the result is added to it.  `mm32`: float32
    matrix arithmetic on the many-feature route (float32 models)."""
    S, rr = R.shape
    rl, p = core.shape[0], core.shape[1]
    acc = out is not None
    if out is None:
        out = torch.empty((S, rl), dtype=F64, device=R.device)
    if S == 0:
        return out
    if DENSE_MIN_P <= p <= DENSE_MAX_P and rr == 1 and not acc:
        # the last site: out[s,l] = (X . core[:, :, 0]^T)[s,l] R[s,0], a plain row GEMM and a row scaling
        T = gemm_rows(X, core.reshape(rl, p), rl, bmode=2, mm32=mm32)
        return kr_rows(T.unsqueeze(1), R, R.new_ones((S, 1))).reshape(S, rl)
    if DENSE_MIN_P <= p <= DENSE_MAX_P and (rr <= 64 or _lib.lib().ttk_env_right_dense_ok(S, rl, p, rr, int(bool(mm32)))):
        fn = _lib.lib().ttk_env_right_dense_mm32 if mm32 else _lib.lib().ttk_env_right_dense_f64
        rc = fn(_chk(R, "R"), _chk(X, "X", (S, p)), _chk(core, "core", (rl, p, rr)), _chk(out, "out", (S, rl)), S, rl, p, rr,
                int(acc), _stream(R))
        _lib.check(rc, "ttk_env_right_dense")
        return out
    rc = _lib.lib().ttk_env_step_right_f64(_chk(R, "R"), _chk(X, "X", (S, p)), _chk(core, "core", (rl, p, rr)),
                                           _chk(out, "out", (S, rl)), S, rl, p, rr, int(acc), _stream(R))
    if rc == -1 and p > FEATURE_BLOCK:      # rows of rr + p doubles beyond the LDS image: sum over feature blocks
        for k, (a, b) in enumerate(_feature_blocks(p)):
            Xb, cb = X[:, a:b].contiguous(), core[:, a:b, :].contiguous()
            rc = _lib.lib().ttk_env_step_right_f64(R.data_ptr(), Xb.data_ptr(), cb.data_ptr(), out.data_ptr(), S, rl, b - a, rr,
                                                   int(acc or k > 0), _stream(R))
            _lib.check(rc, "ttk_env_step_right_f64")
        return out
    _lib.check(rc, "ttk_env_step_right_f64")
    return out


def predict(L, R, out=None):
    """L (S,C,r), R (S,r) -> (S,C); with `out` given the result is added to it."""
    S, C, r = L.shape
    acc = out is not None
    y = out if acc else torch.empty((S, C), dtype=F64, device=L.device)
    if S == 0:
        return y
    rc = _lib.lib().ttk_predict_f64(_chk(L, "L"), _chk(R, "R", (S, r)), _chk(y, "y", (S, C)), S, C, r, int(acc),
                                    _stream(L))
    _lib.check(rc, "ttk_predict_f64")
    return y


LOSS_KINDS = {"square": 0, "mse": 1, "xe": 2, "xe_hard": 3}


def loss(kind, y_pred, y, w=1.0):
    """-> (loss, g (S,C), H (S,C,C)); loss is (S,) for square/xe and (S,C) for mse."""
    S, C = y_pred.shape
    k = LOSS_KINDS[kind]
    ycols = C + 1 if k >= 2 else C
    lo = torch.empty((S, C) if k == 1 else (S,), dtype=F64, device=y_pred.device)
    g = torch.empty((S, C), dtype=F64, device=y_pred.device)
    H = torch.empty((S, C, C), dtype=F64, device=y_pred.device)
    if S == 0:
        return lo, g, H
    rc = _lib.lib().ttk_loss_f64(k, float(w), _chk(y_pred, "y_pred"), _chk(y, "y", (S, ycols)), lo.data_ptr(),
                                 g.data_ptr(), H.data_ptr(), S, C, _stream(y_pred))
    _lib.check(rc, "ttk_loss_f64")
    return lo, g, H


def batch_mean(v, batch):
    """Mean over mini-batches (of `batch` samples) of per-batch means; 0-d device tensor."""
    S = v.shape[0]
    nps = v.numel() // S
    out = torch.empty((), dtype=F64, device=v.device)
    rc = _lib.lib().ttk_batch_mean_f64(_chk(v, "v"), S, nps, int(batch), out.data_ptr(), _stream(v))
    _lib.check(rc, "ttk_batch_mean_f64")
    return out


def kr_rows(L, X, R, out=None):
    """Explicit Jacobian rows (S,C,rl*p*rr) of one Khatri-Rao term; added to `out` when given."""
    S, C, rl = L.shape
    p, rr = X.shape[1], R.shape[1]
    acc = out is not None
    J = out if acc else torch.empty((S, C, rl * p * rr), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_kr_rows_f64(_chk(L, "L"), _chk(X, "X", (S, p)), _chk(R, "R", (S, rr)),
                                    _chk(J, "J", (S, C, rl * p * rr)), S, C, rl, p, rr, int(acc), _stream(L))
    _lib.check(rc, "ttk_kr_rows_f64")
    return J


def hess_left(H, L):
    S, C, rl = L.shape
    HL = torch.empty_like(L)
    rc = _lib.lib().ttk_hess_left_f64(_chk(H, "H", (S, C, C)), _chk(L, "L"), HL.data_ptr(), S, C, rl, _stream(L))
    _lib.check(rc, "ttk_hess_left_f64")
    return HL


def gram_gsize(rl, p, rr, sym):
    """Number of doubles of the compressed Gram [pairs(l,m) -> 48][pairs(p,q)][pairs(r,t) -> 48]."""
    def up48(v):
        return (v + 47) // 48 * 48
    mp = rl * (rl + 1) // 2 if sym else rl * rl
    return up48(mp) * (p * (p + 1) // 2) * up48(rr * (rr + 1) // 2)


def gram_supported(S, C, rl, p, rr, sym=True):
    """Whether ttk_gram_f64 accepts these extents (its packed per-sample row must fit an LDS chunk)."""
    return int(_lib.lib().ttk_gram_ws_bytes(int(S), int(C), int(rl), int(p), int(rr), int(bool(sym)))) > 0


class GramWorkspace:
    """Symmetry-compressed Gram matrix held in a device workspace."""

    def __init__(self, S, C, rl, p, rr, sym, device):
        self.dims = (int(S), int(C), int(rl), int(p), int(rr), int(bool(sym)))
        nbytes = _lib.lib().ttk_gram_ws_bytes(*self.dims)
        if nbytes < 0:
            raise _lib.TTKError("ttk_gram_ws_bytes: invalid extents")
        self.nbytes = int(nbytes)
        self.gsize = gram_gsize(int(rl), int(p), int(rr), bool(sym))
        # room for a right-hand side right behind the compressed Gram, whatever the sample count: a
        # multi-rank run sends [Gram | b] as one message and every rank must lay it out alike
        self.tail = int(rl) * int(p) * int(rr)
        self.buf = torch.empty(max((self.nbytes + 7) // 8, self.gsize + self.tail), dtype=F64, device=device)
        mp = int(rl) * (int(rl) + 1) // 2 if sym else int(rl) * int(rl)
        self.mppad = (mp + 47) // 48 * 48    # leading extent of the compressed Gram [mppad][pairs(p,q)][nrpad]
        self.pair_rep = None                 # feature-pair classes of the last Gram computed into it
        self.generation = 0                  # bumped by every Gram computed into this workspace
        self.cancel_word = torch.zeros(1, dtype=torch.int32, device=device)   # see gram_cancel

    @property
    def compressed(self):
        """The reduced compressed Gram (what a multi-rank all-reduce sums)."""
        return self.buf[:self.gsize]


def gram(L, H, X, R, sym=True, ws=None, pair_rep=None):
    """Compressed Gram of the local normal equations (H = per-sample Hessian (S,C,C)) -> GramWorkspace.
    pair_rep: feature-pair classes as returned by pair_classes(X) (None: every pair computed)."""
    S, C, rl = L.shape
    p, rr = X.shape[1], R.shape[1]
    if ws is None or ws.dims != (S, C, rl, p, rr, int(bool(sym))):
        ws = GramWorkspace(S, C, rl, p, rr, sym, L.device)
    rep = None
    if pair_rep is not None:
        if len(pair_rep) != p * (p + 1) // 2:
            raise _lib.TTKError("pair_rep: expected one entry per feature pair")
        rep = (ctypes.c_int * len(pair_rep))(*[int(v) for v in pair_rep])
    ws.generation += 1                       # (the kernel is told generation % (2^32 - 1) + 1: never the word's initial 0)
    rc = _lib.lib().ttk_gram_cancellable_f64(
        _chk(L, "L"), _chk(H, "H", (S, C, C)), _chk(X, "X", (S, p)), _chk(R, "R", (S, rr)), S, C, rl, p, rr, int(bool(sym)),
        None if rep is None else ctypes.cast(rep, ctypes.c_void_p), ws.buf.data_ptr(), ws.nbytes, ws.cancel_word.data_ptr(),
        ws.generation % 0xFFFFFFFF + 1, _stream(L))
    _lib.check(rc, "ttk_gram_cancellable_f64")
    ws.pair_rep = None if pair_rep is None else [int(v) for v in pair_rep]
    return ws


def gram_cancel(ws):
    """Call off the last Gram queued into `ws` (ops.gram): workgroups of its kernel that have not started yet leave at
    once and the workspace holds nothing of use afterwards.  Queued on the CURRENT stream, which must not be waiting
    for the stream the Gram runs on, or the word arrives after the kernel."""
    rc = _lib.lib().ttk_gram_cancel(ws.cancel_word.data_ptr(), ws.generation % 0xFFFFFFFF + 1, _stream(ws.cancel_word))
    _lib.check(rc, "ttk_gram_cancel")


def gram_scale(ws):
    _, _, rl, p, rr, sym = ws.dims
    scale = torch.empty((), dtype=F64, device=ws.buf.device)
    rc = _lib.lib().ttk_gram_scale_f64(ws.buf.data_ptr(), rl, p, rr, sym, scale.data_ptr(), _stream(ws.buf))
    _lib.check(rc, "ttk_gram_scale_f64")
    return scale


def gram_expand(ws, scale=None, ridge=0.0, out=None):
    """Full (D,D) matrix A/scale + ridge*I."""
    _, _, rl, p, rr, sym = ws.dims
    D = rl * p * rr
    if out is None:
        out = torch.empty((D, D), dtype=F64, device=ws.buf.device)
    rc = _lib.lib().ttk_gram_expand_f64(ws.buf.data_ptr(), rl, p, rr, sym,
                                        None if scale is None else _chk(scale, "scale"), float(ridge),
                                        _chk(out, "A_out", (D, D)), _stream(ws.buf))
    _lib.check(rc, "ttk_gram_expand_f64")
    return out


_rhs_ws = {}


def rhs_supported(S, C, rl, p, rr):
    """Whether ttk_rhs_f64 accepts these extents (a chunk of its sample rows must fit LDS)."""
    return int(_lib.lib().ttk_rhs_ws_bytes(int(S), int(C), int(rl), int(p), int(rr))) > 0


def rhs(L, X, R, g, out=None):
    """b (rl,p,rr) = sum_{s,c} g[s,c] L[s,c,l] X[s,p] R[s,r]; with `out` given the result is added to it."""
    S, C, rl = L.shape
    p, rr = X.shape[1], R.shape[1]
    nbytes = _lib.lib().ttk_rhs_ws_bytes(S, C, rl, p, rr)
    if nbytes < 0:
        raise _lib.TTKError("ttk_rhs_ws_bytes: invalid extents")
    key = (L.device, int(nbytes))
    ws = _rhs_ws.get(key)
    if ws is None:
        _rhs_ws.clear()
        ws = torch.empty((nbytes + 7) // 8, dtype=F64, device=L.device)
        _rhs_ws[key] = ws
    acc = out is not None
    b = out if acc else torch.empty((rl, p, rr), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_rhs_f64(_chk(L, "L"), _chk(X, "X", (S, p)), _chk(R, "R", (S, rr)), _chk(g, "g", (S, C)),
                                _chk(b, "b", (rl, p, rr)), S, C, rl, p, rr, int(acc), ws.data_ptr(), int(nbytes),
                                _stream(L))
    _lib.check(rc, "ttk_rhs_f64")
    return b


def jvp(L, X, R, H, v):
    """hu (S,C) = H . (J v) for the local Jacobian J = L (x) X (x) R."""
    S, C, rl = L.shape
    p, rr = X.shape[1], R.shape[1]
    tmp = torch.empty((S, C, rr), dtype=F64, device=L.device)
    hu = torch.empty((S, C), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_jvp_f64(_chk(L, "L"), _chk(X, "X", (S, p)), _chk(R, "R", (S, rr)), _chk(H, "H", (S, C, C)),
                                _chk(v, "v", (rl, p, rr)), tmp.data_ptr(), hu.data_ptr(), S, C, rl, p, rr, _stream(L))
    if rc == -1 and p > FEATURE_BLOCK:      # too wide for the one call: the same three steps, the first in feature blocks
        u = predict(env_step_left(L, X, v), R)
        return hess_left(H, u.unsqueeze(-1)).squeeze(-1)
    _lib.check(rc, "ttk_jvp_f64")
    return hu


_mvf_ws = {}


def matvec_fused_supported(S, C, rl, p, rr):
    """Whether the one-pass product kernel takes these extents (one output, square bonds of 16 / 32, p <= 8)."""
    return int(_lib.lib().ttk_matvec_fused_ws_bytes(int(S), int(C), int(rl), int(p), int(rr))) > 0


def _mvf_workspace(S, rl, p, rr, device):
    nbytes = int(_lib.lib().ttk_matvec_fused_ws_bytes(S, 1, rl, p, rr))
    key = (device, nbytes)
    ws = _mvf_ws.get(key)
    if ws is None:
        _mvf_ws.clear()
        ws = torch.empty((nbytes + 7) // 8, dtype=F64, device=device)
        _mvf_ws[key] = ws
    return ws, nbytes


def matvec(L, X, R, H, v):
    """(J^T H J) v without forming the Gram matrix: one fused pass where the shape allows it, else
    the Jacobian-vector product followed by the transposed one."""
    S, C, rl = L.shape
    p, rr = X.shape[1], R.shape[1]
    if S > 0 and matvec_fused_supported(S, C, rl, p, rr):
        ws, nbytes = _mvf_workspace(S, rl, p, rr, L.device)
        out = torch.empty((rl, p, rr), dtype=F64, device=L.device)
        rc = _lib.lib().ttk_matvec_fused_f64(_chk(L, "L"), _chk(X, "X", (S, p)), _chk(R, "R", (S, rr)),
                                             _chk(H, "H", (S, 1, 1)), _chk(v, "v", (rl, p, rr)), out.data_ptr(),
                                             S, rl, p, rr, ws.data_ptr(), nbytes, _stream(L))
        _lib.check(rc, "ttk_matvec_fused_f64")
        return out
    return rhs(L, X, R, jvp(L, X, R, H, v))


class CGFused:
    """scipy.sparse.linalg.cg on the operator of one Khatri-Rao term, two launches per iteration and no
    host read inside the loop (see ttk_cg_fused_iter_f64)."""

    def __init__(self, L, X, R, H):
        S, C, rl = L.shape
        self.dims = (S, rl, X.shape[1], R.shape[1])
        self.ptrs = (_chk(L, "L"), _chk(X, "X", (S, X.shape[1])), _chk(R, "R", (S, R.shape[1])), _chk(H, "H", (S, 1, 1)))
        self.keep = (L, X, R, H)
        self.ws, self.nbytes = _mvf_workspace(S, rl, X.shape[1], R.shape[1], L.device)

    def iterate(self, x, r, pbuf, state, it):
        S, rl, p, rr = self.dims
        rc = _lib.lib().ttk_cg_fused_iter_f64(*self.ptrs, _chk(x, "x"), _chk(r, "r"), _chk(pbuf, "pbuf"),
                                              _chk(state, "state"), int(it), S, rl, p, rr, self.ws.data_ptr(),
                                              self.nbytes, _stream(x))
        _lib.check(rc, "ttk_cg_fused_iter_f64")


class LanczosFused:
    """The Lanczos recurrence on the operator of one Khatri-Rao term (extents of the one-pass product kernel): a block
    of iterations is queued by ONE library call, three launches per iteration and no interpreter work between them."""

    def __init__(self, L, X, R, H):
        S, C, rl = L.shape
        self.dims = (S, rl, X.shape[1], R.shape[1])
        self.ptrs = (_chk(L, "L"), _chk(X, "X", (S, X.shape[1])), _chk(R, "R", (S, R.shape[1])), _chk(H, "H", (S, 1, 1)))
        self.keep = (L, X, R, H)
        self.ws, self.nbytes = _mvf_workspace(S, rl, X.shape[1], R.shape[1], L.device)
        self.w = torch.empty(rl * X.shape[1] * R.shape[1], dtype=F64, device=L.device)

    def block(self, V, alphas, betas, stop, j_first, count, tol):
        """Iterations j_first .. j_first + count - 1: row j of V from rows j - 1 and j - 2."""
        S, rl, p, rr = self.dims
        if V.stride(1) != 1 or V.dtype != F64 or not V.is_cuda or V.shape[1] != rl * p * rr or V.shape[0] < j_first + count:
            raise _lib.TTKError("LanczosFused.block: expected a float64 device basis (rows, D) with contiguous rows")
        rc = _lib.lib().ttk_lanczos_fused_block_f64(*self.ptrs, V.data_ptr(), V.stride(0), alphas.data_ptr(), betas.data_ptr(),
                                                    stop.data_ptr(), int(j_first), int(count), float(tol), self.w.data_ptr(),
                                                    S, rl, p, rr, self.ws.data_ptr(), self.nbytes, _stream(V))
        _lib.check(rc, "ttk_lanczos_fused_block_f64")


class CGPlan:
    """The fused CG loop of one problem shape with buffers of its own, so that a block of iterations can be
    captured as a HIP graph ONCE and replayed for every core update of that shape (the operands of an update
    are copied in: 30 MB, 10 us at the headline shape).  Replay takes the per-launch host work and part of
    the gap between dependent launches out of the loop (measured 39.4 vs 42.9 us per iteration); the bits
    are the same as those of eager iterations."""
    BLOCK = 20
    _plans = {}

    @classmethod
    def get(cls, L, X, R, H):
        S, C, rl = L.shape
        key = (L.device, S, rl, X.shape[1], R.shape[1])
        plan = cls._plans.get(key)
        if plan is None:
            if len(cls._plans) > 2:
                cls._plans.clear()
            plan = cls._plans[key] = cls(L, X, R, H)
        return plan

    def __init__(self, L, X, R, H):
        S, C, rl = L.shape
        D = rl * X.shape[1] * R.shape[1]
        dev = L.device
        self.L, self.X, self.R, self.H = (torch.empty_like(t, memory_format=torch.contiguous_format) for t in (L, X, R, H))
        self.x = torch.zeros(D, dtype=F64, device=dev)
        self.r = torch.zeros(D, dtype=F64, device=dev)
        self.pbuf = torch.zeros(2 * D, dtype=F64, device=dev)
        self.state = torch.zeros(5, dtype=F64, device=dev)
        self.fused = CGFused(self.L, self.X, self.R, self.H)
        self.graph = None
        self.broken = False

    def load(self, L, X, R, H):
        self.L.copy_(L)
        self.X.copy_(X)
        self.R.copy_(R)
        self.H.copy_(H)

    def run(self, first, count):
        """Iterations first .. first+count-1 (first >= 2 and even for the captured blocks)."""
        it, end = first, first + count
        if self.graph is None and not self.broken and it % 2 == 0 and end - it >= self.BLOCK:
            # every kernel of the loop has run by now (iterations 0, 1, ...), so the block can be captured as it is;
            # capturing does not run it.  A failure to capture leaves the plan eager for good.
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for k in range(2, 2 + self.BLOCK):
                        self.fused.iterate(self.x, self.r, self.pbuf, self.state, k)
                self.graph = graph
            except Exception:       # noqa: BLE001
                self.broken = True
                self.graph = None
        while self.graph is not None and it % 2 == 0 and end - it >= self.BLOCK:
            self.graph.replay()
            it += self.BLOCK
        for k in range(it, end):
            self.fused.iterate(self.x, self.r, self.pbuf, self.state, k)


def solve_rhs(b, w=None, scale=None, ridge=0.0):
    """-(b/scale + ridge*w), flattened."""
    D = b.numel()
    out = torch.empty((D,), dtype=F64, device=b.device)
    rc = _lib.lib().ttk_solve_rhs_f64(_chk(b, "b"), None if w is None else _chk(w, "w"),
                                      None if scale is None else _chk(scale, "scale"), float(ridge), out.data_ptr(), D,
                                      _stream(b))
    _lib.check(rc, "ttk_solve_rhs_f64")
    return out


_chol_ws = {}


def chol_solve_gram(ws, b, w=None, scale=None, ridge=0.0):
    """x = -(A/scale + ridge I)^-1 (b/scale + ridge w) by the blocked MFMA Cholesky, with A taken
    from a compressed Gram workspace.  Returns (x (D,), info) with info a device int32 scalar:
    0, or the order of the first non-positive pivot."""
    _, _, rl, p, rr, sym = ws.dims
    D = rl * p * rr
    nbytes = int(_lib.lib().ttk_chol_ws_bytes(D))
    key = (ws.buf.device, D)
    cw = _chol_ws.get(key)
    if cw is None:
        _chol_ws.clear()
        cw = torch.empty((nbytes + 7) // 8, dtype=F64, device=ws.buf.device)
        _chol_ws[key] = cw
    st = _stream(ws.buf)
    x = torch.empty((D,), dtype=F64, device=ws.buf.device)
    info = torch.empty((), dtype=torch.int32, device=ws.buf.device)
    rc = _lib.lib().ttk_chol_solve_gram_f64(ws.buf.data_ptr(), rl, p, rr, sym, None if scale is None else _chk(scale, "scale"),
                                            float(ridge), _chk(b, "b"), None if w is None else _chk(w, "w"),
                                            cw.data_ptr(), x.data_ptr(), info.data_ptr(), st)
    _lib.check(rc, "ttk_chol_solve_gram_f64")
    return x, info


def chol_solve_gram_two_calls(ws, b, w=None, scale=None, ridge=0.0):
    """chol_solve_gram through ttk_chol_prepare_f64 + ttk_chol_solve_f64 (the matrix is expanded into the
    Cholesky storage first); what the one-call form is checked against."""
    _, _, rl, p, rr, sym = ws.dims
    D = rl * p * rr
    nbytes = int(_lib.lib().ttk_chol_ws_bytes(D))
    cw = torch.empty((nbytes + 7) // 8, dtype=F64, device=ws.buf.device)
    st = _stream(ws.buf)
    rc = _lib.lib().ttk_chol_prepare_f64(ws.buf.data_ptr(), rl, p, rr, sym, None if scale is None else _chk(scale, "scale"),
                                         float(ridge), _chk(b, "b"), None if w is None else _chk(w, "w"),
                                         cw.data_ptr(), st)
    _lib.check(rc, "ttk_chol_prepare_f64")
    x = torch.empty((D,), dtype=F64, device=ws.buf.device)
    info = torch.empty((), dtype=torch.int32, device=ws.buf.device)
    rc = _lib.lib().ttk_chol_solve_f64(cw.data_ptr(), D, x.data_ptr(), info.data_ptr(), st)
    _lib.check(rc, "ttk_chol_solve_f64")
    return x, info


def chol_mode(mode):
    """How the Cholesky solves factorise: 0 one launch per panel / trailing update, 1 the whole
    factorisation as one launch of tile tasks, -1 (default) the latter from 32 block columns on.
    Returns the previous mode; see ttk_chol_mode."""
    return int(_lib.lib().ttk_chol_mode(int(mode)))


QR_LDS_LIMIT = 150 * 1024


def qr(A):
    """Reduced QR (LAPACK signs) of a device matrix: one-workgroup HIP kernel when it fits in
    LDS, torch.linalg.qr (rocSOLVER) otherwise or for non-float64 / CPU input."""
    m, n = A.shape
    if (not A.is_cuda) or A.dtype != F64 or _lib.lib().ttk_qr_lds_bytes(m, n) > QR_LDS_LIMIT:
        return torch.linalg.qr(A, mode='reduced')
    A = A.contiguous()
    k = min(m, n)
    Q = torch.empty((m, k), dtype=F64, device=A.device)
    R = torch.empty((k, n), dtype=F64, device=A.device)
    rc = _lib.lib().ttk_qr_f64(A.data_ptr(), m, n, Q.data_ptr(), R.data_ptr(), _stream(A))
    _lib.check(rc, "ttk_qr_f64")
    return Q, R


def qr_right_supported(T):
    """Whether qr_right takes this (n, m) matrix: float64 on the device, contiguous, within the one-workgroup QR."""
    return T.is_cuda and T.dtype == F64 and T.ndim == 2 and T.is_contiguous() \
        and _lib.lib().ttk_qr_lds_bytes(T.shape[1], T.shape[0]) <= QR_LDS_LIMIT


def qr_right(T):
    """T (n, m) = Rm Qt with Qt (k, m) of orthonormal rows and Rm (n, k): the QR of the doubly flipped transposed
    matrix (tensor/network.py:685-689) in one launch, nothing copied around it."""
    n, m = T.shape
    k = min(m, n)
    Qt = torch.empty((k, m), dtype=F64, device=T.device)
    Rm = torch.empty((n, k), dtype=F64, device=T.device)
    rc = _lib.lib().ttk_qr_right_f64(T.data_ptr(), n, m, Qt.data_ptr(), Rm.data_ptr(), _stream(T))
    _lib.check(rc, "ttk_qr_right_f64")
    return Qt, Rm


def dot(x, y):
    out = torch.empty((), dtype=F64, device=x.device)
    rc = _lib.lib().ttk_dot_f64(_chk(x, "x"), _chk(y, "y"), out.data_ptr(), x.numel(), _stream(x))
    _lib.check(rc, "ttk_dot_f64")
    return out


def axpy(a, x, y, sign=1.0):
    """y += sign * a * x with `a` a 0-d device tensor (no host sync)."""
    rc = _lib.lib().ttk_axpy_f64(_chk(a, "a"), float(sign), _chk(x, "x"), _chk(y, "y"), x.numel(), _stream(x))
    _lib.check(rc, "ttk_axpy_f64")
    return y


def gram_timing(enable):
    """Arm / disarm HIP-event timing of the Gram kernel (True / 1) or of the fused matrix-free
    product kernel (2); see ttk_gram_timing."""
    _lib.check(_lib.lib().ttk_gram_timing(int(enable)), "ttk_gram_timing")


def gram_timing_read(max_n=2048):
    """Durations (ms) of the Gram kernel launches recorded since the last read."""
    import ctypes
    buf = (ctypes.c_double * max_n)()
    n = _lib.lib().ttk_gram_timing_read(ctypes.cast(buf, ctypes.c_void_p), max_n)
    if n < 0:
        _lib.check(n, "ttk_gram_timing_read")
    return [buf[i] for i in range(n)]


_cg_ws = {}


def cg_update(x, r, p, q, state):
    """One conjugate-gradient vector update in place (q = A p); see ttk_cg_update_f64 / ttk_cg_update_ws_f64 (long
    vectors are updated by many workgroups).  state = [<r,r>, atol, stopped, updates done] on the device."""
    D = x.numel()
    nbytes = int(_lib.lib().ttk_cg_update_ws_bytes(D))
    if nbytes > 0:
        ws = _cg_ws.get(x.device)
        if ws is None or ws.numel() * 8 < nbytes:
            ws = _cg_ws[x.device] = torch.empty((nbytes + 7) // 8, dtype=F64, device=x.device)
        rc = _lib.lib().ttk_cg_update_ws_f64(_chk(x, "x"), _chk(r, "r"), _chk(p, "p"), _chk(q, "q"), _chk(state, "state"), D,
                                             ws.data_ptr(), nbytes, _stream(x))
        _lib.check(rc, "ttk_cg_update_ws_f64")
        return
    rc = _lib.lib().ttk_cg_update_f64(_chk(x, "x"), _chk(r, "r"), _chk(p, "p"), _chk(q, "q"), _chk(state, "state"),
                                      D, _stream(x))
    _lib.check(rc, "ttk_cg_update_f64")


def minres_update(y, r1, r2, v, w, w2, x, state):
    """One MINRES vector update in place (y = A v on entry); see ttk_minres_update_f64 / ttk_minres_update_ws_f64."""
    D = x.numel()
    nbytes = int(_lib.lib().ttk_minres_update_ws_bytes(D))
    if nbytes > 0:
        ws = _cg_ws.get(x.device)
        if ws is None or ws.numel() * 8 < nbytes:
            ws = _cg_ws[x.device] = torch.empty((nbytes + 7) // 8, dtype=F64, device=x.device)
        rc = _lib.lib().ttk_minres_update_ws_f64(_chk(y, "y"), _chk(r1, "r1"), _chk(r2, "r2"), _chk(v, "v"), _chk(w, "w"),
                                                 _chk(w2, "w2"), _chk(x, "x"), _chk(state, "state"), D, ws.data_ptr(), nbytes,
                                                 _stream(x))
        _lib.check(rc, "ttk_minres_update_ws_f64")
        return
    rc = _lib.lib().ttk_minres_update_f64(_chk(y, "y"), _chk(r1, "r1"), _chk(r2, "r2"), _chk(v, "v"), _chk(w, "w"),
                                          _chk(w2, "w2"), _chk(x, "x"), _chk(state, "state"), D, _stream(x))
    _lib.check(rc, "ttk_minres_update_f64")


def lanczos_update(w, v_prev, v_cur, v_next, alphas, betas, j, tol, stop):
    """Iteration j >= 1 of the Lanczos recurrence on w = A v_cur (in place); v_next (a row of the basis) is written.
    alphas[j-1] and betas[j] are written on the device; v_prev may be None (first iteration).  stop (1,) on the
    device: set to j when betas[j] < tol, and once it is set every later call does nothing."""
    rc = _lib.lib().ttk_lanczos_update_f64(_chk(w, "w"), 0 if v_prev is None else _chk(v_prev, "v_prev"),
                                           _chk(v_cur, "v_cur"), _chk(v_next, "v_next"), _chk(alphas, "alphas"),
                                           _chk(betas, "betas"), int(j), w.numel(), float(tol), _chk(stop, "stop"), _stream(w))
    _lib.check(rc, "ttk_lanczos_update_f64")
    return v_next


def lanczos_finish(V, alphas, betas, stop, m_run, x0):
    """x0 + V_m y with T_m y = beta_1 e_1 (m = stop[0] if set, else m_run): the tridiagonal solve in one workgroup
    and one streaming pass over the rows of the basis V (m_cap + 1, D)."""
    D = V.shape[1]
    m_cap = V.shape[0] - 1
    nbytes = int(_lib.lib().ttk_lanczos_finish_ws_bytes(m_cap))
    ws = torch.empty((nbytes + 7) // 8, dtype=F64, device=V.device)
    x = torch.empty(D, dtype=F64, device=V.device)
    rc = _lib.lib().ttk_lanczos_finish_f64(_chk(V, "V"), V.stride(0), _chk(alphas, "alphas"), _chk(betas, "betas"),
                                           _chk(stop, "stop"), int(m_run), m_cap, _chk(x0, "x0"), x.data_ptr(), D, ws.data_ptr(),
                                           nbytes, _stream(V))
    _lib.check(rc, "ttk_lanczos_finish_f64")
    return x


def rows_dot(J, v):
    """u (S, C) = J v for explicit Jacobian rows J (S, C, D)."""
    S, C, D = J.shape
    out = torch.empty((S, C), dtype=F64, device=J.device)
    rc = _lib.lib().ttk_rows_dot_f64(_chk(J, "J"), _chk(v, "v", (D,)), out.data_ptr(), S * C, D, _stream(J))
    _lib.check(rc, "ttk_rows_dot_f64")
    return out


_rowsmv_ws = {}


def rows_matvec_supported(S, C, D):
    return int(_lib.lib().ttk_rows_matvec_ws_bytes(S, C, D)) > 0


def rows_matvec(J, H, v):
    """q (D,) = J^T H J v for explicit rows J (S, C, D) and per-sample Hessians H (S, C, C), one pass over J."""
    S, C, D = J.shape
    nbytes = int(_lib.lib().ttk_rows_matvec_ws_bytes(S, C, D))
    if nbytes <= 0:
        raise _lib.TTKError("ttk_rows_matvec_ws_bytes: extents not supported")
    key = (J.device, nbytes)
    ws = _rowsmv_ws.get(key)
    if ws is None:
        _rowsmv_ws.clear()
        ws = _rowsmv_ws[key] = torch.empty((nbytes + 7) // 8, dtype=F64, device=J.device)
    q = torch.empty((D,), dtype=F64, device=J.device)
    rc = _lib.lib().ttk_rows_matvec_f64(_chk(J, "J"), _chk(H, "H", (S, C, C)), _chk(v, "v", (D,)), q.data_ptr(), S, C, D,
                                        ws.data_ptr(), nbytes, _stream(J))
    _lib.check(rc, "ttk_rows_matvec_f64")
    return q


def rows_rhs(J, g):
    """b (D,) = J^T g for explicit rows J (S, C, D) and weights g (S, C), one pass over J."""
    S, C, D = J.shape
    nbytes = int(_lib.lib().ttk_rows_matvec_ws_bytes(S, C, D))
    if nbytes <= 0:
        raise _lib.TTKError("ttk_rows_matvec_ws_bytes: extents not supported")
    key = (J.device, nbytes)
    ws = _rowsmv_ws.get(key)
    if ws is None:
        _rowsmv_ws.clear()
        ws = _rowsmv_ws[key] = torch.empty((nbytes + 7) // 8, dtype=F64, device=J.device)
    b = torch.empty((D,), dtype=F64, device=J.device)
    rc = _lib.lib().ttk_rows_rhs_f64(_chk(J, "J"), _chk(g, "g", (S, C)), b.data_ptr(), S, C, D, ws.data_ptr(), nbytes,
                                     _stream(J))
    _lib.check(rc, "ttk_rows_rhs_f64")
    return b


def lhl(L, H, z):
    """GL (S, rl) = L_s^T H_s (L_s z_s) per sample: predict + Hessian product + g.L in one kernel."""
    S, C, rl = L.shape
    out = torch.empty((S, rl), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_lhl_f64(_chk(L, "L"), _chk(H, "H", (S, C, C)), _chk(z, "z", (S, rl)), out.data_ptr(), S, C, rl,
                                _stream(L))
    _lib.check(rc, "ttk_lhl_f64")
    return out


def gl(g, L):
    """GL (S,rl) = sum_c g[s,c] L[s,c,l]."""
    S, C, rl = L.shape
    out = torch.empty((S, rl), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_gl_f64(_chk(g, "g", (S, C)), _chk(L, "L"), out.data_ptr(), S, C, rl, _stream(L))
    _lib.check(rc, "ttk_gl_f64")
    return out


_rhsf_ws = {}


def rhs_feature(L, X, R, g, out=None, mm32=False, GL=None):
    """b (rl,p,rr) = sum_{s,c} g[s,c] L[s,c,l] X[s,p] R[s,r] for a site with many features
    (DENSE_MIN_P <= p <= DENSE_MAX_P), features first; with `out` given the result is added to it.
    `mm32`: float32 matrix arithmetic (float32 models).  `GL` = g.L (S, rl) if the caller has it already."""
    S, C, rl = L.shape
    p, rr = X.shape[1], R.shape[1]
    if GL is None:
        GL = gl(g, L)
    nbytes = int(_lib.lib().ttk_rhs_feature_ws_bytes(S, rl, p, rr))
    if nbytes < 0:
        raise _lib.TTKError("ttk_rhs_feature_ws_bytes: invalid extents")
    key = (L.device, nbytes)
    ws = _rhsf_ws.get(key)
    if ws is None:
        _rhsf_ws.clear()
        ws = torch.empty((nbytes + 7) // 8, dtype=F64, device=L.device)
        _rhsf_ws[key] = ws
    acc = out is not None
    b = out if acc else torch.empty((rl, p, rr), dtype=F64, device=L.device)
    fn = _lib.lib().ttk_rhs_feature_mm32 if mm32 else _lib.lib().ttk_rhs_feature_f64
    rc = fn(GL.data_ptr(), _chk(X, "X", (S, p)), _chk(R, "R", (S, rr)), _chk(b, "b", (rl, p, rr)), S, rl, p, rr, int(acc),
            ws.data_ptr(), nbytes, _stream(L))
    _lib.check(rc, "ttk_rhs_feature")
    return b


_filter_ws = {}


def filter_supported(L, Z):
    """True when the factored filter-node kernels take L (ca,S,C,rl) and Z (S,rl,n) (contiguous float64 device
    tensors whose per-sample factors fit the LDS of a wave)."""
    if not (L.is_cuda and L.dtype == F64 and Z.dtype == F64 and L.is_contiguous() and Z.is_contiguous()):
        return False
    ca, S, C, rl = L.shape
    return S > 0 and bool(_lib.lib().ttk_filter_ok(S, C, rl, Z.shape[2], ca))


def _filter_workspace(L, n):
    ca, S, C, rl = L.shape
    nbytes = int(_lib.lib().ttk_filter_ws_bytes(S, C, rl, n, ca))
    if nbytes <= 0:
        raise _lib.TTKError("ttk_filter_ws_bytes: extents not supported")
    key = (L.device, nbytes)
    ws = _filter_ws.get(key)
    if ws is None:
        _filter_ws.clear()
        ws = _filter_ws[key] = torch.empty((nbytes + 7) // 8, dtype=F64, device=L.device)
    return ws, nbytes


def filter_matvec(L, Z, H, v):
    """q (ca n,) = J^T H J v for the rows J[s,c,(a,k)] = sum_l L[a,s,c,l] Z[s,l,k], never formed."""
    ca, S, C, rl = L.shape
    n = Z.shape[2]
    ws, nbytes = _filter_workspace(L, n)
    q = torch.empty((ca * n,), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_filter_matvec_f64(_chk(L, "L"), _chk(Z, "Z", (S, rl, n)), _chk(H, "H", (S, C, C)), _chk(v, "v", (ca * n,)),
                                          q.data_ptr(), S, C, rl, n, ca, ws.data_ptr(), nbytes, _stream(L))
    _lib.check(rc, "ttk_filter_matvec_f64")
    return q


def filter_jv(L, Z, v):
    """u (S, C) = J v for the factored rows of filter_matvec."""
    ca, S, C, rl = L.shape
    n = Z.shape[2]
    u = torch.empty((S, C), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_filter_jv_f64(_chk(L, "L"), _chk(Z, "Z", (S, rl, n)), _chk(v, "v", (ca * n,)), u.data_ptr(), S, C, rl, n,
                                      ca, _stream(L))
    _lib.check(rc, "ttk_filter_jv_f64")
    return u


def filter_rhs(L, Z, g):
    """b (ca n,) = J^T g for the factored rows of filter_matvec."""
    ca, S, C, rl = L.shape
    n = Z.shape[2]
    ws, nbytes = _filter_workspace(L, n)
    b = torch.empty((ca * n,), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_filter_rhs_f64(_chk(L, "L"), _chk(Z, "Z", (S, rl, n)), _chk(g, "g", (S, C)), b.data_ptr(), S, C, rl, n, ca,
                                       ws.data_ptr(), nbytes, _stream(L))
    _lib.check(rc, "ttk_filter_rhs_f64")
    return b


_bond_ws = {}


def _bond_dims(L, G, R):
    ca, S, C, rl = L.shape
    cb, rr = R.shape[0], R.shape[2]
    p = G.shape[3]
    want = (ca, cb, S, p)
    return ca, cb, S, C, rl, p, rr, want


def bond_supported(L, G, R):
    """True when the conv-bond kernels take the stacked factors L (ca,S,C,rl), G (ca,cb,S,p), R (cb,S,rr) of a column
    between two convolution bonds: contiguous float64 device tensors, small bonds, few channels."""
    if not (L.is_cuda and L.dtype == F64 and G.dtype == F64 and R.dtype == F64):
        return False
    if not (L.is_contiguous() and G.is_contiguous() and R.is_contiguous()) or G.dim() != 4:
        return False
    ca, cb, S, C, rl, p, rr, want = _bond_dims(L, G, R)
    return S > 0 and tuple(G.shape) == want and bool(_lib.lib().ttk_bond_ok(S, C, rl, p, rr, ca, cb))


def bond_jv(L, G, R, v, out=None):
    """u (S,C) = sum_{a,b} L_a . (G_ab (x) R_b) . v for the ca cb Khatri-Rao terms of a column between two
    convolution bonds, in one launch; with `out` given the result is added to it."""
    ca, cb, S, C, rl, p, rr, want = _bond_dims(L, G, R)
    acc = out is not None
    u = out if acc else torch.empty((S, C), dtype=F64, device=L.device)
    nbytes = int(_lib.lib().ttk_bond_jv_ws_bytes(S, rl, ca))
    key = (L.device, "z", nbytes)
    ws = _bond_ws.get(key)
    if ws is None:
        for k in [k for k in _bond_ws if len(k) == 3]:
            del _bond_ws[k]
        ws = _bond_ws[key] = torch.empty((nbytes + 7) // 8, dtype=F64, device=L.device)
    rc = _lib.lib().ttk_bond_jv_f64(_chk(L, "L"), _chk(G, "G", want), _chk(R, "R", (cb, S, rr)),
                                    _chk(v, "v", (rl, p, rr)), _chk(u, "u", (S, C)), S, C, rl, p, rr, ca, cb,
                                    int(acc), ws.data_ptr(), nbytes, _stream(L))
    _lib.check(rc, "ttk_bond_jv")
    return u


def _bond_rhs_gl(GL, G, R, dims, out):
    ca, cb, S, C, rl, p, rr, want = dims
    nbytes = int(_lib.lib().ttk_bond_rhs_ws_bytes(S, rl, p, rr))
    key = (GL.device, nbytes)
    ws = _bond_ws.get(key)
    if ws is None:
        for k in [k for k in _bond_ws if len(k) == 2]:
            del _bond_ws[k]
        ws = torch.empty((nbytes + 7) // 8, dtype=F64, device=GL.device)
        _bond_ws[key] = ws
    acc = out is not None
    b = out if acc else torch.empty((rl, p, rr), dtype=F64, device=GL.device)
    rc = _lib.lib().ttk_bond_rhs_f64(GL.data_ptr(), _chk(G, "G", want), _chk(R, "R", (cb, S, rr)),
                                     _chk(b, "b", (rl, p, rr)), S, rl, p, rr, ca, cb, int(acc),
                                     ws.data_ptr(), nbytes, _stream(GL))
    _lib.check(rc, "ttk_bond_rhs")
    return b


def bond_rhs(L, G, R, g, out=None):
    """b (rl,p,rr) = sum_s sum_{a,b} (g . L_a)[s,l] G_ab[s,p] R_b[s,r]: J^T g of a column between two convolution
    bonds (one launch for g . L_a, one for the sum over samples and terms, one to add the slab partials)."""
    dims = _bond_dims(L, G, R)
    ca, cb, S, C, rl, p, rr, want = dims
    GL = torch.empty((ca, S, rl), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_bond_gl_f64(_chk(g, "g", (S, C)), _chk(L, "L"), GL.data_ptr(), S, C, rl, ca, _stream(L))
    _lib.check(rc, "ttk_bond_gl")
    return _bond_rhs_gl(GL, G, R, dims, out)


def bond_z_supported(G, R, core):
    """True when bond_z takes the stacked features G (ca,cb,S,p), right environments R (cb,S,rr) and a core (rl,p,rr)."""
    if not (G.is_cuda and G.dtype == F64 and R.dtype == F64 and core.dtype == F64 and G.dim() == 4 and R.dim() == 3):
        return False
    if not (G.is_contiguous() and R.is_contiguous()):
        return False
    ca, cb, S, p = G.shape
    rl, rr = core.shape[0], core.shape[2]
    return S > 0 and tuple(R.shape) == (cb, S, rr) and bool(_lib.lib().ttk_bond_ok(S, 1, rl, p, rr, ca, cb))


def bond_z(G, R, core):
    """z (ca,S,rl), z_a[s,l] = sum_b sum_{p,r} core[l,p,r] G_ab[s,p] R_b[s,r]: the right-environment step through a
    column between two convolution bonds for all left channels in one launch (also the first half of bond_matvec)."""
    ca, cb, S, p = G.shape
    rl, rr = core.shape[0], core.shape[2]
    z = torch.empty((ca, S, rl), dtype=F64, device=G.device)
    rc = _lib.lib().ttk_bond_z_f64(_chk(G, "G"), _chk(R, "R", (cb, S, rr)), _chk(core.contiguous(), "core", (rl, p, rr)),
                                   z.data_ptr(), S, rl, p, rr, ca, cb, _stream(G))
    _lib.check(rc, "ttk_bond_z")
    return z


def bond_matvec(L, G, R, H, v):
    """q (rl,p,rr) = J^T H J v of a column between two convolution bonds: the first half down to the left bond (z), the
    left environments and the Hessian in one kernel (g . L of the Hessian-weighted prediction), the second half."""
    dims = _bond_dims(L, G, R)
    ca, cb, S, C, rl, p, rr, want = dims
    z = torch.empty((ca, S, rl), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_bond_z_f64(_chk(G, "G", want), _chk(R, "R", (cb, S, rr)), _chk(v, "v", (rl, p, rr)), z.data_ptr(),
                                   S, rl, p, rr, ca, cb, _stream(L))
    _lib.check(rc, "ttk_bond_z")
    GL = torch.empty((ca, S, rl), dtype=F64, device=L.device)
    rc = _lib.lib().ttk_bond_lhl_f64(_chk(L, "L"), _chk(H, "H", (S, C, C)), z.data_ptr(), GL.data_ptr(), S, C, rl, ca,
                                     _stream(L))
    _lib.check(rc, "ttk_bond_lhl")
    return _bond_rhs_gl(GL, G, R, dims, None)


def conv_feature(X, Fc):
    """G (ca, cb, S, P) = X (S, P, px) . Fc (ca, px, cb); X float64 or float32 (read as it is)."""
    S, P, px = X.shape
    ca, _, cb = Fc.shape
    G = torch.empty((ca, cb, S, P), dtype=F64, device=X.device)
    if S == 0:
        return G
    if not X.is_cuda or not X.is_contiguous() or X.dtype not in (F64, torch.float32):
        raise _lib.TTKError("conv_feature: expected a contiguous float64 / float32 CUDA/HIP tensor")
    Fm = Fc.permute(1, 0, 2).reshape(px, ca * cb).contiguous()
    fn = _lib.lib().ttk_conv_feature_f64 if X.dtype == F64 else _lib.lib().ttk_conv_feature_f32
    _lib.check(fn(X.data_ptr(), _chk(Fm, "F"), G.data_ptr(), S * P, px, ca * cb, _stream(X)), "ttk_conv_feature")
    return G


def gemm_rows(A, B, N, bmode=0, rl=0, rr=0, out=None, mm32=False):
    """T (M, N) (+)= A (M, K) B; bmode 0: B (K, N); 1: B = core (rl, K, rr), columns (l, r); 2: B (N, K) transposed.
    `mm32`: float32 matrix arithmetic (float32 models)."""
    M, K = A.shape
    acc = out is not None
    T = out if acc else torch.empty((M, N), dtype=F64, device=A.device)
    if M == 0:
        return T
    fn = _lib.lib().ttk_gemm_rows_mm32 if mm32 else _lib.lib().ttk_gemm_rows_f64
    rc = fn(_chk(A, "A"), _chk(B, "B"), _chk(T, "T", (M, N)), M, K, int(N), int(bmode), int(rl), int(rr), int(acc), _stream(A))
    _lib.check(rc, "ttk_gemm_rows")
    return T


def lcontract(L, T, out=None):
    """out (S, C, n) (+)= sum_l L[s,c,l] T[s,l,:];  L (S, C, rl), T (S, rl, n)."""
    S, C, rl = L.shape
    n = T.shape[2]
    acc = out is not None
    o = out if acc else torch.empty((S, C, n), dtype=F64, device=L.device)
    if S == 0:
        return o
    rc = _lib.lib().ttk_lcontract_f64(_chk(L, "L"), _chk(T, "T", (S, rl, n)), _chk(o, "out", (S, C, n)), S, C, rl, n,
                                      int(acc), _stream(L))
    _lib.check(rc, "ttk_lcontract_f64")
    return o


def lcontract_sum(L, T):
    """out (nb, S, C, n) = sum_a sum_l L[a,s,c,l] T[a,b,s,l,:]  for L (ca, S, C, rl), T (ca, nb, S, rl, n), or None when
    the left environments of a sample do not fit the kernel's LDS."""
    ca, S, C, rl = L.shape
    nb, n = T.shape[1], T.shape[4]
    if ca * C * rl * 8 > 48 * 1024 or nb > 65535:
        return None
    out = torch.empty((nb, S, C, n), dtype=F64, device=L.device)
    if S == 0:
        return out
    rc = _lib.lib().ttk_lcontract_sum_f64(_chk(L, "L"), _chk(T, "T", (ca, nb, S, rl, n)), out.data_ptr(), S, C, rl, n, ca, nb,
                                          _stream(L))
    _lib.check(rc, "ttk_lcontract_sum_f64")
    return out


def lcontract_into(L, T, J, a):
    """J[s, c, a, :] = sum_l L[s,c,l] T[s,l,:]  for J (S, C, ca, n) contiguous: the rows of one left channel of a
    conv filter node's Jacobian, written in place."""
    S, C, rl = L.shape
    n = T.shape[2]
    ca = J.shape[2]
    if not (J.is_contiguous() and J.dtype == F64 and J.is_cuda and tuple(J.shape) == (S, C, ca, n) and 0 <= a < ca):
        raise _lib.TTKError("lcontract_into: expected a contiguous float64 device tensor (S, C, ca, n)")
    if S == 0:
        return J
    rc = _lib.lib().ttk_lcontract_strided_f64(_chk(L, "L"), _chk(T, "T", (S, rl, n)), J.data_ptr() + 8 * a * n, S, C, rl, n,
                                              C * ca * n, ca * n, 0, _stream(L))
    _lib.check(rc, "ttk_lcontract_strided_f64")
    return J


def env_step_left_dense(E, X, core, out=None, mm32=False):
    """env_step_left for a site with many features: T = X . core (one GEMM shared by the C outputs),
    then the left environment's output index."""
    S, C, rl = E.shape
    p, rr = core.shape[1], core.shape[2]
    T = gemm_rows(X, core, rl * rr, bmode=1, rl=rl, rr=rr, mm32=mm32)
    return lcontract(E, T.reshape(S, rl, rr), out=out)


def patch_contract(M, X, J, a=0, b=0):
    """J[s, c, a, :, b] = sum_p M[s,c,p] X[s,p,:]  for J (S, C, ca, px, cb) contiguous; X float64 / float32."""
    S, C, P = M.shape
    px = X.shape[2]
    if not (J.is_contiguous() and J.dtype == F64 and J.is_cuda and X.is_contiguous() and X.dtype in (F64, torch.float32)):
        raise _lib.TTKError("patch_contract: expected contiguous device tensors (J float64, X float64 / float32)")
    ca, cb = J.shape[2], J.shape[4]
    fn = _lib.lib().ttk_patch_contract_f64 if X.dtype == F64 else _lib.lib().ttk_patch_contract_f32
    ptr = J.data_ptr() + 8 * (a * px * cb + b)
    rc = fn(_chk(M, "M"), X.data_ptr(), ptr, S, C, P, px, C * ca * px * cb, ca * px * cb, cb, 0, _stream(M))
    _lib.check(rc, "ttk_patch_contract")
    return J


def patch_contract_multi(M, X):
    """Z (S, C, px, nb) = sum_p M[b,s,c,p] X[s,p,:] for M (nb, S, C, P): the nb channels in one pass over X, or None
    when a sample's blocks do not fit the kernel's LDS (the caller then goes channel by channel)."""
    nb, S, C, P = M.shape
    px = X.shape[2]
    if not bool(_lib.lib().ttk_patch_contract_multi_ok(C, P, px, nb)):
        return None
    if not (X.is_contiguous() and X.is_cuda and X.dtype in (F64, torch.float32)):
        raise _lib.TTKError("patch_contract_multi: expected a contiguous float64 / float32 device tensor X")
    Z = torch.empty((S, C, px, nb), dtype=F64, device=M.device)
    if S == 0:
        return Z
    fn = _lib.lib().ttk_patch_contract_multi_f64 if X.dtype == F64 else _lib.lib().ttk_patch_contract_multi_f32
    _lib.check(fn(_chk(M, "M"), X.data_ptr(), Z.data_ptr(), S, C, P, px, nb, _stream(M)), "ttk_patch_contract_multi")
    return Z


def gemm_tn(A, B, out=None):
    """(D1, D2) (+)= A^T B for A (n, D1), B (n, D2)."""
    n, D1 = A.shape
    D2 = B.shape[1]
    acc = out is not None
    Cm = out if acc else torch.empty((D1, D2), dtype=F64, device=A.device)
    rc = _lib.lib().ttk_gemm_tn_f64(_chk(A, "A"), _chk(B, "B", (n, D2)), _chk(Cm, "C", (D1, D2)), n, D1, D2, int(acc),
                                    _stream(A))
    _lib.check(rc, "ttk_gemm_tn_f64")
    return Cm


def chol_solve_dense(A, b, w=None, ridge=0.0):
    """x = -(A/scale + ridge I)^-1 (b/scale + ridge w), scale = mean |diag A| (1 if 0), by the blocked MFMA
    Cholesky from a dense (D, D) matrix.  Returns (x (D,), info) like chol_solve_gram."""
    D = A.shape[0]
    nbytes = int(_lib.lib().ttk_chol_ws_bytes(D))
    key = (A.device, D)
    cw = _chol_ws.get(key)
    if cw is None:
        _chol_ws.clear()
        cw = torch.empty((nbytes + 7) // 8, dtype=F64, device=A.device)
        _chol_ws[key] = cw
    st = _stream(A)
    scale = torch.empty((), dtype=F64, device=A.device)
    _lib.check(_lib.lib().ttk_dense_scale_f64(_chk(A, "A", (D, D)), D, scale.data_ptr(), st), "ttk_dense_scale_f64")
    rc = _lib.lib().ttk_chol_prepare_dense_f64(A.data_ptr(), D, scale.data_ptr(), float(ridge), _chk(b, "b"),
                                               None if w is None else _chk(w, "w"), cw.data_ptr(), st)
    _lib.check(rc, "ttk_chol_prepare_dense_f64")
    x = torch.empty((D,), dtype=F64, device=A.device)
    info = torch.empty((), dtype=torch.int32, device=A.device)
    _lib.check(_lib.lib().ttk_chol_solve_f64(cw.data_ptr(), D, x.data_ptr(), info.data_ptr(), st), "ttk_chol_solve_f64")
    return x, info


def feature_map(X, kind, p):
    """Per-site feature tensors of a raw input X (S, N) -> (N, S, p) (site i: out[i]); kind "poly": powers
    0 .. p-1, "sincos" (p = 2): cos / sin of pi x / 2."""
    S, N = X.shape
    out = torch.empty((N, S, p), dtype=F64, device=X.device)
    if S == 0:
        return out
    rc = _lib.lib().ttk_feature_map_f64(_chk(X, "X"), S, N, {"poly": 0, "sincos": 1}[kind], int(p), out.data_ptr(), _stream(X))
    _lib.check(rc, "ttk_feature_map_f64")
    return out


def patchify(images, kernel_size, stride, dtype=None):
    """(S, Cin, H, W) images -> (S, patches + 1, Cin k k + 1) patch rows with the zero row / column and the corner
    one the image scripts append; float32 or float64 in, the same or float64 out."""
    if not (isinstance(images, torch.Tensor) and images.is_cuda and images.is_contiguous() and images.ndim == 4
            and images.dtype in (torch.float32, F64)):
        raise _lib.TTKError("patchify: expected a contiguous float32 / float64 CUDA/HIP tensor (S, Cin, H, W)")
    dtype = images.dtype if dtype is None else dtype
    types = {(torch.float32, torch.float32): 0, (F64, F64): 1, (torch.float32, F64): 2}.get((images.dtype, dtype))
    if types is None:
        raise _lib.TTKError(f"patchify: {images.dtype} -> {dtype} is not supported")
    S, Cin, H, W = images.shape
    k, st = int(kernel_size), int(stride)
    if H < k or W < k:
        raise _lib.TTKError("patchify: kernel larger than the image")
    OH, OW = (H - k) // st + 1, (W - k) // st + 1
    out = torch.empty((S, OH * OW + 1, Cin * k * k + 1), dtype=dtype, device=images.device)
    if S == 0:
        return out
    rc = _lib.lib().ttk_patchify(images.data_ptr(), S, Cin, H, W, k, st, types, out.data_ptr(), _stream(images))
    _lib.check(rc, "ttk_patchify")
    return out
