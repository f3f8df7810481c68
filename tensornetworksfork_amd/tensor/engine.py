"""Chain engine: the arithmetic behind TensorNetwork for tensor-train shaped graphs.

Three column shapes are recognised (one column = one site of the train):

  plain     A_k (rl,[c],p,rr) --p-- X_k (S,p)                                   tensor/layers.py:194-221
  conv      A_k (rl,[c],patches,rr) --patches-- X_k (S,patches,pixels) --pixels-- F_k ([a],pixels,[b])
                                                                                tensor/layers.py:791-890
  operator  A_k (rl,[c],u,rr) --u-- F_k ([a],u,d,[b]) --d-- X_k (S,d)           tensor/layers.py:308-343,425-477
            (a linear map per site, or one site of a matrix-product operator such as the cumulative sum)

A conv / operator column is a plain column whose feature vector depends on the sample through
the feature node F_k: G_k[a,b][s,:] = X_k[s] . F_k[a,.,b]; with a bond (a,b) between the
feature nodes the environments carry the composite bond (r,b) and every contraction of the
plain train becomes a sum over (a,b) of the same Khatri-Rao kernels (accumulating launches).
A plain column is the special case a = b = 1 with G = X.

Per-sample environments of a whole data set stay resident on the device (`_Data`) and are
dropped only where a node of an earlier / later column changed.
"""
import torch

from .. import ops

F64 = torch.float64


class _Column:
    __slots__ = ("A", "F", "kind", "xin", "order", "nleft", "out_labels", "sizes", "rl", "p", "rr",
                 "f_order", "f_nleft", "f_sizes", "ca", "cb", "inner")


class _Data:
    """One data set resident on the device with its cached features and environments."""

    def __init__(self, raw, xs, N):
        self.raw = raw
        self.xs = xs
        self.S = xs[0].shape[0]
        self.S_total = self.S
        self.sharded = False
        self.L = [None] * N
        self.R = [None] * N
        self.G = [None] * N
        self.hot = 0               # column that changed last
        self.pairs_done = False
        self.seenA = [None] * N    # (tensor, version) of A_k / F_k when the caches were last valid
        self.seenF = [None] * N

    def matches(self, raw):
        return len(raw) == len(self.raw) and all(a is b for a, b in zip(raw, self.raw))


def _changed(seen, node):
    if node is None:
        return False
    t = node.tensor
    return seen is None or seen[0] is not t or seen[1] != t._version


PAIR_CLASS_MAX_P = 10      # the library honours pair classes up to p(p+1)/2 <= 64 pairs
PAIR_CLASS_TOL = 16.0      # products that agree to this many ulps (of the column's largest entry) are one class


def pair_classes(X):
    """Classes of feature pairs (i<=j, upper-triangle row-major) whose products X_i X_j coincide on
    every sample, as the list rep[b] = first pair of b's class; None when all pairs differ or p is
    above the limit.  A polynomial basis X_i = x^i gives the p(p+1)/2 pairs 2p-1 classes (i+j)."""
    return pair_classes_many([X])[0]


def pair_classes_many(Xs):
    """pair_classes for several feature matrices of one shape at once (two host round trips in all).
    Candidates come from a strided subset of the samples and are then checked on all of them."""
    S, p = Xs[0].shape
    if p < 2 or p > PAIR_CLASS_MAX_P or S == 0:
        return [None] * len(Xs)
    dev = Xs[0].device
    iu = torch.triu_indices(p, p, device=dev)
    Xt = torch.stack([X.T for X in Xs])                        # (n, p, S): reductions run along contiguous rows
    XX = Xt[:, iu[0]] * Xt[:, iu[1]]                           # (n, NX, S), same pair order as the kernels
    n, NX = XX.shape[0], XX.shape[1]
    tol = PAIR_CLASS_TOL * torch.finfo(XX.dtype).eps
    sub = XX[:, :, ::max(1, S // 1024)][:, :, :2048]
    big = sub.abs().amax(dim=2)
    diff = (sub.unsqueeze(2) - sub.unsqueeze(1)).abs().amax(dim=3)
    same = (diff <= tol * torch.maximum(big.unsqueeze(1), big.unsqueeze(2))).cpu()
    reps = []
    for i in range(n):
        rep = list(range(NX))
        for b in range(NX):
            for a in range(b):
                if rep[a] == a and bool(same[i, a, b]):
                    rep[b] = a
                    break
        reps.append(rep)
    if all(rep == list(range(NX)) for rep in reps):
        return [None] * n
    rt = torch.tensor(reps, device=dev)                         # (n, NX)
    ref = torch.gather(XX, 1, rt.unsqueeze(2).expand(n, NX, S))
    ok = ((XX - ref).abs().amax(dim=2) <= tol * XX.abs().amax(dim=2)).cpu()
    out = []
    for i in range(n):
        rep = [reps[i][b] if bool(ok[i, b]) else b for b in range(NX)]
        out.append(None if all(rep[b] == b for b in range(NX)) else rep)
    return out


DENSE_ROWS_KERNEL_MAX_D = 128   # explicit Jacobian rows up to this width go through the Khatri-Rao kernels
TWO_STAGE_MIN_P = 16    # from this many features on, J v and the environment steps contract the features first
TWO_STAGE_MAX_P = 192   # ... up to what the row GEMM keeps in LDS; beyond, the streaming Khatri-Rao kernels


_ONES = {}


def _ones_col(S, device):
    """(S,1) column of ones (a feature factor that contributes nothing), cached per size and device."""
    key = (S, str(device))
    if key not in _ONES:
        if len(_ONES) > 8:
            _ONES.clear()
        _ONES[key] = torch.ones((S, 1), dtype=F64, device=device)
    return _ONES[key]


def _tkey(t):
    """The same memory seen through two view objects is one operand."""
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()))


def _rhs_features_first(rl, p, rr):
    """Whether J^T u of a Khatri-Rao term goes to the features-first kernel.  The three-factor kernel streams
    every slab of samples once per 4 x 96 output columns ("passes"); the features-first one stages a chunk once
    for all its columns but pays a fixed cost (the GL product, the slab sum) that small sites do not earn back.
    Measured (S = 50 000 .. 54 000, C = 9 .. 10; features-first vs three-factor, us): 48 x 65 x 48 (33 passes)
    407 / 1558, 24 x 50 x 24 (7) 149 / 249, 32 x 16 x 32 (3) 105 / 98, 48 x 65 x 1 (1) 180 / 90."""
    if not TWO_STAGE_MIN_P <= p <= TWO_STAGE_MAX_P or rr > 128 or p + rr > 280:
        return False
    if rl <= 16:
        # one row tile: a wave of the three-factor kernel owns up to 12 column tiles (48 per pass); measured
        # (features-first / three-factor, us): 10 x 50 x 10 133 / 73, 16 x 65 x 16 152 / 160, 1 x 50 x 24 124 / 108,
        # 1 x 65 x 48 139 / 306
        return (p * rr + 15) // 16 > 96
    blocks = (((rl + 15) // 16 + 1) // 2) * (((p * rr + 15) // 16 + 2) // 3)
    passes = (blocks + 3) // 4
    return passes >= 6


BOND_LEFT_MAX_BYTES = 2 << 30     # largest stacked intermediate of the one-launch left step through a convolution bond


class LocalProblem:
    """The normal equations of one train node for a bound data set.

    Either a list of Khatri-Rao terms (L (S,C,rl), X (S,p), R (S,rr)) whose sum is the
    Jacobian, or explicit Jacobian rows J (S,C,D).  `core` is the node's tensor in canonical
    (flattened) order; `back` maps a canonical vector to the node's own layout."""

    def __init__(self, S, C, shape, core, back, terms=None, J=None, fz=None):
        self.S, self.C = S, C
        self.shape = tuple(shape)       # canonical shape, prod = D
        self.D = 1
        for n in self.shape:
            self.D *= n
        self.core = core
        self.back = back
        self.terms = terms
        self._J = J
        # (L (ca,S,C,rl), Z (S,rl,n)): rows J[s,c,(a,k)] = sum_l L_a[s,c,l] Z[s,l,k] kept as factors (a conv filter
        # node between convolution bonds); the rows themselves are formed only when something asks for them
        self.fz = fz
        self._Jt = None
        self.mm32 = False               # float32 matrix arithmetic on the many-feature route (set by the engine)
        self.bond = None                # stacked (L, G, R) of a column between convolution bonds the bond kernels take

    @property
    def J(self):
        if self._J is None and self.fz is not None:
            L, Z = self.fz
            ca, S, C, _ = L.shape
            J = torch.empty((S, C, ca, Z.shape[2]), dtype=F64, device=Z.device)
            for a in range(ca):
                ops.lcontract_into(L[a], Z, J, a)
            self._J = J.reshape(S, C, -1)
        return self._J

    # -- first order
    def _jt(self):
        """(D, S*C) copy of the explicit rows: sums over samples then run along contiguous memory."""
        if self._Jt is None:
            self._Jt = self.J.reshape(self.S * self.C, self.D).T.contiguous()
        return self._Jt

    def rhs(self, g):
        """b = J^T g, canonical shape."""
        if self.fz is not None:
            return ops.filter_rhs(*self.fz, g.reshape(self.S, self.C).contiguous()).reshape(self.shape)
        if self.J is not None:
            if self.D <= DENSE_ROWS_KERNEL_MAX_D:
                # explicit rows are a Khatri-Rao term with the rows as left factor and no other factor
                one = _ones_col(self.S, g.device)
                return ops.rhs(self.J.contiguous(), one, one, g.reshape(self.S, self.C).contiguous()).reshape(self.shape)
            if ops.rows_matvec_supported(self.S, self.C, self.D):
                return ops.rows_rhs(self.J.contiguous(), g.reshape(self.S, self.C).contiguous()).reshape(self.shape)
            # wider rows still: element-wise product + row sums (the library's tall-skinny matrix-vector kernels are slow)
            return (self._jt() * g.reshape(1, self.S * self.C)).sum(dim=1).reshape(self.shape)
        if self.bond is not None:
            return ops.bond_rhs(*self.bond, g.reshape(self.S, self.C).contiguous())
        out = None
        # terms of a convolution bond share their left environments (L_a for every b): g . L once per L
        shared = {}
        if self.C > 1 and len(self.terms) > 1:
            for (L, _, _) in self.terms:
                shared[_tkey(L)] = shared.get(_tkey(L), 0) + 1
        gls = {}

        def gl_of(L):
            if _tkey(L) not in gls:
                gls[_tkey(L)] = ops.gl(g, L)
            return gls[_tkey(L)]
        for (L, X, R) in self.terms:
            if _rhs_features_first(L.shape[2], X.shape[1], R.shape[1]):
                # many features and a wide output: the GEMM over the samples with the features as rows
                GLs = gl_of(L) if shared.get(_tkey(L), 0) > 1 else None
                out = ops.rhs_feature(L, X, R, g, mm32=self.mm32, GL=GLs) if out is None else \
                    ops.rhs_feature(L, X, R, g, out=out, mm32=self.mm32, GL=GLs)
            elif not ops.rhs_supported(L.shape[0], L.shape[1], L.shape[2], X.shape[1], R.shape[1]):
                # a sample row too wide for the right-hand-side kernel's LDS chunk: features last,
                # b[l,p,r] = sum_s X[s,p] (GL[s,l] R[s,r]),  GL = g . L, as one transposed product over the samples
                S, rl, rr = L.shape[0], L.shape[2], R.shape[1]
                GL = ops.gl(g, L)                                                        # (S, rl)
                W = ops.kr_rows(GL.unsqueeze(1), _ones_col(S, g.device), R).reshape(S, rl * rr)   # GL[s,l] R[s,r]
                bt = ops.gemm_tn(X, W).reshape(X.shape[1], rl, rr).permute(1, 0, 2)
                out = bt.contiguous() if out is None else out.add_(bt)
            elif shared.get(_tkey(L), 0) > 1:
                GL1 = gl_of(L).unsqueeze(1)                                              # (S, 1, rl): one "output"
                one = _ones_col(L.shape[0], g.device)
                out = ops.rhs(GL1, X, R, one) if out is None else ops.rhs(GL1, X, R, one, out=out)
            else:
                out = ops.rhs(L, X, R, g) if out is None else ops.rhs(L, X, R, g, out=out)
        return out

    def jv(self, v):
        """u[s,c] = J v."""
        if self.fz is not None:
            return ops.filter_jv(*self.fz, v.reshape(self.D).contiguous())
        if self.J is not None:
            return ops.rows_dot(self.J.contiguous(), v.reshape(self.D).contiguous())
        v = v.reshape(self.shape)
        if self.bond is not None:
            return ops.bond_jv(*self.bond, v.contiguous())
        u = None
        tmp = {}
        zs = {}
        for (L, X, R) in self.terms:
            if TWO_STAGE_MIN_P <= X.shape[1] <= TWO_STAGE_MAX_P:
                # many features: contract them first, T[s,(l,r)] = X[s,:] . v[l,:,r] (one GEMM shared by the
                # C outputs of a sample, never stored), then the two environments: C*p/(C+p) times fewer FLOPs;
                # terms that share the left environment (a convolution bond) are summed before it is applied
                if _tkey(L) in zs:
                    ops.env_step_right(R, X, v.contiguous(), out=zs[_tkey(L)][1], mm32=self.mm32)
                else:
                    zs[_tkey(L)] = (L, ops.env_step_right(R, X, v.contiguous(), mm32=self.mm32))   # (S, rl)
                continue
            key = id(R)                  # terms that share R share one environment-step buffer
            if key in tmp:
                ops.env_step_left(L, X, v, out=tmp[key][0])
            else:
                tmp[key] = (ops.env_step_left(L, X, v), R)
        for L, z in zs.values():
            u = ops.predict(L, z) if u is None else ops.predict(L, z, out=u)             # (S, C)
        for t, R in tmp.values():
            u = ops.predict(t, R) if u is None else ops.predict(t, R, out=u)
        return u

    def matvec_fn(self, H):
        """v -> (J^T H J) v, canonical shape, never forming the Gram matrix."""
        if self.terms is not None and len(self.terms) == 1 and not TWO_STAGE_MIN_P <= self.terms[0][1].shape[1] <= TWO_STAGE_MAX_P:
            L, X, R = self.terms[0]
            if ops.rhs_supported(L.shape[0], L.shape[1], L.shape[2], X.shape[1], R.shape[1]):
                return lambda v: ops.matvec(L, X, R, H, v.reshape(self.shape))

        if self.bond is not None and H.shape[1] == self.C:
            Lb, Gb, Rb = self.bond
            Hb = H.contiguous()
            return lambda v: ops.bond_matvec(Lb, Gb, Rb, Hb, v.reshape(self.shape).contiguous())
        if self.fz is not None and H.shape[1] == self.C:
            Lf, Zf = self.fz
            Hc = H.contiguous()
            return lambda v: ops.filter_matvec(Lf, Zf, Hc, v.reshape(self.D).contiguous()).reshape(self.shape)
        if self.J is not None and H.shape[1] == self.C and ops.rows_matvec_supported(self.S, self.C, self.D):
            # explicit rows of a narrow node (conv filter): the whole operator in one pass over the rows
            J = self.J.contiguous()
            return lambda v: ops.rows_matvec(J, H, v.reshape(self.D).contiguous()).reshape(self.shape)
        if self.terms is not None and len(self.terms) == 1 and H.shape[1] == self.C:
            L, X, R = self.terms[0]
            first = _rhs_features_first(L.shape[2], X.shape[1], R.shape[1])
            if (first or ops.rhs_supported(L.shape[0], 1, L.shape[2], X.shape[1], R.shape[1])) and \
                    L.shape[2] <= 128 and L.shape[1] <= 32 and L.shape[1] * L.shape[2] <= 4096:
                # one term with many features: features first (J v), then prediction + Hessian + g.L in one kernel,
                # then the right-hand side of the one "output" GL on the route its shape takes
                one = _ones_col(L.shape[0], L.device)

                def mv1(v):
                    z = ops.env_step_right(R, X, v.reshape(self.shape).contiguous(), mm32=self.mm32)      # (S, rl)
                    GL = ops.lhl(L, H, z)
                    if first:
                        return ops.rhs_feature(L, X, R, None, mm32=self.mm32, GL=GL)
                    return ops.rhs(GL.unsqueeze(1), X, R, one)
                return mv1

        def mv(v):
            u = self.jv(v)
            hu = ops.hess_left(H, u.contiguous().unsqueeze(-1)).squeeze(-1)
            return self.rhs(hu)
        return mv

    def cg_fused(self, H):
        """ops.CGFused for this problem's operator when the one-pass product kernel takes it (one
        Khatri-Rao term, one output, square bonds of 16 / 32, few features), else None."""
        if self.terms is None or len(self.terms) != 1:
            return None
        L, X, R = self.terms[0]
        if not (L.is_cuda and ops.matvec_fused_supported(L.shape[0], L.shape[1], L.shape[2], X.shape[1], R.shape[1])):
            return None
        return ops.CGFused(L.contiguous(), X.contiguous(), R.contiguous(), H)

    def lanczos_fused(self, H):
        """ops.LanczosFused for this problem's operator under the same conditions as cg_fused, else None."""
        if self.terms is None or len(self.terms) != 1:
            return None
        L, X, R = self.terms[0]
        if not (L.is_cuda and L.dtype == F64 and tuple(H.shape) == (L.shape[0], 1, 1) and
                ops.matvec_fused_supported(L.shape[0], L.shape[1], L.shape[2], X.shape[1], R.shape[1])):
            return None
        return ops.LanczosFused(L.contiguous(), X.contiguous(), R.contiguous(), H.contiguous())

    def slice(self, lo, hi):
        """The same problem restricted to samples [lo, hi) (first-axis views, no copies)."""
        if self.J is not None:
            return LocalProblem(hi - lo, self.C, self.shape, self.core, self.back, J=self.J[lo:hi])
        terms = [(L[lo:hi], X[lo:hi], R[lo:hi]) for (L, X, R) in self.terms]
        part = LocalProblem(hi - lo, self.C, self.shape, self.core, self.back, terms=terms)
        part.mm32 = self.mm32
        return part

    # -- Gram
    def single_term(self):
        return self.terms is not None and len(self.terms) == 1

    def compressed_gram(self, sym=True):
        """True when the symmetry-compressed Gram kernel (and the Cholesky fed from it) takes this
        problem: one Khatri-Rao term whose packed sample row fits the kernel's LDS chunk (a site with
        many features, e.g. 50 patches = 1275 feature pairs, does not and goes the dense way)."""
        if not self.single_term():
            return False
        L, X, R = self.terms[0]
        return ops.gram_supported(self.S, self.C, L.shape[2], X.shape[1], R.shape[1], sym)

    def rows(self, lo, hi):
        """Explicit Jacobian rows of samples [lo, hi): (n, C, D)."""
        if self.J is not None:
            return self.J[lo:hi]
        out = None
        for (L, X, R) in self.terms:
            a = (L[lo:hi].contiguous(), X[lo:hi].contiguous(), R[lo:hi].contiguous())
            out = ops.kr_rows(*a) if out is None else ops.kr_rows(*a, out=out)
        return out

    def dense_gram(self, H, chunk_bytes=1 << 30, out=None):
        """A (D,D) = sum_s J^T H J from explicit rows in sample chunks (ttk_gemm_tn_f64); added to `out`
        (a zeroed (D,D) buffer of the caller's) when given."""
        D, C = self.D, self.C
        A = torch.zeros((D, D), dtype=F64, device=H.device) if out is None else out
        step = max(1, int(chunk_bytes // (8 * C * D)))
        for lo in range(0, self.S, step):
            hi = min(self.S, lo + step)
            J = self.rows(lo, hi)
            HJ = ops.hess_left(H[lo:hi].contiguous(), J.contiguous())     # (n,C,D): sum_c H[s,c,d] J[s,c,:]
            ops.gemm_tn(J.reshape(-1, D).contiguous(), HJ.reshape(-1, D), out=A)
        return A


_MISSING = object()


class EngineBase:
    """Shared plumbing of the engines behind TensorNetwork: data-set cache and solver workspaces."""

    def __init__(self, net):
        self.net = net
        self.error = None
        self.datasets = []
        self._gram_ws = {}
        self._abuf = {}
        self._pairs = {}

    def data_for(self, x, shard=None):
        raw = self.site_list(x)
        for d in self.datasets:
            if d.matches(raw) and d.sharded == (shard is not None):
                self.datasets.remove(d)
                self.datasets.append(d)
                return d
        d = self.make_data(raw, shard)
        self.datasets.append(d)
        if len(self.datasets) > 3:
            gone = self.datasets.pop(0)
            self.drop_pairs(getattr(gone, "xs", ()))
        return d

    def output_at(self, d, node, lp, core=None, rows=None):
        """(S, C) prediction through the local problem of `node` (the output is linear in every
        node); `rows` names the sample range a sliced problem covers."""
        return lp.jv(lp.core if core is None else core)

    def prefetch(self, d, node):
        """Bring up to date whatever cached state the local problem of `node` will read."""

    # -- Gram + solve workspaces
    def gram_ws(self, lp, H, sym, shard=None):
        """Compressed Gram of a single-term local problem into the cached workspace.  shard: the
        sample sharding of a multi-rank run; the feature-pair classes are then the ones common to
        all ranks (agreed on once per feature tensor)."""
        L, X, R = lp.terms[0]
        key = (lp.S, lp.C, L.shape[2], X.shape[1], R.shape[1], int(sym))
        ws = self._gram_ws.get(key)
        if ws is None:
            self._gram_ws.clear()
        rep = self.pair_rep(X)
        if shard is not None:
            rep = self._pair_lookup(X, "common")
            if rep is _MISSING:
                # with the feature matrices of the other sites (the same list in the same order on every rank)
                todo = [X] + [P for P in getattr(lp, "pair_peers", ()) if P.shape == X.shape
                              and self._pair_key(P, "common")[1] != self._pair_key(X, "common")[1]
                              and self._pair_lookup(P, "common") is _MISSING]
                reps = shard.common_pair_classes_many([self.pair_rep(P) for P in todo],
                                                      X.shape[1] * (X.shape[1] + 1) // 2, X.device)
                for P, r in zip(todo, reps):
                    self._pair_store(P, "common", r)
                rep = reps[0]
        ws = ops.gram(L, H, X, R, sym=sym, ws=ws, pair_rep=rep)
        self._gram_ws[key] = ws
        return ws

    # Feature-pair classes are cached per feature TENSOR: an entry holds the tensor that owns the
    # memory (the base of a view), so that its address cannot be handed to another tensor while the
    # entry lives, and the version it was computed for; a hit needs the same owner object, the same
    # window into it and the same version (views share their base's version counter).
    @staticmethod
    def _pair_key(X, kind):
        owner = X._base if X._base is not None else X
        return owner, (id(owner), X.storage_offset(), tuple(X.shape), tuple(X.stride()), kind)

    def _pair_lookup(self, X, kind):
        owner, key = self._pair_key(X, kind)
        e = self._pairs.get(key)
        if e is not None and e[0] is owner and e[1] == X._version:
            return e[2]
        return _MISSING

    def _pair_store(self, X, kind, rep):
        if len(self._pairs) > 256:
            self._pairs.clear()
        owner, key = self._pair_key(X, kind)
        self._pairs[key] = (owner, X._version, rep)

    def drop_pairs(self, tensors):
        """Forget the classes of feature tensors that leave the data-set cache."""
        ids = {id(t._base if t._base is not None else t) for t in tensors}
        for key in [k for k in self._pairs if k[0] in ids]:
            del self._pairs[key]

    def pair_rep(self, X):
        """Feature-pair classes of a feature matrix, detected once per tensor (and version)."""
        rep = self._pair_lookup(X, "own")
        if rep is _MISSING:
            rep = pair_classes(X)
            self._pair_store(X, "own", rep)
        return rep

    def prefetch_pairs(self, Xs):
        """Detect the classes of several feature matrices in one go (all sites of a data set when the
        first Gram is asked for), so that a sweep does not stop at every new site for it."""
        todo = {}
        for X in Xs:
            if X.ndim == 2 and 2 <= X.shape[1] <= PAIR_CLASS_MAX_P and self._pair_lookup(X, "own") is _MISSING:
                todo.setdefault(tuple(X.shape), {})[self._pair_key(X, 'own')[1]] = X
        for group in todo.values():
            tensors = list(group.values())
            for X, rep in zip(tensors, pair_classes_many(tensors)):
                self._pair_store(X, "own", rep)

    def dense_buffer(self, D, device):
        buf = self._abuf.get(D)
        if buf is None:
            self._abuf.clear()
            buf = torch.empty((D, D), dtype=F64, device=device)
            self._abuf[D] = buf
        return buf


class ChainEngine(EngineBase):
    def __init__(self, net):
        super().__init__(net)
        self.cols = None
        try:
            self.cols = self._plan()
        except ValueError as e:
            self.error = str(e)

    # ------------------------------------------------------------------ topology
    def _plan(self):
        net = self.net
        N = len(net.main_nodes)
        s = net.sample_dim
        outs = [lab for lab in (net.output_labels or ()) if lab != s]
        if len(net.input_nodes) != N:
            raise ValueError("expected one input node per core")
        cols = []
        used = set()
        for i, node in enumerate(net.main_nodes):
            col = _Column()
            col.A = node
            phys = [(lab, o) for lab, o in node.connections.items() if not node.is_horizontal_bond(lab)]
            if len(phys) != 1:
                raise ValueError(f"{node.name}: expected exactly one vertical edge")
            plab, below = phys[0]
            left = [lab for lab in node.left_labels if lab in node.dim_labels]
            right = [lab for lab in node.right_labels if lab in node.dim_labels]
            col.out_labels = [lab for lab in node.dim_labels if lab in outs]
            if col.out_labels and i != 0:
                raise ValueError("output labels are supported on the first core only")
            if i > 0 and (len(left) != 1 or node.connections.get(left[0]) is not net.main_nodes[i - 1]):
                raise ValueError(f"{node.name}: left bond does not reach the previous core")
            if i < N - 1 and (len(right) != 1 or node.connections.get(right[0]) is not net.main_nodes[i + 1]):
                raise ValueError(f"{node.name}: right bond does not reach the next core")
            col.order = left + col.out_labels + [plab] + right
            col.nleft = len(left) + len(col.out_labels)
            if sorted(col.order) != sorted(node.dim_labels):
                raise ValueError(f"{node.name}: unexpected labels {node.dim_labels}")
            col.F, col.f_order = None, None
            if below in net.input_nodes:
                xin = below
                xl = list(xin.dim_labels)
                others = [(lab, o) for lab, o in xin.connections.items() if o is not node]
                if xl == [s, plab] and not others:
                    col.kind = "plain"
                elif len(xl) == 3 and xl[0] == s and xl[1] == plab and len(others) == 1 and others[0][0] == xl[2]:
                    col.kind = "conv"
                    col.F = others[0][1]
                    inner = [xl[2]]
                else:
                    raise ValueError(f"{xin.name}: unsupported input column {xl}")
            else:
                col.kind = "op"
                col.F = below
                down = [(lab, o) for lab, o in below.connections.items()
                        if o is not node and not below.is_horizontal_bond(lab)]
                if len(down) != 1 or down[0][1] not in net.input_nodes:
                    raise ValueError(f"{below.name}: expected exactly one input edge below the operator node")
                dlab, xin = down[0]
                if list(xin.dim_labels) != [s, dlab] or len(xin.connections) != 1:
                    raise ValueError(f"{xin.name}: expected labels [sample, {dlab}]")
                inner = [plab, dlab]
            col.xin = net.input_nodes.index(xin)
            used.update([node, xin])
            if col.F is not None:
                fn = col.F
                used.add(fn)
                fl = [lab for lab in fn.left_labels if lab in fn.dim_labels]
                fr = [lab for lab in fn.right_labels if lab in fn.dim_labels]
                col.f_order, col.f_nleft = fl + inner + fr, len(fl)
                if sorted(col.f_order) != sorted(fn.dim_labels) or len(fl) > 1 or len(fr) > 1:
                    raise ValueError(f"{fn.name}: unexpected labels {fn.dim_labels}")
                if fl and (i == 0 or fn.connections.get(fl[0]) is not cols[i - 1].F):
                    raise ValueError(f"{fn.name}: left bond does not reach the previous column's feature node")
                if fr and (i == N - 1 or fn.connections.get(fr[0]) is None):
                    raise ValueError(f"{fn.name}: dangling right bond")
            cols.append(col)
        if len(used) != len(net.nodes):
            raise ValueError("nodes outside the recognised columns")
        for t in net.train_nodes:
            if t not in used or t in net.input_nodes:
                raise ValueError("train nodes outside the recognised columns")
        # an output label that no node carries was a size-1 axis squeezed away at construction
        return cols

    def require(self):
        if self.cols is None:
            raise NotImplementedError("this build runs tensor-train shaped networks (plain, conv-featurised or "
                                      f"operator columns) on the HIP path; topology not recognised: {self.error}")
        return self.cols

    # ------------------------------------------------------------------ output layout
    def out_labels(self):
        return list(self.cols[0].out_labels)

    def out_sizes(self):
        col = self.cols[0]
        return [col.A.dim_size(lab) for lab in col.out_labels]

    def out_dim(self):
        C = 1
        for n in self.out_sizes():
            C *= n
        return C

    # ------------------------------------------------------------------ canonical views
    @staticmethod
    def _perm(node, order):
        return [node.dim_labels.index(lab) for lab in order]

    @staticmethod
    def _back(node, order, sizes, t):
        perm = [node.dim_labels.index(lab) for lab in order]
        t = t.reshape(sizes)
        inv = [perm.index(a) for a in range(len(perm))]
        if inv != list(range(len(inv))):
            t = t.permute(inv)
        return t.to(node.tensor.dtype)

    def canonA(self, k):
        col = self.cols[k]
        node = col.A
        perm = self._perm(node, col.order)
        t = node.tensor.permute(perm) if perm != list(range(len(perm))) else node.tensor
        sizes = list(t.shape)
        rl = 1
        for n in sizes[:col.nleft]:
            rl *= n
        rr = 1
        for n in sizes[col.nleft + 1:]:
            rr *= n
        col.sizes, col.rl, col.p, col.rr = sizes, rl, sizes[col.nleft], rr
        return ops.prep(t.reshape(rl, col.p, rr))

    def canonF(self, k):
        """(ca, *inner, cb) view of the feature node of column k: inner = (pixels,) for a conv
        filter, (u, d) for an operator / linear node."""
        col = self.cols[k]
        fn = col.F
        perm = self._perm(fn, col.f_order)
        t = fn.tensor.permute(perm) if perm != list(range(len(perm))) else fn.tensor
        sizes = list(t.shape)
        nl = col.f_nleft
        ni = 1 if col.kind == "conv" else 2
        col.f_sizes = sizes
        col.ca = sizes[0] if nl else 1
        col.inner = tuple(sizes[nl:nl + ni])
        col.cb = sizes[nl + ni] if len(sizes) > nl + ni else 1
        return ops.prep(t.reshape((col.ca,) + col.inner + (col.cb,)))

    def column_of(self, node):
        for k, col in enumerate(self.cols):
            if node is col.A:
                return k, "A"
            if node is col.F:
                return k, "F"
        raise ValueError("node is not a trainable node of this network")

    def layout(self, node):
        """(canonical multi-axis sizes, permutation node axes -> canonical axes) of a train node."""
        k, kind = self.column_of(node)
        col = self.cols[k]
        if kind == "A":
            self.canonA(k)
            return col.sizes, self._perm(node, col.order)
        self.canonF(k)
        return col.f_sizes, self._perm(node, col.f_order)

    def to_canon(self, node, v):
        k, kind = self.column_of(node)
        col = self.cols[k]
        if kind == "A":
            self.canonA(k)
            return ops.prep(v.permute(self._perm(node, col.order)).reshape(col.rl, col.p, col.rr))
        self.canonF(k)
        return ops.prep(v.permute(self._perm(node, col.f_order)).reshape((col.ca,) + col.inner + (col.cb,)))

    def from_canon(self, node, t):
        k, kind = self.column_of(node)
        col = self.cols[k]
        if kind == "A":
            return self._back(node, col.order, col.sizes, t)
        return self._back(node, col.f_order, col.f_sizes, t)

    @property
    def mm32(self):
        """A float32 model (the reference then contracts in float32: image_convolution_lanczos_CIFAR10.py sets no
        dtype) runs the matrix products of its many-feature sites in float32; everything else stays float64."""
        return self.net.main_nodes[0].tensor.dtype == torch.float32

    # ------------------------------------------------------------------ data sets
    def site_list(self, x):
        cols = self.require()
        per_input = list(x) if isinstance(x, (tuple, list)) else [x] * len(self.net.input_nodes)
        return [per_input[col.xin] for col in cols]

    def make_data(self, raw, shard=None):
        device = self.net.main_nodes[0].tensor.device
        done, xs = {}, []
        for col, t in zip(self.cols, raw):
            if id(t) not in done:
                tt = t if shard is None else shard.rows(t)
                tt = tt.to(device) if tt.device != device else tt
                want = 3 if col.kind == "conv" else 2
                if tt.ndim != want:
                    raise ValueError(f"expected a {want}-d input (samples, features{', pixels' if want == 3 else ''}), "
                                     f"got shape {tuple(tt.shape)}")
                # a float32 image set stays float32 (the kernels that read it convert on the fly)
                keep32 = col.kind == "conv" and tt.dtype == torch.float32 and tt.is_cuda
                done[id(t)] = tt.contiguous() if keep32 else ops.prep(tt)
            xs.append(done[id(t)])
        d = _Data(raw, xs, len(raw))
        d.sharded = shard is not None
        d.S_total = raw[0].shape[0]
        return d

    def sync(self, d):
        """Drop cached features / environments that depend on a node that changed."""
        N = len(self.cols)
        first, last = None, None
        for j, col in enumerate(self.cols):
            ca, cf = _changed(d.seenA[j], col.A), _changed(d.seenF[j], col.F)
            if cf:
                d.G[j] = None
                d.seenF[j] = (col.F.tensor, col.F.tensor._version)
            if ca:
                d.seenA[j] = (col.A.tensor, col.A.tensor._version)
            if ca or cf:
                first = j if first is None else first
                last = j
        if first is not None:
            for k in range(first + 1, N):
                d.L[k] = None
            for k in range(0, last):
                d.R[k] = None
            d.hot = last

    # ------------------------------------------------------------------ features and environments
    def feats(self, d, k):
        """G_k: (ca, cb, S, p) feature vectors per feature-bond pair (a plain column: X itself).
        The product with the feature node is one streaming kernel (conv) / one row GEMM (operator)."""
        if d.G[k] is None:
            col = self.cols[k]
            X = d.xs[k]
            if col.kind == "plain":
                d.G[k] = X.reshape(1, 1, *X.shape)
            elif col.kind == "conv":
                d.G[k] = ops.conv_feature(X, self.canonF(k))          # (ca, px, cb) filter -> (ca, cb, S, p)
            else:
                Fc = self.canonF(k)                                   # (ca, u, dd, cb)
                u, dd = col.inner
                Bm = Fc.permute(2, 0, 3, 1).reshape(dd, col.ca * col.cb * u).contiguous()
                flat = ops.gemm_rows(X, Bm, col.ca * col.cb * u, mm32=self.mm32)
                d.G[k] = flat.reshape(X.shape[0], col.ca, col.cb, u).permute(1, 2, 0, 3).contiguous()
        return d.G[k]

    def left(self, d, k):
        """Environment entering column k from the left: (ca_k, S, C, rl_k)."""
        if d.L[k] is None:
            if k == 0:
                C = self.out_dim()
                eye = torch.eye(C, dtype=F64, device=d.xs[0].device)
                d.L[0] = eye.unsqueeze(0).expand(d.S, C, C).contiguous().unsqueeze(0)
            else:
                d.L[k] = self.left_step(d, k - 1, self.canonA(k - 1))
        return d.L[k]

    def left_step(self, d, k, core):
        """Column k applied to the left environment with `core` in place of A_k: (cb_k, S, C, rr_k)."""
        Lk, G = self.left(d, k), self.feats(d, k)
        ca, cb = G.shape[0], G.shape[1]
        rl, p, rr = core.shape
        two_stage = TWO_STAGE_MIN_P <= p <= TWO_STAGE_MAX_P      # many features: contract them first
        if two_stage and ca * cb > 1 and G.is_contiguous() and Lk.is_contiguous() \
                and ca * cb * d.S * rl * rr * 8 <= BOND_LEFT_MAX_BYTES:
            # all terms of a convolution bond at once: one row GEMM over the stacked features, then the sum over the
            # left channels inside one kernel
            T = ops.gemm_rows(G.reshape(ca * cb * d.S, p), core, rl * rr, bmode=1, rl=rl, rr=rr, mm32=self.mm32)
            out = ops.lcontract_sum(Lk, T.reshape(ca, cb, d.S, rl, rr))
            if out is not None:
                return out
        outs = []
        for b in range(cb):
            o = None
            for a in range(ca):
                if two_stage:
                    o = ops.env_step_left_dense(Lk[a], G[a, b], core, mm32=self.mm32) if o is None else \
                        ops.env_step_left_dense(Lk[a], G[a, b], core, out=o, mm32=self.mm32)
                else:
                    o = ops.env_step_left(Lk[a], G[a, b], core) if o is None else \
                        ops.env_step_left(Lk[a], G[a, b], core, out=o)
            outs.append(o)
        return torch.stack(outs) if cb > 1 else outs[0].unsqueeze(0)

    def right(self, d, k):
        """Environment entering column k from the right: (cb_k, S, rr_k)."""
        N = len(self.cols)
        if d.R[k] is None:
            if k == N - 1:
                d.R[k] = torch.ones((1, d.S, 1), dtype=F64, device=d.xs[0].device)
            else:
                Rn, G, core = self.right(d, k + 1), self.feats(d, k + 1), self.canonA(k + 1)
                ca, cb = G.shape[0], G.shape[1]
                if ca * cb > 1 and ops.bond_z_supported(G, Rn, core):
                    # small bonds around a convolution bond: all ca cb terms in one launch
                    d.R[k] = ops.bond_z(G, Rn, core)
                    return d.R[k]
                outs = []
                for a in range(ca):
                    o = None
                    for b in range(cb):   # (a site with many features contracts them first inside the op)
                        o = ops.env_step_right(Rn[b], G[a, b], core, mm32=self.mm32) if o is None else \
                            ops.env_step_right(Rn[b], G[a, b], core, out=o, mm32=self.mm32)
                    outs.append(o)
                d.R[k] = torch.stack(outs) if ca > 1 else outs[0].unsqueeze(0)
        return d.R[k]

    def prefetch(self, d, node):
        self.sync(d)
        k, _ = self.column_of(node)
        self.left(d, k)
        self.right(d, k)
        self.feats(d, k)

    def output_at(self, d, node, lp, core=None, rows=None):
        """The prediction a node's update starts from.  A sweep that arrives at column k from the left still holds
        the right environment that includes column k (built on the way back of the sweep before, or with the
        first environments), one that arrives from the right the left environment that includes it: the prediction
        is then one row-dot kernel over the two cached environments instead of an environment step through the
        node followed by it.  Anything else (a trial core, a sample range, no cached neighbour) goes through the
        local problem."""
        if core is None and rows is None:
            self.sync(d)
            k, _ = self.column_of(node)
            N = len(self.cols)
            if k >= 1 and d.R[k - 1] is not None and d.L[k] is not None:
                Lk, Rk = d.L[k], d.R[k - 1]
            elif k + 1 < N and d.L[k + 1] is not None and d.R[k] is not None:
                Lk, Rk = d.L[k + 1], d.R[k]
            else:
                Lk = None
            if Lk is not None and Lk.shape[0] == Rk.shape[0]:
                y = ops.predict(Lk[0], Rk[0])
                for a in range(1, Lk.shape[0]):
                    ops.predict(Lk[a], Rk[a], out=y)
                return y
        return super().output_at(d, node, lp, core, rows)

    def refresh(self, d):
        """Bring every cached environment up to date (the reference's recompute_all_stacks)."""
        self.sync(d)
        self.left(d, len(self.cols) - 1)
        self.right(d, 0)

    def predict(self, d, k=None):
        """(S, C) prediction, cut at column k (default: the cheapest cut)."""
        self.sync(d)
        N = len(self.cols)
        if k is None:
            vl = 0
            while vl + 1 < N and d.L[vl + 1] is not None:
                vl += 1
            vr = N - 1
            while vr - 1 >= 0 and d.R[vr - 1] is not None:
                vr -= 1
            # among equally cheap cuts take the one next to the column that changed last: a sweep moves on
            # from there, so the environments built up to it stay valid for the following predictions
            k = min(range(N), key=lambda j: (max(0, j - vl) + max(0, vr - j), abs(j - d.hot)))
        Lk1 = self.left_step(d, k, self.canonA(k))
        R = self.right(d, k)
        y = ops.predict(Lk1[0], R[0])
        for b in range(1, R.shape[0]):
            ops.predict(Lk1[b], R[b], out=y)
        return y

    # ------------------------------------------------------------------ local problems
    def local(self, d, node):
        self.sync(d)
        k, kind = self.column_of(node)
        L, R, G = self.left(d, k), self.right(d, k), self.feats(d, k)
        S, C = d.S, L.shape[2]
        core = self.canonA(k)
        col = self.cols[k]
        back = lambda t, node=node: self.from_canon(node, t)   # noqa: E731
        if kind == "A":
            if not d.pairs_done:      # plain columns: G[0, 0] is the input itself
                d.pairs_done = True
                for j, c in enumerate(self.cols):
                    # a feature map that knows its own classes (polynomial_basis) says so on the tensor
                    hint = getattr(d.xs[j], "ttk_pair_rep", None) if c.kind == "plain" else None
                    if hint is not None and hint[1] == d.xs[j]._version:
                        self._pair_store(self.feats(d, j)[0, 0], "own", list(hint[0]))
                d.pair_feats = [self.feats(d, j)[0, 0] for j, c in enumerate(self.cols) if c.kind == "plain"]
                self.prefetch_pairs(d.pair_feats)
            terms = [(L[a], G[a, b], R[b]) for a in range(G.shape[0]) for b in range(G.shape[1])]
            prob = LocalProblem(S, C, (col.rl, col.p, col.rr), core, back, terms=terms)
            prob.mm32 = self.mm32
            prob.pair_peers = getattr(d, "pair_feats", ())
            if len(terms) > 1 and ops.bond_supported(L, G, R):
                prob.bond = (L, G, R)
            return prob
        # feature node: with M_ab[s,c,u] = L_a . (A . R_b) (a row GEMM, then the left environment's output index)
        #   conv filter    J[s,c,a,x,b]   = sum_patch M_ab[s,c,patch] X[s,patch,x]
        #   operator node  J[s,c,a,u,d,b] = M_ab[s,c,u] X[s,d]   (one Khatri-Rao term when a = b = 1)
        Fc = self.canonF(k)
        X = d.xs[k]
        ca, cb = col.ca, col.cb
        A2 = core.reshape(col.rl * col.p, col.rr)

        U_of = {}

        def M(a, b):
            if b not in U_of:      # U_b = A . R_b is shared by the left environments L_a of a convolution bond
                U_of.clear()       # (callers walk b in the outer loop: one U alive at a time)
                U_of[b] = ops.gemm_rows(R[b], A2, col.rl * col.p, bmode=2, mm32=self.mm32).reshape(S, col.rl, col.p)
            return ops.lcontract(L[a], U_of[b])            # (S, C, p)

        if col.kind == "op" and ca == 1 and cb == 1:
            u, dd = col.inner
            ones = torch.ones((S, 1), dtype=F64, device=X.device)
            return LocalProblem(S, C, (u, dd, 1), Fc.reshape(u, dd, 1), back, terms=[(M(0, 0).contiguous(), X, ones)])
        if col.kind == "conv" and ca * cb > 1:
            # the patches are contracted before the left environments: Z[s,l,x,b] = sum_patch U_b[s,l,patch] X[s,patch,x]
            # reads the image rows cb times (not ca cb times), and J[s,c,a,(x,b)] = sum_l L_a[s,c,l] Z[s,l,(x,b)] is one
            # small contraction per left channel, written in place
            px = col.inner[0]
            Z = None
            if R.is_contiguous():
                # U_b = A . R_b for all b as one row GEMM over the stacked right environments, the patches in one pass
                U = ops.gemm_rows(R.reshape(cb * S, -1), A2, col.rl * col.p, bmode=2, mm32=self.mm32)
                Z = ops.patch_contract_multi(U.reshape(cb, S, col.rl, col.p), X)
            if Z is None:
                Z = torch.empty((S, col.rl, 1, px, cb), dtype=F64, device=X.device)
                for b in range(cb):
                    U = ops.gemm_rows(R[b], A2, col.rl * col.p, bmode=2, mm32=self.mm32).reshape(S, col.rl, col.p)
                    ops.patch_contract(U, X, Z, 0, b)
            Zf = Z.reshape(S, col.rl, px * cb)
            if col.rl * px * cb + ca * C * col.rl < C * ca * px * cb and ops.filter_supported(L, Zf):
                # the factors are fewer numbers than the rows: the solver's products stream them instead
                return LocalProblem(S, C, (ca,) + col.inner + (cb,), Fc, back, fz=(L, Zf))
            Jf = torch.empty((S, C, ca, px * cb), dtype=F64, device=X.device)
            for a in range(ca):
                ops.lcontract_into(L[a], Zf, Jf, a)
            return LocalProblem(S, C, (ca,) + col.inner + (cb,), Fc, back, J=Jf.reshape(S, C, -1))
        J = torch.empty((S, C, ca) + col.inner + (cb,), dtype=F64, device=X.device)
        for b in range(cb):
            for a in range(ca):
                if col.kind == "conv":
                    ops.patch_contract(M(a, b), X, J, a, b)                       # J[:, :, a, :, b] = M . X
                else:
                    u, dd = col.inner
                    J[:, :, a, :, :, b] = ops.kr_rows(M(a, b), X, _ones_col(S, X.device)).reshape(S, C, u, dd)
        shape = (ca,) + col.inner + (cb,)
        return LocalProblem(S, C, shape, Fc, back, J=J.reshape(S, C, -1))
