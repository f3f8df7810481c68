/* ttk.h -- C ABI of the MI355X (gfx950) tensor-train core-update kernels.
 *
 * The reference (niccogc/TensorNetworksFork) has no FFI: its hot path is a
 * chain of torch.einsum / torch.linalg calls inside tensor/network.py.  Each
 * entry point below replaces one of those call sites; the citation names the
 * reference lines whose arithmetic it performs.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to dense row-major arrays, 8-byte aligned;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *   - no entry point allocates: scratch comes in through `ws` (size from the
 *     matching *_ws_bytes function);
 *   - return value: 0 ok, negative = rejected argument (nothing was launched).
 *
 * Index names: s sample (S), c/d output (C), l/m left bond (rl), p/q physical
 * (p), r/t right bond (rr), D = rl*p*rr.
 */
#ifndef TTK_H
#define TTK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTK_OK 0
#define TTK_EINVAL (-1)   /* bad extent / null pointer */
#define TTK_EWORKSPACE (-2) /* workspace too small */
#define TTK_ELAUNCH (-3)  /* hipLaunch reported an error */

int ttk_version(void);

/* Left environment step  out[s,c,r] (+)= sum_{l,p} E[s,c,l] core[l,p,r] X[s,p].
 * E (S,C,rl), X (S,p), core (rl,p,rr), out (S,C,rr).  accumulate != 0 adds to `out` (used to sum
 * the conv-bond terms of a conv-featurised column, tensor/layers.py:839-872).
 * Replaces the pair of einsums issued by TensorNode.contract_with for one
 * column of compute_stacks('left') / left_update_stacks
 * (tensor/network.py:55-71,152-161; tensor/node.py:28-45). */
int ttk_env_step_left_f64(const double* E, const double* X, const double* core, double* out,
                          int64_t S, int C, int rl, int p, int rr, int accumulate, void* stream);

/* Right environment step  out[s,l] = sum_{p,r} core[l,p,r] X[s,p] R[s,r].
 * R (S,rr), X (S,p), core (rl,p,rr), out (S,rl).
 * compute_stacks('right') / right_update_stacks (tensor/network.py:55-71,163-172). */
int ttk_env_step_right_f64(const double* R, const double* X, const double* core, double* out,
                           int64_t S, int rl, int p, int rr, int accumulate, void* stream);

/* Prediction  y[s,c] (+)= sum_r L[s,c,r] R[s,r]; L (S,C,r), R (S,r), y (S,C).
 * Last contraction of TensorNetwork.forward (tensor/network.py:121-131). */
int ttk_predict_f64(const double* L, const double* R, double* y, int64_t S, int C, int r, int accumulate, void* stream);

/* Explicit Jacobian rows  J[s,c,(l,p,r)] (+)= L[s,c,l] X[s,p] R[s,r]  (S,C,rl*p*rr), the tensor
 * compute_jacobian_stack materialises (tensor/network.py:101-113).  Only used where the local
 * Jacobian is a SUM of such terms (conv bond > 0) and the Gram matrix is wanted. */
int ttk_kr_rows_f64(const double* L, const double* X, const double* R, double* J, int64_t S, int C, int rl, int p, int rr,
                    int accumulate, void* stream);

/* Losses: per-sample value, gradient g (S,C) and Hessian H (S,C,C) w.r.t. the
 * model output.  kind 0: SquareBregFunction (tensor/bregman.py:9-23,34-52;
 * loss (S), H all-ones*2 as the reference's (S,C,1) broadcast acts inside
 * get_A_b, tensor/network.py:188,212); kind 1: AutogradLoss(MSELoss(none))
 * (bregman.py:266-292; loss (S,C), H = 2 I); kind 2: XEAutogradBregman(w)
 * (bregman.py:189-216; y (S,C+1); loss (S), gradient w(pi - y) as KLDivBregman, bregman.py:140);
 * kind 3: the same with the gradient against the label, w(pi - onehot(argmax y)): what
 * differentiating cross_entropy(z, argmax y) gives (bregman.py:198-204) when y is not one-hot. */
int ttk_loss_f64(int kind, double w, const double* y_pred, const double* y, double* loss, double* g,
                 double* H, int64_t S, int C, void* stream);

/* out[0] = mean over mini-batches of the per-batch mean of v (n_per_sample
 * values per sample), the quantity accumulating_swipe reports
 * (tensor/network.py:472-474,493-494). */
int ttk_batch_mean_f64(const double* v, int64_t S, int n_per_sample, int64_t batch, double* out, void* stream);

/* HL[s,d,l] = sum_c H[s,c,d] L[s,c,l]  (S,C,rl): the Hessian-weighted left
 * factor of the Gram einsum (tensor/network.py:207,212). */
int ttk_hess_left_f64(const double* H, const double* L, double* HL, int64_t S, int C, int rl, void* stream);

/* Gram matrix of the local normal equations
 *   A[(l,p,r),(m,q,t)] = sum_{s,c,d} H[s,c,d] L[s,c,l]X[s,p]R[s,r] L[s,d,m]X[s,q]R[s,t]
 * (get_A_b, tensor/network.py:174-212, summed over mini-batches as in 463-468).
 * Computed in symmetry-compressed form into ws, then expanded:
 *   ttk_gram_f64        -> compressed G in ws  (H (S,C,C) is the per-sample Hessian)
 *   ttk_gram_scale_f64  -> scale[0] = mean_i |A_ii| (1 if 0)      (network.py:298-300)
 *   ttk_gram_expand_f64 -> A_out (D,D) = A / scale[0] + ridge*I   (network.py:301,308,312)
 *                          (scale == NULL means 1).
 * sym_left != 0 asserts H[s] symmetric (exploits l<->m symmetry).
 * x_pair_rep (HOST array of p(p+1)/2 ints, or NULL) names feature-pair classes: entry b, for
 * the pair (i<=j) at upper-triangle row-major position b, is the position of an earlier pair
 * (or b itself) whose product X_i X_j equals this pair's for every sample - e.g. a
 * polynomial basis X_i = x^i (models/tnml.py:11-17) has X_i X_j = x^(i+j).  Only one slice of
 * the Gram per class is computed and the others are copies; the caller vouches for the
 * equality.  Honoured for p <= 10, ignored (everything computed) above. */
int64_t ttk_gram_ws_bytes(int64_t S, int C, int rl, int p, int rr, int sym_left);
int ttk_gram_f64(const double* L, const double* H, const double* X, const double* R,
                 int64_t S, int C, int rl, int p, int rr, int sym_left, const int* x_pair_rep,
                 void* ws, int64_t ws_bytes, void* stream);
/* ttk_gram_f64 that can be called off while it runs.  A sweep queues the next node's Gram before the host knows
 * whether the sweep goes on (convergence_criterion / timeout / a singular system end it:
 * tensor/network.py:435-437,482-484,497-504); when it does not, ttk_gram_cancel writes `generation` into the device word
 * `cancel_word` (4 bytes, on a stream that is not queued behind the Gram) and every workgroup of a
 * ttk_gram_cancellable_f64 launched with that word and that generation that has not started yet leaves at once:
 * the workspace then holds nothing of use.  Each launch that shares a word takes a generation of its own, so that no
 * reset of the word is needed.  cancel_word == NULL: ttk_gram_f64. */
int ttk_gram_cancellable_f64(const double* L, const double* H, const double* X, const double* R,
                             int64_t S, int C, int rl, int p, int rr, int sym_left, const int* x_pair_rep,
                             void* ws, int64_t ws_bytes, const void* cancel_word, unsigned generation, void* stream);
int ttk_gram_cancel(void* cancel_word, unsigned generation, void* stream);
int ttk_gram_scale_f64(const void* ws, int rl, int p, int rr, int sym_left, double* scale, void* stream);
int ttk_gram_expand_f64(const void* ws, int rl, int p, int rr, int sym_left, const double* scale,
                        double ridge, double* A_out, void* stream);

/* Timing of the Gram kernel with HIP events recorded on the launch stream (no
 * counterpart in the reference; used by bench.py for the roofline figure).
 * ttk_gram_timing(1) creates a pool of event pairs and arms recording around every
 * subsequent Gram kernel launch (ttk_gram_timing(2): around every launch of the fused
 * matrix-free product kernel instead; 3 / 4: the features-first right-hand side / the resident right-environment step
 * of a site with many features; 5 / 6: the second / first half of the operator of a column between convolution
 * bonds); ttk_gram_timing(0) disarms.  ttk_gram_timing_read
 * waits for the recorded pairs, writes up to `max` durations (milliseconds, launch
 * order) to ms_out (HOST pointer), clears the list and returns how many it wrote. */
int ttk_gram_timing(int enable);
int ttk_gram_timing_read(double* ms_out, int max);

/* Right-hand side / transposed Jacobian product
 *   b[l,p,r] = sum_{s,c} g[s,c] L[s,c,l] X[s,p] R[s,r]
 * (get_A_b / get_b, tensor/network.py:215,289; second einsum of the matrix-free
 * product, network.py:789).  b (rl,p,rr); accumulate != 0 adds to b. */
int64_t ttk_rhs_ws_bytes(int64_t S, int C, int rl, int p, int rr);
int ttk_rhs_f64(const double* L, const double* X, const double* R, const double* g, double* b,
                int64_t S, int C, int rl, int p, int rr, int accumulate, void* ws, int64_t ws_bytes, void* stream);

/* ttk_rhs_f64 for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same
 * arguments, same kernels, same result bits as ttk_rhs_f64, but the slab partials live in a
 * file-scope device array of ttk_rhs_static_max_bytes() bytes (32 MiB, held by every GPU the
 * library runs on until the process ends) instead of a caller workspace; the call allocates and
 * frees nothing.  Extents whose ttk_rhs_ws_bytes exceeds that size return TTK_EWORKSPACE and launch
 * nothing.  All calls on one GPU share the array: calls that overlap on one GPU (different
 * streams) race; calls on one stream are ordered and safe.  The array's device address is looked up
 * once per device (hipGetSymbolAddress, not a capturable operation): ttk_rhs_static_max_bytes() does
 * it for the current device, so call it once per device before a capture; otherwise the first
 * ttk_rhs_static_f64 on that device does.  The device must be current when either is called. */
int64_t ttk_rhs_static_max_bytes(void);
int ttk_rhs_static_f64(const double* L, const double* X, const double* R, const double* g, double* b,
                       int64_t S, int C, int rl, int p, int rr, int accumulate, void* stream);

/* Jacobian-vector product  u[s,c] = sum_{l,p,r} L[s,c,l] X[s,p] R[s,r] v[l,p,r]
 * followed by hu[s,d] = sum_c H[s,c,d] u[s,c]  (first einsum of the matrix-free
 * product, tensor/network.py:788).  tmp (S,C,rr) scratch; hu (S,C). */
int ttk_jvp_f64(const double* L, const double* X, const double* R, const double* H, const double* v,
                double* tmp, double* hu, int64_t S, int C, int rl, int p, int rr, void* stream);

/* Matrix-free product (J^T H J) v in one pass over the samples, for one output (C = 1, H one number
 * per sample or NULL = 1) and square bonds rl = rr in {16, 32} with p <= 8 features: the two einsums
 * of the reference's matvec closure (tensor/network.py:788-789) fused, the samples read once.
 * ttk_matvec_fused_ws_bytes returns TTK_EINVAL for extents this kernel does not take (callers then
 * use ttk_jvp_f64 + ttk_rhs_f64).
 *
 * ttk_cg_fused_iter_f64: iteration `it` (0, 1, 2, ...) of scipy.sparse.linalg.cg on that operator
 * (scipy_swipe with solver = cg, tensor/network.py:923-924) as two launches and no host read:
 *   p = r + (rho/rho_old) p_old  (p = r for it = 0), q = A p, alpha = rho / <p,q>, x += alpha p,
 *   r -= alpha q, after SciPy's stopping test sqrt(rho) < atol on the current residual.
 * pbuf holds two vectors (2 D doubles: p of even / odd iterations); state (5 doubles on the device) =
 * [rho of even its, rho of odd its, atol, stopped, iterations done]; before it = 0 the caller sets
 * state = [0, <r,r>, atol, 0, 0].  Once the test holds `stopped` is set and every later call is a no-op. */
int64_t ttk_matvec_fused_ws_bytes(int64_t S, int C, int rl, int p, int rr);
int ttk_matvec_fused_f64(const double* L, const double* X, const double* R, const double* H, const double* v,
                         double* out, int64_t S, int rl, int p, int rr, void* ws, int64_t ws_bytes, void* stream);
int ttk_cg_fused_iter_f64(const double* L, const double* X, const double* R, const double* H, double* x, double* r,
                          double* pbuf, double* state, int it, int64_t S, int rl, int p, int rr, void* ws,
                          int64_t ws_bytes, void* stream);

/* Right-hand side of the scaled, ridge-shifted system, negated:
 *   out[i] = -( b[i] / scale[0] + ridge * w[i] )            (network.py:302,309,313,315)
 * scale == NULL means 1; w == NULL means no shift. */
int ttk_solve_rhs_f64(const double* b, const double* w, const double* scale, double ridge, double* out,
                      int64_t D, void* stream);

/* Cholesky solve of the scaled, ridge-shifted normal equations, replacing
 * torch.linalg.cholesky + torch.cholesky_solve in solve_system
 * (tensor/network.py:311-320).  Blocked right-looking factorisation (64-wide panels,
 * fp64 MFMA trailing updates) with the right-hand side carried as an extra block row.
 *   ttk_chol_ws_bytes(n)  -> bytes of the factorisation workspace for order n
 *   ttk_chol_prepare_f64  -> fills the workspace from a compressed Gram (see ttk_gram_f64):
 *                            lower triangle of A/scale[0] + ridge*I and the right-hand side
 *                            -(b/scale[0] + ridge*w)        (network.py:301-302,312-313,315)
 *   ttk_chol_solve_f64    -> x (n) = solution; info[0] (device int) = 0, or the 1-based order
 *                            of the first non-positive pivot (torch.linalg.LinAlgError in
 *                            the reference, network.py:481-484), or -1 when the single-launch
 *                            factorisation gave up waiting (x is then not a solution). */
int64_t ttk_chol_ws_bytes(int n);
int ttk_chol_prepare_f64(const void* gram_ws, int rl, int p, int rr, int sym_left, const double* scale, double ridge,
                         const double* b, const double* w, void* chol_ws, void* stream);
int ttk_chol_solve_f64(void* chol_ws, int n, double* x, int* info, void* stream);
/* ttk_chol_prepare_f64 + ttk_chol_solve_f64 in one call (tensor/network.py:296-316 on the matrix of
 * network.py:207-215).  From order 2048 on, where the factorisation is one launch of tile tasks, each task
 * takes its tile of A/scale[0] + ridge*I straight from the compressed Gram: the expanded lower triangle
 * (4 n^2 bytes written and read back) never exists.  Same values, same operations, same result bits as the
 * two calls; below that order the two calls are what runs. */
int ttk_chol_solve_gram_f64(const void* gram_ws, int rl, int p, int rr, int sym_left, const double* scale, double ridge,
                            const double* b, const double* w, void* chol_ws, double* x, int* info, void* stream);
/* How ttk_chol_solve_f64 factorises (same call, same result up to rounding): 0 = one launch per
 * panel and trailing update; 1 = the whole factorisation as one launch of tile tasks that wait on
 * each other through progress words in the workspace; -1 (default) = 1 from 32 block columns (order 2048) on,
 * where it is the faster one.
 * Returns the previous mode; any other argument only queries.  Two ranges tune the single launch:
 * 100 + u keeps u compute units for its diagonal chain (100: chosen from the order), 1000 + k lets a
 * waiting lane give up after k polls (1000: the default of two million), which aborts the launch with
 * info = -1 and is how the tests check that every wait ends.  200 / 201: the backward substitution behind
 * the single-launch factorisation as one launch per group of four block rows / as one launch of workgroups
 * that wait for each other's block rows (201, the default; same operations, same bits). */
int ttk_chol_mode(int mode);

/* Reduced Householder QR of a row-major (m,n) matrix in one workgroup, LAPACK sign
 * conventions: A = Q R, Q (m,k), R (k,n), k = min(m,n).  Replaces torch.linalg.qr in
 * node_orthonormalize_left/right (tensor/network.py:644,686).  Needs ttk_qr_lds_bytes(m,n)
 * <= 150 KiB of LDS (returns TTK_EINVAL otherwise: the caller keeps its library QR). */
int64_t ttk_qr_lds_bytes(int m, int n);
int ttk_qr_f64(const double* A, int m, int n, double* Q, double* R, void* stream);

/* The gauge move to the left of node_orthonormalize_right (tensor/network.py:672-704) in one launch: the core as it
 * lies in memory, T (n, m) row-major with the left bond n first, is factorised as T = Rm Qt, Qt (k, m) with orthonormal
 * rows (the new core), Rm (n, k) (what the previous core is multiplied with, network.py:704), k = min(m, n) - the
 * reference's QR of the doubly flipped transposed matrix (network.py:685-689, same reflectors in the same order, same
 * signs) read and written through index maps instead of permute / flip / transpose copies.  Same size limit as ttk_qr_f64
 * with (m, n). */
int ttk_qr_right_f64(const double* T, int n, int m, double* Qt, double* Rm, void* stream);

/* Krylov vector kernels with wavefront-shuffle reductions (lanczos_solver,
 * tensor/network.py:796-824): out[0] = <x,y>;  y += a[0]*x  (a on device);
 * Both leave results on the device so that an iteration needs no host sync. */
int ttk_dot_f64(const double* x, const double* y, double* out, int64_t n, void* stream);
int ttk_axpy_f64(const double* a, double sign, const double* x, double* y, int64_t n, void* stream);

/* GL[s,l] = sum_c g[s,c] L[s,c,l]: the output index of the left environment contracted with the
 * gradient (or with H u), the first factor of b = J^T g (get_b, tensor/network.py:258-291) when the
 * features are contracted last (sites with many features). */
int ttk_gl_f64(const double* g, const double* L, double* GL, int64_t S, int C, int rl, void* stream);

/* The vector half of a Krylov iteration as one launch of one workgroup (the vectors have the
 * size of one core), so that an iteration is the operator product plus one more kernel and the
 * host never waits inside the loop.
 *
 * ttk_cg_update_f64: conjugate gradients with the recurrence and stopping rule of
 * scipy.sparse.linalg.cg, which scipy_swipe hands the local system to (tensor/network.py:897-924).
 * state (4 doubles on the device) = [rho = <r,r>, atol = rtol*||b||, stopped, updates done].  If
 * sqrt(rho) < atol the call only sets stopped = 1 and leaves x, r, p untouched (calls queued
 * behind it are no-ops); otherwise  alpha = rho/<p,q>, x += alpha p, r -= alpha q,
 * rho' = <r,r>, p = r + (rho'/rho) p, state[0] = rho', state[3] += 1.   q = A p on entry.
 *
 * ttk_lanczos_update_f64: iteration j >= 1 of the Lanczos recurrence of lanczos_solver
 * (tensor/network.py:796-824) with w = A v_cur on entry:  w -= betas[j-1] v_prev (skipped when
 * v_prev is NULL: the reference's zero vector of the first iteration), alphas[j-1] = <w,v_cur>,
 * w -= alphas[j-1] v_cur, betas[j] = ||w||, v_next = w / betas[j].  The reference leaves its loop
 * when betas[j] < tol (network.py:813): the kernel then writes j to stop[0] (device, 0.0 before the
 * first iteration), and every later call is a no-op once stop[0] != 0, so the host may queue
 * iterations ahead and look at stop[0] only now and then.
 *
 * ttk_lanczos_finish_f64: the end of that solver (network.py:815-824).  m = stop[0] if set, else
 * m_run (the iterations queued); T_m y = betas[0] e_1 with T_m = tridiag(betas[1..m), alphas[0..m),
 * betas[1..m)) by Gaussian elimination with partial pivoting (torch.linalg.solve on the dense T_m in
 * the reference), then x = x0 + sum_{j<m} y[j] V[j,:] over the rows of the basis V (m_cap+1 rows of
 * stride ldv >= D; row j = v_{j+1}): one pass over the basis, no copy of it.  ws: ttk_lanczos_finish_ws_bytes(m_cap).
 *
 * ttk_minres_update_f64: the vector half of one iteration of scipy.sparse.linalg.minres (no shift, no
 * preconditioner; the other solver scipy_swipe is used with, CIFAR_minres.py:112 through
 * tensor/network.py:923-924), recurrence and stopping rules of SciPy's _isolve/minres.py.  On entry
 * y = A v.  state (18 doubles on the device) = [istop, itn, oldb, beta, dbar, epsln, phibar, rhs1, rhs2,
 * tnorm2, gmax, gmin, cs, sn, beta1, rtol, maxiter, machine eps]; istop != 0 (SciPy's codes; 6 =
 * iteration limit) freezes everything, so calls queued behind the last iteration are no-ops.  On exit
 * v is the next Lanczos vector (the next product's operand), x the current solution. */
int ttk_cg_update_f64(double* x, double* r, double* p, const double* q, double* state, int64_t D, void* stream);
/* ttk_cg_update_f64 for long vectors (cores of conv tensor trains): with a workspace of ttk_cg_update_ws_bytes(D) bytes
 * the update of a vector beyond 8192 entries is spread over 128 workgroups (four launches, partial sums in the
 * workspace, fixed summation order); shorter vectors, or ws = NULL, take the one-workgroup kernel. */
int64_t ttk_cg_update_ws_bytes(int64_t D);
int ttk_cg_update_ws_f64(double* x, double* r, double* p, const double* q, double* state, int64_t D, void* ws,
                         int64_t ws_bytes, void* stream);
int ttk_minres_update_f64(double* y, double* r1, double* r2, double* v, double* w, double* w2, double* x, double* state,
                          int64_t D, void* stream);
/* ttk_minres_update_f64 for long vectors, as ttk_cg_update_ws_f64: with a workspace of ttk_minres_update_ws_bytes(D) bytes
 * an iteration on more than 8192 entries runs on 128 workgroups (three vector launches, one scalar launch). */
int64_t ttk_minres_update_ws_bytes(int64_t D);
int ttk_minres_update_ws_f64(double* y, double* r1, double* r2, double* v, double* w, double* w2, double* x, double* state,
                             int64_t D, void* ws, int64_t ws_bytes, void* stream);
int ttk_lanczos_update_f64(double* w, const double* v_prev, const double* v_cur, double* v_next, double* alphas,
                           double* betas, int j, int64_t D, double tol, double* stop, void* stream);
/* Iterations j_first .. j_first + count - 1 of the Lanczos recurrence on the operator of one Khatri-Rao term with one
 * output (the extents ttk_matvec_fused_f64 takes), queued by one call: per iteration the fused product of row j - 1 of
 * the basis V (rows ldv doubles apart) into w (D doubles of scratch), then ttk_lanczos_update_f64 writing row j.  No
 * host work between the launches of an iteration (tensor/network.py:805-814 without the interpreter in the loop). */
int ttk_lanczos_fused_block_f64(const double* L, const double* X, const double* R, const double* H, double* V, int64_t ldv,
                                double* alphas, double* betas, double* stop, int j_first, int count, double tol, double* w,
                                int64_t S, int rl, int p, int rr, void* ws, int64_t ws_bytes, void* stream);
int64_t ttk_lanczos_finish_ws_bytes(int m_cap);
int ttk_lanczos_finish_f64(const double* V, int64_t ldv, const double* alphas, const double* betas, const double* stop,
                           int m_run, int m_cap, const double* x0, double* x, int64_t D, void* ws, int64_t ws_bytes,
                           void* stream);
/* out[row] = <J[row, :], v> for explicit Jacobian rows J (rows = S*C, D entries each): the J v of a narrow node
 * (a conv filter node, tensor/network.py:101-113 with 788) in one pass over J at memory speed. */
int ttk_rows_dot_f64(const double* J, const double* v, double* out, int64_t rows, int D, void* stream);
/* q (D) = J^T H J v for explicit rows J (S, C, D <= 128) and per-sample Hessians H (S, C, C): the operator handed to
 * the Lanczos / Krylov solvers for a narrow node (tensor/network.py:788-789 with the Jacobian of 101-113), one pass
 * over J; the workspace (ttk_rows_matvec_ws_bytes) holds one partial q per workgroup, summed in fixed order. */
int64_t ttk_rows_matvec_ws_bytes(int64_t S, int C, int D);
int ttk_rows_matvec_f64(const double* J, const double* H, const double* v, double* q, int64_t S, int C, int D, void* ws,
                        int64_t ws_bytes, void* stream);
/* b (D) = J^T g = sum_{s,c} g[s,c] J[s,c,:] for explicit rows (get_b, tensor/network.py:258-291, of a narrow node); same
 * kernel, extents and workspace as ttk_rows_matvec_f64. */
int ttk_rows_rhs_f64(const double* J, const double* g, double* b, int64_t S, int C, int D, void* ws, int64_t ws_bytes,
                     void* stream);
/* GL (S, rl) = L_s^T H_s (L_s z_s) per sample, L (S, C, rl <= 128), z (S, rl): the prediction from the contracted
 * environments, the Hessian product and the g.L factor of the right-hand side (the middle of the two einsums at
 * tensor/network.py:788-789 for a site whose features were contracted first) in one kernel. */
int ttk_lhl_f64(const double* L, const double* H, const double* z, double* GL, int64_t S, int C, int rl, void* stream);

/* ---- front-ends: raw inputs -> per-site feature tensors ----
 *
 * ttk_feature_map_f64: the local feature maps of TNMLRegressor for all N sites in one launch
 * (polynomial_basis / fbasis, models/tnml.py:11-23): X (S,N) -> out (N,S,p), out[i] the (S,p) tensor
 * of site i.  kind 0: [x^0 .. x^(p-1)];  kind 1 (p = 2): [cos(pi x/2), sin(pi x/2)].
 *
 * ttk_patchify: k x k patches with the given stride of an image batch (S,Cin,H,W) as rows
 * (S, patches+1, Cin*k*k+1) with a zero last row, a zero last column and a one in the corner:
 * F.unfold(...).transpose(-2,-1) + the two torch.cat + the corner assignment of
 * image_convolution_lanczos_MNIST.py:29-32 / CIFAR_lanczos_swipe.py:35-38.
 * types: 0 float32 -> float32, 1 float64 -> float64, 2 float32 -> float64. */
int ttk_feature_map_f64(const double* X, int64_t S, int N, int kind, int p, double* out, void* stream);
int ttk_patchify(const void* img, int64_t S, int Cin, int H, int W, int k, int stride, int types, void* out, void* stream);

/* ---- sites with many features (conv-featurised / operator columns) and explicit Jacobian rows ----
 *
 * ttk_conv_feature_{f64,f32}: G[ab, row] = sum_x X[row, x] F[x, ab] for rows = (sample, patch): the
 * contraction of the input node with the filter node of a conv column (tensor/layers.py:839-872, done
 * by contract_with inside compute_stacks in the reference).  X (rows, px) float64 or float32 (a float32
 * data set, image_convolution_lanczos_CIFAR10.py, is read as it is), F (px, nab), G (nab, rows).
 *
 * ttk_gemm_rows_f64: T (M,N) (+)= A (M,K) B, rows = samples.  bmode 0: B (K,N) row-major; 1: B(k,(l,r)) =
 * core[l,k,r] with core (rl,K,rr), N = rl*rr: T[s,(l,r)] = sum_p X[s,p] core[l,p,r], the feature
 * contraction done first at a site with many features; 2: B(k,col) = Bt[col,k], Bt (N,K) row-major.
 * K <= 192 (the 64 x K tile of A stays in LDS; 256 for the float32 form), longer K: TTK_EINVAL.
 *
 * ttk_env_right_dense_f64: ttk_env_step_right_f64 for sites with many features (p within the same limit,
 * rr <= 64): the same result by contracting the features first (C p / (C + p) times fewer operations), T
 * never stored.
 *
 * ttk_lcontract_f64: out[s,c,j] (+)= sum_l L[s,c,l] T[s,l,j], T (S,rl,n): second stage of the left
 * environment step of such a site, and M = L . (A . R) of a feature node's Jacobian.
 *
 * ttk_patch_contract_{f64,f32}: J[s,c,x] (+)= sum_p M[s,c,p] X[s,p,x] written at
 * J + s*jstride_s + c*jstride_c + x*xstride: Jacobian rows of a conv filter node (compute_jacobian_stack,
 * tensor/network.py:101-113, for a node C_k).
 *
 * ttk_gemm_tn_f64: Cm (D1,D2) (+)= A^T B with A (n,D1), B (n,D2): the Gram matrix J^T (H J) of a local
 * problem given by explicit rows (get_A_b, tensor/network.py:207,212, for a sum of Khatri-Rao terms).
 *
 * ttk_dense_scale_f64 / ttk_chol_prepare_dense_f64: solve_system's scaling and ridge shift
 * (tensor/network.py:296-316) from a dense (D,D) matrix into the storage of ttk_chol_solve_f64. */
int ttk_conv_feature_f64(const double* X, const double* F, double* G, int64_t rows, int px, int nab, void* stream);
int ttk_conv_feature_f32(const float* X, const double* F, double* G, int64_t rows, int px, int nab, void* stream);
int ttk_gemm_rows_f64(const double* A, const double* B, double* T, int64_t M, int K, int N, int bmode, int rl, int rr,
                      int accumulate, void* stream);
/* 1 if ttk_env_right_dense_f64 / _mm32 take the site: right bonds up to 64 always, up to 128 when the core block of one
 * left-bond index fits the LDS beside the waves' feature tiles (features <= 96). */
int ttk_env_right_dense_ok(int64_t S, int rl, int p, int rr, int mm32);
int ttk_env_right_dense_f64(const double* R, const double* X, const double* core, double* out, int64_t S, int rl, int p,
                            int rr, int accumulate, void* stream);
int ttk_lcontract_f64(const double* L, const double* T, double* out, int64_t S, int C, int rl, int n, int accumulate,
                      void* stream);
/* out[b][s,c,j] = sum_a sum_l L_a[s,c,l] T_ab[s,l,j]: second stage of the left-environment step through a column between
 * two convolution bonds (tensor/layers.py:839-872: the sum over the left channel a), all right channels b in one launch.
 * L (ca,S,C,rl), T (ca,nb,S,rl,n) = the row GEMM G_ab . core of every term, out (nb,S,C,n); ca C rl * 8 <= 48 KiB. */
int ttk_lcontract_sum_f64(const double* L, const double* T, double* out, int64_t S, int C, int rl, int n, int ca, int nb,
                          void* stream);
/* The same with the (S, C, n) result rows inside a wider tensor: row (s, c) starts at out + s ostride_s + c ostride_c
 * (in doubles; ostride_c >= n, ostride_s >= C n), n contiguous.  The Jacobian rows J[s,c,a,(x,b)] of a conv filter
 * node between two convolution bonds (tensor/layers.py:839-872) are written in place, one call per left channel a. */
int ttk_lcontract_strided_f64(const double* L, const double* T, double* out, int64_t S, int C, int rl, int n,
                              int64_t ostride_s, int64_t ostride_c, int accumulate, void* stream);
int ttk_patch_contract_f64(const double* M, const double* X, double* J, int64_t S, int C, int P, int px, int64_t jstride_s,
                           int64_t jstride_c, int64_t xstride, int accumulate, void* stream);
int ttk_patch_contract_f32(const double* M, const float* X, double* J, int64_t S, int C, int P, int px, int64_t jstride_s,
                           int64_t jstride_c, int64_t xstride, int accumulate, void* stream);
/* Z[s,c,x,b] = sum_patch M_b[s,c,patch] X[s,patch,x] for the nb channels b of a convolution bond in one pass over
 * the image rows: M (nb,S,C,P), Z (S,C,px,nb) contiguous, X float64 / float32.  The patch contraction of a conv filter
 * node between convolution bonds (tensor/layers.py:839-872) with c = the left bond index.  The image row and the nb
 * blocks of a sample must fit 64 KiB of LDS (ttk_patch_contract_multi_ok). */
int ttk_patch_contract_multi_ok(int C, int P, int px, int nb);
int ttk_patch_contract_multi_f64(const double* M, const double* X, double* Z, int64_t S, int C, int P, int px, int nb,
                                 void* stream);
int ttk_patch_contract_multi_f32(const double* M, const float* X, double* Z, int64_t S, int C, int P, int px, int nb,
                                 void* stream);
int ttk_gemm_tn_f64(const double* A, const double* B, double* Cm, int64_t n, int D1, int D2, int accumulate, void* stream);

/* Right-hand side / J^T u of a site with many features (16 <= p <= 256, rr <= 128, p + rr <= 280: two chunk
 * images in LDS; other extents return TTK_EINVAL and belong to ttk_rhs_f64; a conv column has one feature per patch):
 *   b (rl,p,rr) (+)= sum_s GL[s,l] X[s,pp] R[s,r],   GL (S,rl) = the gradient-weighted left factor (ttk_gl_f64)
 * i.e. the einsum of get_A_b / get_b (tensor/network.py:205-217, 258-291) after the conv feature contraction
 * (tensor/layers.py:839-872), features first.  ws: ttk_rhs_feature_ws_bytes(S, rl, p, rr) bytes of slab partials. */
int64_t ttk_rhs_feature_ws_bytes(int64_t S, int rl, int p, int rr);
int ttk_rhs_feature_f64(const double* GL, const double* X, const double* R, double* b, int64_t S, int rl, int p, int rr,
                        int accumulate, void* ws, int64_t ws_bytes, void* stream);

/* A train column between two convolution bonds (image_convolution_lanczos.py's CB > 0: the neighbouring filters of
 * tensor/layers.py:839-872 leave ca channels on the left and cb on the right): its Jacobian is the sum of ca cb
 * Khatri-Rao terms (L_a, G_ab, R_b), i.e. the contraction tensor/network.py:205-217, 258-291 carries out over the
 * channel indices as well.  Small bonds (rl, rr <= 16), ca, cb <= 4, p <= 128, C <= 256 and p * 4 ceil(rr / 4) * rl * 8 <= 65536
 * (ttk_bond_ok); one launch per half of the operator whatever ca cb is.  Layouts, all contiguous:
 *   L (ca,S,C,rl)   G (ca,cb,S,p)   R (cb,S,rr)   v, b (rl,p,rr)   u, g (S,C)   GL (ca,S,rl)
 *   ttk_bond_jv_f64:   u[s,c] (+)= sum_{a,b} sum_{l,pp,r} L_a[s,c,l] G_ab[s,pp] v[l,pp,r] R_b[s,r]
 *   ttk_bond_gl_f64:   GL_a[s,l] = sum_c g[s,c] L_a[s,c,l]
 *   ttk_bond_rhs_f64:  b[l,pp,r] (+)= sum_s sum_{a,b} GL_a[s,l] G_ab[s,pp] R_b[s,r]
 * ws: ttk_bond_jv_ws_bytes(S, rl, ca) bytes for z_a[s,l] = sum_{b,pp,r} G_ab[s,pp] v[l,pp,r] R_b[s,r] (ttk_bond_jv_f64:
 * two launches, z then u), ttk_bond_rhs_ws_bytes(S, rl, p, rr) bytes of slab partials, added in fixed order
 * (ttk_bond_rhs_f64). */
int ttk_bond_ok(int64_t S, int C, int rl, int p, int rr, int ca, int cb);
int64_t ttk_bond_jv_ws_bytes(int64_t S, int rl, int ca);
int ttk_bond_jv_f64(const double* L, const double* G, const double* R, const double* v, double* u, int64_t S, int C, int rl,
                    int p, int rr, int ca, int cb, int accumulate, void* ws, int64_t ws_bytes, void* stream);
int ttk_bond_gl_f64(const double* g, const double* L, double* GL, int64_t S, int C, int rl, int ca, void* stream);
/* The two halves of the operator v -> J^T H J v of such a column (tensor/network.py:788-789) meet in the left bond:
 *   ttk_bond_z_f64:    z (ca,S,rl),  z_a[s,l] = sum_{b,pp,r} G_ab[s,pp] v[l,pp,r] R_b[s,r]   (first launch of ttk_bond_jv_f64)
 *   ttk_bond_lhl_f64:  GL_a[s,l] = sum_c hu[s,c] L_a[s,c,l],  hu[s,d] = sum_c H[s,c,d] u[s,c],  u = sum_{a,l} L_a z_a
 * (prediction, Hessian product and g . L in one kernel: the left environments are read once); ttk_bond_rhs_f64
 * finishes.  H (S,C,C); C <= 256, ca rl <= 256. */
int ttk_bond_z_f64(const double* G, const double* R, const double* v, double* z, int64_t S, int rl, int p, int rr, int ca, int cb,
                   void* stream);
int ttk_bond_lhl_f64(const double* L, const double* H, const double* z, double* GL, int64_t S, int C, int rl, int ca,
                     void* stream);
int64_t ttk_bond_rhs_ws_bytes(int64_t S, int rl, int p, int rr);
int ttk_bond_rhs_f64(const double* GL, const double* G, const double* R, double* b, int64_t S, int rl, int p, int rr, int ca,
                     int cb, int accumulate, void* ws, int64_t ws_bytes, void* stream);

/* A conv filter node between convolution bonds (its tensor F[a,x,b]; tensor/layers.py:839-872) in factored form.
 * Its Jacobian rows are J[s,c,(a,k)] = sum_l L_a[s,c,l] Z[s,l,k] with k = (x,b) and
 * Z[s,l,(x,b)] = sum_patch (A . R_b)[s,l,patch] X[s,patch,x] (ttk_gemm_rows + ttk_patch_contract); the products of an
 * iterative solve (tensor/network.py:788-789 for this node) stream L and Z (ca C rl + rl n numbers per sample)
 * instead of the C ca n numbers of the rows.  L (ca,S,C,rl), Z (S,rl,n), H (S,C,C), v / q / b (ca,n), u / g (S,C):
 *   ttk_filter_matvec_f64:  q = J^T H J v        ttk_filter_jv_f64:  u = J v        ttk_filter_rhs_f64:  b = J^T g
 * Extents: C <= 64, ca <= 4, rl <= 16, n <= 256 and one sample's factors within the LDS of a wave
 * (ttk_filter_ok).  ws: ttk_filter_ws_bytes bytes of workgroup partial sums (added in fixed order). */
int ttk_filter_ok(int64_t S, int C, int rl, int n, int ca);
int64_t ttk_filter_ws_bytes(int64_t S, int C, int rl, int n, int ca);
int ttk_filter_matvec_f64(const double* L, const double* Z, const double* H, const double* v, double* q, int64_t S, int C,
                          int rl, int n, int ca, void* ws, int64_t ws_bytes, void* stream);
int ttk_filter_jv_f64(const double* L, const double* Z, const double* v, double* u, int64_t S, int C, int rl, int n, int ca,
                      void* stream);
int ttk_filter_rhs_f64(const double* L, const double* Z, const double* g, double* b, int64_t S, int C, int rl, int n, int ca,
                       void* ws, int64_t ws_bytes, void* stream);

/* float32 matrix arithmetic for float32 models (image_convolution_lanczos_CIFAR10.py and CIFAR_lanczos_swipe.py
 * never set a dtype, so the reference contracts in float32: tensor/network.py:101-113, 788-789 on float32
 * nodes).  Same arguments, layouts and memory types (float64) as the entry points they are named after; the
 * operands are rounded to float32 when they are staged, products and running sums are float32
 * (v_mfma_f32_16x16x4_f32, twice the fp64 matrix rate), partial sums over sample slabs and the row dot of
 * ttk_env_right_dense are added in float64.  Results agree with the _f64 entry points to float32 rounding. */
int ttk_gemm_rows_mm32(const double* A, const double* B, double* T, int64_t M, int K, int N, int bmode, int rl, int rr,
                       int accumulate, void* stream);
int ttk_env_right_dense_mm32(const double* R, const double* X, const double* core, double* out, int64_t S, int rl, int p,
                             int rr, int accumulate, void* stream);
int ttk_rhs_feature_mm32(const double* GL, const double* X, const double* R, double* b, int64_t S, int rl, int p, int rr,
                         int accumulate, void* ws, int64_t ws_bytes, void* stream);

int ttk_dense_scale_f64(const double* A, int D, double* scale, void* stream);
int ttk_chol_prepare_dense_f64(const double* A, int D, const double* scale, double ridge, const double* b, const double* w,
                               void* chol_ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTK_H */
