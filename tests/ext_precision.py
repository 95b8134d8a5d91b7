"""Test infrastructure: the beamformer formulas of pb_bss (call sites
/root/reference/pb_chime5/speech_enhancement/beamforming_wrapper.py:49-75) and the WPE iteration
of nara_wpe (call site core.py:48-58) evaluated in 80-bit
extended precision, one frequency at a time -- the referee for frequency bins where the
float64 evaluation of those formulas (the reference's, hence the oracle's) is itself decided by
rounding: with a nearly singular noise PSD matrix np.linalg.solve loses cond(Phi_N) * eps and
blind_analytic_normalization's four-operand einsum cancels catastrophically once
cond(Phi_N)^2 * eps > 1.  Same formulas, ~3 more decimal digits, own linear algebra
(Gaussian elimination with partial pivoting below; numpy.linalg has no longdouble path)."""
import numpy as np

LD, CLD = np.longdouble, np.clongdouble


def psd(Y, mask):
    """get_power_spectral_density_matrix for one frequency: Y (D, T) complex128, mask (T,)
    -> (D, D) clongdouble, normalised by max(sum mask, 1e-10)."""
    Yl = Y.astype(CLD)
    m = mask.astype(LD)
    m = m / max(m.sum(), LD(1e-10))
    return (Yl * m) @ Yl.conj().T


def solve(A, B):
    """A X = B by Gaussian elimination with partial pivoting, clongdouble."""
    A = A.astype(CLD).copy()
    B = B.astype(CLD).copy()
    n = A.shape[0]
    for j in range(n):
        p = j + int(np.argmax(np.abs(A[j:, j])))
        if p != j:
            A[[j, p]] = A[[p, j]]
            B[[j, p]] = B[[p, j]]
        f = A[j + 1:, j] / A[j, j]
        A[j + 1:] -= f[:, None] * A[j]
        B[j + 1:] -= f[:, None] * B[j]
    X = np.zeros_like(B)
    for j in range(n - 1, -1, -1):
        X[j] = (B[j] - A[j, j + 1:] @ X[j + 1:]) / A[j, j]
    return X


def souden_matrix(cov_x, cov_n, eps=1e-10):
    """Phi_N^-1 Phi_X / max(trace, eps): the columns are the MVDR vectors per reference
    channel (get_mvdr_vector_souden)."""
    phi = solve(cov_n, cov_x)
    return phi / max(np.trace(phi).real, LD(eps))


def snr_terms(mat, cov_x, cov_n):
    """Numerator and denominator of get_optimal_reference_channel's SNR for every candidate
    channel, for one frequency (the reference sums both over the frequencies first)."""
    num = np.real(np.einsum('dr,de,er->r', mat.conj(), cov_x, mat))
    den = np.real(np.einsum('dr,de,er->r', mat.conj(), cov_n, mat))
    return num, den


def ban(w, cov_n):
    """blind_analytic_normalization (eps = 0): w sqrt(w^H Phi_N Phi_N w) / |w^H Phi_N w|."""
    pw = cov_n @ w
    return w * np.sqrt(np.real(np.vdot(pw, pw))) / np.abs(np.vdot(w, pw))


def mvdr_souden_ban_output(Y, target_mask, distortion_mask, ref_channel):
    """X_hat (T,) of one frequency for a given reference channel, complex128."""
    cov_x, cov_n = psd(Y, target_mask), psd(Y, distortion_mask)
    w = ban(souden_matrix(cov_x, cov_n)[:, ref_channel], cov_n)
    return (w.conj() @ Y.astype(CLD)).astype(np.complex128)


def gev_ban_output(Y, target_mask, distortion_mask):
    """beamform_gev_from_masks (beamforming_wrapper.py:77-89,192-208) for one frequency: the
    principal generalised eigenvector of (Phi_X, Phi_N) -- float64 scipy.linalg.eigh as the
    starting point, refined by Rayleigh quotient iteration in extended precision -- with
    v^H Phi_N v = 1, BAN, apply.  The phase is arbitrary (compare magnitudes)."""
    import scipy.linalg
    cov_x, cov_n = psd(Y, target_mask), psd(Y, distortion_mask)
    D = cov_x.shape[0]
    try:
        v = scipy.linalg.eigh(cov_x.astype(np.complex128), cov_n.astype(np.complex128))[1][:, -1]
    except np.linalg.LinAlgError:           # Phi_N not positive definite in float64
        lam, V = np.linalg.eig(np.linalg.solve(cov_n.astype(np.complex128) + 0, cov_x.astype(np.complex128)))
        v = V[:, int(np.argmax(lam.real))]
    v = v.astype(CLD)
    for _ in range(4):
        rho = np.real(np.vdot(v, cov_x @ v)) / np.real(np.vdot(v, cov_n @ v))
        z = solve(cov_x - rho * cov_n + LD(1e-30) * np.trace(cov_x).real * np.eye(D), (cov_n @ v)[:, None])[:, 0]
        v = z / np.sqrt(np.real(np.vdot(z, z)))
    v = v / np.sqrt(np.real(np.vdot(v, cov_n @ v)))
    w = ban(v, cov_n)
    return (w.conj() @ Y.astype(CLD)).astype(np.complex128)


def cholesky_solve(R, P):
    """Hermitian positive definite solve in extended precision, one refinement step."""
    n = R.shape[0]
    L = np.zeros_like(R)
    for j in range(n):
        d = np.sqrt((R[j, j] - np.sum(np.abs(L[j, :j]) ** 2)).real)
        L[j, j] = d
        L[j + 1:, j] = (R[j + 1:, j] - L[j + 1:, :j] @ L[j, :j].conj()) / d
    LH = L.conj().T

    def solve_(B):
        Z = np.zeros_like(B)
        for j in range(n):
            Z[j] = (B[j] - L[j, :j] @ Z[:j]) / L[j, j]
        G = np.zeros_like(B)
        for j in range(n - 1, -1, -1):
            G[j] = (Z[j] - LH[j, j + 1:] @ G[j + 1:]) / LH[j, j]
        return G

    G = solve_(P)
    return G + solve_(P - R @ G)


def wpe(Y, Y_tilde, iterations):
    """nara_wpe.wpe.wpe_v6 (statistics_mode='full', psd_context=0; call site
    /root/reference/pb_chime5/core.py:48-58) for one frequency in extended precision:
    Y (D, T) complex128, Y_tilde (taps D, T) its delayed stack -> list of X after every
    iteration, rounded to complex128."""
    Yl, Yt = Y.astype(CLD), Y_tilde.astype(CLD)
    X = Yl.copy()
    out = []
    for _ in range(iterations):
        power = np.mean(X.real ** 2 + X.imag ** 2, axis=0)
        w = 1 / np.maximum(power, LD(1e-10) * np.max(power))
        Yw = Yt * w[None, :]
        G = cholesky_solve(Yw @ Yt.conj().T, Yw @ Yl.conj().T)
        X = Yl - G.conj().T @ Yt
        out.append(X.astype(np.complex128))
    return out


def _bin_job(job):
    Y, tm, dm, ref, bf = job
    if bf == 'gev_ban':
        return gev_ban_output(Y, tm, dm)
    return mvdr_souden_ban_output(Y, tm, dm, ref)


def beamformer_all_bins(Obs, target_mask, distortion_mask, ref_channel, bf='mvdrSouden_ban',
                        workers=8):
    """Obs (D, T, F), masks (T, F) -> X_hat (T, F) in extended precision, the frequencies
    spread over worker processes."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    F = Obs.shape[-1]
    jobs = [(np.ascontiguousarray(Obs[..., f]), np.ascontiguousarray(target_mask[:, f]),
             np.ascontiguousarray(distortion_mask[:, f]), ref_channel, bf) for f in range(F)]
    with ProcessPoolExecutor(workers, mp_context=mp.get_context('spawn')) as ex:
        return np.array(list(ex.map(_bin_job, jobs, chunksize=8))).T


# ---------------------------------------------------------------- guided CACGMM EM
def eigh(A, max_sweeps=60):
    """Eigendecomposition of one Hermitian matrix in extended precision by cyclic Jacobi
    (numpy.linalg has no longdouble path): eigenvalues (n,) longdouble, eigenvectors as the
    columns of V (n, n) clongdouble.  Pairs are rotated until every off-diagonal entry is below
    eps_longdouble of the larger of its two diagonal entries (or of 1e-30 of the largest
    diagonal entry: pairs inside the rounding noise of the matrix)."""
    A = np.array(A, dtype=CLD)
    n = A.shape[0]
    A = (A + A.conj().T) / 2
    V = np.eye(n, dtype=CLD)
    eps = np.finfo(LD).eps
    for _ in range(max_sweeps):
        d = np.abs(A.diagonal().real)
        scale = np.maximum(np.maximum(d[:, None], d[None, :]), LD(1e-30) * max(d.max(), LD(1e-300)))
        off = np.abs(A) / scale
        np.fill_diagonal(off, 0)
        if off.max() <= 4 * eps:
            break
        for p in range(n - 1):
            for q in range(p + 1, n):
                b = A[p, q]
                babs = np.abs(b)
                if babs == 0:
                    continue
                tau = (A[q, q].real - A[p, p].real) / (2 * babs)
                t = (LD(1) if tau >= 0 else LD(-1)) / (np.abs(tau) + np.sqrt(1 + tau * tau))
                c = 1 / np.sqrt(1 + t * t)
                s = (t * c) * (b / babs)
                for M in (A, V):                       # columns: M <- M J
                    mp, mq = M[:, p].copy(), M[:, q].copy()
                    M[:, p] = c * mp - np.conj(s) * mq
                    M[:, q] = s * mp + c * mq
                ap, aq = A[p, :].copy(), A[q, :].copy()   # rows: A <- J^H A
                A[p, :] = c * ap - s * aq
                A[q, :] = np.conj(s) * ap + c * aq
                A[p, q] = A[q, p] = 0
                A[p, p], A[q, q] = A[p, p].real, A[q, q].real
    return A.diagonal().real.copy(), V


def guided_em(obs, activity, iterations, iterations_post=1):
    """GSS.__call__ (/root/reference/pb_chime5/core.py:154-214: CACGMMTrainer.fit guided by the
    activity, fit without the mask for iterations_post - 1, predict) for ONE frequency in 80-bit
    extended precision: obs (T, D) complex128, activity (K, T) bool -> posteriors (K, T) float64.
    Same formulas as the float64 implementations -- unit-norm observation, gamma / max(q, 10 tiny)
    weights, D sum / max(sum gamma, tiny), eigenvalues / max floored at 1e-10, clip 1e-10 in fit
    and none in predict -- with ~3 more decimal digits and its own eigensolver: the referee for
    EM runs that hinge on the floor (classes with fewer active frames than channels), where two
    float64 programs drift apart by more than either is off the exact iteration."""
    tiny = LD(np.finfo(np.float64).tiny)
    obs = np.asarray(obs)
    T, D = obs.shape
    activity = np.asarray(activity, bool)
    K = activity.shape[0]
    y = obs.astype(CLD)
    nrm = np.sqrt(np.sum(y.real ** 2 + y.imag ** 2, axis=1))
    y = y / np.where(nrm == 0, tiny, nrm)[:, None]                       # (T, D)

    def m_step(gamma, quad):
        model = []
        for k in range(K):
            wgt = gamma[k] / np.maximum(quad[k], 10 * tiny)
            B = (y * wgt[:, None]).T @ y.conj()                            # sum_t w y y^H
            B = B * (LD(D) / max(gamma[k].sum(), tiny))
            lam, V = eigh(B)
            lam = np.maximum(lam / max(lam.max(), tiny), LD(1e-10))
            model.append((lam, V, gamma[k].sum() / LD(T)))
        return model

    def e_step(model, mask, clip):
        lp = np.zeros((K, T), LD)
        quad = np.zeros((K, T), LD)
        pis = np.zeros(K, LD)
        for k, (lam, V, pi) in enumerate(model):
            proj = y.conj() @ V                                            # (T, D): conj(v_j^H y_t)
            quad[k] = np.maximum(np.abs(np.sum((proj.real ** 2 + proj.imag ** 2) / lam[None, :],
                                               axis=1)), tiny)
            lp[k] = -LD(D) * np.log(quad[k]) - np.sum(np.log(lam))
            pis[k] = pi
        g = np.exp(lp - lp.max(axis=0)) * pis[:, None]
        if mask is not None:
            g = g * mask
        g = g / np.maximum(g.sum(axis=0), tiny)
        if clip:
            g = np.clip(g, LD(clip), 1 - LD(clip))
        return g, quad

    gamma = np.where(activity, LD(1), LD(1e-10))
    gamma = gamma / gamma.sum(axis=0)
    quad = np.ones((K, T), LD)
    model = None
    for _ in range(iterations):
        if model is not None:
            gamma, quad = e_step(model, activity, 1e-10)
        model = m_step(gamma, quad)
    for _ in range(max(iterations_post - 1, 0)):
        gamma, quad = e_step(model, None, 1e-10)
        model = m_step(gamma, quad)
    return e_step(model, activity if iterations_post == 0 else None, 0)[0].astype(np.float64)


def em_yardstick(obs_f, activity, iterations, iterations_post, oracle_posteriors, referee, draws=4):
    """What the reference's own float64 arithmetic leaves undecided for one frequency of a
    guided EM: the larger of (oracle - referee) and the oracle's movement when every input
    sample is changed in its last bit (`draws` random sign patterns).  pb_bss forms
    B^-1 = V diag(1 / lambda) V^H explicitly; with eigenvalues on the 1e-10 floor its entries are
    1e10 and q = y^H B^-1 y cancels to 1e10 eps = 1e-6: one float64 run (the oracle) is ONE
    sample of that noise and may happen to land 1e-9 from the exact iteration where the next
    one lands 5e-7 away.  obs_f (D, T, 1) complex128; returns the yardstick (a float)."""
    import gss_oracle as oracle
    rng = np.random.default_rng(20260929)
    eps = np.finfo(np.float64).eps
    d = float(np.max(np.abs(oracle_posteriors - referee)))
    for _ in range(draws):
        pert = obs_f.real * (1 + eps * rng.choice([-1.0, 1.0], size=obs_f.shape)) \
            + 1j * obs_f.imag * (1 + eps * rng.choice([-1.0, 1.0], size=obs_f.shape))
        moved = oracle.gss_block_batched(pert, activity, iterations=iterations,
                                         iterations_post=iterations_post)[..., 0]
        d = max(d, float(np.max(np.abs(moved - oracle_posteriors))))
    return d
