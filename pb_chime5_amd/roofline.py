"""Algorithmic work per kernel launch (SURVEY.md section 8d figures) and the
MI355X peaks they are priced against.  Used by bench.py for the ``roofline``
object and documented in DESIGN.md.

Device arithmetic is float64 / complex128 (16 bytes per STFT bin), so the byte
figures of SURVEY.md section 8d (written for complex64) are doubled:
``B_Y = 16 * F * T * D``.
"""

# MI355X peaks.  HBM3E: /opt/skills/guides/MI355X_MICROARCH.md (8 TB/s spec).
# FP64: AMD MI355X datasheet, 78.6 TFLOP/s for both vector and matrix (f64 MFMA
# issues at 64 cycles per v_mfma_f64_16x16x4_f64 per SIMD: 2048 flop / 64 cyc
# * 1024 SIMDs * 2.4 GHz = 78.6e12); the guide lists no f64 row.
PEAK_HBM_GBS = 8000.0
PEAK_F64_TFLOPS = 78.6


def stft_bin_bytes(F, T, D):
    return 16.0 * F * T * D


def kernel_work(name, *, F, T, D, K, taps, N):
    """Algorithmic (flops, bytes) of ONE launch of kernel ``name`` at the given
    problem size, and the roof that bounds it."""
    BY = stft_bin_bytes(F, T, D)
    n = taps * D
    if name == 'wpe_corr':
        # R = (Yt w) Yt^H : 8 n^2 T ;  P = (Yt w) Y^H : 8 n D T   per frequency (the
        # dense count of SURVEY 8d).  The kernel exploits R = R^H: of its 32 x 32 wave tiles
        # it computes the 16 x 16 sub-tiles that hold an entry of the upper triangle (or of
        # P), `executed` of the dense count ...
        sub = -(-n // 16)
        subtiles = sub * (sub + 1) // 2 + sub * -(-D // 16)
        # ... and forms each complex product with 3 real MFMAs instead of 4 (x 0.75).
        executed = 0.75 * subtiles * 256 / float(n * n + n * D)
        return dict(flops=F * (8.0 * n * n * T + 8.0 * n * D * T), bytes=BY + 8.0 * F * T,
                    bound='mfma', executed=executed)
    if name == 'wpe_solve':
        return dict(flops=F * ((8.0 / 3.0) * n ** 3 + 8.0 * n * n * D),
                    bytes=16.0 * F * (n * n + 2 * n * D), bound='mfma')
    if name == 'wpe_apply':
        return dict(flops=F * 8.0 * D * n * T, bytes=2 * BY + 16.0 * F * n * D, bound='mfma')
    if name == 'wpe_power':
        return dict(flops=F * T * 3.0 * D, bytes=BY + 8.0 * F * T, bound='hbm')
    if name == 'em_estep':
        # per (f, t, k): 8 D^2 (quadratic form  y^H B_k^-1 y)
        return dict(flops=8.0 * D * D * K * F * T, bytes=BY + 8.0 * F * K * T, bound='valu_f64')
    if name == 'em_mstep':
        # per (f, t, k): 4 D^2 (Hermitian outer-product accumulate)
        return dict(flops=4.0 * D * D * K * F * T, bytes=BY + 8.0 * F * K * T, bound='valu_f64')
    if name == 'em_predict':
        return dict(flops=8.0 * D * D * K * F * T, bytes=BY + 8.0 * F * K * T, bound='valu_f64')
    if name in ('em_eig', 'em_chol'):
        # Jacobi eigh ~ 2e6 flop at D = 24, scaled ~ D^3
        return dict(flops=F * K * 2.0e6 * (D / 24.0) ** 3, bytes=16.0 * F * K * D * D * 2,
                    bound='valu_f64')
    if name == 'psd':
        return dict(flops=F * T * 2 * 4.0 * D * D, bytes=BY + 16.0 * F * T, bound='hbm')
    if name == 'mvdr_apply':
        return dict(flops=F * T * 8.0 * D, bytes=BY + 16.0 * F * T, bound='hbm')
    if name == 'stft':
        return dict(flops=D * T * 2.5 * 1024 * 10, bytes=8.0 * D * N + BY, bound='hbm')
    if name in ('istft_frames', 'istft_ola'):
        return dict(flops=T * 2.5 * 1024 * 10, bytes=16.0 * F * T + 8.0 * N, bound='hbm')
    if name == 'masks':
        return dict(flops=F * T * K, bytes=8.0 * F * T * (K + 2), bound='hbm')
    return None


def roofline_entry(name, avg_ms, **size):
    """The ``roofline`` object of the bench line for kernel ``name`` whose launches
    took ``avg_ms`` on average (HIP events on the launch stream)."""
    w = kernel_work(name, **size)
    if w is None or avg_ms <= 0:
        return None
    sec = avg_ms * 1e-3
    if w['bound'] == 'hbm':
        achieved = w['bytes'] / sec / 1e9
        peak, unit = PEAK_HBM_GBS, 'GB/s'
    else:
        achieved = w['flops'] / sec / 1e12
        peak, unit = PEAK_F64_TFLOPS, 'TFLOP/s'
    out = {'kernel': name, 'bound': w['bound'], 'achieved': achieved, 'peak': peak,
           'unit': unit, 'frac': achieved / peak, 'traffic': None,
           'avg_launch_ms': avg_ms, 'algorithmic_flops_per_launch': w['flops'],
           'algorithmic_bytes_per_launch': w['bytes']}
    if 'executed' in w:
        out['executed_over_algorithmic'] = w['executed']
        out['frac_executed'] = out['frac'] * w['executed']
        out['note'] = ('algorithmic = dense count of SURVEY 8d (8 flop per complex MAC); the kernel '
                       'computes only the Hermitian upper triangle and uses 3 real products per '
                       'complex one, so frac can exceed 1; frac_executed prices the MFMA flops '
                       'actually issued against the same peak')
    return out
