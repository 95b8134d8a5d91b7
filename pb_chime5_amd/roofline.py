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
# What v_mfma_f64_16x16x4_f64 sustains on this silicon, MEASURED (tools/micro/mfma_f64_bench.hip,
# profiles/r04b_mfma_f64_bench.txt): 12 independent MFMAs per k-step alone issue every 65.0 cycles
# per SIMD (77.6 TFLOP/s at the 2.40 GHz the kernel holds); with the instruction mix of the
# wpe_corr k-step around them -- 8 f64 VALU operand products and 5 LDS operand reads per 12
# MFMAs -- every 68.3 cycles (73.5 TFLOP/s): f64 VALU instructions run on the lanes the f64 MFMA
# uses, so they are not hidden behind it however many waves share the SIMD.
MEASURED_F64_MFMA_TFLOPS = 77.6
MEASURED_WPE_CORR_MIX_TFLOPS = 73.5
# Scalar data cache: what one CU can feed its waves through s_load_dwordx16, MEASURED
# (tools/micro/smem_bench.hip, profiles/r03_smem_bench.txt: 16 waves per CU streaming 64-byte
# lines that hit the cache, four requests per wait: 122.7 loads / us / CU = 3.27 bytes per cycle
# per CU at 2.4 GHz; one request per wait: 2.67).  The register-form E-step takes its model
# rows -- wave-uniform operands -- down this path, 16 F NE K bytes per 64-frame wave.  It is
# NOT what binds that kernel (with the scalar loads removed it runs 10 % faster, no more;
# DESIGN.md section 8, experiment 11): reported for completeness.
PEAK_SMEM_BYTES_PER_CYCLE_PER_CU = 3.27
NUM_CUS = 256
NUM_SIMDS = 1024
CLOCK_GHZ = 2.4
# VALU issue: a wave64 f64 instruction (v_fma_f64 / v_mul_f64 / v_add_f64, VGPR or SGPR
# operands alike) occupies its SIMD for 4 cycles -- MEASURED 4.3 with two or more waves per
# SIMD, 5.2 with one (tools/micro/valu_f64_bench.hip, profiles/r03_valu_f64_bench.txt), at a
# shader clock that settles at 1.9 - 2.3 GHz under f64 load, not the nominal 2.4.
VALU_F64_ISSUE_CYCLES = 4.0


def estep_scalar_bytes(F, T, D, K):
    """Bytes of model rows the register-form E-step pulls through the scalar cache per launch:
    every wave (64 frames) reads the whole model of its frequency."""
    return 16.0 * (D * (D + 1) // 2) * K * F * -(-T // 64)


def estep_valu_instructions(F, T, D, K):
    """f64 VALU instructions of the triangle walk of the register-form E-step per launch
    (wave64 instructions): per packed entry 4 for Re / Im (y_d1 conj y_d2) and 2 K for the K
    quadratic forms, for every 64-frame wave.  The softmax epilogue (log, exp, division per
    class and frame, ~10 % more at D = 24) is not counted."""
    return (D * (D + 1) // 2) * (4.0 + 2.0 * K) * F * -(-T // 64)


def stft_bin_bytes(F, T, D):
    return 16.0 * F * T * D


def kernel_work(name, *, F, T, D, K, taps, N, iterations=20):
    """Algorithmic (flops, bytes) of ONE launch of kernel ``name`` at the given
    problem size, and the roof that bounds it."""
    BY = stft_bin_bytes(F, T, D)
    n = taps * D
    if name == 'wpe_corr':
        # Needed complex MACs per frame: the upper triangle of R = (Yt w) Yt^H incl. the
        # diagonal, n(n+1)/2, plus P = (Yt w) Y^H, n D.  A complex MAC costs 3 real
        # multiply-adds (Karatsuba / "3M") = 6 flop.  This MINIMUM is what `frac` prices
        # (<= 1 by construction).  SURVEY 8d's dense count (full n x n, 8 flop per
        # complex MAC) is kept as `dense_flops`; what the kernel really issues
        # (16 x 16 sub-tiles that touch the triangle, 3 MFMAs per complex product) as
        # `executed_flops`.
        need = n * (n + 1) // 2 + n * D
        sub = -(-n // 16)
        subtiles = sub * (sub + 1) // 2 + sub * -(-D // 16)
        c = (n // D) + 1 if D else 0          # delay 2: c = taps + 1 frames back
        if D <= 12 and (c * D) // 16 == (c * D + D - 1) // 16 == sub - 1:
            subtiles = sub * (sub + 1) // 2   # P sits in R's last column tile (one array)
        # frames the MFMAs run over: chunks of 64 (16 k-steps); the persistent 32 x 32 kernel
        # (more than 48 sub-tiles) runs a last chunk of at most 48 frames in groups of 16
        rem = T % 64
        t_exec = T - rem + (0 if rem == 0 else 64 if rem > 48 or subtiles <= 48
                            else 16 * -(-rem // 16))
        # bytes: the observation and the frame weights in, the needed entries of R and P out
        return dict(flops=F * 6.0 * need * T, bytes=BY + 8.0 * F * T + 16.0 * F * need,
                    bound='mfma',
                    dense_flops=F * (8.0 * n * n * T + 8.0 * n * D * T),
                    executed_flops=F * 6.0 * subtiles * 256 * t_exec)
    if name == 'wpe_solve':
        return dict(flops=F * ((8.0 / 3.0) * n ** 3 + 8.0 * n * n * D),
                    bytes=16.0 * F * (n * n + 2 * n * D), bound='mfma')
    if name == 'wpe_apply':
        return dict(flops=F * 8.0 * D * n * T, bytes=2 * BY + 16.0 * F * n * D, bound='mfma')
    if name == 'wpe_power':
        return dict(flops=F * T * 3.0 * D, bytes=BY + 8.0 * F * T, bound='hbm')
    if name in ('em_estep', 'em_predict', 'em_mstep'):
        # Hermitian form: per frame and packed upper-triangle entry
        # (NE = D(D+1)/2) one complex product y_d conj(y_e) (2 mul + 2 fma = 6 flop)
        # and 2 real FMAs per class (E: Re/Im of the model entry; M: Re/Im of the
        # accumulator) = 4 K flop.  That is what is executed AND the minimum for this
        # formulation; the dense quadratic-form count of SURVEY 8d (8 D^2 K for the
        # E-step, 4 D^2 K for the M-step) is kept as `dense_flops`.
        NE = D * (D + 1) // 2
        dense = (4.0 if name == 'em_mstep' else 8.0) * D * D * K * F * T
        return dict(flops=F * T * NE * (6.0 + 4.0 * K), bytes=BY + 8.0 * F * K * T,
                    bound='valu_f64', dense_flops=dense)
    if name == 'em_onchip':
        # one array: all iterations + predict in one launch; per pass the E- and M-step flops of
        # the Hermitian form (the predict pass has no M-step: counted as a full pass, < 3 % off)
        NE = D * (D + 1) // 2
        passes = iterations + 1
        return dict(flops=passes * 2 * F * T * NE * (6.0 + 4.0 * K), bytes=passes * BY,
                    bound='valu_f64')
    if name in ('em_eig', 'em_chol'):
        # Cholesky factor + inverse + B^-1 = W^H W of a D x D Hermitian matrix:
        # ~ (8/3 + 8/3 + 8/3) D^3 real flop, plus the chunk reduction
        return dict(flops=F * K * 8.0 * D ** 3, bytes=16.0 * F * K * D * D * 2,
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
    if 'dense_flops' in w:
        out['frac_dense_count'] = w['dense_flops'] / sec / 1e12 / peak
    if 'executed_flops' in w:
        out['frac_executed'] = w['executed_flops'] / sec / 1e12 / peak
        out['executed_flops_per_launch'] = w['executed_flops']
        if name == 'wpe_corr':
            out['frac_executed_of_measured_ceiling'] = (w['executed_flops'] / sec / 1e12
                                                        / MEASURED_WPE_CORR_MIX_TFLOPS)
            out['measured_ceiling_note'] = (
                f'{MEASURED_WPE_CORR_MIX_TFLOPS} TFLOP/s: what back-to-back f64 MFMAs sustain with this '
                'kernel\'s 8 f64 operand products + 5 LDS reads per 12 MFMAs around them '
                f'({MEASURED_F64_MFMA_TFLOPS} alone); tools/micro/mfma_f64_bench.hip, '
                'profiles/r04b_mfma_f64_bench.txt')
    if 'dense_flops' in w:
        out['note'] = ('frac prices the minimum real flops of the formulation (Hermitian upper '
                       'triangle, 3 real products per complex one) and is <= 1 by construction; '
                       'frac_dense_count prices SURVEY 8d\'s dense count (8 flop per complex MAC, '
                       'full matrix) and can exceed 1; frac_executed prices the flops the kernel '
                       'issues (tile-granularity waste included)')
    return out


def utterance_flops(*, F, T, D, K, taps, N, wpe_iterations=3, iterations=20, wpe=True,
                    executed=True):
    """Real float64 operations the pipeline EXECUTES for one utterance (the sum of
    `executed_flops`, else `flops`, of `kernel_work` over the launches of one step): what the
    whole step is priced with against the chip's f64 peak (`f64_peak_frac` of the bench line).
    Kernels without a flop model (activity, masks, layout) do not count."""
    size = dict(F=F, T=T, D=D, K=K, taps=taps, N=N, iterations=iterations)

    def fl(name):
        w = kernel_work(name, **size)
        return w.get('executed_flops', w['flops']) if executed else w['flops']
    total = fl('stft') + fl('istft_frames') + fl('psd') + fl('mvdr_apply')
    if wpe:
        total += wpe_iterations * (fl('wpe_power') + fl('wpe_corr') + fl('wpe_solve') + fl('wpe_apply'))
    if D == 4 and 2 <= K <= 6:
        total += fl('em_onchip')
    else:
        total += iterations * (fl('em_estep') + fl('em_mstep') + fl('em_chol')) + fl('em_predict')
    return total


def step_peak_frac(ms_per_step, **shape):
    """{'executed_gflop_per_step', 'frac', ...}: executed flops of a step / its duration / f64
    peak; `frac_min_flops` prices the minimum of the formulation instead (no tile-granularity
    waste: with one array the 40 unknowns on 16 x 16 MFMA tiles execute 1.57 x the products)."""
    flops = utterance_flops(**shape)
    least = utterance_flops(executed=False, **shape)
    sec = ms_per_step * 1e-3
    return {'executed_gflop_per_step': flops / 1e9,
            'frac': flops / sec / 1e12 / PEAK_F64_TFLOPS,
            'min_gflop_per_step': least / 1e9,
            'frac_min_flops': least / sec / 1e12 / PEAK_F64_TFLOPS,
            'peak_tflops': PEAK_F64_TFLOPS}


# Which translation unit (plus the shared headers) a kernel label of the bench line lives in:
# a PMC traffic figure is valid for the library whose sources hash the same.
def kernel_sources(label):
    unit = ('wpe.hip' if label.startswith('wpe_') else
            'cacgmm.hip' if label.startswith('em_') else
            'stft.hip' if label.startswith(('stft', 'istft', 'activity')) else 'mvdr.hip')
    return [unit, 'gss_internal.h', 'dense_wave.h']


def source_hashes():
    import hashlib
    from pathlib import Path
    csrc = Path(__file__).resolve().parent / 'csrc'
    return {p.name: hashlib.sha256(p.read_bytes()).hexdigest()[:16]
            for p in sorted(csrc.iterdir()) if p.suffix in ('.hip', '.h')}
