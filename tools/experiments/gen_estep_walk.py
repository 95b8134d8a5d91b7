"""REJECTED EXPERIMENT (experiment 11) -- kept so that the numbers
there can be reproduced; not part of the library.

Generator of csrc/estep_walk.inc: the inner walk of the register-form CACGMM E-step as gfx950
assembly, one or two frames per lane.  To rebuild the experiment: copy this file to
pb_chime5_amd/csrc/gen_estep_walk.py, `git apply tools/experiments/estep_asm_walk.patch`
(against the commit that added this file), `python -m pb_chime5_amd.build`; GSS_ESTEP_FRAMES =
0 / 1 / 2 selects the compiler's kernel or the frames per lane of the assembly walk,
GSS_WALK_DIAG=noload at build time removes the model loads (timing only).

    python pb_chime5_amd/csrc/gen_estep_walk.py [out.inc]        (run by pb_chime5_amd.build)

What it was written for turned out not to be the limit: with the scalar loads removed the kernel
is no faster than the compiler's (VALU issue at the sustained clock is what binds, DESIGN.md
section 6).  The original rationale follows.

Why assembly.  A lane of the E-step evaluates  q_k = y^H B_k^-1 y  for its frame: the packed
upper triangle of the K model matrices arrives through the scalar data cache (the rows are
wave uniform), and that cache -- 3.27 B / cycle / CU measured, profiles/r03_smem_bench.txt --
is what binds the one-frame-per-lane kernel.  Two frames per lane use
every scalar byte twice.  The compiler cannot be talked into that form (it hoists the scalar
loads of the unrolled triangle and spills thousands of SGPRs), so the
walk is emitted here with the registers assigned by hand:

  v[YB ..]            y of the lane's FR frames, channel d of frame u at YB + 4 (u D + d): re, im
  v[YA ..]            y_d1 of the current row, 4 registers per frame
  v[TB ..]            Re / Im (y_d1 conj y_d2), 4 registers per frame
  s16 .. s99          ping-pong buffers of G model rows (K complex numbers each), in
                      4-aligned pieces that skip s32 .. s35 (s32 is reserved)
  s[12:13]            running model pointer (the packed triangle is one sequential stream)
  s[14:15]            return address of the row subroutine (channel pointer in the prologue)

Code size matters as much as the schedule: the fully unrolled triangle of D = 24 is 65 KB of
8-byte VOP3 instructions, more than the 64 KB instruction cache two CUs share, and ran 30 %
SLOWER than the one-frame kernel.  Row d1 of the triangle is a suffix of row 0 once y_d1
sits in fixed registers, so the rows share one subroutine with an entry point per column
group (s_call_b64 / s_setpc_b64); it exists twice because the ping-pong parity at a row's
entry alternates.  About 15 KB in all.

Entries (d1 <= d2, row major, as em_chol writes them) are processed in groups of G columns
(G = 2: columns 2p, 2p + 1; an odd row starts with its diagonal entry as a group of one):
wait for the group's rows, request the next group's into the other buffer, advance the model
pointer, then 2 (2 + 2 K) f64 operations per entry and frame pair.  Scalar loads return out
of order, hence lgkmcnt(0) and the ping-pong.  With G = 1 the request after the last entry
of a bin reads one row past the bin's model (the next bin's, or the workspace array that
follows the model in the arena); its value is never used.
"""
import os
import sys
from pathlib import Path

DIAG = os.environ.get('GSS_WALK_DIAG', '')

# first y register per frames-per-lane; the compiler keeps v0 .. v[YB-1]
YBASE = {1: 24, 2: 48}
# ping-pong SGPR buffers of 40 dwords: (first register, dwords) pieces
SBUF = (((16, 16), (36, 16), (52, 8)), ((60, 16), (76, 16), (92, 8)))
SMP, SRET = 12, 14
SHAPES = [(K, D, FR) for FR in (1, 2) for D in (24, 20, 12, 10) for K in (2, 3, 4, 5, 6)]


def group_size(K):
    return 2 if 8 * K <= 40 else 1


def _sgpr(buf, i):
    """SGPR holding dword i of ping-pong buffer buf."""
    for first, n in SBUF[buf]:
        if i < n:
            return first + i
        i -= n
    raise IndexError


def _pieces(buf, ndw):
    """Split the first ndw dwords (multiple of 4) of buffer buf into s_load_dwordx{4,8,16}:
    (dword offset, width, first SGPR)."""
    out, off = [], 0
    for first, n in SBUF[buf]:
        i = 0
        while i < n and off < ndw:
            w = next(w for w in (16, 8, 4) if w <= min(n - i, ndw - off))
            out.append((off, w, first + i))
            i += w
            off += w
    assert off == ndw
    return out


def walk_asm(K, D, FR):
    G = group_size(K)
    assert D % 2 == 0 and G * 4 * K <= 40 and FR in (1, 2)
    YB = YBASE[FR]
    FRAMES = tuple(range(FR))
    YA = YB + 4 * FR * D
    TB = YA + 4 * FR
    top = TB + 4 * FR
    assert top <= (256 if FR == 2 else 128)
    ROW = 16 * K                        # bytes of one entry's model row
    q = lambda u, k: f'%{u * K + k}'
    OP_M, OP_Y, OP_OFF0, OP_OFF1, OP_STRIDE, OP_LANE16 = (f'%{FR * K + i}' for i in range(6))
    OP_OFF = (OP_OFF0, OP_OFF1)
    yreg = lambda u, d: YB + 4 * (u * D + d)
    yre = lambda u, d: f'v[{yreg(u, d)}:{yreg(u, d) + 1}]'
    yim = lambda u, d: f'v[{yreg(u, d) + 2}:{yreg(u, d) + 3}]'
    are = lambda u: f'v[{YA + 4 * u}:{YA + 4 * u + 1}]'
    aim = lambda u: f'v[{YA + 4 * u + 2}:{YA + 4 * u + 3}]'
    pr = lambda u: f'v[{TB + 4 * u}:{TB + 4 * u + 1}]'
    pim = lambda u: f'v[{TB + 4 * u + 2}:{TB + 4 * u + 3}]'
    mp = f's[{SMP}:{SMP + 1}]'
    ret = f's[{SRET}:{SRET + 1}]'
    L = []

    def request(buf, offset_bytes, force=False):
        if DIAG == 'noload' and not force:
            return
        for off, w, s0 in _pieces(buf, G * 4 * K):
            L.append(f's_load_dwordx{w} s[{s0}:{s0 + w - 1}], {mp}, {hex(offset_bytes + 4 * off)}')

    def advance(nbytes):
        L.append(f's_add_u32 s{SMP}, s{SMP}, {hex(nbytes)}')
        L.append(f's_addc_u32 s{SMP + 1}, s{SMP + 1}, 0')

    def entry(buf, j, d2, diag):
        """One entry (current row, column d2) for both frames with row j of buffer buf."""
        m = lambda k, c: (lambda r: f's[{r}:{r + 1}]')(_sgpr(buf, (j * K + k) * 4 + 2 * c))
        for u in FRAMES:
            L.append(f'v_mul_f64 {pr(u)}, {are(u)}, {yre(u, d2)}')
        if not diag:
            for u in FRAMES:
                L.append(f'v_mul_f64 {pim(u)}, {aim(u)}, {yre(u, d2)}')
        for u in FRAMES:
            L.append(f'v_fma_f64 {pr(u)}, {aim(u)}, {yim(u, d2)}, {pr(u)}')
        if not diag:
            for u in FRAMES:
                L.append(f'v_fma_f64 {pim(u)}, -{are(u)}, {yim(u, d2)}, {pim(u)}')
        for k in range(K):
            for u in FRAMES:
                L.append(f'v_fma_f64 {q(u, k)}, {m(k, 0)}, {pr(u)}, {q(u, k)}')
        if not diag:
            for k in range(K):
                for u in FRAMES:
                    L.append(f'v_fma_f64 {q(u, k)}, {m(k, 1)}, {pim(u)}, {q(u, k)}')

    label = lambda v, p: f'.Lestep_walk_{K}_{D}_{FR}_v{v}_g{p}_%='
    ngrp = D // G                       # column groups of the row subroutine

    # ---- prologue.  First pull the bin's model towards this XCD's L2 with vector loads
    # (1 KB each, results unused): em_chol wrote it from other XCDs, and a scalar load that
    # has to go to memory costs ~1000 cycles per group -- more than the ping-pong can hide.
    # They land in the first y register, which the y loads behind them overwrite (vector
    # loads return in order).  Then the lane's frames of every channel and the first group
    # of model rows.
    npf = (D * (D + 1) // 2 * ROW) // 1024
    L.append(f's_mov_b64 {ret}, {OP_M}')
    for i in range(npf):
        L.append(f'global_load_dwordx4 v[{YB}:{YB + 3}], {OP_LANE16}, {ret}')
        if i + 1 < npf:
            L.append(f's_add_u32 s{SRET}, s{SRET}, 0x400')
            L.append(f's_addc_u32 s{SRET + 1}, s{SRET + 1}, 0')
    L.append(f's_mov_b64 {ret}, {OP_Y}')
    for d in range(D):
        for u in FRAMES:
            L.append(f'global_load_dwordx4 v[{yreg(u, d)}:{yreg(u, d) + 3}], {OP_OFF[u]}, {ret}')
        if d + 1 < D:
            L.append(f's_add_u32 s{SRET}, s{SRET}, {OP_STRIDE}')
            L.append(f's_addc_u32 s{SRET + 1}, s{SRET + 1}, 0')
    L.append(f's_mov_b64 {mp}, {OP_M}')
    request(0, 0, True)
    if DIAG == 'noload':
        request(1, 0, True)
    L.append('s_waitcnt vmcnt(0)')

    # ---- rows (static): y_d1 into the row registers, the odd row's leading diagonal
    # entry, then the shared subroutine from the row's first full column group
    g = 0                               # groups done so far: buffer of the next group = g & 1
    for d1 in range(D):
        for u in FRAMES:
            for c in range(4):
                L.append(f'v_mov_b32 v{YA + 4 * u + c}, v{yreg(u, d1) + c}')
        first = d1
        if G == 2 and d1 % 2:
            buf = g & 1
            L.append('s_waitcnt lgkmcnt(0)')
            if d1 + 1 < D:
                request(buf ^ 1, ROW)
                advance(ROW)
            entry(buf, 0, d1, True)
            g += 1
            first = d1 + 1
        p0 = first // G
        if p0 < ngrp:
            v = (p0 ^ g) & 1
            L.append(f's_call_b64 {ret}, {label(v, p0)}')
            g += ngrp - p0
    end = f'.Lestep_walk_{K}_{D}_{FR}_end_%='
    L.append(f's_branch {end}')

    # ---- the row subroutine, once per ping-pong parity
    for v in (0, 1):
        for p in range(ngrp):
            buf = (p & 1) ^ v
            L.append(f'{label(v, p)}:')
            L.append('s_waitcnt lgkmcnt(0)')
            request(buf ^ 1, G * ROW)
            advance(G * ROW)
            for j in range(G):
                entry(buf, j, p * G + j, False)
        L.append(f's_setpc_b64 {ret}')
    L.append(f'{end}:')
    clob = ([f'"v{i}"' for i in range(YB, top)] +
            [f'"s{i}"' for i in range(SMP, 100) if not 32 <= i < 36])
    return L, clob


def emit(K, D, FR):
    lines, clob = walk_asm(K, D, FR)
    outs = ', '.join(f'"+v"(q[{u}][{k}])' for u in range(FR) for k in range(K))
    body = '\n'.join(f'        "{ln}\\n"' for ln in lines)
    clobs = ',\n          '.join(', '.join(clob[i:i + 12]) for i in range(0, len(clob), 12))
    return f'''template <> struct EstepWalk<{K}, {D}, {FR}> {{
    static constexpr bool available = true;
    // q (zero on entry) += y^H M_k y for the frames at byte offsets off0 / off1 of each
    // channel row (stride bytes apart) of Ynf; lane16 = 16 * lane
    static __device__ __forceinline__ void run(const cplx *Mf, const cplx *Ynf, uint32_t off0,
                                               uint32_t off1, uint32_t stride, uint32_t lane16,
                                               double (&q)[{FR}][{K}]) {{
    asm volatile(
{body}
        : {outs}
        : "s"(Mf), "s"(Ynf), "v"(off0), "v"(off1), "s"(stride), "v"(lane16)
        : {clobs}, "scc");
    }}
}};
'''


def generate(path, shapes=SHAPES):
    text = ['// GENERATED by gen_estep_walk.py -- do not edit.\n',
            'template <int K, int D, int FR> struct EstepWalk { static constexpr bool available = false; };\n']
    for K, D, FR in shapes:
        text.append(emit(K, D, FR))
    new = ''.join(text)
    p = Path(path)
    if not p.exists() or p.read_text() != new:
        p.write_text(new)
    return p


if __name__ == '__main__':
    out = sys.argv[1] if len(sys.argv) > 1 else Path(__file__).with_name('estep_walk.inc')
    print(generate(out))
