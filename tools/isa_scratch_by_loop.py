"""Where a kernel's register spills are: scratch loads / stores per loop of its gfx950 ISA.

    python tools/isa_scratch_by_loop.py cacgmm.hip em_onchip4_kernelILi5E [more kernel name parts ...]

Compiles the translation unit to assembly with the library's flags (device only), takes the
kernels whose mangled name contains the given text, finds the loops by their backward branches
(a branch to a label that lies above it: [label, branch] is a loop body) and attributes every
`scratch_load` / `scratch_store` to the innermost loop that contains it.  Per loop: its line
range, instruction count, f64 FMAs / MFMAs / LDS reads (to recognise the hot loop) and the
scratch traffic.  (VERDICT r5 #2a: em_onchip4_kernel<5> spills 84 VGPRs under
__launch_bounds__(256, 3) -- inside or outside the frame loop?)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

R = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(R))
from pb_chime5_amd import build  # noqa: E402


def loops_of(lines):
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r'^(\.LBB\d+_\d+):', l))}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r'\bs_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    # merge loops that share a header (several back edges): the widest
    by_head = {}
    for a, b in loops:
        by_head[a] = max(by_head.get(a, a), b)
    return sorted(by_head.items())


def main():
    unit, wanted = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as tmp:
        asm = Path(tmp) / 'unit.s'
        flags = [f for f in build.FLAGS if f != '-fPIC']
        subprocess.run([build._hipcc(), *flags, '--cuda-device-only', '-S', '-o', str(asm),
                        str(build.CSRC / unit)], check=True, capture_output=True)
        text = asm.read_text().splitlines()
    starts = [i for i, l in enumerate(text) if re.match(r'^_Z\w+:', l)]
    for s in starts:
        name = text[s].split(':')[0]
        if not any(w in name for w in wanted):
            continue
        end = next(i for i in range(s, len(text)) if text[i].startswith('.Lfunc_end'))
        body = text[s:end]
        meta = {k: v for k, v in re.findall(r'\.set ' + re.escape(name) + r'\.(num_vgpr|private_seg_size), (\d+)',
                                              '\n'.join(text))}
        is_instr = [bool(re.match(r'^\s+[a-z]\w+', l)) and not l.strip().startswith(('.', ';')) for l in body]
        loops = loops_of(body)
        print(f'{name}: {sum(is_instr)} instructions, {meta.get("num_vgpr")} VGPRs, '
              f'{meta.get("private_seg_size")} bytes of scratch per lane')

        def innermost(i):
            best = None
            for a, b in loops:
                if a <= i <= b and (best is None or b - a < best[1] - best[0]):
                    best = (a, b)
            return best
        count = {}
        for i, l in enumerate(body):
            if 'scratch_load' in l or 'scratch_store' in l:
                key = innermost(i)
                kind = 'load' if 'scratch_load' in l else 'store'
                count.setdefault(key, {'load': 0, 'store': 0})[kind] += 1
        rows = [(None, 0, len(body))] + [((a, b), a, b) for a, b in loops]
        for key, a, b in rows:
            seg = body[a:b + 1]
            own = count.get(key, {'load': 0, 'store': 0})
            inner = sum(v['load'] + v['store'] for k, v in count.items()
                        if k is not None and key is not None and k != key and key[0] <= k[0] and k[1] <= key[1])
            fma = sum('v_fma_f64' in l or 'v_mul_f64' in l or 'v_add_f64' in l for l in seg)
            print(f'  {"whole kernel, outside every loop" if key is None else f"loop lines {a:5d} - {b:5d}":34s} '
                  f'{sum(is_instr[a:b + 1]):6d} instr  {fma:5d} f64 VALU  '
                  f'{sum("v_mfma" in l for l in seg):4d} MFMA  {sum("ds_read" in l for l in seg):4d} ds_read  |  '
                  f'scratch here: {own["load"]:3d} loads {own["store"]:3d} stores'
                  + (f'  (+{inner} in loops inside it)' if inner else ''))


if __name__ == '__main__':
    main()
