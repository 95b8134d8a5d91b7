"""Random RTTM files through the REFERENCE'S own RTTM front door (database/chime5/rttm.py,
core_chime6_rttm.py, utils/intervall_array.py -- importable in the build container only, with the
stand-ins of make_golden_rttm.py for lazy_dataset / paderbox / soundfile) and through
pb_chime5_amd's: example enumeration, ids, start / end with context, channel selection, audio
lengths and the activity intervals have to agree exactly.

    python tests/golden/fuzz_rttm_vs_reference.py [SEED] [CASES]

Nothing is written; a bug hunt, not a fixture generator."""
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
import make_golden as mg  # noqa: E402
import make_golden_rttm as mgr  # noqa: E402
import make_golden_session as mgs  # noqa: E402

KEYS = ('example_id', 'start', 'end', 'num_samples', 'session_id', 'speaker_id', 'dataset',
        'start_orig', 'end_orig', 'num_samples_orig')


def random_rttm(rng):
    speakers = [f'P{5 + i:02d}' for i in range(int(rng.integers(1, 10)))]
    lines = []
    for spk in speakers:
        for _ in range(int(rng.integers(1, 6))):
            # (everything, context included, stays inside the shortest channel file: past its end
            # the reference relies on ragged np.array() calls that numpy >= 1.24 rejects)
            start = round(float(rng.uniform(0, 1.4)), int(rng.integers(1, 4)))
            dur = round(float(rng.uniform(0.02, 0.6)), int(rng.integers(2, 4)))
            lines.append((start, f'SPEAKER S02_U06.ENH 1 {start} {dur} <NA> <NA> {spk} <NA>'))
    if rng.integers(0, 2):
        lines.sort()
    return ''.join(l + '\n' for _, l in lines)


def summary(enh):
    ds = enh.get_dataset('dev')
    out = []
    for ex in ds:
        row = {k: mgr._tree(ex[k]) for k in KEYS}
        row['audio_files'] = [Path(p).name for p in ex['audio_path']]
        row['audio_shape'] = list(ex['audio_data'].shape)
        out.append(row)
    act = {k: ([list(map(int, iv)) for iv in v.normalized_intervals]
               if hasattr(v, 'normalized_intervals') else 'ones')
           for k, v in enh.activity['S02'].items()}
    return out, act


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(seed)
    import test_rttm_frontdoor as fixture_dir
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        root, rttm_file, _ = fixture_dir._make_chime6_dir(tmp)
        ref = mg._prepare_reference(tmp)
        mg._register_stubs()
        mg._module('lazy_dataset', from_dict=lambda d: mgs._Dataset(d.values()),
                   concatenate=mgs._concatenate)
        mg._module('lazy_dataset.database', Database=mgr._Database)
        mg._module('soundfile', SoundFile=mgs._SoundFile)
        sys.path.insert(0, str(ref))
        pkg = types.ModuleType('pb_chime5')
        pkg.__path__ = [str(ref / 'pb_chime5')]
        pkg.git_root = ref
        sys.modules['pb_chime5'] = pkg
        import gss_oracle as oracle
        import pb_chime5.utils.intervall_array as ref_ia
        import pb_chime5.io as ref_io
        pb = mg._module('paderbox')
        pb.utils = mg._module('paderbox.utils')
        pb.utils.nested = mg._module('paderbox.utils.nested', deflatten=mgr._deflatten)
        pb.array = mg._module('paderbox.array')
        pb.array.intervall = mg._module(
            'paderbox.array.intervall', from_rttm=ref_ia.ArrayIntervalls_from_rttm,
            zeros=lambda: ref_ia.ArrayIntervall(shape=None), ones=mgr._Ones)
        pb.io = mg._module('paderbox.io', load_audio=ref_io.load_audio)
        pb.transform = mg._module('paderbox.transform')
        pb.transform.module_stft = mg._module('paderbox.transform.module_stft',
                                              stft=oracle.stft, istft=oracle.istft)
        import pb_chime5.core_chime6_rttm as ref_core
        from pb_chime5_amd.core_chime6_rttm import get_enhancer as amd_get_enhancer

        for case in range(cases):
            text = random_rttm(rng)
            rttm_file.write_text(text)
            kw = dict(context_samples=int(rng.integers(0, 15000)), wpe=True, wpe_tabs=2,
                      wpe_iterations=1, bss_iterations=1,
                      multiarray=['outer_array_mics', 'first_array_mics', True][int(rng.integers(0, 3))])
            res = {}
            for side, fn in (('reference', ref_core.get_enhancer), ('amd', amd_get_enhancer)):
                try:
                    enh = fn(database_rttm=str(rttm_file), activity_rttm=str(rttm_file),
                             chime6_dir=str(root), **kw)
                    res[side] = summary(enh)
                except Exception as e:
                    res[side] = type(e).__name__ + ': ' + str(e)[:100]
            r, a = res['reference'], res['amd']
            if isinstance(r, str) or isinstance(a, str):
                if not (isinstance(r, str) and isinstance(a, str) and r.split(':')[0] == a.split(':')[0]):
                    print('case', case, 'exceptions differ:', r if isinstance(r, str) else 'ok', '|',
                          a if isinstance(a, str) else 'ok')
                    print(text)
                    bad += 1
                continue
            if r != a:
                bad += 1
                print('case', case, kw, 'differs;', len(r[0]), 'vs', len(a[0]), 'examples')
                for x, y in zip(r[0], a[0]):
                    if x != y:
                        print('  reference', x, '\n  amd      ', y)
                        break
                if r[1] != a[1]:
                    print('  activity differs:', {k: (r[1].get(k), a[1].get(k)) for k in r[1] if r[1].get(k) != a[1].get(k)})
                print(text)
    print('rttm fuzz: seed', seed, 'cases', cases, 'failures', bad)


if __name__ == '__main__':
    main()
