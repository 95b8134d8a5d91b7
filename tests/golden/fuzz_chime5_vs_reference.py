"""Random CHiME-5 / CHiME-6 flavoured corpora (pb_chime5_amd.synthetic_corpus: utterances per
speaker, redacted segments, recording length, seed) and enhancer settings (context, array
selection) through the REFERENCE'S own JSON front door (database.py, activity.py, core.py /
core_chime6.py -- build container only, stand-ins as in make_golden_session.py) and through
pb_chime5_amd's: iterator bookkeeping and activity intervals have to agree exactly, and so do the
observation shape and the per-example activity that enhance_example cuts out.

    python tests/golden/fuzz_chime5_vs_reference.py [SEED] [CASES]"""
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
import make_golden as mg  # noqa: E402
import make_golden_session as mgs  # noqa: E402

KEYS = ('start', 'end', 'num_samples', 'start_orig', 'end_orig', 'num_samples_orig')


class _NumpyStaging:
    """Stands in for ops.HostStaging (page-locked memory needs a GPU): same `shape` contract."""

    def shape(self, D, N, K, N_act):
        self.obs = np.full((D, N), 12345, dtype=np.int16)
        self.act = np.full((K, N_act), 7, dtype=np.uint8)
        return self.obs, self.act


def bookkeeping(enh, session_id, chime6, probe):
    it = enh.get_iterator(session_id)
    examples = [{'example_id': ex['example_id'], 'speaker_id': ex['speaker_id'],
                 'reference_array': ex.get('reference_array'),
                 **{k: mgs._tree(ex[k]) for k in KEYS}} for ex in it]
    activity = enh.activity[session_id]
    if chime6:
        act = {spk: [list(map(int, iv)) for iv in t.normalized_intervals] for spk, t in activity.items()}
    else:
        act = {a: {spk: [list(map(int, iv)) for iv in t.normalized_intervals] for spk, t in tr.items()}
               for a, tr in activity.items()}
    cut = {}
    for idx in probe:
        if idx < len(it):
            loader = ()
            if hasattr(enh, '_prepare_example'):      # pb_chime5_amd: the host side alone (no GPU here)
                obs, ex_act, speaker = enh._prepare_example(it[idx])
                # ... and the session driver's loader (WAV slices straight into int16 rows,
                # activity tracks sliced into uint8 rows) must hand the device the same thing
                from pb_chime5_amd.io.wav_slices import WavSliceReader
                st, reader = _NumpyStaging(), WavSliceReader()
                target, _, _, keep = enh._prepare_into(it[idx], st, reader)
                reader.close()
                same = (st.obs.shape == obs.shape and np.array_equal(st.obs / 2.0 ** 15, obs)
                        and np.array_equal(st.act, np.array(list(ex_act.values())).astype(np.uint8))
                        and target == tuple(ex_act.keys()).index(speaker))
                x = np.arange(st.obs.shape[-1] + 1000)
                trimmed = enh._trim_context(x, it[idx])
                same = same and np.array_equal(trimmed, x if keep is None else x[keep[0]:keep[1]])
                if not same:
                    loader = ('the session loader differs from _prepare_example',)
            else:
                enh.enhance_example(it[idx], debug=True)
                loc = enh.enhance_example_locals
                obs, ex_act = loc['obs'], loc['ex_array_activity']
            cut[idx] = (tuple(obs.shape), list(ex_act.keys()),
                        np.packbits(np.array(list(ex_act.values())), axis=-1).tobytes(),
                        float(np.abs(obs).sum())) + loader
    return examples, act, cut


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rng = np.random.default_rng(seed)
    from pb_chime5_amd.synthetic_corpus import write_chime5_corpus
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        ref = mg._prepare_reference(tmp)
        mg._register_stubs()
        mg._module('lazy_dataset', from_dict=lambda d: mgs._Dataset(d.values()),
                   concatenate=mgs._concatenate)
        mg._module('soundfile', SoundFile=mgs._SoundFile)
        sys.path.insert(0, str(ref))
        pkg = types.ModuleType('pb_chime5')
        pkg.__path__ = [str(ref / 'pb_chime5')]
        pkg.git_root = ref
        sys.modules['pb_chime5'] = pkg
        import pb_chime5.core as ref_core
        import pb_chime5.core_chime6 as ref_core6
        import pb_chime5.mapping as ref_mapping
        import pb_chime5_amd.core as amd_core
        import pb_chime5_amd.core_chime6 as amd_core6
        # the stages themselves are not under test here: one cheap EM iteration, no WPE
        for case in range(cases):
            chime6 = bool(rng.integers(0, 2))
            corpus = dict(session_id='S02', seconds=float(rng.integers(6, 12)), seed=int(rng.integers(0, 10 ** 6)),
                          utts_per_speaker=int(rng.integers(1, 6)), num_redacted=int(rng.integers(0, 4)),
                          rir_taps=64)
            enhancer = dict(context_samples=int(rng.integers(0, 40000)),
                            multiarray=[False, True, 'outer_array_mics', 'first_array_mics'][int(rng.integers(0, 4))],
                            wpe=False, bss_iterations=1, bss_iterations_post=1)
            root = tmp / f'corpus{case}'
            json_path = write_chime5_corpus(root, **corpus, chime6=chime6)
            n_total = int(corpus['seconds'] * 16000)
            for key in list(ref_mapping.session_array_to_num_samples):
                if key.startswith('S02_'):
                    ref_mapping.session_array_to_num_samples[key] = n_total
            res = {}
            for side, mod in (('reference', ref_core6 if chime6 else ref_core),
                              ('amd', amd_core6 if chime6 else amd_core)):
                try:
                    enh = mod.get_enhancer(database_path=str(json_path), **enhancer)
                    res[side] = bookkeeping(enh, 'S02', chime6, probe=(0, 3))
                except Exception as e:
                    res[side] = type(e).__name__ + ': ' + str(e)[:120]
            r, a = res['reference'], res['amd']
            tag = dict(case=case, chime6=chime6, **corpus, **enhancer)
            if isinstance(r, str) or isinstance(a, str):
                if not (isinstance(r, str) and isinstance(a, str) and r.split(':')[0] == a.split(':')[0]):
                    print('exceptions differ:', r if isinstance(r, str) else 'ok', '|', a if isinstance(a, str) else 'ok', tag)
                    bad += 1
                continue
            if r != a:
                bad += 1
                what = [n for n, x, y in zip(('examples', 'activity', 'example cut'), r, a) if x != y]
                print('differs in', what, tag)
                if r[0] != a[0]:
                    for x, y in zip(r[0], a[0]):
                        if x != y:
                            print('  reference', x, '\n  amd      ', y)
                            break
                    print('  counts', len(r[0]), len(a[0]))
                if r[2] != a[2]:
                    for idx in r[2]:
                        if r[2][idx] != a[2].get(idx):
                            print('  cut', idx, r[2][idx][:2], (a[2].get(idx) or (None, None))[:2])
    print('chime5/6 front door fuzz: seed', seed, 'cases', cases, 'failures', bad)


if __name__ == '__main__':
    main()
