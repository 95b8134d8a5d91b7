"""Golden fixture for the RTTM-driven front door.

Runs ONLY in the build container (needs /root/reference).  It builds the small CHiME-6 style
directory of tests/test_rttm_frontdoor.py (12 channel files of different length, one RTTM
file) and drives the reference's REAL code on it:

* ``ArrayIntervalls_from_rttm`` (utils/intervall_array.py:45-101, decimal-exact seconds ->
  samples) standing in for ``paderbox.array.intervall.from_rttm``, which it was copied to,
* ``get_chime6_files`` / ``RTTMDatabase`` / ``get_dataset_for_session`` /
  ``recursive_load_audio`` (database/chime5/rttm.py:70-632),
* ``get_database`` / ``get_enhancer`` / ``Activity`` / ``Enhancer.enhance_example``
  (core_chime6_rttm.py:31-422),

with the numeric third-party calls delegated to the CPU oracle as in make_golden.py, a
list-backed stand-in for ``lazy_dataset`` (its ``Database`` base class included) and the
``wave``-based ``soundfile`` stand-in of make_golden_session.py.

Output: rttm_session.json (examples, activity; bit exact) and rttm_session.npz (enhanced
signals).  Usage:  python tests/golden/make_golden_rttm.py
"""
import json
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

ENHANCER = dict(context_samples=4000, wpe=True, wpe_tabs=2, wpe_iterations=2, bss_iterations=3)
MULTIARRAY = ('outer_array_mics', 'first_array_mics', True)


class _Database:
    """lazy_dataset.database.Database: datasets by name or alias from ``self.data``."""

    def get_dataset(self, names):
        if isinstance(names, str):
            names = [names]
        data = self.data
        alias = data.get('alias') or {}
        out = []
        for name in names:
            for ds_name in alias.get(name, [name]):
                for example_id, ex in data['datasets'][ds_name].items():
                    ex = dict(ex)
                    ex['example_id'] = example_id
                    ex['dataset'] = ds_name
                    out.append(ex)
        return mgs._Dataset(out)


class _Ones:
    """paderbox.array.intervall.ones(): active everywhere, no fixed length."""

    def __getitem__(self, item):
        return np.ones(item.stop - item.start, dtype=bool)


def _deflatten(d, sep=None):
    out = {}
    for key, value in d.items():
        cur = out
        for part in key[:-1]:
            cur = cur.setdefault(part, {})
        cur[key[-1]] = value
    return out


def _tree(x):
    if isinstance(x, dict):
        return {k: _tree(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_tree(v) for v in x]
    if isinstance(x, (int, np.integer)):
        return int(x)
    return x


def main():
    import test_rttm_frontdoor as fixture_dir
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        root, rttm_file, _ = fixture_dir._make_chime6_dir(tmp)
        # the reference builds examples for every session of the RTTM and needs audio for
        # each of them: keep the session that has audio
        rttm_file.write_text(''.join(l + '\n' for l in fixture_dir.RTTM.splitlines() if ' S02' in l))
        ref = mg._prepare_reference(tmp)
        mg._register_stubs()
        mg._module('lazy_dataset', from_dict=lambda d: mgs._Dataset(d.values()),
                   concatenate=mgs._concatenate)
        mg._module('lazy_dataset.database', Database=_Database)
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
        pb.utils.nested = mg._module('paderbox.utils.nested', deflatten=_deflatten)
        pb.array = mg._module('paderbox.array')
        pb.array.intervall = mg._module(
            'paderbox.array.intervall', from_rttm=ref_ia.ArrayIntervalls_from_rttm,
            zeros=lambda: ref_ia.ArrayIntervall(shape=None), ones=_Ones)
        pb.io = mg._module('paderbox.io', load_audio=ref_io.load_audio)
        pb.transform = mg._module('paderbox.transform')
        pb.transform.module_stft = mg._module('paderbox.transform.module_stft',
                                              stft=oracle.stft, istft=oracle.istft)

        import pb_chime5.core_chime6_rttm as core_rttm

        fixture = {'enhancer': ENHANCER, 'cases': {}}
        arrays = {}
        for multiarray in MULTIARRAY:
            tag = str(multiarray)
            enh = core_rttm.get_enhancer(database_rttm=str(rttm_file), activity_rttm=str(rttm_file),
                                         chime6_dir=str(root), multiarray=multiarray, **ENHANCER)
            ds = enh.get_dataset('dev')
            examples = []
            for i, ex in enumerate(ds):
                meta = {k: _tree(v) for k, v in ex.items() if k not in ('audio_data', 'audio_path')}
                meta['audio_files'] = [Path(p).name for p in ex['audio_path']]
                meta['audio_shape'] = list(ex['audio_data'].shape)
                examples.append(meta)
                if multiarray == 'outer_array_mics' or i == 1:     # keep the fixture small
                    arrays[f'{tag}/x_hat/{i}'] = enh.enhance_example(ex)
            activity = enh.activity['S02']
            act = {k: ([list(map(int, iv)) for iv in v.normalized_intervals]
                       if hasattr(v, 'normalized_intervals') else 'ones')
                   for k, v in activity.items()}
            fixture['cases'][tag] = {'examples': examples, 'activity': act}
        fixture['dataset_names'] = sorted(set(enh.db.data['datasets']) | set(enh.db.data['alias']))
        fixture['example_id'] = core_rttm.RTTMDatabase.example_id('S02', '1', 100, 200)
        (HERE / 'rttm_session.json').write_text(json.dumps(fixture, indent=1))
        mg._save('rttm_session.npz', **arrays)
        print('rttm_session.json', {k: len(v['examples']) for k, v in fixture['cases'].items()})


if __name__ == '__main__':
    main()
