"""End-to-end session throughput (WAV files -> enhanced WAV files) on a synthetic CHiME-5
style session: 24 channels, reference-default context of 15 s on both sides."""
import sys, time, tempfile, shutil
sys.path.insert(0, '.')
from pathlib import Path
from pb_chime5_amd.synthetic_corpus import write_chime5_corpus
from pb_chime5_amd.core import get_enhancer
root = Path(tempfile.mkdtemp(prefix='sess_'))
t = time.time()
jp = write_chime5_corpus(root / 'corpus', seconds=90.0, utts_per_speaker=4, num_redacted=1, seed=5)
print('corpus written in', round(time.time() - t, 1), 's')
for inflight in (1, 2, 3, 3, 4):
    enh = get_enhancer(database_path=str(jp), multiarray=True, context_samples=240000)
    enh.inflight = inflight
    it = enh.get_iterator('S02')
    secs = sum(ex['num_samples']['observation']['U01'] for ex in it) / 16000
    out = root / f'out{inflight}_{time.time_ns()}'
    t = time.time()
    enh.enhance_session('S02', out)
    e = time.time() - t
    print(f'inflight={inflight}: {len(it)} utterances, {secs:.0f} s of audio incl. context, wall {e:.2f} s '
          f'-> {1e3 * e / len(it):.1f} ms/utterance, {secs / e:.0f} x real time')
shutil.rmtree(root)
