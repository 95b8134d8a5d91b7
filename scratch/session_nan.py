import sys, tempfile, warnings
sys.path.insert(0, '.')
import numpy as np
from pathlib import Path
from pb_chime5_amd.synthetic_corpus import write_chime5_corpus
from pb_chime5_amd.core import get_enhancer
root = Path(tempfile.mkdtemp(prefix='sess_'))
jp = write_chime5_corpus(root / 'corpus', seconds=90.0, utts_per_speaker=4, num_redacted=1, seed=5)
enh = get_enhancer(database_path=str(jp), multiarray=True, context_samples=240000)
for i, ex in enumerate(enh.get_iterator('S02')):
    x = enh.enhance_example(ex)
    print(i, ex['example_id'], x.shape, 'finite' if np.isfinite(x).all() else 'NON-FINITE', 'max', np.abs(x).max() if np.isfinite(x).all() else None)
