"""Dump the CHiME-5 session tables of the reference (pb_chime5/mapping.py:12-79) to
mapping_tables.json.  Runs only in the build container (needs /root/reference).
Usage:  python tests/golden/make_golden_mapping.py"""
import importlib.util
import json
from pathlib import Path

HERE = Path(__file__).resolve().parent
spec = importlib.util.spec_from_file_location('ref_mapping', '/root/reference/pb_chime5/mapping.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
tables = {name: dict(getattr(ref, name))
          for name in ('session_to_speakers', 'session_to_dataset', 'session_to_arrays')}
(HERE / 'mapping_tables.json').write_text(json.dumps(tables, indent=1, sort_keys=True))
print('mapping_tables.json', {k: len(v) for k, v in tables.items()})
