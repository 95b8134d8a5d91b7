"""The slice of sacred's command line the reference's scripts rely on
(/root/reference/pb_chime5/scripts/run.py:19-33,126-142, kaldi_run.py:1-57):

    python -m <script> [command] [with key=value ... named_config ...] [-F DIR]

* configuration keys = the keyword arguments of ``get_enhancer`` plus the script's own;
  ``key=value`` updates one (values are Python literals, anything else is a string;
  unknown keys are an error, as with sacred),
* named configs are preset updates selected by bare name after ``with``,
* ``command`` picks another entry point (``test_run``; ``print_config``).

sacred itself (observers, seeds, captured functions) is not reproduced.
"""
import ast
import inspect
import json
from pathlib import Path


def enhancer_defaults(get_enhancer, drop=()):
    return {k: v.default for k, v in inspect.signature(get_enhancer).parameters.items()
            if k not in drop and v.default is not inspect.Parameter.empty}


def _literal(text):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def parse(argv, defaults, commands=('test_run', 'print_config'), named_configs=None):
    """-> (command or None, config dict, file_storage or None)."""
    named_configs = named_configs or {}
    argv = list(argv)
    file_storage = None
    rest = []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a in ('-F', '--file_storage'):
            file_storage = argv[i + 1]
            i += 2
        elif a.startswith('--file_storage='):
            file_storage = a.split('=', 1)[1]
            i += 1
        else:
            rest.append(a)
            i += 1
    command = None
    if rest and rest[0] != 'with':
        command = rest.pop(0)
        if command not in commands:
            raise SystemExit(f'Unknown command {command!r}; available: {commands}')
    config = dict(defaults)
    if rest:
        if rest[0] != 'with':
            raise SystemExit(f'Expected "with" before config updates, got {rest[0]!r}')
        for item in rest[1:]:
            if '=' in item:
                key, value = item.split('=', 1)
                if key not in config:
                    raise SystemExit(
                        f'Unknown config key {key!r}; available: {sorted(config)}')
                config[key] = _literal(value)
            elif item in named_configs:
                config.update(named_configs[item])
            else:
                raise SystemExit(f'Unknown named config {item!r}; available: '
                                 f'{sorted(named_configs)}')
    return command, config, file_storage


def print_config(config):
    print('Configuration:')
    for k in config:
        print(f'  {k} = {config[k]!r}')


def new_run_dir(basedir, config):
    """sacred's FileStorageObserver layout: ``basedir/<next integer>/config.json``."""
    basedir = Path(basedir).expanduser().resolve()
    basedir.mkdir(parents=True, exist_ok=True)
    taken = [int(p.name) for p in basedir.iterdir() if p.is_dir() and p.name.isdigit()]
    run_dir = basedir / str(max(taken, default=0) + 1)
    run_dir.mkdir()
    (run_dir / 'config.json').write_text(json.dumps(config, indent=2, default=str))
    return run_dir


def broadcast_path(path):
    """The master's run directory for every process (dlp_mpi.bcast in the reference)."""
    from pb_chime5_amd import parallel
    if parallel.world_size() == 1:
        return path
    return Path(parallel.broadcast_object(str(path) if path is not None else None, src=0))
