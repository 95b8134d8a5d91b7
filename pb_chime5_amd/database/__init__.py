"""JSON example database (/root/reference/pb_chime5/database/__init__.py:109-263).

The reference keeps its examples in ``lazy_dataset`` objects; the session driver only
uses a handful of their operations -- ``map``, ``filter(lazy=False)``, ``groupby``,
slicing, ``len`` and iteration (core.py:323-381, activity.py:119, database.py:83-131).
``ExampleList`` provides exactly those on a plain list of example dicts; every
access hands out a deep copy so that mapped functions may modify their example.

Database layout (the JSON written by the reference's ``create_json``):
``{"datasets": {name: {example_id: example}}, "alias": {name: [names...]}}``.
"""
import copy
import json
from pathlib import Path

DATASETS = 'datasets'
ALIAS = 'alias'
EXAMPLE_ID = 'example_id'
DATASET_NAME = 'dataset'


class ExampleList:
    """Ordered examples + a chain of map functions applied on access."""

    def __init__(self, examples, maps=()):
        self._examples = list(examples)
        self._maps = tuple(maps)

    def _get(self, ex):
        ex = copy.deepcopy(ex)
        for fn in self._maps:
            ex = fn(ex)
        return ex

    def __len__(self):
        return len(self._examples)

    def __iter__(self):
        for ex in self._examples:
            yield self._get(ex)

    def __getitem__(self, item):
        if isinstance(item, slice):
            return ExampleList(self._examples[item], self._maps)
        if isinstance(item, str):
            for ex in self._examples:
                if ex.get(EXAMPLE_ID) == item:
                    return self._get(ex)
            raise KeyError(item)
        return self._get(self._examples[item])

    def keys(self):
        return tuple(ex.get(EXAMPLE_ID) for ex in self._examples)

    def map(self, fn):
        return ExampleList(self._examples, self._maps + (fn,))

    def filter(self, fn, lazy=True):
        # the predicate sees the mapped example; the unmapped one is kept
        return ExampleList([ex for ex in self._examples if fn(self._get(ex))], self._maps)

    def groupby(self, fn):
        groups = {}
        for ex in self._examples:
            groups.setdefault(fn(self._get(ex)), []).append(ex)
        return {k: ExampleList(v, self._maps) for k, v in groups.items()}

    def sort(self, key):
        return ExampleList(sorted(self._examples, key=lambda ex: key(self._get(ex))),
                           self._maps)

    def __repr__(self):
        return f'{type(self).__name__}(len={len(self)}, maps={len(self._maps)})'


def concatenate(*lists):
    if len(lists) == 1:
        return lists[0]
    assert all(len(l._maps) == 0 for l in lists), 'concatenate before mapping'
    return ExampleList([ex for l in lists for ex in l._examples])


def to_list(x, item_type=None):
    if item_type is not None and isinstance(x, item_type):
        return [x]
    if isinstance(x, (list, tuple)):
        return list(x)
    return [x]


class DictDatabase:
    def __init__(self, database_dict):
        self.database_dict = database_dict

    @property
    def dataset_names(self):
        return (tuple(self.database_dict[DATASETS].keys())
                + tuple(self.database_dict.get(ALIAS, {}).keys()))

    def _examples_of(self, dataset_name):
        if dataset_name in self.database_dict.get(ALIAS, {}):
            examples = {}
            for name in self.database_dict[ALIAS][dataset_name]:
                new = self.database_dict[DATASETS][name]
                clash = set(examples) & set(new)
                assert len(clash) == 0, clash
                examples.update(new)
            return examples
        return self.database_dict[DATASETS][dataset_name]

    def get_datasets(self, dataset_names):
        """One ExampleList over the named datasets (or aliases); every example gets
        its ``example_id`` and ``dataset`` (the name it was requested under)."""
        lists = []
        for dataset_name in to_list(dataset_names, item_type=str):
            try:
                examples = self._examples_of(dataset_name)
            except KeyError:
                import difflib
                similar = difflib.get_close_matches(dataset_name, self.dataset_names, n=5,
                                                    cutoff=0)
                raise KeyError(dataset_name, f'close_matches: {similar}', self) from None
            if len(examples) == 0:
                raise RuntimeError(f'The requested dataset {dataset_name!r} is empty. ')
            for example_id, ex in examples.items():
                ex[EXAMPLE_ID] = example_id
                ex[DATASET_NAME] = dataset_name
            lists.append(ExampleList(examples.values()))
        return concatenate(*lists)


class JsonDatabase(DictDatabase):
    def __init__(self, json_path):
        self._json_path = json_path
        self._dict = None

    @property
    def database_dict(self):
        if self._dict is None:
            with open(Path(self._json_path).expanduser()) as fd:
                self._dict = json.load(fd)
        return self._dict

    def __repr__(self):
        return f'{type(self).__name__}({str(self._json_path)!r})'
