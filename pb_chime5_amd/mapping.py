"""Static CHiME-5 session facts the hot path touches: the output sub-directory of
a session (/root/reference/pb_chime5/core.py:366,385 via mapping.session_to_dataset).
The corpus split itself is public CHiME-5 metadata: S02/S09 = dev, S01/S21 = eval,
everything else = train."""


class Dispatcher(dict):
    """dict with a more helpful KeyError."""

    def __getitem__(self, item):
        try:
            return super().__getitem__(item)
        except KeyError:
            raise KeyError(
                f'Invalid option {item!r}. Possible keys are {self.keys()!r}.') from None


_SESSIONS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 16, 17, 18, 19, 20, 21, 22, 23, 24]

session_to_dataset = Dispatcher({
    f'S{s:02d}': {2: 'dev', 9: 'dev', 1: 'eval', 21: 'eval'}.get(s, 'train')
    for s in _SESSIONS
})
