"""Speaker / segment delimiter factories accepted by `whisperer(delimiter=...)`.

Public names, call signatures, produced strings and the `.regex` attribute of each factory are the interface of
urro_whisper/delimiters.py:3-29 (callers write `delimiter=HYPHEN` or `delimiter=SPEAKER()`); they are generated
here from one table.  The reference closes a bracket through python-bidi's MIRRORED map, which is not a
dependency of this package: the usual pairs are listed."""
_CLOSERS = dict(zip("[({<«‹「『【〈《〔⟨⟦", "])}>»›」』】〉》〕⟩⟧"))
_CLOSERS.update({v: k for k, v in list(_CLOSERS.items())[:4]})


class _Delimiter:
    """Callable returning the delimiter text; `.regex` matches the delimiter inside decoded text."""

    def __init__(self, name, regex, make):
        self.__name__ = name
        self.regex = regex
        self._make = make

    def __call__(self, *args, **kwargs):
        return self._make(*args, **kwargs)

    def __repr__(self):
        return f"<delimiter {self.__name__}>"


def _repeated(ch):
    return lambda count=1: " " + ch * count


def _labelled(long_label, short_label):
    return lambda ch="[", short=False: f" {ch}{short_label if short else long_label}{_CLOSERS.get(ch, ch)}"


_TABLE = (
    ("HYPHEN", r"\-", None, _repeated("-")),
    ("GREATER_THAN", r"\>", None, _repeated(">")),
    ("SPEAKER", None, "SPEAKER", _labelled("SPEAKER 1", "S1")),
    ("PERSON", None, "PERSON", _labelled("PERSON 1", "P1")),
)
for _name, _mark, _word, _make in _TABLE:
    _regex = rf"(\s)?{_mark}\s" if _mark else rf"(\s)?\[{_word}\s\d\]\s"
    globals()[_name] = _Delimiter(_name, _regex, _make)
del _name, _mark, _word, _make, _regex
