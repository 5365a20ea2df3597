"""gtsam.symbol_shorthand: `from ...gtsam.symbol_shorthand import B, V, X, L` (batch.py:26).
A key is (ord(character) << 56) | index, as in gtsam::Symbol."""

_CHR_SHIFT = 56
_INDEX_MASK = (1 << _CHR_SHIFT) - 1


def symbol(c: str, j: int) -> int:
    j = int(j)
    if j < 0 or j > _INDEX_MASK:
        raise RuntimeError("Symbol index is too large")
    return (ord(c) << _CHR_SHIFT) | j


def symbolChr(key: int) -> str:
    return chr((int(key) >> _CHR_SHIFT) & 0xFF)


def symbolIndex(key: int) -> int:
    return int(key) & _INDEX_MASK


def key_string(key: int) -> str:
    c = (int(key) >> _CHR_SHIFT) & 0xFF
    return f"{chr(c)}{symbolIndex(key)}" if c else str(int(key))


def _make(c):
    def f(j):
        return symbol(c, j)
    f.__name__ = c.upper()
    f.__doc__ = f"Key for character '{c}' and index j."
    return f


for _c in "abcdefghijklmnopqrstuvwxyz":
    globals()[_c.upper()] = _make(_c)
del _c
