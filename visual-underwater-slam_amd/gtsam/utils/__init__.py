"""`from gtsam.utils import plot` (batch.py:27): the name exists; plotting is outside the hot path."""
from . import plot  # noqa: F401
