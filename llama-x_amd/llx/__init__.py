"""llx: MI355X-native kernels + host glue behind the modelling/ and subclasses/ drop-in packages."""
from . import _lib  # noqa: F401
