from . import upfirdn2d  # noqa: F401
