"""Small utilities mirrored from pseudo_3D_interpolation/functions/utils.py (only what steps 12-14 use)."""
import numpy as np


def xprint(*args, kind: str = 'info', verbosity: int = 0, **kwargs) -> None:
    """print() with a coloured prefix, gated by verbosity (reference: functions/utils.py:57-76)."""
    verbosity = 1 if verbosity is True else verbosity
    table = {
        'info': ('\033[39m', '[INFO]  ', 1),
        'warning': ('\033[33m\033[1m', '[WARN]  ', 0),
        'error': ('\033[31m\033[1m', '[ERROR]  ', 0),
        'success': ('\033[32m', '[SUCCESS]  ', 1),
        'debug': ('\033[36m', '[DEBUG]  ', 2),
    }
    entry = table.get(kind)
    level = 1
    if entry is not None:
        color, label, level = entry
        args = [f'{color}{label}'] + [f'{a}' for a in args] + ['\033[0m']
    if level <= verbosity:
        print(*args, **kwargs)


def rescale(a, vmin=0, vmax=1):
    """Rescale to [vmin, vmax] using the array's own extrema (functions/utils.py:413-441)."""
    a = np.asarray(a)
    return rescale_dask(a, vmin=np.nanmin(a) if vmin is None else vmin, vmax=np.nanmax(a) if vmax is None else vmax)


def rescale_dask(a, vmin=0, vmax=1, amin=None, amax=None):
    """Rescale to [vmin, vmax] with given (global) extrema (functions/utils.py:444-473)."""
    a = np.asarray(a)
    amin = np.nanmin(a) if amin is None else amin
    amax = np.nanmax(a) if amax is None else amax
    if amin == amax:
        return a
    return vmin + (a - amin) * ((vmax - vmin) / (amax - amin))
