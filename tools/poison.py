"""Debug aid: every ``torch.empty`` / ``empty_like`` / ``new_empty`` on the GPU comes back FILLED instead of holding
whatever the caching allocator hands out (often the previous, almost identical, result at the same address -- which
hides a read of memory that this run never wrote).

    SSA_POISON=nan   every byte 0xFF: NaN in float32 and float64, -1 in the integer types
    SSA_POISON=big   every byte 0x7E: 8.4e37 in float32, 1.9e300 in float64 (a huge FINITE value: a stale word that
                     is only ever multiplied by zero stays harmless, anything else explodes)

``install()`` is called by ``tests/conftest.py`` and by the tools when the variable is set.  The product never imports
this file.
"""
import os

_installed = False


def mode():
    m = os.environ.get("SSA_POISON", "").strip().lower()
    return m if m in ("nan", "big") else ""


def install(which: str = "") -> bool:
    global _installed
    which = which or mode()
    if not which or _installed:
        return _installed
    import torch

    byte = 0xFF if which == "nan" else 0x7E
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def fill(t):
        if t.is_cuda and t.numel():
            # (as_strided storage view: padded leading dimensions included)
            raw = t.untyped_storage()
            torch.tensor([], dtype=torch.uint8, device=t.device).set_(raw).fill_(byte)
        return t

    def empty(*args, **kwargs):
        return fill(real_empty(*args, **kwargs))

    def empty_like(*args, **kwargs):
        return fill(real_empty_like(*args, **kwargs))

    torch.empty, torch.empty_like = empty, empty_like
    _installed = True
    return True
