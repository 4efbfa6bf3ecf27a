"""Plugin implementations backed by libsmqtk_hip.so (no CPU fallbacks)."""
from .. import _lib


def _require_usable(obj: object) -> None:
    """Fail loudly when the HIP library or a GPU is missing."""
    if not _lib.usable():
        raise _lib.HipError(
            "%s needs libsmqtk_hip.so and a visible MI355X; there is no CPU "
            "fallback (library present: %s, devices: %d)"
            % (type(obj).__name__, _lib.library_present(), _lib.device_count()))
