"""Error types of the host layer.

Mirrors ``VqError`` (reference src/core/error.rs:4-28) with the same ``Display`` text.  The
reference's Python binding turns every ``VqError`` into ``ValueError(e.to_string())``
(pyvq/src/pq.rs:86), so all of them derive from ``ValueError`` and tests written against
pyvq (``pytest.raises(ValueError, match="Dimension mismatch")``) read the same here.
"""


class VqError(ValueError):
    """Base of every error raised by this package."""


class DimensionMismatch(VqError):
    def __init__(self, expected: int, found: int):
        self.expected, self.found = int(expected), int(found)
        super().__init__(f"Dimension mismatch: expected {self.expected}, found {self.found}")


class EmptyInput(VqError):
    def __init__(self):
        super().__init__("Empty input: at least one vector is required")


class InvalidParameter(VqError):
    def __init__(self, parameter: str, reason: str):
        self.parameter, self.reason = parameter, reason
        super().__init__(f"Invalid parameter '{parameter}': {reason}")


class InvalidData(VqError):
    def __init__(self, msg: str):
        super().__init__(f"Invalid data: {msg}")


class FfiError(VqError):
    """Device / library failure (``VqError::FfiError``, src/core/error.rs:26-27)."""

    def __init__(self, msg: str, status: int = -99):
        self.status = status
        super().__init__(f"FFI error: {msg}")
