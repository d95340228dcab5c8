"""Exception taxonomy of the reference (lib/src/internal/lz4_exception.dart:2-35).

Class names and messages are the reference's, so callers (and the parity tests)
can match on either.  ``str(e)`` follows the Dart ``toString`` form
``'<runtimeType>: <message>'`` (:10).
"""


class Lz4Exception(Exception):
    def __init__(self, message: str):
        super().__init__(message)
        self.message = message

    def __str__(self):
        return f"{type(self).__name__}: {self.message}"


class Lz4FormatException(Lz4Exception):
    pass


class Lz4CorruptDataException(Lz4Exception):
    pass


class Lz4OutputLimitException(Lz4Exception):
    pass


class Lz4UnsupportedFeatureException(Lz4Exception):
    pass


class Lz4BackendError(RuntimeError):
    """HIP/runtime failure or a request outside the device path (not a reference error)."""


class RangeError(ValueError):
    """Stands in for Dart's RangeError (argument validation)."""
