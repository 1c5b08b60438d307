"""Exception types of the routed-module contract.

Names, constructor signatures and messages mirror ultralytics/utils/errors.py:45-76 so that
callers and tests written against the reference catch the same types.
"""


class YOLOMasterError(Exception):
    """Base error of the YOLO-Master path."""


class MoERouterError(YOLOMasterError):
    """Raised when a routed module receives invalid input or configuration."""


class ShapeMismatchError(YOLOMasterError):
    """Raised when a routed tensor violates an expected shape contract."""

    def __init__(self, expected, actual, context: str = ""):
        self.expected = expected
        self.actual = actual
        self.context = context
        message = f"Shape mismatch: expected {expected}, got {actual}"
        if context:
            message += f" [{context}]"
        super().__init__(message)
