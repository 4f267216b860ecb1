"""Exception types of the plugin API, as /root/reference/riemann/sampling_errors.py:11-29."""


class RiemannBaseError(Exception):
    """Base class; ``str()`` gives ``"<ClassName>: <msg>"`` (sampling_errors.py:11-20)."""

    def __init__(self, msg):
        self.msg = msg

    def __str__(self):
        return "{}: {}".format(self.__class__.__name__, self.msg)


class ParameterError(RiemannBaseError):
    """Faulty parameters / shape mismatch (sampling_errors.py:23-27)."""
    pass
