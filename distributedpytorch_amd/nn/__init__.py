from .engine import Act, Context, ParamStore, Tape  # noqa: F401
