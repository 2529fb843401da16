from .build import get_model, get_optimizer  # noqa: F401
