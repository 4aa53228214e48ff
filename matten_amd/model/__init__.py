"""Training-loop shell of the models (mirror of the reference's matten.model package)."""
from .model import BaseModel  # noqa: F401
