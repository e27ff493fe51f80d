from .base_trainer import Multi_BaseTrainer_dist  # noqa: F401
