from .pretrain import PretrainEncoderTrainer, WarmupCosine  # noqa: F401
