from .pretrain import PretrainEncoderTrainer, WarmupCosine  # noqa: F401
from .finetune import FineTuneTrainer  # noqa: F401
