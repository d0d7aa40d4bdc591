from .pretrain import PretrainDecoderTrainer, PretrainEncoderTrainer, WarmupCosine  # noqa: F401
from .finetune import FineTuneTrainer  # noqa: F401
