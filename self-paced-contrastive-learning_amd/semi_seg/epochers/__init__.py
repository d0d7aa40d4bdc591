from .pretrain import PretrainDecoderEpocher, PretrainEncoderEpocher, unzip_twice_transformed  # noqa: F401
from .finetune import EvalEpocher, FineTuneEpocher  # noqa: F401
from .legacy import ContrastiveProjectorWrapper, InfoNCEPretrainEpocher  # noqa: F401
