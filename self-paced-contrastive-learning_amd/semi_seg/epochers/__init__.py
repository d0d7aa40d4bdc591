from .pretrain import PretrainEncoderEpocher, unzip_twice_transformed  # noqa: F401
