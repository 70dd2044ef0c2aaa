from .batch_train import (BatchProcessor, DefaultBatchProcessor, GradientAccumulationBatchProcessor,  # noqa: F401
                          get_batch_processor_from_config)
from .step import DefaultStep, GraphStep, MixedPrecisionStep, Step  # noqa: F401
