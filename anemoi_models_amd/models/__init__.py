from .encoder_processor_decoder import AnemoiModelEncProcDec
from .hierarchical import AnemoiModelEncProcDecHierarchical

__all__ = ["AnemoiModelEncProcDec", "AnemoiModelEncProcDecHierarchical"]
