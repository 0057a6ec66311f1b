from .encoder_processor_decoder import AnemoiModelEncProcDec

__all__ = ["AnemoiModelEncProcDec"]
