"""Host-side mirror of ``anemoi.models.layers``: same class names, constructor kwargs and ``state_dict`` layout,
forward passes executed by the gfx950 kernels of ``libanemoi_amd.so``."""
