"""Host-side helpers: config containers, ``instantiate`` and the data-indices stand-in."""
