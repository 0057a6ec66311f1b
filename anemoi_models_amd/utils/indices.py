"""Stand-in for the part of ``anemoi.models.data_indices.IndexCollection`` the forward path reads.

The model root only touches ``data_indices.internal_model.{input,output}`` for
``len()``, ``.prognostic``, ``.full``, ``.diagnostic`` and ``.name_to_index``
(reference models/encoder_processor_decoder.py:103,108-125).  A real
``IndexCollection`` can be passed instead; this class exists so tests and
``bench.py`` need no dataset configuration.

Variable layout: input = [prognostic..., forcing...], output = [prognostic..., diagnostic...].
"""

from __future__ import annotations

import torch


class _ModelIndex:
    def __init__(self, names, prognostic, diagnostic=(), forcing=()):
        self.name_to_index = {n: i for i, n in enumerate(names)}
        self.full = torch.arange(len(names), dtype=torch.int64)
        self.prognostic = torch.as_tensor(list(prognostic), dtype=torch.int64)
        self.diagnostic = torch.as_tensor(list(diagnostic), dtype=torch.int64)
        self.forcing = torch.as_tensor(list(forcing), dtype=torch.int64)

    def __len__(self) -> int:
        return len(self.name_to_index)


class _Internal:
    def __init__(self, inp, out):
        self.input = inp
        self.output = out


class _DataIndex:
    """Dataset-side view (reference data_indices/index.py:46-63): positions in the FULL variable list."""

    def __init__(self, name_to_index, full, prognostic, diagnostic=(), forcing=()):
        self.name_to_index = dict(name_to_index)
        self.full = torch.as_tensor(list(full), dtype=torch.int64)
        self.prognostic = torch.as_tensor(list(prognostic), dtype=torch.int64)
        self.diagnostic = torch.as_tensor(list(diagnostic), dtype=torch.int64)
        self.forcing = torch.as_tensor(list(forcing), dtype=torch.int64)

    def __len__(self) -> int:
        return int(self.full.numel())


class SimpleDataIndices:
    """Dataset layout ``[prognostic..., forcing..., diagnostic...]``; input = all but diagnostic, output = all but
    forcing (reference data_indices/collection.py:24-103 without remapping: ``internal_*`` == the plain views)."""

    def __init__(self, n_prognostic: int, n_forcing: int = 0, n_diagnostic: int = 0, names=None) -> None:
        prog = [f"prog_{i}" for i in range(n_prognostic)]
        forc = [f"forc_{i}" for i in range(n_forcing)]
        diag = [f"diag_{i}" for i in range(n_diagnostic)]
        if names is not None:  # variable names in dataset order (prognostic, forcing, diagnostic)
            names = list(names)
            assert len(names) == n_prognostic + n_forcing + n_diagnostic
            prog, forc = names[:n_prognostic], names[n_prognostic:n_prognostic + n_forcing]
            diag = names[n_prognostic + n_forcing:]
        inp = _ModelIndex(prog + forc, range(n_prognostic), forcing=range(n_prognostic, n_prognostic + n_forcing))
        out = _ModelIndex(prog + diag, range(n_prognostic),
                          diagnostic=range(n_prognostic, n_prognostic + n_diagnostic))
        self.internal_model = _Internal(inp, out)
        self.model = self.internal_model
        names = prog + forc + diag
        self.name_to_index = {n: i for i, n in enumerate(names)}
        np_, nf = n_prognostic, n_forcing
        d_in = _DataIndex(self.name_to_index, range(np_ + nf), range(np_), forcing=range(np_, np_ + nf))
        d_out = _DataIndex(self.name_to_index, list(range(np_)) + list(range(np_ + nf, len(names))), range(np_),
                           diagnostic=range(np_ + nf, len(names)))
        self.data = _Internal(d_in, d_out)
        self.internal_data = self.data
        self.num_input = len(inp)
        self.num_output = len(out)
