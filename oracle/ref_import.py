"""Import the *reference* ProteinReDiff package in this container (oracle harness only).

Works only where /root/reference exists (the build container).  Four third-party modules the
reference imports at module load are absent here; none is on the arithmetic path, so inert
stand-ins are registered in ``sys.modules`` first.  Nothing from the
reference is copied: it is imported from where it lies, read-only.
"""
import contextlib
import sys
import types

import torch
from torch import nn

REFERENCE_ROOT = "/root/reference"


def _install_stubs():
    if "pytorch_lightning" in sys.modules:
        return
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    pl.LightningModule = LightningModule
    pl.LightningDataModule = object
    pl.Trainer = object
    pl.seed_everything = lambda *a, **k: None
    sys.modules["pytorch_lightning"] = pl

    ema = types.ModuleType("torch_ema")

    class ExponentialMovingAverage:
        def __init__(self, params, decay):
            pass

        def to(self, *a, **k):
            return self

        def update(self, *a, **k):
            pass

        def state_dict(self):
            return {}

        def load_state_dict(self, sd):
            pass

        def average_parameters(self):
            return contextlib.nullcontext()

    ema.ExponentialMovingAverage = ExponentialMovingAverage
    sys.modules["torch_ema"] = ema

    rdkit = types.ModuleType("rdkit")
    chem = types.ModuleType("rdkit.Chem")
    chem.Mol = chem.Atom = chem.Bond = object
    rdkit.Chem = chem
    sys.modules["rdkit"] = rdkit
    sys.modules["rdkit.Chem"] = chem

    bio = types.ModuleType("Bio")
    pdb = types.ModuleType("Bio.PDB")
    parser = types.ModuleType("Bio.PDB.PDBParser")
    parser.PDBParser = object
    bio.PDB = pdb
    pdb.PDBParser = parser
    sys.modules["Bio"] = bio
    sys.modules["Bio.PDB"] = pdb
    sys.modules["Bio.PDB.PDBParser"] = parser


def import_reference():
    """Returns the reference's ``ProteinReDiff.model`` and ``ProteinReDiff.modules`` modules."""
    sys.dont_write_bytecode = True
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import ProteinReDiff.model as ref_model      # noqa: E402
    import ProteinReDiff.modules as ref_modules  # noqa: E402
    return ref_model, ref_modules
