"""Callers and data formats on either side of the hot path (SURVEY.md §8f "next" #2 and #3), host-side Python.

* batch assembly: ``collate_fn`` / ``RepeatDataset`` / ``PDBDataset`` with the contract of the reference's
  ProteinReDiff/data.py:80-185 (atoms first, residues after, ``residue_type + 1``, padded to the longest complex);
* output handling of ``generate.py``: sequence decoding (generate.py:75-91), CA placement (:65-74), multi-model PDB
  text (protein.py:124-174);
* ``generate_samples``: the ``Trainer.predict`` loop of generate.py:138-159 without Lightning.

rdkit / Biopython / TM-align are not needed: ligands are handled as coordinate arrays, proteins as plain arrays.
Nothing here is on the arithmetic path; the model call goes to the HIP path through ``ProteinReDiffModel``.
"""
from __future__ import annotations

import dataclasses
import os
from pathlib import Path
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from .constants import NUM_RESIDUE_ATOMS, RESIDUE_TYPES

UNKNOWN_RESIDUE_NAME = "UNK"          # PDB name written for aatype -1 ('X': undetermined / undecoded residue)
RESIDUE_NAMES = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE",
                 "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL"]
RESIDUE_ATOMS = ["N", "CA", "C", "CB", "O", "CG", "CG1", "CG2", "OG", "OG1", "SG", "CD", "CD1", "CD2", "ND1", "ND2",
                 "OD1", "OD2", "SD", "CE", "CE1", "CE2", "CE3", "NE", "NE1", "NE2", "OE1", "OE2", "CH2", "NH1", "NH2",
                 "OH", "CZ", "CZ2", "CZ3", "NZ", "OXT"]
PDB_CHAIN_IDS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"
assert len(RESIDUE_ATOMS) == NUM_RESIDUE_ATOMS


# ---------------------------------------------------------------------------------------------------
# batch assembly (data.py:80-185)
# ---------------------------------------------------------------------------------------------------

def collate_fn(data_list: Sequence[Mapping[str, Any]]) -> Dict[str, Any]:
    """Pad and stack per-complex dicts.  Key families decide the placement, as in data.py:80-142:
    ``atom_*`` -> rows [0, na); ``bond_*`` -> the [0, na)^2 corner; ``residue_*`` -> rows [na, na+nr)
    (and ``residue_type`` shifted by +1 so that 0 means pad / atom slot); ``*_mol`` -> python lists."""
    n_max = max(d["num_atoms"] + d["num_residues"] for d in data_list)
    out: Dict[str, Any] = {}
    for key, first in data_list[0].items():
        if key.startswith("atom_"):
            tail = (0, 0) * (first.dim() - 1)
            out[key] = torch.stack([F.pad(d[key], tail + (0, n_max - d["num_atoms"])) for d in data_list])
        elif key.startswith("bond_"):
            tail = (0, 0) * (first.dim() - 2)
            out[key] = torch.stack([F.pad(d[key], tail + (0, n_max - d["num_atoms"]) * 2) for d in data_list])
        elif key.startswith("residue_"):
            tail = (0, 0) * (first.dim() - 1)
            shift = 1 if key.endswith("_type") else 0
            out[key] = torch.stack([
                F.pad(d[key] + shift, tail + (d["num_atoms"], n_max - d["num_atoms"] - d["num_residues"]))
                for d in data_list])
        elif key.endswith("_mol"):
            out[key] = [d[key] for d in data_list]
        elif torch.is_tensor(first):
            out[key] = torch.stack([d[key] for d in data_list])
        elif isinstance(first, (int, float)):
            out[key] = torch.tensor([d[key] for d in data_list])
        else:
            out[key] = [d[key] for d in data_list]
    return out


class RepeatDataset(torch.utils.data.Dataset):
    """data.py:145-155: the same complex ``repeat`` times (one entry per requested sample)."""

    def __init__(self, data: Mapping[str, Any], repeat: int):
        self.data, self.repeat = data, repeat

    def __len__(self):
        return self.repeat

    def __getitem__(self, index: int):
        return self.data


class InferenceDataset(torch.utils.data.Dataset):
    """data.py:157-168: a LIST of featurised complexes, one entry per index (the ``predict_batch_*`` scripts' dataset);
    ``repeat`` is the length the caller declares, as in the reference."""

    def __init__(self, data: Sequence[Mapping[str, Any]], repeat: int):
        self.data, self.repeat = data, repeat

    def __len__(self):
        return self.repeat

    def __getitem__(self, index: int):
        return self.data[index]


class PDBDataset(torch.utils.data.Dataset):
    """data.py:170-185: ``root/<pdb_id>/{ligand,protein}_data.pt`` as written by preprocess_pdbbind.py:79-83."""

    def __init__(self, root_dir: Union[str, Path], pdb_ids: Sequence[str]):
        self.root_dir, self.pdb_ids = Path(root_dir), list(pdb_ids)

    def __len__(self):
        return len(self.pdb_ids)

    def __getitem__(self, index: int):
        pdb_id = self.pdb_ids[index]
        ligand = torch.load(self.root_dir / pdb_id / "ligand_data.pt", weights_only=False)
        protein = torch.load(self.root_dir / pdb_id / "protein_data.pt", weights_only=False)
        return {"pdb_id": pdb_id, **ligand, **protein}


class BucketBatchSampler(torch.utils.data.Sampler):
    """Batches of complexes of SIMILAR size for data-parallel training.

    The reference shuffles complexes freely (data.py:238-246: ``DataLoader(shuffle=True)``), so a batch is padded to its
    longest complex and, under DDP, every optimisation step lasts as long as the rank that drew the largest one (N ranges over
    100 .. 384 in PDBbind: the pair track costs O(N^3)).  This sampler keeps the reference's statistics -- every complex once per
    epoch, order reshuffled every epoch -- but draws each batch from one size bucket and hands the ranks of a step batches of
    the SAME bucket:

    * ``sizes[i]`` = num_atoms + num_residues of complex i; complexes are sorted into buckets ``bucket_width`` nodes wide;
    * per epoch (``set_epoch``) the members of every bucket are shuffled, cut into batches of ``batch_size``, the batches of a
      bucket grouped into steps of ``world_size`` batches (an incomplete last group is completed by wrapping around inside the
      bucket, as DistributedSampler pads), and the steps shuffled; rank r takes batch r of every step;
    * deterministic in ``(seed, epoch)``, identical on every rank, no communication.

    Use as ``DataLoader(dataset, batch_sampler=BucketBatchSampler(...), collate_fn=collate_fn)``."""

    def __init__(self, sizes: Sequence[int], batch_size: int, world_size: int = 1, rank: int = 0, bucket_width: int = 32,
                 seed: int = 0, shuffle: bool = True):
        if not 0 <= rank < world_size:
            raise ValueError(f"rank {rank} outside world of {world_size}")
        if batch_size < 1 or bucket_width < 1:
            raise ValueError("batch_size and bucket_width must be positive")
        self.sizes = [int(v) for v in sizes]
        self.batch_size, self.world_size, self.rank = batch_size, world_size, rank
        self.bucket_width, self.seed, self.shuffle, self.epoch = bucket_width, seed, shuffle, 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _steps(self) -> List[List[List[int]]]:
        g = torch.Generator().manual_seed((self.seed * 1_000_003 + self.epoch) & 0x7FFFFFFF)
        buckets: Dict[int, List[int]] = {}
        for i, n in enumerate(self.sizes):
            buckets.setdefault(n // self.bucket_width, []).append(i)
        steps: List[List[List[int]]] = []
        for key in sorted(buckets):
            members = buckets[key]
            if self.shuffle:
                members = [members[j] for j in torch.randperm(len(members), generator=g).tolist()]
            batches = [members[k:k + self.batch_size] for k in range(0, len(members), self.batch_size)]
            n0, k = len(batches), 0
            while len(batches) % self.world_size:                 # complete the last step by wrapping around inside the bucket:
                batches.append(batches[k % n0])                   # distinct batches as long as the bucket has them
                k += 1
            steps.extend(batches[k:k + self.world_size] for k in range(0, len(batches), self.world_size))
        if self.shuffle:
            steps = [steps[j] for j in torch.randperm(len(steps), generator=g).tolist()]
        return steps

    def __iter__(self):
        for step in self._steps():
            yield step[self.rank]

    def __len__(self) -> int:
        return len(self._steps())

    def padding_waste(self) -> Tuple[float, float]:
        """(this sampler, free shuffling with the same seed): fraction of the O(N^2) pair positions of an epoch that are
        padding or idle time of a rank waiting for the largest batch of its step."""
        def waste(steps):
            used = total = 0
            for step in steps:
                nmax = max(self.sizes[i] for bt in step for i in bt)
                for bt in step:
                    total += len(bt) * nmax * nmax
                    used += sum(self.sizes[i] ** 2 for i in bt)
            return 1.0 - used / max(total, 1)
        g = torch.Generator().manual_seed((self.seed * 1_000_003 + self.epoch) & 0x7FFFFFFF)
        order = torch.randperm(len(self.sizes), generator=g).tolist()
        free = [order[k:k + self.batch_size] for k in range(0, len(order), self.batch_size)]
        free_steps = [free[k:k + self.world_size] for k in range(0, len(free), self.world_size)]
        return waste(self._steps()), waste(free_steps)


class PDBDataModule:
    """data.py:206-259 without Lightning: the three id lists under ``data_dir`` and loaders over the preprocessed cache.
    ``bucket_width`` > 0 makes the TRAINING loader draw size-bucketed batches (BucketBatchSampler; sizes are read once from
    the cache); 0 reproduces the reference's free shuffling."""

    def __init__(self, data_dir: Union[str, Path] = "data", batch_size: int = 1, num_workers: int = 1, bucket_width: int = 32,
                 world_size: int = 1, rank: int = 0, seed: int = 0, group=None):
        self.data_dir = Path(data_dir)
        self.group = group                  # process group of the data-parallel ranks (None: the default group) -- _train_sizes
        self.cache_dir = self.data_dir / "PDB_processed_cache"
        self.batch_size, self.num_workers, self.bucket_width = batch_size, num_workers, bucket_width
        self.world_size, self.rank, self.seed = world_size, rank, seed
        self.train_sampler: Optional[BucketBatchSampler] = None

    def setup(self, stage: Optional[str] = None) -> None:
        def ids(name):
            with open(self.data_dir / name, "r") as f:
                return [line.strip() for line in f if line.strip()]
        self.train_pdb_ids, self.val_pdb_ids, self.test_pdb_ids = ids("PRD_train_pdb_ids"), ids("PRD_val_pdb_ids"), ids("PRD_test_pdb_ids")

    def train_dataloader(self):
        ds = PDBDataset(self.cache_dir, self.train_pdb_ids)
        if self.bucket_width <= 0:
            if self.world_size > 1:         # what Lightning injects under DDP (train.py:38): a disjoint shard per rank, reshuffled per epoch
                self.train_sampler = torch.utils.data.distributed.DistributedSampler(
                    ds, num_replicas=self.world_size, rank=self.rank, shuffle=True, seed=self.seed)
                return torch.utils.data.DataLoader(ds, batch_size=self.batch_size, sampler=self.train_sampler,
                                                   num_workers=self.num_workers, collate_fn=collate_fn)
            return torch.utils.data.DataLoader(ds, batch_size=self.batch_size, shuffle=True, num_workers=self.num_workers, collate_fn=collate_fn)
        self.train_sampler = BucketBatchSampler(self._train_sizes(ds), self.batch_size, self.world_size, self.rank, self.bucket_width, self.seed)
        return torch.utils.data.DataLoader(ds, batch_sampler=self.train_sampler, num_workers=self.num_workers, collate_fn=collate_fn)

    def set_epoch(self, epoch: int) -> None:
        """Reshuffle for a new epoch (both samplers are deterministic in (seed, epoch), identical on every rank)."""
        if self.train_sampler is not None:
            self.train_sampler.set_epoch(epoch)

    def _train_sizes(self, ds) -> List[int]:
        """num_atoms + num_residues of every training complex.  Read from ``<cache>/sizes_index.json`` where an entry's
        FINGERPRINT -- (mtime_ns, size) of the complex's two cache files -- still matches: a cache that was re-preprocessed
        (other cropping / featurisation) is re-scanned instead of driving the buckets with stale sizes.  Under an initialised
        process group rank 0 alone scans and writes the index and broadcasts its verdict, the other ranks then read the file (ranks that
        start together on a shared cache neither scan it N times nor overwrite each other's file; a scan that raises on rank 0 raises
        on every rank instead of leaving the others in a barrier).  Best effort on a read-only cache:
        it is scanned every time."""
        import json
        index_path = self.cache_dir / "sizes_index.json"

        def fingerprint(pid):
            fp = []
            for name in ("ligand_data.pt", "protein_data.pt"):
                try:
                    st = os.stat(self.cache_dir / pid / name)
                    fp += [int(st.st_mtime_ns), int(st.st_size)]
                except OSError:
                    fp += [0, 0]
            return fp

        def load():
            if not index_path.exists():
                return {}
            try:
                with open(index_path, "r") as f:
                    raw = json.load(f)
                if not isinstance(raw, dict) or raw.get("version") != 2:
                    return {}           # round-4 files had no fingerprints: rebuilt once
                return {str(k): (int(v[0]), [int(x) for x in v[1]]) for k, v in raw["entries"].items()}
            except (OSError, ValueError, KeyError, TypeError, IndexError):
                return {}

        def scan_and_store(index):
            stale = [i for i, pid in enumerate(ds.pdb_ids) if pid not in index or index[pid][1] != fingerprint(pid)]
            for i in stale:
                d = ds[i]
                index[ds.pdb_ids[i]] = (int(d["num_atoms"]) + int(d["num_residues"]), fingerprint(ds.pdb_ids[i]))
            if stale:
                try:
                    tmp = index_path.with_suffix(".json.tmp%d" % os.getpid())
                    with open(tmp, "w") as f:
                        json.dump({"version": 2, "entries": {k: [v[0], v[1]] for k, v in index.items()}}, f)
                    os.replace(tmp, index_path)
                except OSError:
                    pass
            return index

        distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        if not distributed:
            index = scan_and_store(load())
        else:
            # Rank 0 scans; its verdict travels as an object broadcast (which is also the rendezvous: a bare barrier would leave the
            # other ranks waiting for ever when rank 0's scan raises -- a corrupt cache item -- and, with NCCL, needs a device already
            # set).  ``self.group`` (None: the default group) is the group the loaders of this module are sharded over.
            group = getattr(self, "group", None)
            rank = torch.distributed.get_rank(group)
            verdict = [None]
            index = None
            if rank == 0:
                try:
                    index = scan_and_store(load())
                    verdict = [("ok", None)]
                except Exception as exc:                # noqa: BLE001 -- re-raised on every rank below
                    verdict = [("error", f"{type(exc).__name__}: {exc}")]
            src = torch.distributed.get_global_rank(group, 0) if group is not None else 0
            torch.distributed.broadcast_object_list(verdict, src=src, group=group)
            if verdict[0][0] != "ok":
                raise RuntimeError(f"PDBDataModule: rank 0 failed while scanning the cache for complex sizes ({verdict[0][1]})")
            if rank != 0:
                index = load()
                if any(pid not in index or index[pid][1] != fingerprint(pid) for pid in ds.pdb_ids):
                    index = scan_and_store(index)       # rank 0 could not write (read-only cache): scan here too
        return [index[pid][0] for pid in ds.pdb_ids]

    def val_dataloader(self):
        return torch.utils.data.DataLoader(PDBDataset(self.cache_dir, self.val_pdb_ids), batch_size=self.batch_size,
                                           num_workers=self.num_workers, collate_fn=collate_fn)

    def test_dataloader(self):
        return torch.utils.data.DataLoader(PDBDataset(self.cache_dir, self.test_pdb_ids), batch_size=self.batch_size,
                                           num_workers=self.num_workers, collate_fn=collate_fn)


# ---------------------------------------------------------------------------------------------------
# proteins and output handling (protein.py:50-202, generate.py:65-91)
# ---------------------------------------------------------------------------------------------------

@dataclasses.dataclass(frozen=True)
class Protein:
    chain_index: np.ndarray
    residue_index: np.ndarray
    aatype: np.ndarray
    atom_pos: np.ndarray
    atom_mask: np.ndarray


def protein_from_sequence(sequence: str) -> Protein:
    """protein.py:177-191: CA-only placeholder protein for a sequence ('X' -> -1)."""
    idx = {name: i for i, name in enumerate(RESIDUE_TYPES)}
    idx["X"] = -1
    aatype = np.array([idx[s] for s in sequence], dtype=np.int64)
    n = len(aatype)
    mask = np.zeros((n, NUM_RESIDUE_ATOMS), dtype=np.float32)
    mask[:, 1] = 1.0
    return Protein(np.zeros((n,), np.int64), np.arange(n, dtype=np.int64), aatype,
                   np.zeros((n, NUM_RESIDUE_ATOMS, 3), np.float32), mask)


def protein_to_data(prot: Protein, **extra) -> Dict[str, Any]:
    """data.py:59-77 without the rdkit CA molecule."""
    return {"num_residues": len(prot.aatype), "residue_type": torch.from_numpy(prot.aatype),
            "residue_mask": torch.ones(len(prot.aatype)), "residue_chain_index": torch.from_numpy(prot.chain_index),
            "residue_index": torch.from_numpy(prot.residue_index), "residue_atom_pos": torch.from_numpy(prot.atom_pos),
            "residue_atom_mask": torch.from_numpy(prot.atom_mask), **extra}


def protein_to_pdb_string(prot: Protein) -> str:
    """Fixed-column ATOM records, one per present atom (protein.py:124-156)."""
    lines, serial = [], 1
    for i in range(prot.chain_index.shape[0]):
        aa = int(prot.aatype[i])
        chain, resi = PDB_CHAIN_IDS[prot.chain_index[i]], prot.residue_index[i]
        resn = RESIDUE_NAMES[aa] if aa >= 0 else UNKNOWN_RESIDUE_NAME       # -1 must not wrap around to VAL
        for xyz, present, name in zip(prot.atom_pos[i], prot.atom_mask[i], RESIDUE_ATOMS):
            if present < 0.5:
                continue
            field = name if len(name) >= 4 else " " + name.ljust(3)
            lines.append(f"{'ATOM':<6}{serial:>5} {field}{'':>1}{resn:>3} {chain:>1}{resi:>4}{'':>1}   "
                         f"{xyz[0]:>8.3f}{xyz[1]:>8.3f}{xyz[2]:>8.3f}{1.0:>6.2f}{0.0:>6.2f}          {name[0]:>2}{'':>2}".ljust(80))
            serial += 1
    return "\n".join(lines) + "\n"


def proteins_to_pdb_file(proteins: Iterable[Protein], pdb_path: Union[str, Path]) -> None:
    """Multi-model PDB (protein.py:165-174)."""
    text = ""
    for model_id, prot in enumerate(proteins, 1):
        text += f"MODEL      {model_id:>3}".ljust(80) + "\n" + protein_to_pdb_string(prot) + "ENDMDL".ljust(80) + "\n"
    Path(pdb_path).write_text(text)


def predict_seq(logits) -> List[str]:
    """generate.py:75-80: argmax over the 21 classes with alphabet ["X"] + RESIDUE_TYPES."""
    tokens = torch.argmax(torch.softmax(torch.as_tensor(logits), dim=-1), dim=-1)
    alphabet = ["X"] + RESIDUE_TYPES
    return [alphabet[int(i)] for i in tokens]


def update_seq(protein: Protein, logits) -> Protein:
    """generate.py:82-91: decoded sequence with leading / trailing X stripped becomes the new aatype."""
    sequence = "".join(predict_seq(logits)).lstrip("X").rstrip("X")
    return dataclasses.replace(protein, aatype=np.array([RESIDUE_TYPES.index(s) for s in sequence], dtype=np.int64))


def update_pos(protein: Protein, num_ligand_atoms: int, pos: np.ndarray) -> Tuple[Protein, np.ndarray]:
    """generate.py:65-74: rows [na:] are the CA trace, rows [:na] the ligand atoms (returned as an array)."""
    atom_pos = np.zeros_like(protein.atom_pos)
    atom_pos[:, 1] = pos[num_ligand_atoms: num_ligand_atoms + atom_pos.shape[0]]
    atom_mask = np.zeros_like(protein.atom_mask)
    atom_mask[:, 1] = 1.0
    return dataclasses.replace(protein, atom_pos=atom_pos, atom_mask=atom_mask), np.asarray(pos[:num_ligand_atoms])


# ---------------------------------------------------------------------------------------------------
# generate.py:138-195 without Lightning / rdkit / TM-align
# ---------------------------------------------------------------------------------------------------

@torch.inference_mode()
def generate_samples(model, data: Mapping[str, Any], num_samples: int, batch_size: int = 1, seed: int = 0,
                     output_dir: Optional[Union[str, Path]] = None):
    """Draw ``num_samples`` samples of one featurised complex ``data`` (the dict of ligand_to_data ∪ protein_to_data).

    Returns (positions [S,N,3] in Angstrom, logits [S,N,21], proteins, ligand_positions).  With ``output_dir`` the CA
    models go to ``sample_protein.pdb`` and the ligand coordinates to ``sample_ligand_pos.npy`` (the reference writes an
    SDF through rdkit and aligns every sample with TM-align first -- both out of scope here).

    Every sample carries its DECODED sequence (generate.py:83-91 decodes every sample): residues decoded as 'X' become
    aatype -1 and are written as UNK; the input sequence is never silently kept.  The reference strips leading / trailing X
    and raises on an inner one (``RESIDUE_TYPES.index("X")``); keeping the length and marking the residue instead keeps the
    CA trace and the sequence aligned.  A ``UserWarning`` names the samples that contain undetermined residues."""
    import warnings

    from .synthetic import NoiseSource, batch_to
    device = model.device
    positions, logits = [], []
    for start in range(0, num_samples, batch_size):
        idx = list(range(start, min(start + batch_size, num_samples)))
        batch = collate_fn([data] * len(idx))
        batch = batch_to({k: v for k, v in batch.items() if torch.is_tensor(v)}, device)
        pos, lg = model.sample(batch, sources=[NoiseSource(seed, k) for k in idx])
        positions.append(pos.cpu())
        logits.append(lg.cpu())
    positions, logits = torch.cat(positions).numpy(), torch.cat(logits).numpy()
    na, nr = int(data["num_atoms"]), int(data["num_residues"])
    template = Protein(np.asarray(data["residue_chain_index"]), np.asarray(data["residue_index"]),
                       np.asarray(data["residue_type"]), np.asarray(data["residue_atom_pos"], dtype=np.float32),
                       np.asarray(data["residue_atom_mask"], dtype=np.float32))
    proteins, ligands = [], []
    for pos, lg in zip(positions, logits):
        prot, lig = update_pos(template, na, pos)
        seq = predict_seq(lg[na: na + nr])
        prot = dataclasses.replace(prot, aatype=np.array([RESIDUE_TYPES.index(s) if s != "X" else -1 for s in seq], dtype=np.int64))
        proteins.append(prot)
        ligands.append(lig)
    undetermined = [k for k, p in enumerate(proteins) if (p.aatype < 0).any()]
    if undetermined:
        warnings.warn(f"samples {undetermined} decode to 'X' at some residues; those are written as UNK", UserWarning)
    if output_dir is not None:
        out = Path(output_dir)
        out.mkdir(parents=True, exist_ok=True)
        proteins_to_pdb_file(proteins, out / "sample_protein.pdb")
        np.save(out / "sample_ligand_pos.npy", np.stack(ligands))
    return positions, logits, proteins, ligands
