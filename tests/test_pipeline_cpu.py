"""CPU: the callers / data formats around the hot path against fixtures captured from the
imported reference (tests/golden/host.npz, oracle/gen_golden.py): collate_fn, PDB text, sequence decoding."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from protein_redesign_amd import pipeline as PL
from protein_redesign_amd.synthetic import synthetic_sample


def host():
    return np.load(os.path.join(ROOT, "tests", "golden", "host.npz"), allow_pickle=False)


def test_collate_matches_reference():
    z = host()
    samples = [synthetic_sample(4, 9, esm_dim=8, seed=31), synthetic_sample(6, 5, esm_dim=8, seed=32)]
    batch = PL.collate_fn(samples)
    keys = [k[len("collate_"):] for k in z.files if k.startswith("collate_")]
    assert len(keys) >= 15
    for k in keys:
        assert torch.is_tensor(batch[k]), k
        assert batch[k].shape == z["collate_" + k].shape, k
        assert np.array_equal(batch[k].numpy(), z["collate_" + k]), k
    # layout facts the kernels rely on: atoms first, residues after, residue_type shifted by one
    assert batch["atom_mask"][0, :4].sum() == 4 and batch["residue_mask"][0, 4:13].sum() == 9
    assert int(batch["residue_type"][0, :4].abs().sum()) == 0 and int(batch["residue_type"][0, 4:13].min()) >= 1


def test_pdb_text_matches_reference(tmp_path):
    z = host()
    prot = PL.Protein(z["pdb_chain_index"], z["pdb_residue_index"], z["pdb_aatype"], z["pdb_atom_pos"], z["pdb_atom_mask"])
    text = PL.protein_to_pdb_string(prot)
    assert text == str(z["pdb_text"])
    PL.proteins_to_pdb_file([prot, prot], tmp_path / "m.pdb")
    written = (tmp_path / "m.pdb").read_text()
    assert written.count("MODEL") == 2 and written.count("ENDMDL") == 2 and all(len(l) == 80 for l in written.splitlines())


def test_sequence_decoding_matches_reference():
    z = host()
    assert "".join(PL.predict_seq(z["seq_logits"])) == str(z["seq_pred"])
    prot = PL.update_seq(PL.protein_from_sequence("A" * 9), z["seq_logits"])
    assert np.array_equal(prot.aatype, z["seq_aatype"])
    fs = PL.protein_from_sequence("ACDXW")
    assert np.array_equal(fs.aatype, z["fromseq_aatype"]) and np.array_equal(fs.atom_mask, z["fromseq_mask"])


def test_pdb_dataset_roundtrip(tmp_path):
    s = synthetic_sample(3, 6, esm_dim=4, seed=5)
    d = tmp_path / "1abc"
    d.mkdir()
    lig = {k: v for k, v in s.items() if k.startswith(("atom_", "bond_")) or k == "num_atoms"}
    pro = {k: v for k, v in s.items() if k.startswith("residue_") or k == "num_residues"}
    torch.save(lig, d / "ligand_data.pt")
    torch.save(pro, d / "protein_data.pt")
    ds = PL.PDBDataset(tmp_path, ["1abc"])
    item = ds[0]
    assert item["pdb_id"] == "1abc" and torch.equal(item["atom_feats"], s["atom_feats"])
    batch = PL.collate_fn([item, item])
    assert batch["residue_esm"].shape == (2, 9, 4) and batch["pdb_id"] == ["1abc", "1abc"]
    assert len(PL.RepeatDataset(item, 5)) == 5


def test_checkpoint_roundtrip_with_ema_and_num_steps_override(tmp_path):
    """generate.py:103: load_from_checkpoint(ckpt, num_steps=...) + ema_state_dict hook (model.py:197-201)."""
    from protein_redesign_amd.constants import make_args
    from protein_redesign_amd.diffusion_model import ProteinReDiffModel
    args = make_args(single_dim=32, pair_dim=32, num_blocks=1, esm_dim=8, num_steps=64)
    m = ProteinReDiffModel(args)
    m.ema.update(m.parameters())                       # make the shadow weights active
    with torch.no_grad():
        for s in m.ema.shadow:
            s.add_(1.0)
    ckpt = {"state_dict": m.state_dict(), "hyper_parameters": dict(args)}
    m.on_save_checkpoint(ckpt)
    torch.save(ckpt, tmp_path / "last.ckpt")
    m2 = ProteinReDiffModel.load_from_checkpoint(tmp_path / "last.ckpt", num_steps=1000)
    assert m2.num_steps == 1000 and m2.single_dim == 32
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    p0 = next(p for p in m2.parameters() if p.requires_grad)
    before = p0.detach().clone()
    with m2.ema.average_parameters(m2.parameters()):   # predict_step swaps the EMA weights in (model.py:249-252)
        assert not torch.equal(p0.detach(), before)
    assert torch.equal(p0.detach(), before)


def test_inference_dataset_and_unknown_residues():
    """data.py:157-168 (one featurised complex per index) and the PDB name of an undetermined residue."""
    from protein_redesign_amd import pipeline as PL
    from protein_redesign_amd.synthetic import synthetic_sample
    items = [synthetic_sample(2, 4, esm_dim=8, seed=1), synthetic_sample(3, 5, esm_dim=8, seed=2)]
    ds = PL.InferenceDataset(items, repeat=2)
    assert len(ds) == 2 and ds[1] is items[1]
    batch = PL.collate_fn([ds[0], ds[1]])
    assert batch["atom_mask"].shape == (2, 8)
    prot = PL.protein_from_sequence("AXW")
    text = PL.protein_to_pdb_string(prot)
    assert [ln[17:20] for ln in text.splitlines()] == ["ALA", "UNK", "TRP"]


def test_bucket_batch_sampler_covers_every_complex_and_shards_by_size():
    """size-bucketed batches for DDP -- every complex exactly once per epoch (up to the wrap-around that
    completes a step, as DistributedSampler pads), ranks of one step draw from ONE size bucket, deterministic in (seed, epoch),
    and far less padding than free shuffling on a PDBbind-like size distribution (N in [100, 384])."""
    from protein_redesign_amd.pipeline import BucketBatchSampler
    g = torch.Generator().manual_seed(0)
    sizes = torch.randint(100, 385, (203,), generator=g).tolist()
    world, bs = 4, 2
    samplers = [BucketBatchSampler(sizes, bs, world, r, bucket_width=32, seed=5) for r in range(world)]
    per_rank = [list(s_) for s_ in samplers]
    assert len({len(p_) for p_ in per_rank}) == 1 and len(per_rank[0]) == len(samplers[0])
    seen = sorted(i for p_ in per_rank for bt in p_ for i in bt)
    assert set(seen) == set(range(len(sizes)))                           # everything is drawn ...
    assert len(seen) - len(sizes) <= 9 * world * bs                      # ... with at most one wrap-around step per bucket
    for step in zip(*per_rank):                                          # the ranks of a step share a bucket
        assert len({sizes[i] // 32 for bt in step for i in bt}) == 1
    again = [list(BucketBatchSampler(sizes, bs, world, r, bucket_width=32, seed=5)) for r in range(world)]
    assert again == per_rank
    for s_ in samplers:
        s_.set_epoch(1)
    assert [list(s_) for s_ in samplers] != per_rank                     # reshuffled every epoch
    bucketed, free = samplers[0].padding_waste()
    assert bucketed < 0.15 and free > 2 * bucketed, (bucketed, free)
    with pytest.raises(ValueError):
        BucketBatchSampler(sizes, bs, world, world)


def test_bucket_batch_sampler_pads_a_step_with_distinct_batches():
    """The incomplete last step of a bucket is completed by wrapping around INSIDE the bucket: the ranks of a padded step get
    distinct batches as long as the bucket has that many (9 complexes on 8 ranks used to give 7 ranks the same complex)."""
    from protein_redesign_amd.pipeline import BucketBatchSampler
    world = 8
    for n_members, bs in ((9, 1), (5, 1), (11, 2), (3, 1)):
        sizes = [200] * n_members
        steps = list(zip(*[list(BucketBatchSampler(sizes, bs, world, r, bucket_width=32, seed=1)) for r in range(world)]))
        n_batches = -(-n_members // bs)
        for step in steps:
            distinct = len({tuple(bt) for bt in step})
            assert distinct == min(world, n_batches), (n_members, bs, step)
        drawn = [i for step in steps for bt in step for i in bt]
        assert set(drawn) == set(range(n_members))


def test_pdb_datamodule_round_trip(tmp_path):
    """data.py:206-259: id lists + preprocessed cache -> collated batches; the training loader is size-bucketed."""
    from protein_redesign_amd.pipeline import PDBDataModule, PDBDataset
    from protein_redesign_amd.synthetic import synthetic_sample
    ids = [f"c{k:02d}" for k in range(10)]
    cache = tmp_path / "PDB_processed_cache"
    for k, pid in enumerate(ids):
        d = synthetic_sample(3 + k % 4, 6 + 5 * (k % 5), esm_dim=8, seed=k)
        (cache / pid).mkdir(parents=True)
        torch.save({kk: v for kk, v in d.items() if kk.startswith(("atom", "bond", "num_atoms"))}, cache / pid / "ligand_data.pt")
        torch.save({kk: v for kk, v in d.items() if kk.startswith(("residue", "num_residues"))}, cache / pid / "protein_data.pt")
    for name, sel in (("PRD_train_pdb_ids", ids[:6]), ("PRD_val_pdb_ids", ids[6:8]), ("PRD_test_pdb_ids", ids[8:])):
        (tmp_path / name).write_text("\n".join(sel) + "\n")
    dm = PDBDataModule(tmp_path, batch_size=2, num_workers=0, bucket_width=8)
    dm.setup()
    batches = list(dm.train_dataloader())
    assert sorted(pid for bt in batches for pid in bt["pdb_id"]) == sorted(ids[:6]) or len(batches) >= 3
    for bt in batches:
        n = bt["atom_mask"].shape[1]
        assert bt["residue_mask"].shape[1] == n and bt["bond_mask"].shape[1:] == (n, n)
    assert len(list(dm.val_dataloader())) == 1 and len(list(dm.test_dataloader())) == 1
    # the sizes found by the first scan are persisted next to the cache and re-used
    import json
    raw = json.loads((cache / "sizes_index.json").read_text())
    assert raw["version"] == 2
    index = {k: v[0] for k, v in raw["entries"].items()}
    assert sorted(index) == sorted(ids[:6]) and all(v > 0 for v in index.values())
    dm2 = PDBDataModule(tmp_path, batch_size=2, num_workers=0, bucket_width=8)
    dm2.setup()
    no_scan = type("D", (), {"pdb_ids": ids[:6], "__getitem__": lambda self, i: 1 / 0})()
    assert dm2._train_sizes(no_scan) == [index[p] for p in ids[:6]]
    # a complex that was re-preprocessed (other cropping) is re-read, not served from the index: the entry's fingerprint is the
    # (mtime, size) of its cache files
    d = synthetic_sample(3, 40, esm_dim=8, seed=77)
    torch.save({kk: v for kk, v in d.items() if kk.startswith(("residue", "num_residues"))}, cache / ids[2] / "protein_data.pt")
    with pytest.raises(ZeroDivisionError):
        dm2._train_sizes(no_scan)                                   # the stale entry forces a read of that complex
    sizes = dm2._train_sizes(PDBDataset(cache, ids[:6]))
    assert sizes[2] == int(torch.load(cache / ids[2] / "ligand_data.pt", weights_only=False)["num_atoms"]) + 40
    assert [s_ for k, s_ in enumerate(sizes) if k != 2] == [index[p] for k, p in enumerate(ids[:6]) if k != 2]
    assert dm2._train_sizes(no_scan) == sizes                       # ... and the refreshed index is persisted
    # free shuffling under data parallelism: disjoint shards per rank (what Lightning's DistributedSampler gives the reference)
    seen = []
    for r in range(2):
        dmr = PDBDataModule(tmp_path, batch_size=1, num_workers=0, bucket_width=0, world_size=2, rank=r, seed=3)
        dmr.setup()
        loader = dmr.train_dataloader()
        dmr.set_epoch(1)
        seen.append(sorted(pid for bt in loader for pid in bt["pdb_id"]))
    assert len(seen[0]) == len(seen[1]) == 3 and sorted(seen[0] + seen[1]) == sorted(ids[:6])


def _sizes_worker(rank, world, port, data_dir, out_dir):
    import torch.distributed as dist
    from protein_redesign_amd.pipeline import PDBDataModule, PDBDataset
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dm = PDBDataModule(data_dir, batch_size=1, num_workers=0, bucket_width=8, world_size=world, rank=rank)
    dm.setup()
    out = {}
    # (1) a scan that raises on rank 0 (a corrupt cache item) raises on EVERY rank -- nobody is left waiting in a collective
    broken = type("D", (), {"pdb_ids": dm.train_pdb_ids, "__getitem__": lambda self, i: 1 / 0})()
    try:
        dm._train_sizes(broken)
        out["raised"] = None
    except RuntimeError as exc:
        out["raised"] = str(exc)
    # (2) the healthy path: rank 0 scans and writes, the others read the same sizes
    out["sizes"] = dm._train_sizes(PDBDataset(dm.cache_dir, dm.train_pdb_ids))
    torch.save(out, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_train_sizes_scan_failure_reaches_every_rank(tmp_path):
    """rank 0 scanned behind a bare barrier; when its scan raised, the other ranks waited for ever.  The verdict now
    travels by broadcast_object_list and is re-raised everywhere (gloo, world size 2)."""
    import torch.multiprocessing as mp
    from protein_redesign_amd.synthetic import synthetic_sample
    data = tmp_path / "data"
    cache = data / "PDB_processed_cache"
    ids = [f"c{k:02d}" for k in range(4)]
    for k, pid in enumerate(ids):
        d = synthetic_sample(3 + k, 6 + 2 * k, esm_dim=8, seed=k)
        (cache / pid).mkdir(parents=True)
        torch.save({kk: v for kk, v in d.items() if kk.startswith(("atom", "bond", "num_atoms"))}, cache / pid / "ligand_data.pt")
        torch.save({kk: v for kk, v in d.items() if kk.startswith(("residue", "num_residues"))}, cache / pid / "protein_data.pt")
    for name in ("PRD_train_pdb_ids", "PRD_val_pdb_ids", "PRD_test_pdb_ids"):
        (data / name).write_text("\n".join(ids) + "\n")
    out = tmp_path / "out"
    out.mkdir()
    port = 29700 + (os.getpid() % 200)
    mp.spawn(_sizes_worker, args=(2, port, str(data), str(out)), nprocs=2, join=True)
    r0, r1 = (torch.load(out / f"s{r}.pt") for r in range(2))
    for r in (r0, r1):
        assert r["raised"] is not None and "rank 0 failed" in r["raised"] and "ZeroDivisionError" in r["raised"]
    assert r0["sizes"] == r1["sizes"] == [3 + k + 6 + 2 * k for k in range(4)]
