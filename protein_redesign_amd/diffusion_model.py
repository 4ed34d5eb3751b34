"""``ProteinReDiffModel`` with the reference's LightningModule surface and a HIP hot path.

Mirrors reference ProteinReDiff/model.py:55-549 for everything ``generate.py`` / ``train.py`` touch:
constructor from Namespace|Mapping, ``add_argparse_args``, ``run_setup_schedule``, ``prepare_batch``
(eval branch), ``forward`` / ``sample_step`` (same signature, returns (noise_pred, seq_pred)),
``sample``, ``predict_step``, checkpoint hooks and ``state_dict`` key names.  The arithmetic of
``sample_step`` and of the reverse-diffusion loop runs in libprd_hip.so; nothing falls back to
PyTorch ops on the hot path.

Differences that are deliberate and documented:
* randomness is injected: every sample k draws its mask permutation and noise from a CPU fp32
  generator keyed (seed, global sample index) (``synthetic.NoiseSource``), so results do not depend
  on device RNG, batch size or on how samples are sharded over GPUs;
* the redesign mask is drawn per sample (identical to the reference at its default batch_size=1).
"""
from __future__ import annotations

import contextlib
import math
from argparse import ArgumentParser, Namespace
from typing import Dict, List, Mapping, Optional, Sequence, Union

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, ops
from .constants import ATOM_FEATURE_CARDS, BOND_FEATURE_CARDS, NUM_RESIDUE_CLASSES
from .schedule import reverse_coefficients, schedule_tables
from .synthetic import NoiseSource
from .trunk import Denoiser, Linear

try:  # the reference subclasses pl.LightningModule (model.py:55); absent in this image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover - depends on the environment
    pl = None

    class _Base(nn.Module):
        """Minimal stand-in for pl.LightningModule: what model.py actually calls on ``self``."""

        def save_hyperparameters(self, args=None):
            self.hparams = args

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

        @classmethod
        def load_from_checkpoint(cls, path, map_location="cpu", **overrides):
            ckpt = torch.load(path, map_location=map_location, weights_only=False)
            args = ckpt.get("hyper_parameters", {})
            args = dict(vars(args)) if isinstance(args, Namespace) else dict(args)
            if "args" in args and len(args) == 1:
                inner = args["args"]
                args = dict(vars(inner)) if isinstance(inner, Namespace) else dict(inner)
            args.update(overrides)
            model = cls(args)
            model.load_state_dict(ckpt["state_dict"])
            model.on_load_checkpoint(ckpt)
            return model


class _TableSum(nn.Module):
    """AtomEmbedding / BondEmbedding parameter container (modules.py:35-70): ``embeddings.{f}.weight``.
    The lookups are fused into prd_atom_embed / prd_static_pair."""

    def __init__(self, cards: Sequence[int], embed_dim: int):
        super().__init__()
        self.embeddings = nn.ModuleList([nn.Embedding(n, embed_dim) for n in cards])
        self.num_features = len(cards)
        self.scale = 1.0 / math.sqrt(self.num_features)


class AtomEmbedding(_TableSum):
    def __init__(self, embed_dim: int):
        super().__init__(ATOM_FEATURE_CARDS, embed_dim)


class BondEmbedding(_TableSum):
    def __init__(self, embed_dim: int):
        super().__init__(BOND_FEATURE_CARDS, embed_dim)


class RadialBasisProjection(nn.Module):
    """modules.py:73-82 (centres only; exp(-scale (d-c)^2) is generated inside prd_pair_init)."""

    def __init__(self, embed_dim: int, min_val: float = 0.0, max_val: float = 2.0):
        super().__init__()
        self.scale = (embed_dim - 1) / (max_val - min_val)
        self.center = nn.Parameter(torch.linspace(min_val, max_val, embed_dim), requires_grad=False)


class SinusoidalProjection(nn.Module):
    """modules.py:85-97 (frequencies only; sin/cos evaluated inside prd_time_embed)."""

    def __init__(self, embed_dim: int):
        super().__init__()
        if embed_dim % 2 != 0:
            raise ValueError(f"embed_dim must be even: {embed_dim}.")
        self.embed_dim = embed_dim
        self.weight = nn.Parameter(torch.logspace(-4.0, 0.0, embed_dim // 2), requires_grad=False)


class _Ema:
    """Shadow-parameter EMA with torch-ema 0.3's call surface and checkpoint format (model.py:124,194-201,217,238,250;
    environment.yml pins torch_ema==0.3): ``state_dict()`` carries decay / num_updates / shadow_params / collected_params,
    ``update()`` uses the warm-up decay min(decay, (1 + n) / (10 + n)).  The shadow list covers EVERY parameter in
    registration order (242 tensors for the reference configuration, as torch-ema 0.3 keeps them); checkpoints whose list
    only holds the 240 trainable tensors (the 0.2 behaviour) load as well."""

    def __init__(self, params, decay: float, use_num_updates: bool = True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow = [p.detach().clone() for p in params]
        self.collected = None
        self.active = False          # becomes True once updated or loaded: until then shadow == init copy

    def to(self, device=None, dtype=None):
        self.shadow = [s.to(device=device) if device is not None else s for s in self.shadow]
        return self

    def _match(self, params):
        """Pairs (parameter, shadow) for a shadow list holding all parameters or only the trainable ones."""
        params = list(params)
        if len(self.shadow) == len(params):
            return list(zip(params, self.shadow))
        train = [p for p in params if p.requires_grad]
        if len(self.shadow) != len(train):
            raise ValueError(f"EMA holds {len(self.shadow)} shadow tensors, the model has {len(params)} parameters "
                             f"({len(train)} trainable)")
        return list(zip(train, self.shadow))

    def update(self, params):
        self.active = True
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            pairs = [(p.detach(), s) for p, s in self._match(params) if p.requires_grad]
            if pairs and all(p.device == s.device and p.is_cuda for p, s in pairs):
                # the same arithmetic as the loop below (s -= (s - p) * (1 - decay)) in three multi-tensor launches
                # instead of three launches per tensor (720 per optimisation step)
                shadow = [s for _, s in pairs]
                tmp = torch._foreach_sub(shadow, [p for p, _ in pairs])
                torch._foreach_mul_(tmp, 1.0 - decay)
                torch._foreach_sub_(shadow, tmp)
            else:
                for p, s in pairs:
                    s.sub_((s - p.to(s.device)) * (1.0 - decay))

    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": self.shadow,
                "collected_params": self.collected}

    def load_state_dict(self, sd):
        shadow = sd.get("shadow_params")
        if shadow is not None:
            self.shadow = [s.detach().clone() for s in shadow]
            self.active = True
        self.decay = sd.get("decay", self.decay)
        self.num_updates = sd.get("num_updates", self.num_updates)
        self.collected = sd.get("collected_params")

    @contextlib.contextmanager
    def average_parameters(self, params=None):
        if not self.active or params is None:
            yield
            return
        pairs = self._match(params)
        saved = [p.detach().clone() for p, _ in pairs]
        with torch.no_grad():
            for p, s in pairs:
                p.copy_(s.to(p.device))
        try:
            yield
        finally:
            with torch.no_grad():
                for (p, _), s in zip(pairs, saved):
                    p.copy_(s)


class ProteinReDiffModel(_Base):
    def __init__(self, args: Union[Namespace, Mapping]):
        super().__init__()
        if isinstance(args, Mapping):
            args = Namespace(**args)
        self.pair_dim, self.single_dim = args.pair_dim, args.single_dim
        self.dist_dim, self.time_dim = args.dist_dim, args.time_dim
        self.max_bond_distance, self.max_relpos, self.esm_dim = args.max_bond_distance, args.max_relpos, args.esm_dim
        self.setup_schedule = False
        self.setup_esm = False
        self.mask_prob, self.num_steps = args.mask_prob, args.num_steps
        self.diffusion_schedule = args.diffusion_schedule
        self.learning_rate, self.warmup_steps, self.ema_decay = args.learning_rate, args.warmup_steps, args.ema_decay
        self.n_recycles, self.training_mode = args.n_recycles, args.training_mode
        # key of the injected randomness (see module docstring).  None = taken from torch's seeded global RNG
        # (torch.initial_seed(), i.e. what pl.seed_everything(args.seed) / generate.py --seed set) when a sample is drawn,
        # so different seeds give different samples like in the reference; set an int to pin it regardless of the global seed.
        self.sample_seed = None
        self.use_hip_graph = True               # replay one captured step graph inside sample()
        # What to do when a sampling loop / optimisation step ends with inf / NaN (the reference is fp32 end to end and cannot
        # overflow at 65504; the default split-16 arithmetic can -- include/prd_hip.h, OPERAND RANGE):
        #   "fp32"  (default) run the CALL again under PRD_ARITH_FP32 (a fresh call in this process, with a RuntimeWarning), and keep
        #           fp32 for this model's later calls; if that is non-finite too -- or the call already ran in fp32 -- raise;
        #   "raise" raise _lib.NonFiniteError naming the arithmetic;   "off" return whatever came out (no check, no host sync).
        self.nonfinite_policy = "fp32"
        self.arithmetic = None                  # None: the process default (_lib); "fp32" / "split16": this model's calls pin it
        self.arith_fallbacks = 0                # how many calls were repeated in fp32 by the policy above
        self.nonfinite_group = None             # process group over which a non-finite verdict is agreed (None: the default group)
        self._side = None                       # ops.SideStream of the device the model runs on (created on first use)
        self._sample_counter = 0

        self.Denoiser = Denoiser(args)
        S, P = self.single_dim, self.pair_dim
        self.embed_atom_feats = AtomEmbedding(S)
        self.embed_beta = nn.Sequential(SinusoidalProjection(self.time_dim), Linear(self.time_dim, P, bias=False, init="normal"))
        self.embed_residue_type = nn.Sequential(
            nn.LayerNorm(NUM_RESIDUE_CLASSES, elementwise_affine=False),
            Linear(NUM_RESIDUE_CLASSES, S, bias=False, init="normal"), nn.ReLU())
        self.embed_bond_feats = BondEmbedding(P)
        self.embed_bond_distance = nn.Embedding(self.max_bond_distance + 1, P)
        self.embed_residue_esm = nn.Sequential(nn.LayerNorm(self.esm_dim, elementwise_affine=False),
                                               Linear(self.esm_dim, S, bias=False, init="normal"))
        self.embed_relpos = nn.Embedding(self.max_relpos * 2 + 1, P)
        self.embed_dist = nn.Sequential(RadialBasisProjection(self.dist_dim), Linear(self.dist_dim, P, bias=False, init="normal"))
        self.weight_radial = nn.Sequential(nn.LayerNorm(P, elementwise_affine=False), Linear(P, P, init="relu"),
                                           nn.ReLU(), Linear(P, 1, bias=False, init="final"))
        self.seq_mlp = nn.Sequential(nn.LayerNorm(S, elementwise_affine=False), Linear(S, S, init="relu"), nn.ReLU(),
                                     Linear(S, NUM_RESIDUE_CLASSES, bias=False, init="final"))
        self.ema = _Ema(self.parameters(), decay=self.ema_decay)
        self.save_hyperparameters(args)

    # ------------------------------------------------------------------ argparse (model.py:130-170)
    @staticmethod
    def add_argparse_args(parent_parser: ArgumentParser) -> ArgumentParser:
        g = parent_parser.add_argument_group("DiffusionModel")
        g.add_argument("--training_mode", action="store_true")
        for name, typ, default in (
                ("mask_prob", float, 1.0), ("esm_dim", int, 1280), ("time_dim", int, 256), ("dist_dim", int, 256),
                ("single_dim", int, 512), ("pair_dim", int, 64), ("head_dim", int, 16), ("num_heads", int, 4),
                ("transition_factor", int, 4), ("num_blocks", int, 12), ("max_bond_distance", int, 7),
                ("max_relpos", int, 32), ("num_steps", int, 64), ("diffusion_schedule", str, "linear"),
                ("learning_rate", float, 4e-4), ("warmup_steps", int, 1000), ("ema_decay", float, 0.999)):
            g.add_argument(f"--{name}", type=typ, default=default)
        g2 = parent_parser.add_argument_group("IterativeDenoiser")   # parsed, unused
        for name, typ, default in (
                ("n_recycles", int, 4), ("top_k_neighbors", int, 30), ("dropout", float, 0.3),
                ("num_gvp_encoder_layers", int, 3), ("num_positional_embeddings", int, 16),
                ("gvp_edge_hidden_dim_scalar", int, 32), ("gvp_edge_hidden_dim_vector", int, 32)):
            g2.add_argument(f"--{name}", type=typ, default=default)
        return parent_parser

    # ------------------------------------------------------------------ schedule / hooks
    def run_setup_schedule(self):
        dev = self.device
        for k, v in schedule_tables(self.num_steps, self.diffusion_schedule).items():
            setattr(self, k, v.to(dev))
        self._coef = reverse_coefficients({k: getattr(self, k).cpu() for k in
                                           ("alphas", "sqrt_one_minus_alphas_cumprod", "sqrt_alphas", "sqrt_betas")}).to(dev)

    def to(self, *args, **kwargs):
        out = torch._C._nn._parse_to(*args, **kwargs)
        self.ema.to(device=out[0], dtype=out[1])
        self.setup_schedule = False
        return super().to(*args, **kwargs)

    def on_save_checkpoint(self, checkpoint):
        checkpoint["ema_state_dict"] = self.ema.state_dict()

    def on_load_checkpoint(self, checkpoint):
        if "ema_state_dict" in checkpoint:
            self.ema.load_state_dict(checkpoint["ema_state_dict"])

    def configure_optimizers(self):
        # the reference's Adam (model.py:203-217); on the GPU its fused single-kernel form (same update rule, one launch instead of a
        # dozen multi-tensor passes over the 16 M parameters: 0.6 -> 0.2 ms per step)
        params = list(self.parameters())
        fused = bool(params) and all(p.is_cuda and p.is_floating_point() for p in params)
        optimizer = torch.optim.Adam(params, lr=self.learning_rate, **({"fused": True} if fused else {}))
        sched = torch.optim.lr_scheduler.LinearLR(optimizer, start_factor=1.0 / self.warmup_steps,
                                                  total_iters=self.warmup_steps - 1)
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": sched, "interval": "step"}}

    def optimizer_step(self, *args, **kwargs):
        if pl is not None:
            super().optimizer_step(*args, **kwargs)
        self.ema.update(self.parameters())

    # ------------------------------------------------------------------ loss path (model.py:471-549), forward only
    def q(self, x, seq, t, noise_z, noise_seq, batch):
        """Forward noising of structure / sequence at step t and of the sequence at t-1 (model.py:471-488)."""
        ac, om = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod
        extra, inv = batch["residue_extra_mask"], batch["residue_inv_extra_mask"]
        z_t = ac[t][:, None, None] * x + om[t][:, None, None] * noise_z
        seq_t = ac[t][:, None, None] * seq + om[t][:, None, None] * noise_seq
        seq_t = extra.unsqueeze(-1) * seq + inv.unsqueeze(-1) * seq_t
        t1 = (t - 1).clamp(min=0)
        seq_t1 = ac[t1][:, None, None] * seq + om[t1][:, None, None] * noise_seq
        return z_t, seq_t, seq_t1, t1

    def diffusion_loss(self, batch, x, mask, t, noise_z=None, noise_seq=None):
        """model.py:490-526.  The network forward runs on the HIP path; the O(N) loss reductions are torch ops.
        ``noise_z`` / ``noise_seq`` (mean-free) may be injected; otherwise they are drawn on the device."""
        seq, rm = batch["residue_one_hot"], batch["residue_mask"]
        if noise_z is None:
            noise_z = ops.remove_mean(torch.randn_like(x), mask.contiguous())
        if noise_seq is None:
            noise_seq = ops.remove_mean(torch.randn_like(seq), rm.contiguous())
        z_t, seq_t, seq_t1, t1 = self.q(x, seq, t, noise_z, noise_seq, batch)
        noise_pred, seq_pred = self(batch, z_t, seq_t, mask, t)
        ac, om = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod
        seq_pred_t1 = ac[t1][:, None, None] * seq_pred + om[t1][:, None, None] * noise_seq
        loss = (mask.unsqueeze(-1) * torch.square(noise_pred - noise_z)).sum(dim=(1, 2))
        loss = loss + F.kl_div(torch.log_softmax(seq_pred_t1, dim=-1) * rm.unsqueeze(-1),
                               torch.softmax(seq_t1, dim=-1) * rm.unsqueeze(-1), reduction="none").sum()
        ce = F.cross_entropy(((seq_pred + 1) / 2).view(-1, NUM_RESIDUE_CLASSES), batch["residue_type"].view(-1),
                             reduction="none", ignore_index=0)
        return loss + (ce * mask.view(-1)).sum()

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """model.py:226-247: loss under the EMA weights, logged as ``val_loss``."""
        if not self.setup_schedule:
            self.run_setup_schedule()
            self.setup_schedule = True
        batch = self.prepare_batch(batch, batch_idx)
        x, mask = batch["x"], batch["residue_and_atom_mask"]
        num_nodes = (mask > 0.5).sum(-1)
        t = torch.randint(0, self.num_steps, size=(x.size(0),), device=x.device)
        with self.ema.average_parameters(self.parameters()):
            diff_loss = self.diffusion_loss(batch, x, mask, t)
        loss = torch.mean(diff_loss / num_nodes)
        self.log("val_loss", loss, on_epoch=True, sync_dist=True, batch_size=x.size(0))
        return loss

    def training_step(self, batch, batch_idx, t=None, noise_z=None, noise_seq=None, sources=None, check_finite=True):
        """model.py:528-549: mean over the batch of diffusion_loss / node count, differentiable with respect to every trainable
        parameter (training.network: HIP forward, per-operator backward, per-block recompute).  ``t`` / the noises / the mask
        sources may be injected (parity tests); otherwise they are drawn like the reference draws them."""
        if not self.setup_schedule:
            self.run_setup_schedule()
            self.setup_schedule = True
        if sources is None:
            # the reference draws a fresh torch.randperm at every optimisation step (model.py:460 -> mask_utils.py:87), so the
            # redesign mask of a complex differs from epoch to epoch: key the draw on a running count of training steps
            # (batch_idx restarts every epoch; keyed determinism on batch_idx is for validation / predict only)
            sources = self._sources(batch["atom_mask"].shape[0], None)
        batch = self.prepare_batch(batch, batch_idx, sources=sources)
        x, mask = batch["x"], batch["residue_and_atom_mask"]
        num_nodes = (mask > 0.5).sum(-1)
        if t is None:
            t = torch.randint(0, self.num_steps, size=(x.size(0),)).to(x.device)
        if noise_z is None:                     # drawn HERE so that a repeated forward (non-finite policy) sees the same noise
            noise_z = ops.remove_mean(torch.randn_like(x), mask.contiguous())
        if noise_seq is None:
            noise_seq = ops.remove_mean(torch.randn_like(batch["residue_one_hot"]), batch["residue_mask"].contiguous())

        def forward_loss():
            return torch.mean(self.diffusion_loss(batch, x, mask, t, noise_z, noise_seq) / num_nodes)

        with _lib.arithmetic(self.arithmetic):
            loss = forward_loss()
        if check_finite and self.nonfinite_policy != "off" and loss.is_cuda and self._any_rank(not bool(torch.isfinite(loss.detach()))):
            loss = self._nonfinite_training_step(forward_loss)
        self.log("train_loss", loss, on_step=True, on_epoch=True, sync_dist=True, batch_size=x.size(0))
        return loss

    def _current_arith(self) -> int:
        return _lib.GEMM_MODES[self.arithmetic] if self.arithmetic is not None else _lib.arith()

    def _any_rank(self, flag: bool) -> bool:
        """``flag`` OR-ed over the data-parallel ranks (``self.nonfinite_group``, default: the default process group) so that every
        rank takes the SAME branch of the non-finite policy: a rank that alone switched arithmetic would train a different model, and
        a rank that alone raised would leave the others waiting in the gradient all-reduce."""
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return flag
        group = getattr(self, "nonfinite_group", None)
        if torch.distributed.get_world_size(group) < 2:
            return flag
        dev = self.device if torch.distributed.get_backend(group) == "nccl" else "cpu"
        f = torch.tensor([1.0 if flag else 0.0], device=dev)
        torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.MAX, group=group)
        return bool(f.item() != 0)

    def _nonfinite_training_step(self, forward_loss):
        """The loss of a training step came out inf / NaN (on this rank or on another one: ``_any_rank``).  Under split-16 arithmetic
        with policy "fp32": THIS MODEL is pinned to PRD_ARITH_FP32 from here on (``self.arithmetic``, as sample()'s fallback does; the
        process default is left alone) and the forward is repeated under it.  The backward of the returned loss runs after this
        function returns; every autograd node of training.py records the arithmetic of its forward and re-enters it in its backward
        (training._pin_arithmetic), so the fp32 forward gets fp32 recomputes and fp32 backward kernels.  Anything else raises."""
        import warnings
        cur = self._current_arith()
        if cur == 1 and self.nonfinite_policy == "fp32":
            warnings.warn("training_step: non-finite loss under split-16 arithmetic (an operand beyond the fp16 range: include/prd_hip.h, "
                          "OPERAND RANGE); pinning this model to PRD_ARITH_FP32 and repeating the step", RuntimeWarning, stacklevel=3)
            self.arithmetic = "fp32"
            self.arith_fallbacks += 1
            with _lib.arithmetic("fp32"):
                loss = forward_loss()
            if not self._any_rank(not bool(torch.isfinite(loss.detach()))):
                return loss
            cur = 0
        raise _lib.NonFiniteError(f"training_step: non-finite loss under {_lib.ARITH_NAMES[cur]} arithmetic"
                                  + (" (the fp32 repeat of a non-finite split-16 step)" if self.arith_fallbacks and cur == 0 else "")
                                  + (": outside the operand range of the split arithmetic (include/prd_hip.h); set model.arithmetic = 'fp32' "
                                     "or nonfinite_policy = 'fp32'" if cur == 1 else ": the weights / inputs themselves produce inf or NaN"))

    def predict_step(self, batch, batch_idx):
        with self.ema.average_parameters(self.parameters()):
            return self.sample(batch, batch_idx=batch_idx)

    # ------------------------------------------------------------------ batch preparation (model.py:424-468)
    def _sources(self, b: int, batch_idx: Optional[int] = None) -> List[NoiseSource]:
        """Default noise sources of a batch of b samples: keyed (seed, global sample index).  The seed follows the seeded
        global RNG unless ``sample_seed`` pins it; the index interleaves the ranks of an initialised process group
        (dataloader batch ``batch_idx`` of rank r holds samples (batch_idx * world + r) * b ...), so that ranks of a
        multi-GPU ``Trainer.predict`` never draw the same sample twice (the reference's identically seeded ranks do)."""
        seed = self.sample_seed if self.sample_seed is not None else (torch.initial_seed() & 0x7FFFFFFF)
        world, rank = 1, 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
        if batch_idx is None:           # direct sample() calls: running count of the samples this object has drawn
            first = self._sample_counter * world + rank * b
            self._sample_counter += b
        else:
            first = (int(batch_idx) * world + rank) * b
        return [NoiseSource(seed, first + k) for k in range(b)]

    def prepare_batch(self, batch, id=None, sources: Optional[Sequence] = None):
        """Eval branch (:459-468): ``int(n_res * mask_prob)`` residues per sample leave the known set.  The training-mode
        branches (:441-458) need ``residue_esm_tokens`` that no published code produces and are out of scope:
        ``training_step`` therefore runs with ``training_mode=False`` masking, which is also what the reference does when it is
        trained with its default flags (``--training_mode`` is a store_true option, model.py:133)."""
        if self.training_mode:
            raise NotImplementedError("training-mode masking needs residue_esm_tokens, which no published code produces "
                                      "(SURVEY.md §2/§5); run with training_mode=False")
        am, rm = batch["atom_mask"], batch["residue_mask"]
        dev = am.device
        b = am.shape[0]
        if sources is None:
            sources = self._sources(b, id if isinstance(id, int) else None)
        one_hot = F.one_hot(batch["residue_type"], num_classes=NUM_RESIDUE_CLASSES) * 2.0 - 1.0
        pos = am.unsqueeze(-1) * batch["atom_pos"] + rm.unsqueeze(-1) * batch["residue_atom_pos"][:, :, 1]
        rm_cpu = rm.detach().cpu()
        extra = rm_cpu.clone()
        inv = torch.zeros_like(rm_cpu)
        for k in range(b):        # RandomMaskingModule(stochastic=False), per sample (mask_utils.py:77-102)
            ones = torch.where(rm_cpu[k] == 1)[0]
            n = int(ones.numel() * self.mask_prob)
            sel = ones[sources[k].randperm(ones.numel())[:n]]
            extra[k, sel] = 0
            inv[k, sel] = 1
        extra, inv = extra.to(dev), inv.to(dev)
        batch["residue_one_hot"] = one_hot * extra.unsqueeze(-1)
        batch["residue_esm"] = batch["residue_esm"] * extra.unsqueeze(-1)
        batch["residue_type_masked"] = (batch["residue_type"] * extra).long()
        batch["residue_extra_mask"] = extra
        batch["residue_inv_extra_mask"] = inv
        batch["x"] = 0.1 * pos
        batch["residue_and_atom_mask"] = am + rm
        return batch

    # ------------------------------------------------------------------ the network (model.py:254-375)
    def _static_inputs(self, batch) -> Dict[str, torch.Tensor]:
        """Step-invariant embeddings: computed once per ``sample``."""
        am, rm = batch["atom_mask"].contiguous(), batch["residue_mask"].contiguous()
        S, P = self.single_dim, self.pair_dim
        bond_tabs = [e.weight for e in self.embed_bond_feats.embeddings]
        sp = ops.static_pair({k: batch[k].contiguous() for k in (
            "atom_mask", "residue_mask", "bond_mask", "bond_feats", "bond_distance", "residue_index",
            "residue_chain_index")}, bond_tabs + [self.embed_bond_distance.weight, self.embed_relpos.weight],
            self.max_bond_distance, self.max_relpos, P)
        tabs = torch.cat([e.weight for e in self.embed_atom_feats.embeddings], dim=0).contiguous()
        offsets = getattr(self, "_atom_table_offsets", None)
        if offsets is None or offsets.device != am.device:             # constant: uploaded once (not a host copy per call)
            offs, acc = [], 0
            for n in ATOM_FEATURE_CARDS:
                offs.append(acc)
                acc += n
            offsets = self._atom_table_offsets = torch.tensor(offs, dtype=torch.int32, device=am.device)
        ss = ops.atom_embed(batch["atom_feats"].contiguous(), am, tabs, offsets, S)
        esm = ops.layer_norm(batch["residue_esm"].contiguous())
        ss = ops.linear(esm, self.embed_residue_esm[1].weight, rowmask=rm, resid=ss)
        return {"pair": sp, "single": ss}

    def _step_inputs(self, static, seq_t, rm, t):
        """Per-step inputs of the network that do not depend on the coordinates: single (model.py:343-346) and the time
        embedding (model.py:341, 360).  Inside ``sample()`` the fused step-boundary kernel produces them for the next step."""
        single = ops.single_init(static["single"], seq_t.contiguous(), rm, self.embed_residue_type[1].weight)
        eb = ops.time_embed(t.contiguous(), self.embed_beta[0].weight, self.embed_beta[1].weight, self.num_steps)
        return single, eb

    def _network(self, batch, z, seq_t, mask, t, static=None, step_inputs=None, raw_noise=False, defer_seq_head=False):
        """``step_inputs`` = (single, ebeta) if already computed; ``raw_noise``: return the coordinate head's output before
        remove_mean (the step-boundary kernel removes the mean itself); ``defer_seq_head``: return the sequence head's HIDDEN units
        instead of the logits (the step-boundary kernel applies the last layer itself)."""
        if static is None:
            static = self._static_inputs(batch)
        rm = batch["residue_mask"].contiguous()
        mask = mask.contiguous()
        z = z.contiguous()
        side = self._side
        if side is None or side.stream is not None and side.stream.device != z.device:
            side = self._side = ops.SideStream(z.device)
        if step_inputs is None:
            step_inputs = self._step_inputs(static, seq_t, rm, t)
        single, eb = step_inputs
        # (opt-in, measured slower: ops.SideStream) the single-only chain beside the pair input stage, joined at the OPM
        with side.fork():
            pre = self.Denoiser.project_single(single, mask)
        sm = self.seq_mlp
        # the pair input stage runs inside Denoiser.run_ (fused with the outer-product update and the first bias heads); the
        # sequence head's first layer is linear in LN(single_out), like the last block's outer-linear term: one launch for both
        single, pair, h = self.Denoiser.run_(single, None, mask, pre=pre, join=side.join, tail=(sm[1].weight, sm[1].bias),
                                             pair_init=(static["pair"], z, self.embed_dist[0].center, self.embed_dist[1].weight, eb))
        with side.fork():
            if h is None:
                h = ops.linear(single, sm[1].weight, sm[1].bias, act=1, ln_a=True)    # LayerNorm (no affine) fused into the linear
            seq_pred = h if defer_seq_head else ops.linear(h, sm[3].weight)
        wr = self.weight_radial
        eps_raw = ops.coord_head(pair, z, mask, wr[1].weight, wr[1].bias, wr[3].weight)
        noise_pred = eps_raw if raw_noise else ops.remove_mean(eps_raw, mask)
        side.join()
        return noise_pred, seq_pred

    def forward(self, batch, z, seq_t, mask, t):
        """Inference (no_grad / inference_mode): the fused HIP path.  With autograd enabled: the same HIP operators, one
        autograd node each (training.py), so that ``diffusion_loss(...).backward()`` reaches every parameter."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from . import training
            return training.network(self, batch, z, seq_t, mask, t)
        return self._network(batch, z, seq_t, mask, t)

    def sample_step(self, batch, z, seq_t, mask, t):
        return self._network(batch, z, seq_t, mask, t)

    # ------------------------------------------------------------------ reverse diffusion (model.py:377-422)
    @torch.inference_mode()
    def sample(self, batch, sources: Optional[Sequence] = None, batch_idx: Optional[int] = None):
        if sources is None:
            sources = self._sources(batch["atom_mask"].shape[0], batch_idx)
        # the keyed generators are consumed by a loop: remember where they stood so that a repeat draws the same noise
        states = [s.g.get_state() if hasattr(s, "g") else None for s in sources]
        with _lib.arithmetic(self.arithmetic):
            loop = ReverseDiffusion(self, batch, sources)
            loop.run()
            cur = _lib.arith()
            if self.nonfinite_policy == "off" or loop.finite():
                return loop.result()
        if cur == 1 and self.nonfinite_policy == "fp32":
            import warnings
            warnings.warn(f"sample: inf / NaN after {loop.steps_done} denoising steps under split-16 arithmetic (an operand beyond the "
                          "fp16 range: include/prd_hip.h, OPERAND RANGE); running the call again under PRD_ARITH_FP32 and keeping fp32 "
                          "for this model", RuntimeWarning, stacklevel=2)
            if any(st is None for st in states):
                raise _lib.NonFiniteError("sample: non-finite under split-16 arithmetic and the noise sources cannot be rewound")
            for s, st in zip(sources, states):
                s.g.set_state(st)
            self.arith_fallbacks += 1
            self.arithmetic = "fp32"
            with _lib.arithmetic("fp32"):
                loop = ReverseDiffusion(self, batch, sources)
                loop.run()
                if loop.finite():
                    return loop.result()
            cur = 0
        raise _lib.NonFiniteError(f"sample: inf / NaN coordinates or logits under {_lib.ARITH_NAMES[cur]} arithmetic"
                                  + (" (the fp32 repeat of a non-finite split-16 call)" if self.arith_fallbacks and cur == 0 else "")
                                  + (": outside the operand range of the split arithmetic (include/prd_hip.h); set model.arithmetic = 'fp32' "
                                     "or nonfinite_policy = 'fp32'" if cur == 1 else ": the weights / inputs themselves produce inf or NaN"))


class ReverseDiffusion:
    """State and driver of one ``sample()`` call (reference model.py:377-422) on one GPU.

    Everything the loop needs is resident in HBM before the first step: the step-invariant
    embeddings, the complete noise table ``[T-1, b, N, 3]`` (drawn on the host from the keyed
    generators, in the reference's per-sample order) and the per-step scalar
    table.  ``z``, ``seq_t`` and the step counter ``t`` live on the device and are advanced by the
    kernels themselves, so ONE captured hipGraph of a whole step (network forward + reverse update)
    is replayed for every remaining step: no per-step host work and no ``(t == 0).all()``
    device->host sync (model.py:415)."""

    def __init__(self, model: "ProteinReDiffModel", batch, sources: Optional[Sequence] = None):
        m = self.model = model
        if not m.setup_schedule:
            m.run_setup_schedule()
            m.setup_schedule = True
        dev = m.device
        b, N = batch["atom_mask"].shape
        if sources is None:
            sources = m._sources(b)
        self.batch = batch = m.prepare_batch(batch, sources=sources)
        self.mask = batch["residue_and_atom_mask"].contiguous()
        self.rm = batch["residue_mask"].contiguous()
        self.T = T = m.num_steps
        z0 = torch.stack([s.randn(N, 3) for s in sources])
        s0 = torch.stack([s.randn(N, NUM_RESIDUE_CLASSES) for s in sources])
        if T > 1:
            noise = torch.stack([torch.stack([s.randn(N, 3) for _ in range(T - 1)]) for s in sources], dim=1)
        else:
            noise = torch.zeros(1, b, N, 3)
        self.noise = noise.to(dev).contiguous()                      # [T-1, b, N, 3]
        self.z = ops.remove_mean(z0.to(dev).contiguous(), self.mask)
        seq_t = ops.remove_mean(s0.to(dev).contiguous(), self.rm)
        self.seq_t = (batch["residue_extra_mask"].unsqueeze(-1) * batch["residue_one_hot"]
                      + batch["residue_inv_extra_mask"].unsqueeze(-1) * seq_t).contiguous()
        self.static = m._static_inputs(batch)
        self.t = torch.full((b,), T - 1, dtype=torch.int64, device=dev)
        self._init = (self.z.clone(), self.seq_t.clone())
        self.steps_done = 0
        self.graph = None
        self.seq_pred = None
        # inputs of the coming step that the previous step's boundary kernel prepares (single, time embedding)
        import os
        self.fused_boundary = os.environ.get("PRD_FUSED_BOUNDARY", "1") != "0"
        self.sync = torch.zeros(2, dtype=torch.int32, device=dev)     # [0] arrival counter, [1] sticky non-finite flag (prd_step_boundary)
        self._seq_pred_buf = None               # logits of the last step (written by the step-boundary kernel)
        self.single_in, self.eb_in = m._step_inputs(self.static, self.seq_t, self.rm, self.t)

    def _refresh_step_inputs(self):
        single, eb = self.model._step_inputs(self.static, self.seq_t, self.rm, self.t)
        self.single_in.copy_(single)
        self.eb_in.copy_(eb)

    def reset(self):
        """Back to step T-1 with the same initial noise (bench / repeated runs)."""
        self.z.copy_(self._init[0])
        self.seq_t.copy_(self._init[1])
        self.t.fill_(self.T - 1)
        self.steps_done = 0
        self._refresh_step_inputs()

    def restart(self, step: int, z: torch.Tensor, seq_t: torch.Tensor):
        """Continue from a given state: ``z`` / ``seq_t`` are the tensors ENTERING denoising step number ``step`` (0 = the
        first, t = T-1).  The pre-drawn noise table is indexed by the step counter, so the continuation consumes exactly the
        noise the uninterrupted loop would (checkpointed sampling; segment-wise parity tests)."""
        if not 0 <= step < self.T:
            raise ValueError(f"step must be in [0, {self.T}), got {step}")
        self.z.copy_(z.to(self.z.device))
        self.seq_t.copy_(seq_t.to(self.seq_t.device))
        self.t.fill_(self.T - 1 - step)
        self.steps_done = step
        self._refresh_step_inputs()

    def _enqueue_step(self):
        """Network forward on the inputs the previous boundary prepared, then ONE step-boundary launch: remove_mean, reverse
        update, t <- t - 1 and the next step's single / time-embedding inputs (ops.step_boundary_)."""
        m = self.model
        if not self.fused_boundary:             # the four separate launches (kept for A/B measurements: PRD_FUSED_BOUNDARY=0)
            noise_pred, seq_pred = m._network(self.batch, self.z, self.seq_t, self.mask, self.t, static=self.static)
            ops.reverse_update_(self.z, self.seq_t, self.t, noise_pred, seq_pred, self.noise, self.mask, m._coef, self.T)
            return seq_pred
        eps_raw, seq_h = m._network(self.batch, self.z, self.seq_t, self.mask, self.t, static=self.static,
                                    step_inputs=(self.single_in, self.eb_in), raw_noise=True, defer_seq_head=True)
        if self._seq_pred_buf is None:
            self._seq_pred_buf = torch.empty(*self.seq_t.shape, device=self.seq_t.device, dtype=torch.float32)
        ops.step_boundary_(self.z, self.seq_t, self.t, eps_raw, self._seq_pred_buf, self.noise, self.mask, m._coef, self.T,
                           self.single_in, self.static["single"], self.rm, m.embed_residue_type[1].weight,
                           self.eb_in, m.embed_beta[0].weight, m.embed_beta[1].weight, self.sync,
                           seq_h=seq_h, w_seq=m.seq_mlp[3].weight)
        return self._seq_pred_buf

    @torch.inference_mode()
    def step(self):
        """One denoising step: network forward + reverse update (asynchronous)."""
        if self.steps_done >= self.T:
            raise RuntimeError("all num_steps denoising steps already done; call reset()")
        if self.graph is not None:
            self.graph.replay()
        elif self.model.use_hip_graph and self.steps_done >= 1:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.seq_pred = self._enqueue_step()      # capture only; the replay below executes it
            self.graph.replay()
        else:
            self.seq_pred = self._enqueue_step()          # first step eager: warms every kernel up
        self.steps_done += 1

    def run(self):
        while self.steps_done < self.T:
            self.step()

    def finite(self) -> bool:
        """False when the loop produced inf / NaN: the sticky flag the step-boundary kernel keeps (any step, prd_hip.h) or'ed with a
        check of the current state and logits (covers the un-fused boundary and direct step() use).  ONE host read per call --
        after the loop, never inside it."""
        bad = self.sync[1] != 0
        bad = bad | ~torch.isfinite(self.z).all()
        if self.seq_pred is not None:
            bad = bad | ~torch.isfinite(self.seq_pred).all()
        return not bool(bad)

    def result(self):
        """(positions in Angstrom, residue-masked logits) as in model.py:421-422."""
        return 10.0 * self.z, self.rm.unsqueeze(-1) * self.seq_pred
