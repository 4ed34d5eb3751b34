"""CPU oracle for the ProteinReDiff denoiser hot path.  TEST INFRASTRUCTURE ONLY.

A functional, pure-PyTorch (CPU, fp32) restatement of the reference algorithm:
every function works on a flat ``state_dict`` (reference key names, SURVEY.md
Appendix B) and cites the reference lines it follows.  It exists so that

* tests/ can check the HIP path against it (and it against the golden vectors
  generated from the imported reference by ``oracle/gen_golden.py``),
* ``__graft_entry__.smoke()`` can check one step, and
* ``bench.py`` can time it as the ``cpu_baseline`` ("port").

Nothing under ``protein_redesign_amd/`` imports this file: the product path is
HIP only and fails loudly without its extension.

Parity pin: checked against the imported reference itself (fixtures under
tests/golden/, generator committed) -- the reference ships no tests or golden
vectors of its own.  EMA / checkpoint loading stay
"parity unpinned" (no reference checkpoint is obtainable offline).
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Mapping[str, torch.Tensor]
ROW_CHUNK_ELEMS = 1 << 28      # triangle attention: at most this many logits (1 GiB of fp32) are materialised at a time


# ---------------------------------------------------------------------------
# leaf helpers
# ---------------------------------------------------------------------------

def ln(x: torch.Tensor, w: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.LayerNorm over the last dim, eps=1e-5 (torch default; AF2_modules.py:161-182)."""
    return F.layer_norm(x, x.shape[-1:], w, b, 1e-5)


def lin(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    """nn.Linear with weight [out,in] and optional bias (modules.py:129, AF2_modules.py:94)."""
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


def table_sum(p: Params, prefix: str, feats: torch.Tensor, n_tables: int) -> torch.Tensor:
    """AtomEmbedding / BondEmbedding: sum_f scale * table_f[idx_f], scale = 1/sqrt(F)
    accumulated left to right from 0.0 (modules.py:47-51, 66-70)."""
    scale = 1.0 / math.sqrt(n_tables)
    acc = 0.0
    for f in range(n_tables):
        acc = acc + scale * F.embedding(feats[..., f], p[f"{prefix}.embeddings.{f}.weight"])
    return acc


def radial_basis(p: Params, d: torch.Tensor) -> torch.Tensor:
    """RadialBasisProjection: exp(-(K-1)/2 * (d - c_k)^2), c = linspace(0,2,K) (modules.py:73-82)."""
    center = p["embed_dist.0.center"]
    scale = (center.numel() - 1) / 2.0
    return torch.exp(-scale * torch.square(d.unsqueeze(-1) - center))


def sinusoid(p: Params, tau: torch.Tensor) -> torch.Tensor:
    """SinusoidalProjection: [sin(w*tau), cos(w*tau)], w = logspace(-4,0,K/2) (modules.py:85-97)."""
    wx = p["embed_beta.0.weight"] * tau.unsqueeze(-1)
    return torch.cat([torch.sin(wx), torch.cos(wx)], dim=-1)


def remove_mean(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """utils.py:32-36: subtract the masked mean over the node axis, per sample and channel."""
    m = mask.unsqueeze(-1).expand_as(x)
    s = (m * x).sum(dim=1, keepdim=True)
    n = m.sum(dim=1, keepdim=True)
    return x - m * s / n


# ---------------------------------------------------------------------------
# trunk blocks
# ---------------------------------------------------------------------------

def gated_attention(p: Params, prefix: str, x: torch.Tensor, mask: torch.Tensor,
                    heads: int, head_dim: int, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """modules.py:185-225 (``Attention.forward``): LN (no affine), q/k/v (no bias), sigmoid gate,
    query pre-scaled by 1/sqrt(c), additive bias, key mask filled with -2**15, softmax,
    gated, output projection."""
    x = ln(x)
    lead = x.shape[:-2]
    n = x.shape[-2]

    def split(t):  # "... i (h c) -> ... h i c"
        return t.reshape(*lead, n, heads, head_dim).transpose(-2, -3)

    q = split(lin(p, prefix + ".q_proj", x))
    k = split(lin(p, prefix + ".k_proj", x))
    v = split(lin(p, prefix + ".v_proj", x))
    g = split(torch.sigmoid(lin(p, prefix + ".gate_proj", x)))
    logits = torch.matmul((1.0 / math.sqrt(head_dim)) * q, k.transpose(-1, -2))
    if bias is not None:
        logits = logits + bias
    key_mask = mask.unsqueeze(-2).unsqueeze(-2)          # "... j -> ... 1 1 j"
    logits = logits.masked_fill(key_mask < 0.5, -(2.0 ** 15))
    attn = torch.softmax(logits, dim=-1)
    out = g * torch.matmul(attn, v)
    out = out.transpose(-2, -3).reshape(*lead, n, heads * head_dim)
    return lin(p, prefix + ".out_proj", out)


def triangle_attention(p: Params, prefix: str, pair: torch.Tensor, mask2d: torch.Tensor,
                       heads: int, head_dim: int, ending: bool) -> torch.Tensor:
    """modules.py:236-243: row-wise gated attention; "ending" = same op on the transposed pair."""
    if ending:
        pair = pair.transpose(-2, -3)
        mask2d = mask2d.transpose(-1, -2)
    n = pair.shape[-2]
    rows = max(1, ROW_CHUNK_ELEMS // (heads * n * n))
    if rows >= pair.shape[-3]:
        out = gated_attention(p, prefix + ".attn", pair, mask2d, heads, head_dim)
    else:       # same arithmetic, row blocks at a time: the [b,N,H,N,N] logits of N = 769 would need 7.3 GB (x3 temporaries)
        out = torch.cat([gated_attention(p, prefix + ".attn", pair[..., r:r + rows, :, :], mask2d[..., r:r + rows, :],
                                         heads, head_dim) for r in range(0, pair.shape[-3], rows)], dim=-3)
    if ending:
        out = out.transpose(-2, -3)
    return out


def triangle_multiplication(p: Params, prefix: str, pair: torch.Tensor, mask2d: torch.Tensor,
                            incoming: bool) -> torch.Tensor:
    """modules.py:262-274 with equations :250-252."""
    x = ln(pair)
    ab = mask2d.unsqueeze(-1) * torch.sigmoid(lin(p, prefix + ".ab_gate", x)) * lin(p, prefix + ".ab_proj", x)
    a, b = torch.chunk(ab, 2, dim=-1)
    eq = "...kid,...kjd->...ijd" if incoming else "...ikd,...jkd->...ijd"
    o = torch.einsum(eq, a, b)
    return torch.sigmoid(lin(p, prefix + ".out_gate", x)) * lin(p, prefix + ".out_proj", ln(o))


def outer_linear(p: Params, prefix: str, single: torch.Tensor) -> torch.Tensor:
    """modules.py:283-287: Linear(cat[x_i * x_j, x_i - x_j]) on LN(single)."""
    x = ln(single)
    xi = x.unsqueeze(-2)
    xj = x.unsqueeze(-3)
    return lin(p, prefix + ".linear", torch.cat([xi * xj, xi - xj], dim=-1))


def transition(p: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """single_fc / pair_fc: LN -> Linear -> ReLU -> Linear (modules.py:306-311, 321-326)."""
    return lin(p, prefix + ".3", torch.relu(lin(p, prefix + ".1", ln(x))))


def pair_bias(p: Params, prefix: str, pair: torch.Tensor) -> torch.Tensor:
    """FoldingBlock.attn_bias: LN -> Linear(P,H) -> "... i j h -> ... h i j" (modules.py:300-304)."""
    return lin(p, prefix + ".1", ln(pair)).permute(0, 3, 1, 2)


def folding_block(p: Params, prefix: str, single: torch.Tensor, pair: torch.Tensor, mask: torch.Tensor,
                  heads: int, head_dim: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """modules.py:328-343: the eight residual updates, strictly in this order."""
    mask2d = mask.unsqueeze(-1) * mask.unsqueeze(-2)
    single = single + gated_attention(p, prefix + ".single_attn", single, mask, heads, head_dim,
                                      bias=pair_bias(p, prefix + ".attn_bias", pair))
    single = single + transition(p, prefix + ".single_fc", single)
    pair = pair + outer_linear(p, prefix + ".outer_linear", single)
    pair = pair + triangle_multiplication(p, prefix + ".pair_mul_outgoing", pair, mask2d, incoming=False)
    pair = pair + triangle_multiplication(p, prefix + ".pair_mul_incoming", pair, mask2d, incoming=True)
    pair = pair + triangle_attention(p, prefix + ".pair_attn_starting", pair, mask2d, heads, head_dim, ending=False)
    pair = pair + triangle_attention(p, prefix + ".pair_attn_ending", pair, mask2d, heads, head_dim, ending=True)
    pair = pair + transition(p, prefix + ".pair_fc", pair)
    return single, pair


def outer_product_update(p: Params, prefix: str, single: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """AF2_modules.py:503-545.  With a [b,N,C] input the einsum "...abc,...adc->...abdc" keeps the
    channel axis: out[b,i,j,c] = a[b,i,c] * b[b,j,c]; then Linear(C,P) and division by (m_i m_j + 1e-3)."""
    x = ln(single, p[prefix + ".layer_norm.weight"], p[prefix + ".layer_norm.bias"])
    m = mask.unsqueeze(-1)
    a = lin(p, prefix + ".linear_1", x) * m
    b = lin(p, prefix + ".linear_2", x) * m
    outer = a.unsqueeze(-2) * b.unsqueeze(-3)
    outer = lin(p, prefix + ".linear_out", outer)
    norm = m.unsqueeze(-2) * m.unsqueeze(-3) + 1e-3
    return outer / norm


def single_pair_attention(p: Params, prefix: str, single: torch.Tensor, pair: torch.Tensor, heads: int) -> torch.Tensor:
    """AF2_modules.py:421-473 + Attention :251-367 + _attention :613-628.
    The mask is computed and never applied there (:447 vs :461-463); the residual is taken on the
    LayerNorm-ed input (:465-470); per-head width equals single_dim (modules.py:366-368)."""
    z = ln(pair, p[prefix + ".linear_z.0.weight"], p[prefix + ".linear_z.0.bias"])
    z = F.linear(z, p[prefix + ".linear_z.1.weight"]).permute(0, 3, 1, 2)          # [b,H,N,N]
    m = ln(single, p[prefix + ".layer_norm_m.weight"], p[prefix + ".layer_norm_m.bias"])
    b_, n, _ = m.shape

    def heads_of(t):
        return t.view(b_, n, heads, -1).transpose(-2, -3)

    q = heads_of(lin(p, prefix + ".mha.linear_q", m))
    k = heads_of(lin(p, prefix + ".mha.linear_k", m))
    v = heads_of(lin(p, prefix + ".mha.linear_v", m))
    q = q / math.sqrt(q.shape[-1])
    a = torch.matmul(q, k.transpose(-1, -2)) + z
    a = torch.softmax(a, dim=-1)
    o = torch.matmul(a, v).transpose(-2, -3)                                         # [b,N,H,C]
    g = torch.sigmoid(lin(p, prefix + ".mha.linear_g", m)).view(b_, n, heads, -1)
    o = (o * g).reshape(b_, n, -1)
    return m + lin(p, prefix + ".mha.linear_o", o)


def denoiser(p: Params, cfg: Mapping, single: torch.Tensor, pair: torch.Tensor, mask: torch.Tensor):
    """modules.py:391-404: OPM (masked add), SPA, num_blocks folding blocks, pair symmetrisation."""
    h, c = cfg["num_heads"], cfg["head_dim"]
    mask2d = mask.unsqueeze(-1) * mask.unsqueeze(-2)
    pair = pair + mask2d.unsqueeze(-1) * outer_product_update(p, "Denoiser.opm", single, mask)
    single = single_pair_attention(p, "Denoiser.SPAAttnBlock", single, pair, h)
    for i in range(cfg["num_blocks"]):
        single, pair = folding_block(p, f"Denoiser.folding_blocks.{i}", single, pair, mask, h, c)
    pair = 0.5 * (pair + pair.transpose(1, 2))
    return single, pair


# ---------------------------------------------------------------------------
# network forward (== sample_step == forward) and the diffusion loop
# ---------------------------------------------------------------------------

def embed_inputs(p: Params, cfg: Mapping, batch: Mapping[str, torch.Tensor], z: torch.Tensor,
                 seq_t: torch.Tensor, mask: torch.Tensor, t: torch.Tensor):
    """model.py:332-361: single and pair inputs of the trunk."""
    am, rm = batch["atom_mask"], batch["residue_mask"]
    am2 = am.unsqueeze(-1) * am.unsqueeze(-2)
    rm2 = rm.unsqueeze(-1) * rm.unsqueeze(-2)
    ri, ci = batch["residue_index"], batch["residue_chain_index"]
    relpos = ri.unsqueeze(-1) - ri.unsqueeze(-2)
    chain = (ci.unsqueeze(-1) == ci.unsqueeze(-2)).float()
    mask2d = mask.unsqueeze(-1) * mask.unsqueeze(-2)
    zij = z.unsqueeze(-2) - z.unsqueeze(-3)
    dist = torch.linalg.norm(zij, dim=-1)
    tau = t / cfg["num_steps"]

    single = am.unsqueeze(-1) * table_sum(p, "embed_atom_feats", batch["atom_feats"], 9)
    single = single + rm.unsqueeze(-1) * (
        torch.relu(F.linear(ln(seq_t), p["embed_residue_type.1.weight"]))
        + F.linear(ln(batch["residue_esm"]), p["embed_residue_esm.1.weight"])
    )
    pair = am2.unsqueeze(-1) * (
        batch["bond_mask"].unsqueeze(-1) * table_sum(p, "embed_bond_feats", batch["bond_feats"], 3)
        + F.embedding(batch["bond_distance"].clamp(max=cfg["max_bond_distance"]), p["embed_bond_distance.weight"])
    )
    mr = cfg["max_relpos"]
    pair = pair + rm2.unsqueeze(-1) * (
        chain.unsqueeze(-1) * F.embedding(mr + relpos.clamp(min=-mr, max=mr), p["embed_relpos.weight"])
    )
    pair = pair + mask2d.unsqueeze(-1) * (
        F.linear(radial_basis(p, dist), p["embed_dist.1.weight"])
        + F.linear(sinusoid(p, tau[:, None, None]), p["embed_beta.1.weight"])
    )
    return single, pair, zij, mask2d


def heads(p: Params, single: torch.Tensor, pair: torch.Tensor, zij: torch.Tensor,
          mask2d: torch.Tensor, mask: torch.Tensor):
    """model.py:364-374: SE(3)-equivariant coordinate update and sequence logits."""
    w = F.linear(torch.relu(lin(p, "weight_radial.1", ln(pair))), p["weight_radial.3.weight"])
    r = zij * torch.rsqrt(torch.sum(torch.square(zij), -1, keepdim=True) + 1e-4)
    noise_pred = (mask2d.unsqueeze(-1) * w * r).sum(dim=2)
    noise_pred = remove_mean(noise_pred, mask)
    seq_pred = F.linear(torch.relu(lin(p, "seq_mlp.1", ln(single))), p["seq_mlp.3.weight"])
    return noise_pred, seq_pred


def network_step(p: Params, cfg: Mapping, batch: Mapping[str, torch.Tensor], z: torch.Tensor,
                 seq_t: torch.Tensor, mask: torch.Tensor, t: torch.Tensor):
    """model.py:318-375 (``sample_step``; identical body to ``forward`` :254-316)."""
    single, pair, zij, mask2d = embed_inputs(p, cfg, batch, z, seq_t, mask, t)
    single, pair = denoiser(p, cfg, single, pair, mask)
    return heads(p, single, pair, zij, mask2d, mask)


def get_betas(num_steps: int, schedule: str) -> torch.Tensor:
    """difffusion.py:8-26."""
    if schedule == "linear":
        return torch.linspace(0.0001, 0.02, num_steps)
    if schedule == "cosine":
        steps = num_steps + 1
        x = torch.linspace(0, num_steps, steps)
        ac = torch.cos((x / steps) * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
    raise ValueError(f"Invalid schedule: {schedule}")


def schedule_tables(num_steps: int, schedule: str = "linear") -> Dict[str, torch.Tensor]:
    """model.py:172-190 (only the tables the sampling / loss paths read)."""
    betas = get_betas(num_steps, schedule)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, 0)
    return {
        "betas": betas, "alphas": alphas, "alphas_cumprod": ac,
        "sqrt_betas": torch.sqrt(betas), "sqrt_alphas": torch.sqrt(alphas),
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
    }


def q_sample(sch: Mapping[str, torch.Tensor], batch: Mapping[str, torch.Tensor], x, seq, t, noise_z, noise_seq):
    """model.py:471-488: forward noising of structure and sequence at step t, and the sequence at t-1."""
    ac, om = sch["sqrt_alphas_cumprod"], sch["sqrt_one_minus_alphas_cumprod"]
    extra, inv = batch["residue_extra_mask"], batch["residue_inv_extra_mask"]
    z_t = ac[t][:, None, None] * x + om[t][:, None, None] * noise_z
    seq_t = ac[t][:, None, None] * seq + om[t][:, None, None] * noise_seq
    seq_t = extra.unsqueeze(-1) * seq + inv.unsqueeze(-1) * seq_t
    t1 = (t - 1).clamp(min=0)
    seq_t1 = ac[t1][:, None, None] * seq + om[t1][:, None, None] * noise_seq
    return z_t, seq_t, seq_t1, t1


def diffusion_loss(p: Params, cfg: Mapping, batch: Mapping[str, torch.Tensor], t: torch.Tensor,
                   noise_z: torch.Tensor, noise_seq: torch.Tensor, network=None) -> torch.Tensor:
    """model.py:490-526 with injected noise (already mean-free, :495-496): squared error on the predicted noise,
    KL between the re-noised predicted and true sequence at t-1 (summed over the WHOLE batch, as the reference does),
    and cross-entropy of (logits+1)/2 against residue_type with ignore_index 0, times the node mask."""
    sch = schedule_tables(cfg["num_steps"], cfg.get("diffusion_schedule", "linear"))
    x, mask, rm = batch["x"], batch["residue_and_atom_mask"], batch["residue_mask"]
    seq = batch["residue_one_hot"]
    z_t, seq_t, seq_t1, t1 = q_sample(sch, batch, x, seq, t, noise_z, noise_seq)
    net = network or (lambda z, s, m, tt: network_step(p, cfg, batch, z, s, m, tt))
    noise_pred, seq_pred = net(z_t, seq_t, mask, t)
    ac, om = sch["sqrt_alphas_cumprod"], sch["sqrt_one_minus_alphas_cumprod"]
    seq_pred_t1 = ac[t1][:, None, None] * seq_pred + om[t1][:, None, None] * noise_seq
    loss = (mask.unsqueeze(-1) * torch.square(noise_pred - noise_z)).sum(dim=(1, 2))
    loss = loss + F.kl_div(torch.log_softmax(seq_pred_t1, dim=-1) * rm.unsqueeze(-1),
                           torch.softmax(seq_t1, dim=-1) * rm.unsqueeze(-1), reduction="none").sum()
    ce = F.cross_entropy(((seq_pred + 1) / 2).view(-1, seq_pred.shape[-1]), batch["residue_type"].view(-1),
                         reduction="none", ignore_index=0)
    loss = loss + (ce * mask.view(-1)).sum()
    return loss


def redesign_mask(residue_mask: torch.Tensor, mask_prob: float, perms: Sequence[torch.Tensor]):
    """RandomMaskingModule.forward(stochastic=False) (mask_utils.py:77-102), applied per sample:
    ``int(n_res * mask_prob)`` residues, chosen by ``perms[k]`` (a permutation of range(n_res_k)),
    are removed from the known set.  Equals the reference for batch size 1."""
    extra = residue_mask.clone()
    inv = torch.zeros_like(residue_mask)
    for k in range(residue_mask.shape[0]):
        ones = torch.where(residue_mask[k] == 1)[0]
        n = int(ones.numel() * mask_prob)
        sel = ones[perms[k][:n]]
        extra[k, sel] = 0
        inv[k, sel] = 1
    return extra, inv


def prepare_batch(batch: Dict[str, torch.Tensor], mask_prob: float, perms: Sequence[torch.Tensor]):
    """model.py:424-440 + eval branch :459-468."""
    am, rm = batch["atom_mask"], batch["residue_mask"]
    one_hot = F.one_hot(batch["residue_type"], num_classes=21) * 2.0 - 1.0
    pos = am.unsqueeze(-1) * batch["atom_pos"] + rm.unsqueeze(-1) * batch["residue_atom_pos"][:, :, 1]
    extra, inv = redesign_mask(rm, mask_prob, perms)
    out = dict(batch)
    out["residue_esm"] = batch["residue_esm"] * extra.unsqueeze(-1)
    out["residue_type_masked"] = (batch["residue_type"] * extra).long()
    out["residue_one_hot"] = one_hot * extra.unsqueeze(-1)
    out["residue_extra_mask"] = extra
    out["residue_inv_extra_mask"] = inv
    out["x"] = 0.1 * pos
    out["residue_and_atom_mask"] = am + rm
    return out


@torch.inference_mode()
def sample(p: Params, cfg: Mapping, batch: Dict[str, torch.Tensor], noise_sources: Sequence,
           return_trajectory: bool = False):
    """model.py:377-422 with injected randomness.

    ``noise_sources[k]`` (one per sample) provides ``randperm(n)`` and ``randn(*shape)`` and is
    consumed in the reference's order: mask permutation, z_T, seq_T, then one [N,3] draw per step
    with t > 0."""
    T = cfg["num_steps"]
    sch = schedule_tables(T, cfg.get("diffusion_schedule", "linear"))
    b, N = batch["atom_mask"].shape
    n_res = batch["residue_mask"].sum(-1).long().tolist()
    perms = [noise_sources[k].randperm(n_res[k]) for k in range(b)]
    batch = prepare_batch(batch, cfg["mask_prob"], perms)
    mask = batch["residue_and_atom_mask"]
    rm = batch["residue_mask"]
    seq = batch["residue_one_hot"]
    z = remove_mean(torch.stack([noise_sources[k].randn(N, 3) for k in range(b)]), mask)
    seq_t = remove_mean(torch.stack([noise_sources[k].randn(N, 21) for k in range(b)]), rm)
    seq_t = batch["residue_extra_mask"].unsqueeze(-1) * seq + batch["residue_inv_extra_mask"].unsqueeze(-1) * seq_t
    traj = []
    seq_pred = None
    for i in range(T):
        tt = T - 1 - i
        t = torch.full((b,), tt, dtype=torch.long)
        w_noise = (1.0 - sch["alphas"][t]) / sch["sqrt_one_minus_alphas_cumprod"][t]
        noise_pred, seq_pred = network_step(p, cfg, batch, z, seq_t, mask, t)
        mean = (1.0 / sch["sqrt_alphas"][t])[:, None, None] * (z - w_noise[:, None, None] * noise_pred)
        seq_t = torch.softmax(seq_pred, dim=-1) * 2 - 1
        if tt == 0:
            z = mean
        else:
            noise = remove_mean(torch.stack([noise_sources[k].randn(N, 3) for k in range(b)]), mask)
            z = mean + sch["sqrt_betas"][t][:, None, None] * noise
        if return_trajectory:
            traj.append((noise_pred.clone(), seq_pred.clone(), z.clone()))
    pos = 10.0 * z
    out = (pos, rm.unsqueeze(-1) * seq_pred)
    return out + (traj,) if return_trajectory else out
