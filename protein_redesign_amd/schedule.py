"""Diffusion schedule tables (host side, fp32 torch ops exactly as the reference computes them:
ProteinReDiff/difffusion.py:8-26 and model.py:172-190)."""
from __future__ import annotations

import math
from typing import Dict

import torch


def linear_beta_schedule(n_timestep: int, start: float = 0.0001, end: float = 0.02) -> torch.Tensor:
    return torch.linspace(start, end, n_timestep)


def cosine_beta_schedule(n_timestep: int) -> torch.Tensor:
    grid = torch.linspace(0, n_timestep, n_timestep + 1)
    abar = torch.cos((grid / (n_timestep + 1)) * math.pi * 0.5) ** 2
    abar = abar / abar[0]
    return torch.clip(1 - (abar[1:] / abar[:-1]), 0, 0.999)


def get_betas(n_timestep: int, schedule: str) -> torch.Tensor:
    """The reference prints and exits on an unknown name (difffusion.py:13-15); raising is the
    library-friendly equivalent."""
    if schedule == "linear":
        return linear_beta_schedule(n_timestep)
    if schedule == "cosine":
        return cosine_beta_schedule(n_timestep)
    raise ValueError(f"Invalid schedule: {schedule}")


def schedule_tables(num_steps: int, schedule: str, device="cpu") -> Dict[str, torch.Tensor]:
    """All tables of ``run_setup_schedule`` (model.py:172-190), attribute names unchanged."""
    t: Dict[str, torch.Tensor] = {}
    t["betas"] = get_betas(num_steps, schedule).to(device)
    t["alphas"] = 1.0 - t["betas"]
    t["alphas_cumprod"] = torch.cumprod(t["alphas"], 0)
    t["alphas_cumprod_prev"] = torch.cat([torch.ones(1, device=device), t["alphas_cumprod"][:-1]])
    t["one_minus_alphas_cumprod"] = 1.0 - t["alphas_cumprod"]
    t["one_minus_alphas_cumprod_prev"] = 1.0 - t["alphas_cumprod_prev"]
    t["sqrt_betas"] = torch.sqrt(t["betas"])
    t["sqrt_alphas"] = torch.sqrt(t["alphas"])
    t["sqrt_alphas_cumprod"] = torch.sqrt(t["alphas_cumprod"])
    t["sqrt_alphas_cumprod_prev"] = torch.sqrt(t["alphas_cumprod_prev"])
    t["sqrt_one_minus_alphas_cumprod"] = torch.sqrt(1.0 - t["alphas_cumprod"])
    t["sqrt_recip_alphas_cumprod"] = 1.0 / t["sqrt_alphas_cumprod"]
    t["sqrt_recipm1_alphas_cumprod"] = torch.sqrt(1.0 / t["alphas_cumprod"] - 1)
    t["posterior_mean_coef1"] = t["betas"] * t["sqrt_alphas_cumprod_prev"] / t["one_minus_alphas_cumprod"]
    t["posterior_mean_coef2"] = t["one_minus_alphas_cumprod_prev"] * t["sqrt_alphas"] / t["one_minus_alphas_cumprod"]
    t["posterior_variance"] = t["betas"] * t["one_minus_alphas_cumprod_prev"] / t["one_minus_alphas_cumprod"]
    return t


def reverse_coefficients(tab: Dict[str, torch.Tensor]) -> torch.Tensor:
    """[T,4] = (w_noise, 1/sqrt_alpha, sqrt_beta, 0) per step, the scalars of model.py:407-419."""
    w_noise = (1.0 - tab["alphas"]) / tab["sqrt_one_minus_alphas_cumprod"]
    inv_sa = 1.0 / tab["sqrt_alphas"]
    return torch.stack([w_noise, inv_sa, tab["sqrt_betas"], torch.zeros_like(w_noise)], dim=1).contiguous()
