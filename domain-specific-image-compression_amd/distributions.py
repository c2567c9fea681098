"""Mirror of code/modelv2/distributions.py: densities in bits, on the GPU."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops

LOG2E = 1.0 / math.log(2.0)


class StudentT(nn.Module):
    """Per-channel Student-t, returns -log2 p(x) (distributions.py:11-31)."""

    def __init__(self, eps=1e-9):
        super().__init__()
        self.eps = eps

    @torch.no_grad()
    def neg_log2_prob(self, x, sigma, nu):
        return ops.student_t_bits(x, sigma, nu)


class FactorizedGaussian(nn.Module):
    """Zero-mean factorised Gaussian prior with per-channel log_sigma (distributions.py:33-46)."""

    def __init__(self, C):
        super().__init__()
        self.log_sigma = nn.Parameter(torch.zeros(C), requires_grad=False)

    @torch.no_grad()
    def neg_log2_prob(self, x):
        return ops.gaussian_bits(x, self.log_sigma)
