"""Rate-distortion reporting on the host: the two CSV tables of the reference's evaluation script and
its Bjontegaard-delta metrics.  Pure Python/NumPy/SciPy; no GPU work happens here.

Reference behaviour restated:
  * modelseval.py:196-256 — one row per (lambda, image) with columns lambda,image,bpp,mse,psnr,msssim
    ("per_image_per_lambda_results.csv") and one row per lambda with the plain means,
    lambda,count,bpp,mse,psnr,msssim, sorted by bpp ("agg_model_rd_summary.csv").
  * writeupbdcurvesjpegALL.py:114-171 — BD-rate (%) and BD-quality between two RD curves: sort each
    curve by quality, force strictly increasing quality (+1e-9) and log-rate (+1e-12), interpolate
    log-rate over quality with a shape-preserving cubic (PCHIP), integrate the RATE difference (not the
    log-rate difference) over the common quality range and divide by the mean rate of the second
    curve; BD-quality integrates the quality difference over the common log-rate range.
The reference's committed outputs (batch_bd_results_jpeg/*.csv) pin bd_metrics in
tests/test_report.py.
"""
from __future__ import annotations

import csv
import os
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
from scipy.integrate import quad
from scipy.interpolate import PchipInterpolator

DETAIL_COLUMNS = ("lambda", "image", "bpp", "mse", "psnr", "msssim")
SUMMARY_COLUMNS = ("lambda", "count", "bpp", "mse", "psnr", "msssim")


def _strictly_increasing(values: np.ndarray, tiny: float) -> np.ndarray:
    out = np.asarray(values, dtype=float).copy()
    for i in range(1, out.size):
        if out[i] <= out[i - 1]:
            out[i] = out[i - 1] + tiny
    return out


def bd_metrics(rate_a: Sequence[float], quality_a: Sequence[float], rate_b: Sequence[float],
               quality_b: Sequence[float]) -> Tuple[float, float]:
    """(BD-rate of curve a against curve b in percent, BD-quality a minus b).

    Rates in bpp, quality in dB (PSNR) or linear units (MS-SSIM).  Raises ValueError when the two
    curves share no quality range (writeupbdcurvesjpegALL.py:144-145)."""
    def prepared(rate, quality):
        rate = np.asarray(rate, dtype=float)
        quality = np.asarray(quality, dtype=float)
        order = np.argsort(quality)
        q = _strictly_increasing(quality[order], 1e-9)
        log_r = _strictly_increasing(np.log(rate[order]), 1e-12)
        return q, log_r

    qa, la = prepared(rate_a, quality_a)
    qb, lb = prepared(rate_b, quality_b)
    q_lo, q_hi = max(qa.min(), qb.min()), min(qa.max(), qb.max())
    if q_hi <= q_lo:
        raise ValueError("No overlap in quality range between curves -- cannot compute BD metrics.")
    rate_of_q_a, rate_of_q_b = PchipInterpolator(qa, la), PchipInterpolator(qb, lb)
    span = q_hi - q_lo
    mean_diff = quad(lambda q: np.exp(rate_of_q_a(q)) - np.exp(rate_of_q_b(q)), q_lo, q_hi)[0] / span
    mean_b = quad(lambda q: np.exp(rate_of_q_b(q)), q_lo, q_hi)[0] / span
    bd_rate = mean_diff / mean_b * 100.0

    r_lo, r_hi = max(la.min(), lb.min()), min(la.max(), lb.max())
    if r_hi <= r_lo:
        return bd_rate, float("nan")
    q_of_rate_a, q_of_rate_b = PchipInterpolator(la, qa), PchipInterpolator(lb, qb)
    bd_quality = quad(lambda r: q_of_rate_a(r) - q_of_rate_b(r), r_lo, r_hi)[0] / (r_hi - r_lo)
    return bd_rate, bd_quality


class RDReport:
    """Collects per-image results per lambda and writes the reference's two CSV tables."""

    def __init__(self):
        self.rows: List[Dict[str, object]] = []

    def add_batch(self, lam, names: Iterable[str], bpp, mse, psnr, msssim) -> None:
        """Per-image vectors of one evaluated batch (e.g. evaluate.evaluate_batch output) at trade-off lam."""
        cols = [np.asarray(v, dtype=float).reshape(-1) for v in (bpp, mse, psnr, msssim)]
        names = list(names)
        if any(c.size != len(names) for c in cols):
            raise ValueError("add_batch: every metric needs one value per image")
        for i, name in enumerate(names):
            self.rows.append({"lambda": lam, "image": name, "bpp": float(cols[0][i]), "mse": float(cols[1][i]),
                              "psnr": float(cols[2][i]), "msssim": float(cols[3][i])})

    def summary(self) -> List[Dict[str, object]]:
        """One row per lambda with plain means, sorted by bpp (modelseval.py:219-232, 254)."""
        by_lam: Dict[object, List[Dict[str, object]]] = {}
        for r in self.rows:
            by_lam.setdefault(r["lambda"], []).append(r)
        out = []
        for lam, rs in by_lam.items():
            out.append({"lambda": lam, "count": len(rs),
                        **{k: float(np.mean([r[k] for r in rs])) for k in ("bpp", "mse", "psnr", "msssim")}})
        return sorted(out, key=lambda r: r["bpp"])

    def write(self, output_dir: str) -> Tuple[str, str]:
        os.makedirs(output_dir, exist_ok=True)
        detail = os.path.join(output_dir, "per_image_per_lambda_results.csv")
        summary = os.path.join(output_dir, "agg_model_rd_summary.csv")
        _write_csv(detail, DETAIL_COLUMNS, self.rows)
        _write_csv(summary, SUMMARY_COLUMNS, self.summary())
        return detail, summary


def _write_csv(path: str, columns: Sequence[str], rows: Iterable[Dict[str, object]]) -> None:
    with open(path, "w", newline="") as f:
        w = csv.writer(f, lineterminator="\n")
        w.writerow(columns)
        for r in rows:
            w.writerow([repr(r[c]) if isinstance(r[c], float) else r[c] for c in columns])


def read_rd_csv(path: str) -> Dict[str, np.ndarray]:
    """Columns of an RD table written by RDReport or by the reference, as float arrays (image names as str)."""
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    out: Dict[str, np.ndarray] = {}
    for k in (rows[0].keys() if rows else ()):
        vals = [r[k] for r in rows]
        out[k] = np.array(vals) if k == "image" else np.array([float(v) for v in vals])
    return out
