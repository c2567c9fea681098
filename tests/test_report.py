"""RD reporting: CSV layout of modelseval.py:244-256 and the BD metrics of
writeupbdcurvesjpegALL.py:114-171, pinned by the reference's own committed result files
(code/modelv2/batch_bd_results_jpeg/*.csv, batch_eval_model/*.csv; copied as data under
tests/golden/bd/)."""
import os

import numpy as np
import pytest

from dsic_amd import report

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bd")


def test_bd_metrics_reproduce_the_reference_summary():
    model = report.read_rd_csv(os.path.join(G, "agg_model_rd.csv"))
    jpeg = report.read_rd_csv(os.path.join(G, "agg_jpeg_rd.csv"))
    want = report.read_rd_csv(os.path.join(G, "bd_metrics_summary_jpeg.csv"))
    rate_psnr, d_psnr = report.bd_metrics(model["bpp"], model["psnr"], jpeg["bpp"], jpeg["psnr"])
    rate_mss, d_mss = report.bd_metrics(model["bpp"], model["msssim"], jpeg["bpp"], jpeg["msssim"])
    assert rate_psnr == pytest.approx(want["bd_rate_psnr_pct"][0], rel=1e-7, abs=1e-9)
    assert d_psnr == pytest.approx(want["bd_psnr_db"][0], rel=1e-7, abs=1e-9)
    assert rate_mss == pytest.approx(want["bd_rate_mss_pct"][0], rel=1e-7, abs=1e-9)
    assert d_mss == pytest.approx(want["bd_mss_diff"][0], rel=1e-7, abs=1e-9)


def test_bd_metrics_properties():
    r = np.array([0.2, 0.5, 1.0, 2.0])
    q = np.array([28.0, 31.0, 33.5, 35.0])
    rate, dq = report.bd_metrics(r, q, r, q)
    assert abs(rate) < 1e-9 and abs(dq) < 1e-9
    rate, _ = report.bd_metrics(1.1 * r, q, r, q)          # 10 % more bits at every quality
    assert rate == pytest.approx(10.0, rel=1e-9)
    with pytest.raises(ValueError):
        report.bd_metrics(r, q, r, q + 100.0)               # no common quality range


def test_csv_tables_match_the_reference_layout(tmp_path):
    rep = report.RDReport()
    rep.add_batch(25, ["a.png", "b.png"], [0.6, 0.5], [1e-3, 2e-3], [30.0, 27.0], [0.9, 0.8])
    rep.add_batch(10, ["a.png", "b.png"], [0.3, 0.2], [4e-3, 2e-3], [24.0, 27.0], [0.7, 0.75])
    detail, summary = rep.write(str(tmp_path))
    ref_detail_header = open(os.path.join(G, "per_image_head.csv")).readline().strip()
    ref_summary_header = open(os.path.join(G, "agg_model_rd_summary.csv")).readline().strip()
    assert open(detail).readline().strip() == ref_detail_header
    assert open(summary).readline().strip() == ref_summary_header
    s = report.read_rd_csv(summary)
    assert list(s["lambda"]) == [10.0, 25.0]                 # sorted by bpp
    assert list(s["count"]) == [2.0, 2.0]
    assert s["bpp"][1] == pytest.approx(0.55) and s["psnr"][0] == pytest.approx(25.5)
    d = report.read_rd_csv(detail)
    assert list(d["image"]) == ["a.png", "b.png", "a.png", "b.png"]
    # floats are written with their shortest round-trip representation, like pandas' to_csv
    assert "0.6," in open(detail).read()


def test_reference_summary_file_parses_and_orders_by_bpp():
    s = report.read_rd_csv(os.path.join(G, "agg_model_rd_summary.csv"))
    assert np.all(np.diff(s["bpp"]) > 0)
    with pytest.raises(ValueError):
        report.RDReport().add_batch(1, ["x"], [0.1, 0.2], [0.0], [0.0], [0.0])
