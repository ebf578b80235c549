"""bench.py's accounting helpers (no GPU): the algorithmic bytes behind `roofline.achieved` against SURVEY.md section 8d's
closed form, and the JSON contract's static fields."""
import types

import bench


def test_attention_bytes_match_survey_8d():
    cfg = types.SimpleNamespace(dim=1024, n_head=16, n_layer=24)          # GPT-L
    nbytes, launches = bench.attention_bytes_per_generate(cfg, 64, 576)    # config 2: 32 images, CFG -> 64 rows
    # kvb * B2 * [T + (N-1)(T+1) + (N-2)(N-1)/2], kvb = 98 304 B per token (24 layers x 2 x 16 x 64 x 2 B), T = 1
    kvb, N, T = 24 * 2 * 16 * 64 * 2, 576, 1
    assert nbytes == kvb * 64 * (T + (N - 1) * (T + 1) + (N - 2) * (N - 1) // 2) == 98304 * 64 * 166176
    assert launches == 576 * 24
    assert abs(nbytes / launches - 75.63e6) < 0.01e6                       # 75.6 MB per average launch
    twice, _ = bench.attention_bytes_per_generate(cfg, 128, 576)           # two batches per chain
    assert twice == 2 * nbytes


def test_constants_name_baseline_config_2():
    assert (bench.BATCH, bench.LAT, bench.IMG, bench.CFG, bench.TOPK) == (32, 24, 384, 4.0, 2000)
    assert bench.HBM_PEAK_GBS == 8000.0
