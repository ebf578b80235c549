"""avg decode-attention launch time over a generate() (bench.py's own measure_attention) per kernel variant / cache policy.
    ROWS=128 python tools/attn_sweep.py [variants ...]      (ROWS: rows of the decode chain incl. CFG, default 128)"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from llamagen_amd import _lib as L

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
gpt, _ = bench.build_models(dev, 0)
N = 576
ROWS = int(os.environ.get("ROWS", "128"))
gpt.setup_caches(ROWS, 1 + N, torch.bfloat16)
nbytes, nl = bench.attention_bytes_per_generate(gpt.config, ROWS, N)
for v in [int(a) for a in sys.argv[1:]] or [2, 3, 1, 0, 4, 5]:
    gpt._engine.attn_variant = v
    sec, launches, per_pos = bench.measure_attention(gpt, ROWS, N)
    print(f"variant {v}: avg {sec / launches * 1e6:6.2f} us  {nbytes / sec / 1e9:7.1f} GB/s  frac {nbytes / sec / 8e12:.3f}  {per_pos}", flush=True)
