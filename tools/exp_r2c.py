"""Full hot path (generate + decode_code) throughput for (images per chain) x (chains in flight), each configuration measured
twice, interleaved, over >= 384 images (development aid: is carrying two reference batches per chain worth building?)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from llamagen_amd.pipeline import SamplingPipeline
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); torch.set_grad_enabled(False)
gpt, vq = bench.build_models(dev, 0)
N = 576
skw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
cfgs = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(32, 3), (64, 2), (64, 3), (32, 2)]
pipes = {}
for B, lanes in cfgs:
    views = gpt.lane_view()  # own engines per configuration
    p = SamplingPipeline(views, vq, lanes=lanes)
    p.prepare(B, N, **skw)
    pipes[(B, lanes)] = p
for rep in range(2):
    for (B, lanes), p in pipes.items():
        K = max(lanes * 2, 384 // B)
        K -= K % lanes
        conds = [torch.randint(0, 1000, (B,), device=dev) for _ in range(K)]
        torch.cuda.synchronize(); t = time.perf_counter()
        p.run(conds, N, **skw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"rep {rep} B={B} lanes={lanes} K={K}: {B * K / dt:6.1f} img/s", flush=True)
