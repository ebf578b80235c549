"""Throughput of token-level continuous batching (llamagen_amd/serve.py) against the batch-level pipeline on the same workload
(development aid, round 3): GPT-L 384 px class-conditional requests, cfg 4.0, top-k 2000, decode only (no VQ).
    python tools/serve_bench.py [slots ...]        (default 32 64)
Prints requests/s and tokens/s for ContinuousBatcher(slots) draining 3 x slots requests that are all queued at t = 0, and for
SamplingPipeline on the same number of images (one chain of `slots` images at a time).
    python tools/serve_bench.py buckets
compares a 64-slot batcher with and without the captured slot counts 8 / 16 / 32 at LOW load (4 and 16 requests in flight):
time per decode step, i.e. the per-token latency a lightly loaded server gives each request."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from llamagen_amd.pipeline import SamplingPipeline
from llamagen_amd.serve import ContinuousBatcher

dev = torch.device("cuda:0")
N = 576
skw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)


def low_load(gpt):
    for buckets in (None, (8, 16, 32)):
        cb = ContinuousBatcher(gpt, 64, N, slot_buckets=buckets, **skw)
        for live in (4, 16, 64):
            res = []
            for rep in range(2):                      # first round warms kernels and captures that bucket's graph
                for l in torch.randint(0, 1000, (live,)).tolist():
                    cb.submit(l)
                s0 = cb.steps_run
                torch.cuda.synchronize(); t = time.perf_counter()
                out = cb.run()
                torch.cuda.synchronize(); dt = time.perf_counter() - t
                assert len(out) == live
                res.append((dt, cb.steps_run - s0))
            dt, steps = res[-1]
            print(f"slots=64 buckets={buckets}: {live:2d} requests in flight -> {dt / steps * 1e6:7.1f} us per step "
                  f"({live / dt:.1f} images/s, bucket {cb.B}, {cb.switches} switches so far)", flush=True)
        del cb
        torch.cuda.empty_cache()


def main():
    torch.set_grad_enabled(False)
    gpt, _ = bench.build_models(dev, 0)
    if sys.argv[1:] == ["buckets"]:
        return low_load(gpt)
    for slots in [int(a) for a in sys.argv[1:]] or [32, 64]:
        nreq = 3 * slots
        labels = torch.randint(0, 1000, (nreq,)).tolist()
        cb = ContinuousBatcher(gpt, slots, N, **skw)
        for l in labels[:slots]:                      # warm-up round: kernels, graph capture
            cb.submit(l)
        cb.run()
        torch.cuda.synchronize()
        for l in labels:
            cb.submit(l)
        t = time.perf_counter()
        out = cb.run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        assert len(out) == nreq
        print(f"ContinuousBatcher slots={slots}: {nreq} requests in {dt:.2f} s = {nreq / dt:.1f} requests/s, {nreq * N / dt / 1e3:.1f} k tokens/s "
              f"({cb.steps_run} steps)", flush=True)
        del cb
        gpt._engine = None
        torch.cuda.empty_cache()
        pipe = SamplingPipeline(gpt, None, lanes=1)
        pipe.prepare(slots, N, **skw)
        conds = [torch.randint(0, 1000, (slots,), device=dev) for _ in range(3)]
        torch.cuda.synchronize(); t = time.perf_counter()
        pipe.run(conds, N, **skw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"SamplingPipeline  batch={slots}: {nreq} images in {dt:.2f} s = {nreq / dt:.1f} images/s, {nreq * N / dt / 1e3:.1f} k tokens/s", flush=True)
        del pipe
        gpt._engine = None
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
