"""Do G independent decode chains (B/G images each) on G streams overlap on this GPU?  (development aid)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, generate  # noqa: E402


def build(name, B, N, dev, stream):
    torch.manual_seed(0)
    m = GPT_models[name](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(m.output.weight, 0, 0.02)
    m = m.to(device=dev, dtype=torch.bfloat16).eval()
    c = torch.randint(0, 1000, (B,), device=dev)
    kw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
    with torch.cuda.stream(stream):
        generate(m, c, N, **kw)
    torch.cuda.synchronize()
    e = m._engine
    e.k_cache.normal_(0, 1)
    e.v_cache.normal_(0, 1)
    e.noise = torch.empty(N, B, 16384, device=dev).exponential_(1.0)
    return m, e


def capture(e, B, pos, stream):
    sp = dict(use_cfg=True, cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
    st = torch.tensor([pos, pos], dtype=torch.int32, device=e.dev)
    with torch.cuda.stream(stream):
        e.state.copy_(st)
        e.decode_step(B, sp)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        e.decode_step(B, sp)
    torch.cuda.synchronize()
    return g, st


def main(name="GPT-L", B=32, img=384):
    dev = torch.device("cuda:0")
    N = (img // 16) ** 2
    from llamagen_amd import _lib as L
    tilesets = {
        "default": {},
        "lean qkv(1,2) w13(1,2)": {"qkv": (1, 2, 8), "w13": (1, 2, 8)},
        "lean qkv(1,2) w13(2,2)": {"qkv": (1, 2, 8), "w13": (2, 2, 8)},
        "lean qkv(1,2)": {"qkv": (1, 2, 8)},
    }
    for G in (3,):
        streams = [torch.cuda.Stream() for _ in range(G)]
        engs = [build(name, B, N, dev, s) for s in streams]
        for tname, tiles in tilesets.items():
            for m, e in engs:
                e.tile_override = dict(tiles)
            res = []
            for pos in (8, N // 2):
                gs = [capture(e, B, pos, s) for (m, e), s in zip(engs, streams)]
                best = 1e9
                for _ in range(3):
                    for (g, st), (m, e) in zip(gs, engs):
                        e.state.copy_(st)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        for (g, st), s in zip(gs, streams):
                            with torch.cuda.stream(s):
                                g.replay()
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
                res.append(f"pos {pos}: {best / G:7.1f} us/batch-step")
                del gs
            print(f"{G} concurrent batches, tiles {tname:24s}: " + "   ".join(res), flush=True)
        del engs


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0] if a else "GPT-L", int(a[1]) if len(a) > 1 else 32, int(a[2]) if len(a) > 2 else 384)
