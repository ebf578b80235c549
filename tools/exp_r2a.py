"""Round-2 experiment batch A (development aid; results are quoted in DESIGN.md).  E1 needs the experimental
`lgen_touch_lines` kernel, which was removed from the library after the experiment came out negative (git history:
commit "Headline-config parity tests ..." has it); E2 / E3 run on the current library.
  E1  does the decode attention run faster when its K/V rows were pulled into the memory-side cache first
      (lgen_touch_lines), and what does the pull cost alone / next to a GEMM chain on another stream?
  E2  rows per decode chain x chains in flight (decode only): is a CFG batch better run as two half chains?
  E3  GEMM workgroup shapes under 3 lanes (LGEN_TILES)
"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, generate
from llamagen_amd import _lib as L
from llamagen_amd.pipeline import SamplingPipeline

dev = torch.device("cuda:0")
N = 576
skw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)


def build():
    torch.manual_seed(0)
    gpt = GPT_models["GPT-L"](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(gpt.output.weight, 0, 0.02)
    return gpt.to(device=dev, dtype=torch.bfloat16).eval()


def ev():
    return torch.cuda.Event(enable_timing=True)


def e1(gpt):
    B = 32
    gpt._engine = None
    gpt.setup_caches(2 * B, 1 + N, torch.bfloat16)
    torch.cuda.synchronize()
    e = gpt._engine
    lib = e.lib
    e.k_cache.normal_(0, 1)
    e.v_cache.normal_(0, 1)
    nl = e.L
    row_stride = e.S8 * e.kvs * 2  # bytes between (b, h) rows
    rows = e.B2 * e.H

    def attn(i):
        L.check(lib.lgen_attn_decode(L.ptr(e.qbuf), L.ptr(e.k_cache[i]), L.ptr(e.v_cache[i]), L.ptr(e.ap), L.ptr(e.state), 0, 0,
                                     e.B2, e.MTs, e.H, e.hd, e.hdp, e.S8, e.kvs, e.dt, L.stream()), "attn")

    def touch(i, pos, blocks=1024):
        for t in (e.k_cache[i], e.v_cache[i]):
            L.check(lib.lgen_touch_lines(L.ptr(t), (pos + 1) * e.hdp * 2, row_stride, rows, 0, 0, blocks, L.stream()), "touch")

    for nt in (1,):  # (the K/V cache policy was a run-time knob when this ran; it is fixed to non-temporal now)
        for pos in (64, 144, 288, 575):
            e.state.copy_(torch.tensor([pos, pos], dtype=torch.int32, device=dev))
            res = {}
            for mode in ("plain", "touched"):
                best = 1e9
                for rep in range(3):
                    evs = []
                    tevs = []
                    for i in range(nl):
                        if mode == "touched":
                            t0, t1 = ev(), ev()
                            t0.record(); touch(i, pos); t1.record()
                            tevs.append((t0, t1))
                        a0, a1 = ev(), ev()
                        a0.record(); attn(i); a1.record()
                        evs.append((a0, a1))
                    torch.cuda.synchronize()
                    tot = sum(a.elapsed_time(b) for a, b in evs) * 1e3 / nl
                    best = min(best, tot)
                    if tevs:
                        res["touch_us"] = min(res.get("touch_us", 1e9), sum(a.elapsed_time(b) for a, b in tevs) * 1e3 / nl)
                res[mode] = best
            mb = (pos + 1) * 2 * e.H * e.hd * 2 * e.B2 / 1e6
            print(f"E1 kv_nt={nt} pos={pos:3d} ({mb:6.1f} MB): attn plain {res['plain']:6.2f} us  after touch {res['touched']:6.2f} us  "
                  f"(touch K+V itself {res['touch_us']:6.2f} us = {mb / res['touch_us'] * 1e3 / 1e3:.2f} TB/s)", flush=True)

    # E1b: touch stream next to a GEMM chain
    pos = 288
    e.state.copy_(torch.tensor([pos, pos], dtype=torch.int32, device=dev))
    e.ssq_parts = e.d // 16
    tq, to, t13, t2 = e._tiles("qkv", 3 * e.d, e.d), e._tiles("wo", e.d, e.d), e._tiles("w13", 2 * e.F, e.d), e._tiles("w2", e.d, e.F)

    def gemm_chain():
        for w in e.layers:
            e.gemm(w["wo"], e.ap, e.hp, e.B2, e.MTs, e.d, e.d, L.EPI_RES, to, ssq_out=e.ssq)
            e.gemm(w["w13"], e.hp, e.gp, e.B2, e.MTs, 2 * e.F, e.d, L.EPI_SWIGLU, t13, norm_w=w["fn"])
            e.gemm(w["w2"], e.gp, e.hp, e.B2, e.MTs, e.d, e.F, L.EPI_RES, t2, ssq_out=e.ssq)
            L.check(lib.lgen_gemm_qkv_rope(L.ptr(w["wqkv"]), L.ptr(e.hp), L.ptr(e.qbuf), L.ptr(e.k_cache[0]), L.ptr(e.v_cache[0]),
                                           L.ptr(e.freqs_cis), L.ptr(e.state), e.B2, e.MTs, e.d, e.H, e.hd, e.hdp, e.S8, e.kvs, e.dt,
                                           tq[0], tq[1], tq[2], L.ptr(w["an"]), L.ptr(e.ssq), e.ssq_parts, e.eps, L.stream()), "qkv")

    def touch_chain(blocks):
        for i in range(nl):
            touch(i, pos, blocks)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    graphs = {}
    for name, fn, s in (("gemm", gemm_chain, s1), ("touch1024", lambda: touch_chain(1024), s2), ("touch256", lambda: touch_chain(256), s2),
                        ("touch64", lambda: touch_chain(64), s2)):
        with torch.cuda.stream(s):
            fn()
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                fn()
            graphs[name] = (g, s)
    torch.cuda.synchronize()

    def run(names, reps=5):
        best = {}
        for _ in range(3):
            marks = {}
            torch.cuda.synchronize()
            for n in names:
                g, s = graphs[n]
                with torch.cuda.stream(s):
                    a, b = ev(), ev()
                    a.record(s)
                    for _ in range(reps):
                        g.replay()
                    b.record(s)
                    marks[n] = (a, b)
            torch.cuda.synchronize()
            for n, (a, b) in marks.items():
                best[n] = min(best.get(n, 1e9), a.elapsed_time(b) * 1e3 / reps / nl)
        return best

    print("E1b per-layer us, alone:", {k: round(v, 2) for k, v in run(["gemm"]).items()}, {k: round(v, 2) for k, v in run(["touch1024"]).items()},
          {k: round(v, 2) for k, v in run(["touch256"]).items()}, {k: round(v, 2) for k, v in run(["touch64"]).items()}, flush=True)
    for t in ("touch1024", "touch256", "touch64"):
        print(f"E1b per-layer us, gemm chain + {t} concurrently:", {k: round(v, 2) for k, v in run(["gemm", t]).items()}, flush=True)


def e2(gpt, grid, tag="E2"):
    for B, lanes in grid:
        gpt._engine = None
        pipe = SamplingPipeline(gpt, None, lanes=lanes)
        pipe.prepare(B, N, **skw)
        K = max(2 * lanes, 64 // B)
        K -= K % lanes
        conds = [torch.randint(0, 1000, (B,), device=dev) for _ in range(K)]
        torch.cuda.synchronize(); t = time.perf_counter()
        pipe.run(conds, N, **skw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"{tag} decode only B={B} lanes={lanes} K={K}: {B * K / dt:6.1f} img/s  ({dt / K * lanes / N * 1e6:7.1f} us per step of a chain)"
              f"  tiles='{os.environ.get('LGEN_TILES', '')}'", flush=True)
        del pipe
        torch.cuda.empty_cache()


def main():
    which = sys.argv[1:] or ["e2", "e3"]
    gpt = build()
    if "e1" in which:
        try:
            e1(gpt)
        except Exception as ex:  # keep going: the other experiments are independent
            print("E1 failed:", repr(ex), flush=True)
    if "e2" in which:
        e2(gpt, [(32, 1), (16, 2), (8, 4), (16, 4), (32, 2), (16, 6), (32, 3), (32, 4)])
    if "e3" in which:
        for name, spec in {"fat_res": "wo=4,1,8;w2=4,1,8", "mid": "wo=2,1,8;w2=2,1,8",
                           "fat_all": "qkv=2,4,8;wo=4,1,8;w13=4,4,8;w2=4,1,8;head=4,4,8",
                           "fat_all_res2": "qkv=2,4,8;wo=2,1,8;w13=4,4,8;w2=2,1,8;head=4,4,8"}.items():
            os.environ["LGEN_TILES"] = spec
            e2(gpt, [(32, 3), (32, 1)], tag=f"E3 {name}")
        os.environ["LGEN_TILES"] = ""


if __name__ == "__main__":
    main()
