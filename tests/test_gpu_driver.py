"""a24: the reference's sample_c2i.py call sequence, driven end to end on the GPU through examples/sample_c2i.py --
reference-format checkpoints on disk (wrapper-key sniffing, stray freqs_cis), the reference's flags, generate() + decode_code(),
image grid on disk -- with the token ids held to the oracle bit for bit (fp32) and the pixels to 1e-3."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import llamagen_oracle as O  # noqa: E402
from llamagen_amd import GPT_models, VQ_models  # noqa: E402
from llamagen_amd.testing import synth_for_module  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example():
    spec = importlib.util.spec_from_file_location("example_sample_c2i", os.path.join(ROOT, "examples", "sample_c2i.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_sample_c2i_driver_matches_oracle(tmp_path):
    ex = _example()
    dev = torch.device("cuda:0")
    lat = 16  # --image-size 256
    gpt = GPT_models["GPT-B"](vocab_size=16384, block_size=lat * lat, num_classes=1000, cls_token_num=1, model_type="c2i")
    gsd = synth_for_module(gpt, seed=31, lin_std=0.02)
    disk = dict(gsd)
    disk["freqs_cis"] = torch.zeros(1 + lat * lat, 32, 2)
    torch.save({"model": disk, "optimizer": {}}, tmp_path / "gpt.pt")  # DDP layout; the other three: tests/test_checkpoint.py
    vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8)
    vsd = synth_for_module(vq, seed=32)
    torch.save({"model": vsd}, tmp_path / "vq.pt")
    argv = ["--gpt-model", "GPT-B", "--gpt-ckpt", str(tmp_path / "gpt.pt"), "--vq-ckpt", str(tmp_path / "vq.pt"), "--image-size", "256",
            "--precision", "none", "--seed", "5", "--cfg-scale", "4.0", "--top-k", "2000", "--out", str(tmp_path / "grid.png")]
    labels = [207, 360, 387]
    N = lat * lat
    args = ex.build_parser().parse_args(argv)
    res = ex.main(args, class_labels=labels)
    assert res["missing"] == [] and res["unexpected"] == []
    toks = res["index_sample"]
    assert toks.dtype == torch.int32 and tuple(toks.shape) == (3, N)
    # the oracle on the noise the GPU generator produced after torch.manual_seed(5)
    torch.manual_seed(5)
    qs = iter([torch.empty(3, 16384, device=dev).exponential_(1).cpu() for _ in range(N)])
    cfg = O.GPTConfig(**O.GPT_SIZES["GPT-B"], vocab_size=16384, block_size=lat * lat, num_classes=1000, cls_token_num=1)
    ref = O.generate(O.GPTOracle(cfg, gsd, torch.float32), torch.tensor(labels), N, cfg_scale=4.0, cfg_interval=-1, temperature=1.0,
                     top_k=2000, top_p=1.0, sample_logits=True, noise_fn=lambda s: next(qs))
    np.testing.assert_array_equal(toks.cpu().numpy(), ref.numpy())
    img_ref = O.vq_decode_code(vsd, ref.long(), [3, 8, lat, lat])
    assert (res["samples"].cpu() - img_ref).abs().max().item() < 1e-3
    from PIL import Image
    grid = np.asarray(Image.open(res["path"]))
    assert grid.shape == (256 + 4, 3 * 258 + 2, 3) and grid.dtype == np.uint8
    np.testing.assert_array_equal(grid[2:258, 2:258], O.to_uint8_hwc(res["samples"][:1].cpu())[0].numpy())


def test_sample_c2i_ddp_driver_equals_sequential_loop(tmp_path):
    """examples/sample_c2i_ddp.py (world 1, 2 lanes in flight, 3 iterations of 2 images, resize 256 -> ... kept at 256) writes the
    .npz the reference's loop would: labels and noise in the reference's RNG order, uint8 HWC, global index order."""
    spec = importlib.util.spec_from_file_location("example_sample_c2i_ddp", os.path.join(ROOT, "examples", "sample_c2i_ddp.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    from llamagen_amd import generate
    from llamagen_amd.postprocess import to_uint8_hwc
    dev = torch.device("cuda:0")
    lat = 16
    gpt = GPT_models["GPT-B"](vocab_size=16384, block_size=lat * lat, num_classes=1000, cls_token_num=1, model_type="c2i")
    gsd = synth_for_module(gpt, seed=41, lin_std=0.02)
    torch.save({"model": gsd}, tmp_path / "c2i_B_256.pt")
    vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8)
    vsd = synth_for_module(vq, seed=42)
    torch.save({"model": vsd}, tmp_path / "vq.pt")
    argv = ["--gpt-model", "GPT-B", "--gpt-ckpt", str(tmp_path / "c2i_B_256.pt"), "--vq-ckpt", str(tmp_path / "vq.pt"), "--image-size", "256",
            "--image-size-eval", "256", "--precision", "bf16", "--global-seed", "3", "--cfg-scale", "2.0", "--top-k", "500",
            "--per-proc-batch-size", "2", "--num-fid-samples", "5", "--sample-dir", str(tmp_path / "samples"), "--lanes", "2",
            "--batches-per-chain", "2"]  # iterations 0 and 1 share a 8-row chain; labels / noise still in the reference's RNG order
    path = ex.main(ex.build_parser().parse_args(argv))
    arr = np.load(path)["arr_0"]
    assert arr.shape == (5, 256, 256, 3) and arr.dtype == np.uint8 and "GPT-B-c2i_B_256-size-256-size-256-VQ-16-topk-500" in path
    # the reference's loop, sequentially, on the same seed
    gpt.load_state_dict(gsd, strict=False)
    gpt = gpt.to(device=dev, dtype=torch.bfloat16).eval()
    vq.load_state_dict(vsd)
    vq = vq.to(dev).eval()
    torch.manual_seed(3)  # rank_seed(3, 0, 1)
    ref = []
    for _ in range(3):
        c = torch.randint(0, 1000, (2,), device=dev)
        ids = generate(gpt, c, lat * lat, cfg_scale=2.0, cfg_interval=-1, temperature=1.0, top_k=500, top_p=1.0, sample_logits=True)
        ref.append(to_uint8_hwc(vq.decode_code(ids, [2, 8, lat, lat])).cpu())
    ref = torch.cat(ref).numpy()[:5]
    np.testing.assert_array_equal(arr, ref)


def test_sample_t2i_driver_matches_oracle(tmp_path):
    """examples/sample_t2i.py (reference flags, T5 features from a file): the left-padding rotation, `embs * masks`, generate() with
    emb_masks and decode_code of sample_t2i.py:95-123, token ids held to the oracle bit for bit (fp32)."""
    spec = importlib.util.spec_from_file_location("example_sample_t2i", os.path.join(ROOT, "examples", "sample_t2i.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    dev = torch.device("cuda:0")
    lat, T, C = 16, 120, 2048
    kw = dict(n_layer=2, n_head=4, dim=256, vocab_size=16384, block_size=lat * lat, cls_token_num=T, caption_dim=C, model_type="t2i")
    from llamagen_amd.gpt import ModelArgs, Transformer
    gpt = Transformer(ModelArgs(**kw))
    gsd = synth_for_module(gpt, seed=51, lin_std=0.05)
    torch.save({"model": gsd}, tmp_path / "t2i.pt")
    vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8)
    vsd = synth_for_module(vq, seed=52)
    torch.save({"model": vsd}, tmp_path / "vq.pt")
    g = torch.Generator().manual_seed(9)
    lens = [7, 33, 120]
    embs = torch.randn(3, T, C, generator=g)
    masks = torch.zeros(3, T, dtype=torch.int64)
    for b, n in enumerate(lens):
        masks[b, :n] = 1                                   # right-padded, as the T5 tokenizer emits them
    np.savez(tmp_path / "t5.npz", caption_embs=embs.numpy(), emb_masks=masks.numpy())
    # the example builds its model from the registry; give it this small architecture under a registry name
    ex.GPT_models["GPT-tiny-t2i"] = lambda **k: Transformer(ModelArgs(**{**kw, **{a: b for a, b in k.items() if a in ("block_size", "cls_token_num", "model_type")}}))
    argv = ["--gpt-model", "GPT-tiny-t2i", "--gpt-ckpt", str(tmp_path / "t2i.pt"), "--vq-ckpt", str(tmp_path / "vq.pt"), "--image-size", "256",
            "--precision", "none", "--seed", "11", "--cfg-scale", "7.5", "--top-k", "1000", "--t5-feature-path", str(tmp_path / "t5.npz"),
            "--out", str(tmp_path / "t2i.png")]
    try:
        res = ex.main(ex.build_parser().parse_args(argv))   # the registry key above is a valid --gpt-model choice now
    finally:
        ex.GPT_models.pop("GPT-tiny-t2i", None)
    toks = res["index_sample"]
    N = lat * lat
    assert toks.dtype == torch.int32 and tuple(toks.shape) == (3, N)
    # independent restatement of sample_t2i.py:95-108 for the oracle
    lp = torch.stack([torch.cat([embs[b, n:], embs[b, :n]]) for b, n in enumerate(lens)])
    lm = torch.flip(masks, dims=[-1])
    c = lp * lm[:, :, None]
    assert torch.equal(res["c_indices"].cpu().float(), c) and torch.equal(res["c_emb_masks"].cpu(), lm)
    torch.manual_seed(11)
    qs = iter([torch.empty(3, 16384, device=dev).exponential_(1).cpu() for _ in range(N)])
    cfg = O.GPTConfig(**{k: v for k, v in kw.items()})
    ref = O.generate(O.GPTOracle(cfg, gsd, torch.float32), c, N, emb_masks=lm, cfg_scale=7.5, cfg_interval=-1, temperature=1.0,
                     top_k=1000, top_p=1.0, sample_logits=True, noise_fn=lambda s: next(qs))
    np.testing.assert_array_equal(toks.cpu().numpy(), ref.numpy())
    img_ref = O.vq_decode_code(vsd, ref.long(), [3, 8, lat, lat])
    assert (res["samples"].cpu() - img_ref).abs().max().item() < 1e-3
    assert os.path.exists(res["path"])
