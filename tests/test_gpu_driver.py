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
            "--per-proc-batch-size", "2", "--num-fid-samples", "5", "--sample-dir", str(tmp_path / "samples"), "--lanes", "2"]
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
