"""llamagen_amd -- MI355X-native LlamaGen sampling engine (GPT decode loop + VQ tokenizer).

Drop-in surface of the reference (FoundationVision/LlamaGen):
    from llamagen_amd import GPT_models, VQ_models, generate
mirrors autoregressive/models/gpt.py:464-467, tokenizer/tokenizer_image/vq_model.py:424 and
autoregressive/models/generate.py:126.  All arithmetic runs in the HIP library behind the C ABI
declared in include/lgen.h; importing the package does not need a GPU, running it does.
"""
from .gpt import GPT_models, ModelArgs, Transformer  # noqa: F401
from .vq_model import VQ_models, VQModel  # noqa: F401
from .generate import generate  # noqa: F401

__all__ = ["GPT_models", "VQ_models", "generate", "Transformer", "VQModel", "ModelArgs"]
