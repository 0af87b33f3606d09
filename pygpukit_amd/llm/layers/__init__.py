from .attention import Attention
from .block import TransformerBlock
from .linear import Linear, LinearBF16, LinearFP8, quantize_linear_fp8
from .mlp import MLP
from .norm import Norm
from .rope import apply_rotary_pos_emb_numpy, precompute_freqs_cis

__all__ = ["Attention", "TransformerBlock", "Linear", "LinearBF16", "LinearFP8", "MLP", "Norm", "precompute_freqs_cis", "apply_rotary_pos_emb_numpy",
           "quantize_linear_fp8"]
