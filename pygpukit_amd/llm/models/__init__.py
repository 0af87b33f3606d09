from .causal import CausalTransformerModel, GPT2Model, LlamaModel, QwenModel

__all__ = ["CausalTransformerModel", "GPT2Model", "LlamaModel", "QwenModel"]
