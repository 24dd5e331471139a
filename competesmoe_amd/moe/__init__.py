"""LLaVA-stack MoE layers (drop-in for moe_model/model/moe)."""
from .register import register_moe, get_moe, MOE_REGISTRY
from .moe import MoeLayer
from .smoe import SMoeLayer
from .competesmoe import CompeteSMoE
from .shard_smoe import MoEShareLayer, DeepSeekV3ShareLayer
from .block import MoEBlock

__all__ = ["register_moe", "get_moe", "MOE_REGISTRY", "MoeLayer", "SMoeLayer", "CompeteSMoE", "MoEShareLayer",
           "DeepSeekV3ShareLayer", "MoEBlock"]
