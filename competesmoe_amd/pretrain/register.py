"""`@register_moe(name)` / `get_moe(name)` of the pretrain stack (moe_pretrain_model/layers/moe/register.py), on the shared Registry."""
from ..registry import Registry

_REGISTRY = Registry("pretrain-stack MoE layer")
MOE_REGISTRY = _REGISTRY.classes
register_moe = _REGISTRY.register
get_moe = _REGISTRY.get
