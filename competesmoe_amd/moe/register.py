"""`@register_moe(name)` / `get_moe(name)` of the LLaVA stack (moe_model/model/moe/register.py:1-21), on the shared Registry."""
from ..registry import Registry

_REGISTRY = Registry("LLaVA-stack MoE layer")
MOE_REGISTRY = _REGISTRY.classes
register_moe = _REGISTRY.register
get_moe = _REGISTRY.get
