"""Registry of the pretrain-stack MoE classes (same contract as moe_pretrain_model/layers/moe/register.py:1-21)."""

MOE_REGISTRY = {}


def register_moe(*names):
    def decorate(cls):
        for name in names:
            if name in MOE_REGISTRY and MOE_REGISTRY[name] is not cls:
                raise AssertionError(f"Model named '{name}' conflicts with existing model! \n {cls} \n Models: {MOE_REGISTRY}")
            MOE_REGISTRY[name] = cls
        return cls
    return decorate


def get_moe(model_name):
    try:
        return MOE_REGISTRY[model_name]
    except KeyError:
        raise ValueError(f"Attempted to load moe method'{model_name}', but no model for this name found! "
                         f"Supported model names: {', '.join(MOE_REGISTRY.keys())}")
