"""Pretrain-stack MoE layers (drop-in for moe_pretrain_model/layers/moe + layers/cvmm.py)."""
from .register import register_moe, get_moe, MOE_REGISTRY
from .framework_layers import LoggingLayer, RegularizedLayer, OncePerIterLayer
from .cvmm import CVMMSel, cvmm, cvmm_prepare_sel2
from .moe import MoE
from .smoe import SMoeLayer
from .competesmoe import CompeteSMoE
from .deepseek import DeepSeekV2, DeepSeekV3
from .smoe_perturbed import MoEPerturbedCosingGating, Selection
from .block import MoEBlock
from .smoe_ep import EPSMoeLayer

__all__ = ["register_moe", "get_moe", "MOE_REGISTRY", "LoggingLayer", "RegularizedLayer", "OncePerIterLayer", "CVMMSel", "cvmm",
           "cvmm_prepare_sel2", "MoE", "SMoeLayer", "CompeteSMoE", "DeepSeekV2", "DeepSeekV3", "MoEPerturbedCosingGating", "Selection", "MoEBlock", "EPSMoeLayer"]
