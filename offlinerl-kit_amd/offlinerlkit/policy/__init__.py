"""offlinerlkit.policy — the four model-free policies of the hot path plus SAC and the model-based callers whose ``learn`` reuses
its kernels (MOPO, COMBO, MCQ: SURVEY §8(f)3), engine-backed.
(The reference package also exports MOBILE / RAMBO / RCSL / diffusion policies; those are out of scope here.)"""
from .base_policy import BasePolicy, EnginePolicy
from .iql import IQLPolicy
from .sac_family import CQLPolicy, EDACPolicy
from .td3bc import TD3BCPolicy
from .model_based import SACPolicy, MOPOPolicy, COMBOPolicy
from .mcq import MCQPolicy

__all__ = ["BasePolicy", "EnginePolicy", "CQLPolicy", "IQLPolicy", "TD3BCPolicy", "EDACPolicy", "SACPolicy", "MOPOPolicy", "COMBOPolicy", "MCQPolicy"]
