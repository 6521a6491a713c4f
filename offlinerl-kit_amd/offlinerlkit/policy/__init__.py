"""offlinerlkit.policy — the four model-free policies of the hot path, engine-backed.
(The reference package also exports model-based / RCSL / diffusion policies; those are out of scope here.)"""
from .base_policy import BasePolicy, EnginePolicy
from .iql import IQLPolicy
from .sac_family import CQLPolicy, EDACPolicy
from .td3bc import TD3BCPolicy

__all__ = ["BasePolicy", "EnginePolicy", "CQLPolicy", "IQLPolicy", "TD3BCPolicy", "EDACPolicy"]
