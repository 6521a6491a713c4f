"""MI355X-native drop-in for the model-free hot path of OfflineRL-Kit.

Mirrors the reference's API surface for that path only:
  offlinerlkit.buffer.ReplayBuffer, offlinerlkit.policy.{CQL,IQL,TD3BC,EDAC}Policy,
  offlinerlkit.policy_trainer.MFPolicyTrainer, offlinerlkit.nets / offlinerlkit.modules.
All updates run in the HIP engine (liborlengine.so) through the C ABI in include/orl_engine.h.
"""
__version__ = "0.5.0"      # = the engine version reported by orl_version()
