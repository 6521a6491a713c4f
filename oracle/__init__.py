"""CPU oracle for the policy.learn() hot path (CQL / IQL / TD3+BC / EDAC).

TEST INFRASTRUCTURE ONLY.  This package is a numpy (fp32) restatement of the
reference algorithm with hand-written backward passes.  It exists so that the
HIP engine can be checked on identical replay batches and identical noise.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it; the product path (``offlinerl-kit_amd/``) never does and
fails loudly when the HIP extension is missing.

Parity pin: every function here is checked against golden vectors produced by
running the real reference (``/root/reference``, imported in the build
container by ``tests/golden/make_golden.py``) — see ``tests/test_oracle_golden.py``.

Reference anchors (file:line under /root/reference):
  nets/mlp.py:9-33, modules/critic_module.py:17-28, modules/actor_module.py:9-51,
  modules/dist_module.py:6-127, nets/ensemble_linear.py:9-41,
  policy/model_free/{sac,cql,iql,td3,td3bc,edac}.py, buffer/buffer.py:96-106.
"""
from . import nn  # noqa: F401
