"""CPU: pins the numpy oracle against golden vectors produced by the real
reference (tests/golden/make_golden.py).  Gate: losses / Q-values within 1e-4
relative over teacher-forced windows, parameters within a few
1e-6 absolute (Adam moves them by ~lr per step)."""
import numpy as np
import pytest

import synth
from helpers import load_golden, cql_oracle_setup, rel_err, rel_err_keys, key_scales, scale_err, check_state_against_golden


@pytest.mark.parametrize("case", list(synth.CQL_CASES) + list(synth.CQL_EXTRA_CASES))
def test_cql_oracle_matches_reference(case):
    from oracle import cql as ocql
    g = load_golden(case)
    cfg, st, batches, noises = cql_oracle_setup(case)
    keys = [str(k) for k in g["loss_keys"]]
    for k, (b, n) in enumerate(zip(batches, noises)):
        res, aux = ocql.learn(st, cfg, b, n)
        assert list(res.keys()) == keys
        got = np.array([res[x] for x in keys])
        ref = g[f"step{k}/losses"]
        assert rel_err(got, ref, floor=1e-2) < 1e-4, (case, k, got, ref)
        assert rel_err_keys(got, ref, key_scales(g)) < 1e-4, (case, k, got, ref)      # every key against its OWN scale over the window
        if k == 0:
            assert scale_err(aux["q1a"], g["step0/c1_qa"]) < 1e-5
            assert scale_err(aux["q2a"], g["step0/c2_qa"]) < 1e-5
            assert scale_err(aux["q1"], g["step0/c1_q"]) < 1e-5
            assert scale_err(aux["q2"], g["step0/c2_q"]) < 1e-5
            if "step0/target_q" in g.files:
                assert scale_err(aux["target_q"], g["step0/target_q"]) < 1e-5
        if k in (0, len(batches) - 1):
            nets = {nm: st[nm] for nm in ("actor", "critic1", "critic2", "critic1_old", "critic2_old")}
            check_state_against_golden(g, f"state{k}", nets, atol=2e-6 * (k + 1))
            if cfg["auto_alpha"]:
                assert abs(float(st["log_alpha"][0]) - float(g[f"state{k}/log_alpha"][0])) < 1e-6
            assert abs(float(st["cql_log_alpha"][0]) - float(g[f"state{k}/cql_log_alpha"][0])) < 1e-6


def _run_generic(algo, case, net_names, aux_checks):
    from helpers import generic_oracle_setup
    g = load_golden(case)
    mod, cfg, st, batches, noises = generic_oracle_setup(algo, case)
    keys = [str(k) for k in g["loss_keys"]]
    for k, (b, n) in enumerate(zip(batches, noises)):
        res, aux = mod.learn(st, cfg, b, n)
        assert list(res.keys()) == keys
        got = np.array([res[x] for x in keys])
        assert rel_err(got, g[f"step{k}/losses"], floor=1e-2) < 1e-4, (case, k, got, g[f"step{k}/losses"])
        if k == 0:
            for okey, gkey in aux_checks:
                if okey in aux:
                    assert scale_err(aux[okey], g[gkey]) < 1e-5, (okey,)
        if f"state{k}/{net_names[0]}/{next(iter(st[net_names[0]]))}/digest" in g.files:
            check_state_against_golden(g, f"state{k}", {nm: st[nm] for nm in net_names}, atol=2e-6 * (k + 1))


@pytest.mark.parametrize("case", list(synth.IQL_CASES))
def test_iql_oracle_matches_reference(case):
    _run_generic("iql", case, ("actor", "critic_q1", "critic_q2", "critic_v", "critic_q1_old", "critic_q2_old"),
                 (("q1", "step0/q1"), ("v", "step0/v")))


@pytest.mark.parametrize("case", list(synth.TD3BC_CASES))
def test_td3bc_oracle_matches_reference(case):
    _run_generic("td3bc", case, ("actor", "critic1", "critic2", "actor_old", "critic1_old", "critic2_old"),
                 (("q1", "step0/q1"), ("q_pi", "step0/q_pi")))


@pytest.mark.parametrize("case", list(synth.EDAC_CASES))
def test_edac_oracle_matches_reference(case):
    """Includes the analytic restatement of the double-backward gradient-diversity term."""
    from helpers import generic_oracle_setup
    g = load_golden(case)
    mod, cfg, st, batches, noises = generic_oracle_setup("edac", case)
    keys = [str(k) for k in g["loss_keys"]]
    for k, (b, n) in enumerate(zip(batches, noises)):
        res, aux = mod.learn(st, cfg, b, n)
        assert list(res.keys()) == keys
        got = np.array([res[x] for x in keys])
        assert rel_err(got, g[f"step{k}/losses"], floor=1e-2) < 1e-4, (case, k, got, g[f"step{k}/losses"])
        if k == 0:
            assert scale_err(aux["qas"], g["step0/qas"]) < 1e-5
            assert scale_err(aux["qs"], g["step0/qs"]) < 1e-5
        if k in (0, len(batches) - 1):
            check_state_against_golden(g, f"state{k}", {nm: st[nm] for nm in ("actor", "critics", "critics_old")}, atol=2e-6 * (k + 1))
            assert abs(float(st["log_alpha"][0]) - float(g[f"state{k}/log_alpha"][0])) < 1e-6


@pytest.mark.parametrize("case", list(synth.CQL_CASES))
def test_torch_cpu_counterpart_of_cql_matches_reference(case):
    """oracle/torch_cql.py (stock torch autograd + torch.optim.Adam, written against SURVEY Appendix A.1) is bench.py's
    `cpu_baseline` of kind "port" timed with torch threads: it must reproduce the real reference's losses on the
    fixtures over the whole teacher-forced window."""
    import torch
    from oracle.torch_cql import TorchCQL
    torch.set_num_threads(4)
    g = load_golden(case)
    cfg, st, batches, noises = cql_oracle_setup(case)
    keys = [str(k) for k in g["loss_keys"]]
    pol = TorchCQL(st, cfg)
    for k, (b, n) in enumerate(zip(batches[:6], noises[:6])):
        res = pol.learn(b, n)
        assert list(res.keys()) == keys
        got = np.array([res[x] for x in keys])
        assert rel_err(got, g[f"step{k}/losses"], floor=1e-2) < 2e-5, (case, k, got, g[f"step{k}/losses"])


NETS5 = ("actor", "critic1", "critic2", "critic1_old", "critic2_old")


@pytest.mark.parametrize("case", list(synth.MOPO_CASES))
def test_sac_oracle_matches_reference_mopo_learn(case):
    """MOPOPolicy.learn = SACPolicy.learn on torch.cat([real, fake]) (mopo.py:81-84, sac.py:88-140)"""
    from helpers import mopo_oracle_setup
    from oracle import sac as osac
    g = load_golden(case)
    cfg, st, batches, noises = mopo_oracle_setup(case)
    keys = [str(k) for k in g["loss_keys"]]
    for k, (b, n) in enumerate(zip(batches, noises)):
        res, aux = osac.learn(st, cfg, synth.mix_batch(b), n)
        assert list(res.keys()) == keys
        got = np.array([res[x] for x in keys])
        assert rel_err(got, g[f"step{k}/losses"], floor=1e-2) < 1e-4, (case, k, got, g[f"step{k}/losses"])
        if k == 0:
            assert scale_err(aux["q1"], g["step0/c1_q"]) < 1e-5 and scale_err(aux["q1a"], g["step0/c1_qa"]) < 1e-5
        if k in (0, len(batches) - 1):
            check_state_against_golden(g, f"state{k}", {nm: st[nm] for nm in NETS5}, atol=2e-6 * (k + 1))
            if cfg["auto_alpha"]:
                assert abs(float(st["log_alpha"][0]) - float(g[f"state{k}/log_alpha"][0])) < 1e-6


@pytest.mark.parametrize("case", list(synth.COMBO_CASES))
def test_cql_oracle_matches_reference_combo_learn(case):
    """COMBOPolicy.learn (combo.py:110-241): the CQL update on the mixed batch, conservative rows from rho_s, data term on real rows"""
    from helpers import combo_oracle_setup
    from oracle import cql as ocql
    g = load_golden(case)
    cfg, st, batches, noises = combo_oracle_setup(case)
    keys = [str(k) for k in g["loss_keys"]]
    Br = cfg["real_rows"]
    for k, (b, n) in enumerate(zip(batches, noises)):
        res, aux = ocql.learn(st, cfg, synth.mix_batch(b), n)
        assert list(res.keys()) == keys
        got = np.array([res[x] for x in keys])
        assert rel_err(got, g[f"step{k}/losses"], floor=1e-2) < 1e-4, (case, k, got, g[f"step{k}/losses"])
        if k == 0:
            assert scale_err(aux["q1"], g["step0/c1_q"]) < 1e-5 and scale_err(aux["q1a"], g["step0/c1_qa"]) < 1e-5
            assert scale_err(aux["q1"][:Br], g["step0/c1_q_real"]) < 1e-5          # the separate forward on the real rows = the first Br rows
            assert scale_err(aux["cat_q1"][:, 2] + np.log(0.5 ** cfg["act_dim"]), g["step0/c1_q_rand"][:, 0]) < 1e-5
        if k in (0, len(batches) - 1):
            check_state_against_golden(g, f"state{k}", {nm: st[nm] for nm in NETS5}, atol=2e-6 * (k + 1))
            assert abs(float(st["cql_log_alpha"][0]) - float(g[f"state{k}/cql_log_alpha"][0])) < 1e-6


@pytest.mark.parametrize("case", list(synth.MCQ_CASES))
def test_mcq_oracle_matches_reference(case):
    """MCQPolicy.learn (mcq.py:48-126): VAE behaviour-policy step, in-distribution + OOD critic targets, SAC actor / temperature"""
    from helpers import mcq_oracle_setup
    from oracle import mcq as omcq
    g = load_golden(case)
    cfg, st, batches, noises = mcq_oracle_setup(case)
    keys = [str(k) for k in g["loss_keys"]]
    for k, (b, n) in enumerate(zip(batches, noises)):
        res, aux = omcq.learn(st, cfg, b, n)
        assert list(res.keys()) == keys
        got = np.array([res[x] for x in keys])
        assert rel_err(got, g[f"step{k}/losses"], floor=1e-2) < 1e-4, (case, k, got, g[f"step{k}/losses"])
        if k == 0:
            assert scale_err(aux["q1"], g["step0/c1_q"]) < 1e-5 and scale_err(aux["q1_ood"], g["step0/c1_q_ood"]) < 1e-5
            assert scale_err(aux["q1a"], g["step0/c1_qa"]) < 1e-5
        if k in (0, len(batches) - 1):
            check_state_against_golden(g, f"state{k}", {nm: st[nm] for nm in NETS5 + ("behavior_policy",)}, atol=4e-6 * (k + 1))
