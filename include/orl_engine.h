/*
 * orl_engine.h — C ABI of the MI355X-native offline-RL update engine.
 *
 * This is the drop-in boundary for the policy.learn() hot path of the
 * reference (zhaoyizhou1123/OfflineRL-Kit).  The reference has no FFI layer:
 * its boundary is the duck-typed Python interface
 *     BasePolicy.learn(batch) -> Dict[str,float]   (offlinerlkit/policy/base_policy.py:8-26)
 *     ReplayBuffer.sample(batch_size) -> Dict      (offlinerlkit/buffer/buffer.py:96-106)
 *     MFPolicyTrainer.train()                      (offlinerlkit/policy_trainer/mf_policy_trainer.py:41-90)
 * Each entry point below names the reference code it replaces.  Signatures
 * are plain C: pointers, sizes, no torch types.  The Python mirror
 * (offlinerl-kit_amd/offlinerlkit) binds them with ctypes; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returning int returns 0 on success, non-zero on error;
 *     the message is available from orl_last_error() (thread-local).
 *   - one engine = one device + one HIP stream; an engine is not thread-safe.
 *   - an engine carries `n_runs` independent runs (seeds) that are updated
 *     together by every kernel launch (run-batched, like an ensemble).  All
 *     host-side arrays have a leading run dimension [n_runs][...].
 *   - all floating point is fp32.  `precision` selects the MFMA scheme used by
 *     the GEMMs only: 0 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32),
 *     1 = split operands: every operand as hi + lo 16-bit planes (IEEE half:
 *     22 significand bits per operand, power-of-two operand scales folded back
 *     into the fp32 accumulators), 3 MFMAs per product, fp32 accumulate;
 *     orl_split_bits() reports the operand width of the loaded build (22; 16
 *     for the bf16-plane variant build).  In this mode hidden activations
 *     and inputs must stay below 65504 in magnitude (fp16 range) and weights
 *     below 1023 (they enter the products times 2^6): see orl_health.
 *     2 = three fp16 planes per operand (hi + mid + lo = 33 significand bits:
 *     an fp32 operand is represented exactly) and the six products down to
 *     2^-33 with fp32 accumulation -- the arithmetic class of the fp32 MFMA at
 *     more than its rate -- in the launches that have such a flavour: the
 *     weight-stationary forward / dgrad / wgrad launches of nets with
 *     256-wide hidden layers from 4096 batched rows (CQL's dominant launches
 *     at two and three hidden layers, the 256-row phases of CQL / IQL /
 *     TD3+BC / SAC at many runs, the forwards and dgrads of EDAC's ensemble
 *     critics); every other launch (tiled and few-row passes) runs the
 *     precision-0 kernels.  The fp16 operand range of
 *     precision 1 applies to those launches.
 *   - orl_step / orl_learn_n additionally return ORL_RC_UNHEALTHY (1) when the
 *     step(s) ran but a run's health flag is raised (non-finite loss or
 *     gradient, an operand beyond the split-precision range): results are
 *     delivered, orl_last_error() describes the runs, orl_health() has the flags.
 */
#ifndef ORL_ENGINE_H
#define ORL_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORL_ALGO_CQL 0   /* policy/model_free/cql.py:87-207   */
#define ORL_ALGO_IQL 1   /* policy/model_free/iql.py:86-139   */
#define ORL_ALGO_TD3BC 2 /* policy/model_free/td3bc.py:83-124 */
#define ORL_ALGO_EDAC 3  /* policy/model_free/edac.py:88-166  */
#define ORL_ALGO_SAC 4   /* policy/model_free/sac.py:88-140 (MOPOPolicy.learn on the real+model batch, model_based/mopo.py:81-84) */
#define ORL_ALGO_MCQ 5   /* policy/model_free/mcq.py:48-126 (SAC critics / actor + the VAE behaviour policy of nets/vae.py) */

/* per-run health flags (orl_health).  The reference raises nothing when a run diverges (its losses simply turn nan); here a diverging run
 * can additionally be MASKED by the arithmetic -- the ReLU of the matrix kernels works on the integer view of the activations and maps a NaN
 * whose sign bit is set to +0, and at precision 1 an operand beyond the fp16-plane range multiplies to NaN -- so the engine watches for it. */
#define ORL_HEALTH_NONFINITE_LOSS 1 /* a metric of a step (loss, alpha, Q statistic) was inf / nan */
#define ORL_HEALTH_NONFINITE_GRAD 2 /* a summed parameter gradient was inf / nan when Adam consumed it */
#define ORL_HEALTH_SPLIT_RANGE 4    /* precision 1: an input, stored hidden activation (|x| >= 65504) or weight (|w| >= 65504 / 2^6) is out of the operand range */
#define ORL_RC_UNHEALTHY 1

#define ORL_MAX_HIDDEN 4
#define ORL_MAX_METRICS 8
#define ORL_MAX_NOISE 6

/* network ids (per algorithm; state_dict prefixes of SURVEY.md Appendix B) */
#define ORL_NET_ACTOR 0
#define ORL_NET_CRITIC1 1     /* IQL: critic_q1 ; EDAC: critics (ensemble) */
#define ORL_NET_CRITIC2 2     /* IQL: critic_q2 */
#define ORL_NET_CRITIC1_OLD 3 /* EDAC: critics_old */
#define ORL_NET_CRITIC2_OLD 4
#define ORL_NET_CRITIC_V 5    /* IQL only  */
#define ORL_NET_ACTOR_OLD 6   /* TD3BC only */
#define ORL_NET_VAE_ENC 7     /* MCQ behaviour policy (nets/vae.py): e1, e2, [mean; log_std] */
#define ORL_NET_VAE_DEC 8     /*                                      d1, d2, d3            */
#define ORL_NUM_NETS 9

/* per-run scalars (not nn.Parameters in the reference: run_cql.py:102, cql.py:57) */
#define ORL_SCALAR_LOG_ALPHA 0
#define ORL_SCALAR_CQL_LOG_ALPHA 1
#define ORL_SCALAR_ALPHA 2 /* read-only: the alpha the next learn() will use */
/* optimizer state of the scalars (torch.optim.Adam exp_avg / exp_avg_sq of alpha_optim, cql_alpha_optim) and TD3's
 * _last_actor_loss (td3.py:59): readable / writable so a policy can be re-bound or checkpointed without losing them */
#define ORL_SCALAR_LOG_ALPHA_M 3
#define ORL_SCALAR_LOG_ALPHA_V 4
#define ORL_SCALAR_CQL_LOG_ALPHA_M 5
#define ORL_SCALAR_CQL_LOG_ALPHA_V 6
#define ORL_SCALAR_LAST_ACTOR_LOSS 7

/* optimizer ids for orl_set_lr (run_iql.py:133 mutates actor_optim's lr per epoch) */
#define ORL_OPT_ACTOR 0
#define ORL_OPT_CRITIC 1
#define ORL_OPT_ALPHA 2
#define ORL_OPT_CQL_ALPHA 3
#define ORL_OPT_CRITIC_V 4
#define ORL_OPT_VAE 5      /* MCQ behavior_policy_optim */

typedef struct orl_config {
  int32_t algo;
  int32_t obs_dim, act_dim;
  int32_t n_hidden;
  int32_t hidden[ORL_MAX_HIDDEN];
  int32_t batch_size;
  int32_t n_runs;    /* independent runs carried by this engine (>=1) */
  int32_t device;    /* HIP device ordinal */
  int32_t precision; /* 0 fp32 MFMA; 1 split-fp16 MFMA (hi + lo planes); 2 three fp16 planes (exact fp32 operands, six products) in the many-row critic launches, fp32 MFMA elsewhere */
  uint64_t seed;     /* device Philox seed for orl_learn_n */
  float gamma, tau;
  float actor_lr, critic_lr, alpha_lr;
  float adam_beta1, adam_beta2, adam_eps;
  /* SAC family (sac.py:42-48) */
  int32_t auto_alpha;
  float alpha;
  float target_entropy;
  /* CQL (cql.py:16-60) */
  float cql_weight, temperature;
  int32_t max_q_backup, deterministic_backup, with_lagrange;
  float lagrange_threshold, cql_alpha_lr;
  int32_t num_repeat_actions;
  float act_low, act_high;
  /* IQL (iql.py:16-50) */
  float expectile, iql_temperature, critic_v_lr;
  /* TD3+BC (td3bc.py:17-53) */
  float policy_noise, noise_clip, td3bc_alpha, max_action;
  int32_t update_actor_freq;
  /* EDAC (edac.py:15-52) */
  int32_t num_critics;
  float eta;
  /* COMBO (policy/model_based/combo.py:110-241): the CQL update on a batch whose first `cql_real_rows` rows are real data and the rest
   * model rollouts.  The conservative term repeats rows [cql_cons_row0, cql_cons_row0 + cql_cons_rows) ("model": the model part,
   * "mix": the whole batch) and its -w mean Q term runs over the real rows only.  0 = the whole batch (plain CQL). */
  int32_t cql_cons_row0, cql_cons_rows, cql_real_rows;
  /* MCQ (mcq.py:19-46, run_mcq.py:34-36, 93-101): VAE hidden width / latent size, lambda, behaviour-policy lr; the number of sampled
   * actions is num_repeat_actions, the VAE's max_action is max_action */
  int32_t vae_hidden, vae_latent;
  float mcq_lambda, behavior_lr;
  /* launch geometry of the weight-stationary kernels (csrc/ws_gemm.h), per engine:
   *   ws_one_round  0 (default): as many workgroups per net as fill whole rounds of the CUs; 1: CUs / nets workgroups per net, one round
   *                 (what a process that runs SEVERAL engines per GPU wants: the CUs one engine's launch leaves idle are where the
   *                 other engines' kernels run; bench.py's two-engine default);
   *   ws_cus        CUs one launch spreads over, 8..256 (0 = 256).
   * The environment variables ORL_WS_ONE_ROUND / ORL_WS_CUS, when set, override these fields; they are read once, in orl_engine_create. */
  int32_t ws_one_round, ws_cus;
  /* nn.Dropout(p) behind every hidden ReLU of the ACTOR backbone (nets/mlp.py:16-24; run_iql.py:34,106 builds only the actor backbone with
   * --dropout_rate).  IQL only; 0 = none.  Active in learn() (policy.train() mode, iql.py:122), never in select_action (eval mode).
   * Teacher-forced runs pass the keep masks (0 / 1) of the reference's draws as noise slots 0 .. n_hidden - 1. */
  float actor_dropout;
  /* optional caller-owned parameter arena (device pointer, orl_arena_floats()
   * floats) so that framework tensors can alias engine parameters; NULL = the
   * engine allocates with hipMalloc. */
  float* external_arena;
} orl_config;

/* Replay minibatch: the dict ReplayBuffer.sample returns (buffer.py:96-106).
 * Arrays are [n_runs][batch][dim] row-major; rewards/terminals [n_runs][batch]. */
typedef struct orl_batch {
  const float* observations;
  const float* actions;
  const float* next_observations;
  const float* rewards;
  const float* terminals;
  int32_t on_device; /* 0: host pointers (copied in), 1: device pointers */
} orl_batch;

/* Explicit noise for a teacher-forced step, in the reference's draw order.
 * CQL: [0] eps_actor (B,A) N(0,1); [1] eps_next (B,A) or (B*N,A) with
 * max_q_backup; [2] u_rand (B*N,A) U[low,high); [3] eps_pi (B*N,A); [4] eps_next_pi (B*N,A).
 * EDAC: [0] eps_actor, [1] eps_next.  TD3BC: [0] eps_target (B,A).  IQL: none.  SAC: [0] eps_next, [1] eps_actor (B,A).
 * MCQ: [0] eps_vae (B,Z), [1] eps_next (B,A), [2] z_ood (2B*N,Z) N(0,1) (clamped to +-0.5 by the engine like VAE.decode), [3] eps_ood (2B,A),
 * [4] eps_actor (B,A).
 * Each array has a leading n_runs dimension. */
typedef struct orl_noise {
  const float* slot[ORL_MAX_NOISE];
  int32_t on_device;
} orl_noise;

typedef struct orl_engine orl_engine;

/* -- lifecycle --------------------------------------------------------------- */
const char* orl_last_error(void);
const char* orl_version(void);
int orl_split_bits(void); /* significand bits an operand carries at precision 1 (22: fp16 hi + lo planes; 16: the bf16-plane variant build) */
void orl_config_default(orl_config* cfg, int32_t algo); /* script defaults: run_{cql,iql,td3bc,edac}.py get_args() */
int64_t orl_arena_floats(const orl_config* cfg);        /* size of the parameter arena for external_arena */
int orl_engine_create(const orl_config* cfg, orl_engine** out); /* replaces <Algo>Policy.__init__ + deepcopy of targets (sac.py:29-33) */
void orl_engine_destroy(orl_engine* e);
int orl_engine_sync(orl_engine* e);                     /* hipStreamSynchronize on the engine stream */

/* -- parameters (nn.Module.state_dict() view) -------------- */
int orl_net_present(orl_engine* e, int net);
int64_t orl_net_floats(orl_engine* e, int net);
int orl_net_num_tensors(orl_engine* e, int net);
/* name: reference state_dict key relative to the net prefix (e.g. "backbone.model.0.weight") */
int orl_net_tensor(orl_engine* e, int net, int idx, char* name, int name_cap, int64_t* offset_floats,
                   int32_t* ndim, int64_t shape[4]);
float* orl_net_ptr(orl_engine* e, int run, int net);    /* device pointer to the net's flat fp32 parameters */
int orl_net_set(orl_engine* e, int run, int net, const float* host, int64_t n_floats); /* load_state_dict */
int orl_net_get(orl_engine* e, int run, int net, float* host, int64_t n_floats);       /* state_dict */
int orl_scalar_set(orl_engine* e, int run, int which, float v);
int orl_scalar_get(orl_engine* e, int run, int which, float* v);
int orl_set_lr(orl_engine* e, int opt, float lr);       /* optim.param_groups[0]["lr"] = lr */
int orl_reset_optimizers(orl_engine* e);                /* fresh torch.optim.Adam state (step=0, m=v=0) */
/* torch.optim.Adam state_dict()["state"] of a trainable net's optimizer: exp_avg / exp_avg_sq flat in state_dict order
 * (orl_net_floats values each); the shared step count is orl_step_count / orl_set_step_count (every optimizer of a policy steps
 * once per learn(); TD3BC's actor optimizer steps on every update_actor_freq-th call and derives its own count from it). */
int orl_adam_get(orl_engine* e, int run, int net, float* exp_avg, float* exp_avg_sq, int64_t n_floats);
int orl_adam_set(orl_engine* e, int run, int net, const float* exp_avg, const float* exp_avg_sq, int64_t n_floats);
int orl_set_step_count(orl_engine* e, int64_t steps);   /* resume: Adam's t, TD3BC's _cnt and the device RNG offsets continue from here */

/* -- replay buffer (buffer/buffer.py): its own object, like the reference's ReplayBuffer ------- */
typedef struct orl_buffer orl_buffer;
/* ReplayBuffer.__init__ (:8-32): an empty HBM-resident SoA store on `device` */
int orl_buffer_create(int32_t obs_dim, int32_t act_dim, int32_t device, orl_buffer** out);
void orl_buffer_destroy(orl_buffer* b);
/* load_dataset (:72-86): host arrays -> HBM SoA; obs/next_obs [n][obs_dim], act [n][act_dim], rew/term [n] */
int orl_buffer_load(orl_buffer* b, const float* obs, const float* act, const float* next_obs, const float* rew,
                    const float* term, int64_t n);
/* normalize_obs (:88-94): (x - mean) / (std + eps) in place on the device; mean/std(+eps) (obs_dim each) to host */
int orl_buffer_normalize_obs(orl_buffer* b, float eps, float* mean_out, float* std_out);
int64_t orl_buffer_size(orl_buffer* b);
/* sample (:96-106): gather `batch` rows into caller-owned DEVICE arrays (packed [batch][dim], rew/term [batch]).
 * idx: host int64[batch] (the np.random.randint draw of the reference) or NULL = device Philox(seed, call counter). */
int orl_buffer_sample(orl_buffer* b, const int64_t* idx, int32_t batch, uint64_t seed, float* obs_out, float* act_out,
                      float* next_obs_out, float* rew_out, float* term_out);
/* lets orl_learn_n sample this buffer on the device (the buffer must outlive the engine's use of it) */
int orl_engine_attach_buffer(orl_engine* e, orl_buffer* b);

/* -- the hot path ---------------------------------------------------------------- */
/* policy.learn(batch) with explicit noise: one gradient step for every run.
 * metrics: host [n_runs][ORL_MAX_METRICS] in the reference's result-dict order
 * (orl_metric_name); synchronous, like the reference's .item() calls. */
int orl_step(orl_engine* e, const orl_batch* batch, const orl_noise* noise, float* metrics);
/* MFPolicyTrainer inner loop (mf_policy_trainer.py:52-60): n x {sample -> learn -> logkv_mean},
 * sampling and noise on device; metrics_mean: host [n_runs][ORL_MAX_METRICS] epoch means;
 * elapsed_ms (optional): HIP-event time of the n steps on the engine stream. */
int orl_learn_n(orl_engine* e, int n_steps, float* metrics_mean, float* elapsed_ms);
/* Sticky per-run health flags (ORL_HEALTH_* bits), flags_out: host uint32[n_runs] (may be NULL); returns the OR over the runs, < 0 on
 * error.  orl_step / orl_learn_n update the flags from what they already read back (metrics; one word per run that k_adam raises on a
 * non-finite gradient) and, at precision 1, scan the step's MFMA operands for the fp16-plane range when a run turned non-finite.
 * orl_health_check runs that range scan on demand (inputs, stored hidden activations and parameters of the LAST step; a pass over
 * the workspaces, not for the inner loop: MFPolicyTrainer calls it once per epoch); orl_health_clear resets the flags (after
 * load_state_dict / a restart of the diverged runs). */
int orl_health(orl_engine* e, uint32_t* flags_out);
int orl_health_check(orl_engine* e, uint32_t* flags_out);
int orl_health_clear(orl_engine* e);
int orl_num_metrics(orl_engine* e);
const char* orl_metric_name(orl_engine* e, int idx);
int64_t orl_step_count(orl_engine* e);

/* -- test / profiling taps --------------------------------------------------------- */
/* copies an intermediate of the LAST step to host: returns number of floats written or <0.
 * names: "q1","q2","target_q","q1a","q2a","logp_a", ... (algorithm specific); every engine also has the minibatch of the last
 * step ("b_obs","b_nobs","b_act","b_rew","b_term": what ReplayBuffer.sample returned / the device sampler drew) and its noise
 * arrays under their orl_noise slot names ("n_eps_actor", ...). */
int64_t orl_debug_read(orl_engine* e, int run, const char* name, float* host, int64_t cap);
/* packed ReLU-mask words (bit b of word w of a row <-> unit 32 w + b is > 0) of a hidden-activation workspace of the LAST step,
 * e.g. "ch0" / "ch1" = the CQL critics' hidden layers: [members][rows][width / 32] words; < 0 when the kernels that ran did not
 * emit bits for it.  What the backward kernels read instead of the activation (autograd's threshold_backward mask). */
int64_t orl_debug_read_bits(orl_engine* e, int run, const char* name, uint32_t* host, int64_t cap_words);
/* gradient of the LAST step w.r.t. the parameters of a trainable net, flat in state_dict order (orl_net_floats values): what
 * autograd leaves in param.grad before optimizer.step() (cql.py:180-190 etc.); the split-K slabs of the backward kernels summed. */
int orl_debug_grads(orl_engine* e, int run, int net, float* host, int64_t n_floats);
/* runs one generic GEMM tile configuration on host data (kernel unit tests): see csrc/gemm.h */
int orl_debug_gemm(int cfg, int mode, int M, int N, int K, const float* A, const float* B, const float* v0,
                   const float* v1, float* C, int ksplit, int precision);
/* times `reps` launches of one GEMM tile configuration on random data (kind 0 forward, 1 dgrad, 2 wgrad) */
int orl_debug_gemm_time(int cfg, int kind, int M, int N, int K, int nz, int ksplit, int reps, float* ms_out);
/* average duration (ms) of the kernel with the largest accumulated time during the last orl_learn_n
 * when profiling was enabled with orl_profile_enable(e,1); name copied to `name`. */
int orl_profile_enable(orl_engine* e, int on);
int orl_profile_query(orl_engine* e, int idx, char* name, int name_cap, double* total_ms, int64_t* launches,
                      double* flops_per_launch, double* bytes_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* ORL_ENGINE_H */
