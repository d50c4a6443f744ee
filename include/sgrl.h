/* sgrl.h -- C ABI of libsgrl_hip.so, the MI355X-native batched rollout engine for the SGRL hot path.
 *
 * The reference has no FFI for this path (it is pure Python over mujoco-py / PyTorch); the entry points below
 * are what a binding for the path would bind, each replacing one reference interface:
 *
 *   sgrl_engine_create    <- SubprocVecEnv.__init__(env_fns)            reference src/subproc_vec_env.py:34-52
 *                            + utils.makeEnvWrapper / registerEnvs       reference src/utils.py:14-80
 *   sgrl_reset            <- SubprocVecEnv.reset()                       reference src/subproc_vec_env.py:65-68
 *                            -> ModularEnvWrapper.reset -> reset_model   reference src/wrappers.py:56-65,
 *                                                                        src/environments/3d_walker_7_full.py:150-164
 *   sgrl_step             <- SubprocVecEnv.step_async + step_wait        reference src/subproc_vec_env.py:54-63 (+ worker :12-15)
 *                            -> ModularEnvWrapper.step -> ModularEnv.step -> do_simulation -> _get_obs
 *                                                                        reference src/wrappers.py:39-54,
 *                                                                        src/environments/3d_walker_7_full.py:15-148
 *   sgrl_get_records / sgrl_set_records / sgrl_refresh
 *                         <- sim.get_state / MujocoEnv.set_state (+ sim.forward)   [gym 0.17.2, 3P]
 *   sgrl_set_actor_*      <- SEPolicy.forward under torch.no_grad()      reference src/SEActor.py:334-347,
 *                                                                        src/agent.py:189-198  (see sgrl_set.h)
 *
 * Conventions: plain pointers and sizes, no exceptions, int return codes (0 = ok, <0 = error; message via
 * sgrl_last_error()).  Pointers marked DEV are device (HIP) pointers owned by the caller (e.g. torch
 * tensor.data_ptr()); HOST pointers are ordinary host memory.  `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  A handle is not thread-safe; calls on one handle must not overlap.
 */
#ifndef SGRL_H
#define SGRL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgrl_engine sgrl_engine;

enum {
  SGRL_OK = 0,
  SGRL_ERR_ARG = -1,      /* bad argument (null pointer, sizes that do not match the morphologies) */
  SGRL_ERR_MODEL = -2,    /* malformed model blob */
  SGRL_ERR_HIP = -3,      /* HIP runtime error (no device, allocation, launch) */
  SGRL_ERR_LIMIT = -4     /* a morphology exceeds the engine's limits (nv > 64, LDS footprint > 160 KiB) */
};

/* Create a batch of environments.
 *   n_morph            number of distinct morphologies
 *   ib/ib_len, fb/fb_len   HOST: per-morphology model blobs (include/sgrl_model.h; sgrl_amd.model_pack.pack_model)
 *   morph_count        HOST: envs per morphology; env indices are assigned contiguously in morphology order
 *   obs_max_len        row length of observation outputs  (reference main.py:108-117: 41 * max_limbs)
 *   action_max_len     row length of action inputs        (3 * max_limbs; first 3 slots are torso dummies)
 *   seed, env_id_base  counter-RNG key: env i draws from stream (seed, env_id_base + i, episode)
 *   max_episode_steps  gym TimeLimit (reference arguments.py:109-114, default 1000); <= 0 disables
 */
int sgrl_engine_create(int n_morph, const int32_t* const* ib, const int32_t* ib_len, const double* const* fb,
                       const int32_t* fb_len, const int32_t* morph_count, int obs_max_len, int action_max_len,
                       uint64_t seed, uint32_t env_id_base, int max_episode_steps, sgrl_engine** out);
void sgrl_engine_destroy(sgrl_engine* e);

int sgrl_num_envs(const sgrl_engine* e);
int sgrl_record_stride(const sgrl_engine* e);   /* doubles per env record: max over morphs of nq+nv+4 */
int sgrl_lds_bytes(const sgrl_engine* e);       /* dynamic LDS per workgroup used by the step kernel */
int sgrl_launch_groups(const sgrl_engine* e);   /* concurrent k_env_step dispatches one sgrl_step issues (one per LDS occupancy class / kernel family) */
/* How many of those dispatches run on a FIXED-DIMENSION kernel (sgrl_amd/csrc/step_spec.hip: the dimension sets of a shipped
 * morphology family as compile-time constants) instead of the generic one; 0 for custom XMLs / row caps, or with SGRL_SPECS=0. */
int sgrl_fixed_dim_groups(const sgrl_engine* e);
/* How many environments step TWO to a wavefront (sgrl_amd/csrc/wave_half.h): environments of the light morphologies (walker_2 / _3 / _4,
 * hopper_3 / _4: at most 15 dofs) on a fixed-dimension kernel, paired with a neighbour of the same morphology; 0 with SGRL_PAIR=0.  Same results to rounding. */
int sgrl_paired_envs(const sgrl_engine* e);

/* VecEnv.reset(): every env starts a new episode.  obs: DEV float[n_env*obs_max_len]; obs64: DEV double[...] or NULL. */
int sgrl_reset(sgrl_engine* e, float* obs, double* obs64, void* stream);

/* VecEnv.step(actions).  actions: DEV float[n_env*action_max_len] (policy order, zero padded).
 * Outputs (DEV, any may be NULL except obs): obs float[n_env*obs_max_len] -- the post-step observation, or the
 * reset observation for envs that finished and auto_reset != 0 (reference subproc_vec_env.py:12-15);
 * reward float[n_env]; done uint8[n_env]; dist float[n_env] (info["dist"]); truncated uint8[n_env]
 * (gym's info["TimeLimit.truncated"]); obs64/reward64: double-precision copies for parity tests. */
int sgrl_step(sgrl_engine* e, const float* actions, float* obs, float* reward, uint8_t* done, float* dist,
              uint8_t* truncated, double* obs64, double* reward64, int auto_reset, void* stream);

/* Raw state access (teacher-forced parity, checkpointing).  HOST buffers:
 *   rec[n_env * stride]: qpos[nq] | qvel[nv] | torso_xy_stale[2] | target[2] (rest of the row unused)
 *   cnt[n_env * 4]:      step_count, episode, constraint-row overflow count, reserved */
int sgrl_get_records(sgrl_engine* e, double* rec, int32_t* cnt);
int sgrl_set_records(sgrl_engine* e, const double* rec, const int32_t* cnt);
/* After sgrl_set_records: recompute kinematics (gym set_state -> sim.forward) and emit observations. */
int sgrl_refresh(sgrl_engine* e, float* obs, double* obs64, void* stream);

/* Time `reps` back-to-back step launches with HIP events on `stream`; returns mean milliseconds per launch in
 * *ms_out (used by bench.py's roofline block; state advances as in sgrl_step). */
int sgrl_time_steps(sgrl_engine* e, const float* actions, float* obs, float* reward, uint8_t* done, int reps,
                    void* stream, float* ms_out);

/* The replay push's row format (reference common/buffer.py:75-84 `add_transition` arguments + the bookkeeping of
 * trainer.py:205-232), written for all environments in one launch:
 *   block[n_env][2 * obs_len + act_len + 4] = obs | action | next_obs | reward | done | store | morph_id   (float32)
 * Sources are device pointers with row strides ld_*; a NULL source leaves its columns untouched (the observation half can be
 * written before the step overwrites the observation buffer, the rest after it).  `done` comes as float32 OR as bytes (non-zero =
 * 1.0), not both; `store` as bytes; `morph_id` as int64 (small integers: exact in float32). */
int sgrl_pack_transitions(const float* obs, int ld_obs, const float* action, int ld_act, const float* next_obs, int ld_next,
                          const float* reward, const float* done_f32, const uint8_t* done_u8, const uint8_t* store,
                          const int64_t* morph_id, float* block, int n_env, int obs_len, int act_len, void* stream);

/* The collection-round bookkeeping of the reference's loop (reference src/trainer.py:205-232: episode_timesteps_list, done_list, the
 * time-limit rule `done_bool = 0 if episode_timesteps + 1 == max_episode_steps`, episode_reward_list / its buffer) for all n_env
 * environments of a rank in one launch, on DEVICE arrays the caller owns: reward f32 / done u8 as sgrl_step wrote them; state
 * done_list u8, ep_steps int64, ep_reward f32, reward_buf f32 (all zero at the start of a round); outputs store u8 (the row belongs to
 * the environment's first episode of the round) and done_bool f32 (the `done` to store) -- what sgrl_pack_transitions takes -- and
 * all_done (one int32: non-zero when every environment has finished its first episode). */
int sgrl_round_record(const float* reward, const uint8_t* done, uint8_t* done_list, int64_t* ep_steps, float* ep_reward, float* reward_buf,
                      uint8_t* store, float* done_bool, int32_t* all_done, int n_env, int max_episode_steps, void* stream);

/* The learner's side of the replay push: the kept rows of one gathered block (format above) go into the ring buffers of their
 * morphologies -- reference common/buffer.py:75-84 `add_transition` once per stored row, one buffer per morphology
 * (main.py:141-155), the block walked in global environment order (trainer.py:205-236) -- in ONE launch.  Row r is written to
 * ring slot `slot[r]` of ring `morph_id(r)` (its columns cut to that ring's obs_dim / act_dim); slot[r] < 0 skips the row.  The
 * caller computes the slots (write pointer + rank of the row among the stored rows of its morphology, modulo the capacity):
 * sgrl_amd/rollout.py TransitionSink.ingest.  `rings` is a DEVICE array of n_rings descriptors holding device pointers. */
typedef struct sgrl_ring {
  float* obs; float* action; float* next_obs; float* reward; float* done;   /* [capacity][obs_dim] | [capacity][act_dim] | ... */
  int32_t obs_dim, act_dim;
} sgrl_ring;
int sgrl_ingest_rows(const float* block, int n_rows, int obs_len, int act_len, const int64_t* slot, const sgrl_ring* rings,
                     int n_rings, void* stream);
/* The same with the slot arithmetic on the device too (three launches for ANY number of rows -- one rank's block or the learner's
 * whole gather of N ranks, laid out contiguously in rank order -- and no host involvement): `pos` / `cap` / `pending` are DEVICE
 * arrays of n_rings (<= 32) int64 -- the rings' write pointers (read and advanced), capacities, and a running count of rows stored
 * per ring that the host folds into its own pointers when it next looks (rollout.py TransitionSink.fold_counters); `slot_ws` is a
 * DEVICE workspace of at least sgrl_ingest_ws_words(n_rows) int64 (a ticket, the write pointers before the block, per-chunk
 * counts, row keys, the rows' slots); its contents on entry do not matter (the call zeroes its ticket on the stream), but two
 * calls in flight on different streams need a workspace each.  Library version 0.1.0 took an n_rows-word buffer here.  Same rings,
 * bit for bit, as sgrl_ingest_rows with the caller's slots (row order = the reference's `for i in range(num_envs)` order,
 * common/buffer.py:75-84 per row). */
int64_t sgrl_ingest_ws_words(int n_rows);
int sgrl_ingest_block(const float* block, int n_rows, int obs_len, int act_len, const sgrl_ring* rings, int n_rings, int64_t* pos,
                      const int64_t* cap, int64_t* pending, int64_t* slot_ws, void* stream);

const char* sgrl_last_error(void);
const char* sgrl_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SGRL_H */
