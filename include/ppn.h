/*
 * ppn.h -- C ABI of the MI355X-native pypownet load-flow step engine (libppn.so).
 *
 * Drop-in boundary.  The reference has no FFI of its own: its load-flow "backend" is the
 * Python call  pypower.api.runpf(mpc, ppopt, '', '') / rundcpf(...)  selected by the `loadflow_backend`
 * string (reference pypownet/grid.py:48-66, 212-242) and the game rules around it live in
 * pypownet/game.py.  This header is what a third `loadflow_backend` value ("hip") binds through ctypes;
 * each entry point names the reference interface it replaces.  Plain pointers and sizes only; no Python
 * or torch types cross this boundary; no exceptions: every call returns 0 or a negative PPN_E_* code and
 * ppn_last_error() gives the text.
 *
 * Ownership / threading: the engine owns all device memory; the caller owns every host buffer; one engine
 * per GPU; calls are NOT thread-safe and are stream-ordered on the engine's HIP stream
 * (ppn_read(..., to_host=1) and ppn_sync() synchronise).
 *
 * Index conventions: a grid has nS substations, each with two busbars ("nodes" 0/1).  Bus ROW r of the
 * reference's doubled bus table is (substation r % nS, node r / nS), i.e. row i+nS is the '666'-twin of
 * row i (reference pypownet/__init__.py:10).  Elements (productions, loads, line origins/extremities) are
 * attached to (substation, node).
 */
#ifndef PPN_H
#define PPN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ppn_engine ppn_engine;

/* error codes */
enum {
  PPN_OK = 0,
  PPN_E_INVALID = -1,   /* bad argument */
  PPN_E_HIP = -2,       /* HIP runtime error (text in ppn_last_error) */
  PPN_E_NODEVICE = -3,  /* no usable GPU */
  PPN_E_CAPACITY = -4,  /* case does not fit the LDS capacities requested */
  PPN_E_STATE = -5      /* call not valid in the current state (e.g. no chronic loaded) */
};

/* per-environment game flag (ppn_field PPN_F_FLAG): mirrors the 4th element of RunEnv.step()'s tuple
 * (reference pypownet/game.py:856-883, environment.py:893-907). */
enum {
  PPN_FLAG_OK = 0,
  PPN_FLAG_DIVERGED = 1,          /* DivergingLoadflowException */
  PPN_FLAG_TOO_MANY_LOADS = 2,    /* TooManyConsumptionsCut (tested first, game.py:868) */
  PPN_FLAG_TOO_MANY_PRODS = 3,    /* TooManyProductionsCut */
  PPN_FLAG_ENGINE_CAPACITY = 4    /* engine error: active buses / LU fill exceed the configured capacity */
};
/* illegal-action bits (PPN_F_ILLEGAL): IllegalActionException contents, game.py:650-753 */
enum {
  PPN_ILL_TOO_MANY = 1, PPN_ILL_BROKEN_LINE = 2, PPN_ILL_LINE_COOLDOWN = 4, PPN_ILL_NODE_COOLDOWN = 8
};

enum { PPN_MODE_AC = 0, PPN_MODE_DC = 1 };
enum { PPN_SOLVER_NEWTON = 1, PPN_SOLVER_FDXB = 2 };     /* = PYPOWER PF_ALG values (grid.py:63) */
enum { PPN_LOOP_NATURAL = 0, PPN_LOOP_FIXED = 1, PPN_LOOP_RANDOM = 2 };   /* chronic looping (chronic.py:266-291) */
/* PPN_LOOP_RANDOM: the reference draws the next chronic with np.random.choice on the host (chronic.py:283-289).  Here every
 * environment draws it on the device from a counter-based generator: slot = ppn_mix32(rng_seed, env, draw) % n_slots, draw =
 * number of chronics this environment has drawn so far (uniform like the reference's, reproducible under rng_seed, not the
 * same stream as numpy's). */
/* what happened to a line during the last step (PPN_F_LINE_EVENTS): the events the reference logs line by line,
 * game.py:436-469 (maintenance / hazards), 541-578 (overflow cuts), 627-648 (agent switches) */
enum { PPN_EV_SWITCHED = 1, PPN_EV_MAINTENANCE = 2, PPN_EV_HAZARD = 4, PPN_EV_HARD_OVERFLOW = 8, PPN_EV_SOFT_OVERFLOW = 16 };
/* outcome of the last load-flow solve of the step's cascade (PPN_F_SOLVE_OUTCOME): which text the reference's
 * DivergingLoadflowException carries, grid.py:231, 238 ('The grid is not connexe') vs grid.py:264 ('Power grid outage') */
enum { PPN_SOLVE_CONVERGED = 0, PPN_SOLVE_OUTAGE = 1, PPN_SOLVE_NOT_CONNEXE = 2, PPN_SOLVE_CAPACITY = 4 };

/* The case: MATPOWER-v2 arrays exactly as `loadcase` returns them (reference grid.py:65), float64 row-major.
 * bus: [2*nS x bus_cols>=13], gen: [nP x gen_cols>=8], branch: [nl x branch_cols>=11]. */
typedef struct ppn_case {
  int32_t n_bus_rows, bus_cols;
  int32_t n_gen, gen_cols;
  int32_t n_branch, branch_cols;
  double base_mva;
  const double* bus;
  const double* gen;
  const double* branch;
} ppn_case;

/* The 17 scalars of configuration.yaml (reference pypownet/parameters.py:89-153) + solver knobs. */
typedef struct ppn_rules {
  int32_t mode;                       /* loadflow_mode: PPN_MODE_AC | PPN_MODE_DC */
  int32_t solver;                     /* PPN_SOLVER_NEWTON | PPN_SOLVER_FDXB */
  double tol;                         /* PF_TOL (reference: 1e-6) */
  int32_t max_it;                     /* PF_MAX_IT (10) / PF_MAX_IT_FD (25) */
  double hard_overflow_coefficient;
  int32_t n_timesteps_hard_overflow_is_broken;
  double n_timesteps_consecutive_soft_overflow_breaks;   /* double: 1e12 when overflow cut-off is disabled */
  int32_t n_timesteps_soft_overflow_is_broken;
  int32_t n_timesteps_horizon_maintenance;
  int32_t max_number_prods_game_over;
  int32_t max_number_loads_game_over;
  int32_t n_timesteps_actionned_line_reactionable;
  int32_t n_timesteps_actionned_node_reactionable;
  int32_t max_number_actionned_substations;
  int32_t max_number_actionned_lines;
  int32_t max_number_actionned_total;
  int32_t game_over_mode_hard;        /* RunEnv(game_over_mode='hard'): next chronic after a game over */
  int32_t chronic_looping;            /* PPN_LOOP_* */
  /* engine capacities (0 = safe default: every busbar may be active).  max_active_buses bounds the busbars with a line
   * attached in any one topology (at least the number of substations; at most 254) and sizes the LDS working set; a
   * topology that exceeds a capacity reports PPN_FLAG_ENGINE_CAPACITY for that environment. */
  int32_t max_active_buses;
  int32_t lu_capacity;                /* capacity of the filled pattern, as doubles of uniform 2 x 2 blocks (4 per entry), 0 = auto */
  int32_t rng_seed;                   /* PPN_LOOP_RANDOM: seed of the per-environment chronic draws */
  /* Q plane of the Newton storage (rows of the Q equations: one per PQ bus).  0 (default): never short -- the exact bound over
   * every row of every loaded chronic when no spare busbar exists (max_active_buses = number of substations), the full plane
   * otherwise.  1: with spare busbars, reserve the chronic-derived PQ share + 12 % of the pattern capacity only (one LDS
   * granule class less: more environments per CU); a topology that splits many production substations at once may then need
   * more rows than reserved and reports PPN_FLAG_ENGINE_CAPACITY for that environment (PPN_SOLVE_CAPACITY). */
  int32_t q_plane_auto;
} ppn_rules;

/* One chronic as parsed by the reference reader (pypownet/chronic.py:173-229): float32 [T x n] row-major,
 * planned series already shifted by one row, T = zip-truncated length.  ids: int32[T].  dates: int32[T x 6]
 * (year, month, day, hour, minute, second). */
typedef struct ppn_chronic {
  int32_t T;
  const float* prods_p;          /* [T x nP] */
  const float* prods_v;          /* [T x nP], kV, <=0 => production off */
  const float* loads_p;          /* [T x nL] */
  const float* loads_q;
  const float* prods_p_planned;
  const float* prods_v_planned;
  const float* loads_p_planned;
  const float* loads_q_planned;
  const float* maintenance;      /* [T x nl] */
  const float* hazards;          /* [T x nl] */
  const int32_t* ids;            /* [T] */
  const int32_t* dates;          /* [T x 6] */
} ppn_chronic;

/* Fields readable (and, where noted W, writable) per environment; shapes are [batch x n]. */
typedef enum ppn_field {
  PPN_F_VM = 0,            /* f64 [2nS]  bus voltage magnitude p.u. per bus row            (W) */
  PPN_F_VA,                /* f64 [2nS]  bus voltage angle, DEGREES (MATPOWER convention)  (W) */
  PPN_F_PG,                /* f64 [nP]   gen[:,PG]                                         (W) */
  PPN_F_QG,                /* f64 [nP]   gen[:,QG]                                         (W) */
  PPN_F_VG,                /* f64 [nP]   gen[:,VG] (0 => production off)                   (W) */
  PPN_F_PD,                /* f64 [nL]   load active power                                 (W) */
  PPN_F_QD,                /* f64 [nL]                                                     (W) */
  PPN_F_PF, PPN_F_QF, PPN_F_PT, PPN_F_QT,   /* f64 [nl] branch[:,13:17]                        */
  PPN_F_AMPS,              /* f64 [nl]   extract_flows_a (grid.py:112-138)                     */
  PPN_F_PRODS_NODES,       /* u8 [nP]                                                      (W) */
  PPN_F_LOADS_NODES,       /* u8 [nL]                                                      (W) */
  PPN_F_LINES_OR_NODES,    /* u8 [nl]                                                      (W) */
  PPN_F_LINES_EX_NODES,    /* u8 [nl]                                                      (W) */
  PPN_F_LINES_STATUS,      /* u8 [nl]                                                      (W) */
  PPN_F_RECONNECTABLE,     /* i32 [nl]  timesteps_before_lines_reconnectable               (W) */
  PPN_F_LINE_COOLDOWN,     /* i32 [nl]  timesteps_before_lines_reactionable                (W) */
  PPN_F_NODE_COOLDOWN,     /* i32 [nS]  timesteps_before_nodes_reactionable                (W) */
  PPN_F_SOFT_COUNT,        /* i32 [nl]  n_timesteps_soft_overflowed_lines                  (W) */
  PPN_F_DONE,              /* u8 [1]                                                           */
  PPN_F_FLAG,              /* i32 [1]   PPN_FLAG_*                                             */
  PPN_F_ILLEGAL,           /* i32 [1]   PPN_ILL_* bits of the last step                        */
  PPN_F_CASCADE_DEPTH,     /* i32 [1]   depth reached by the cascade of the last step          */
  PPN_F_N_SOLVES,          /* i32 [1]   cumulative number of load-flow solves                  */
  PPN_F_N_ITERS,           /* i32 [1]   cumulative solver iterations (NR its / FD half-its)     */
  PPN_F_CHRONIC_SLOT,      /* i32 [1]                                                          */
  PPN_F_CHRONIC_ROW,       /* i32 [1]   row index of the current timestep in its chronic       */
  PPN_F_N_LOADS_CUT,       /* i32 [1]                                                          */
  PPN_F_N_PRODS_CUT,       /* i32 [1]                                                          */
  PPN_F_SUCCESS,           /* u8 [1]    success flag of the last solve                         */
  PPN_F_OBSERVATION,       /* f64 [obs_len] Observation.as_array() (environment.py:583-595)    */
  PPN_F_BUS_TYPE,          /* u8 [2nS]  bus type of the last solve: 1 PQ, 2 PV, 3 REF, 4 isolated */
  PPN_F_REWARD,            /* f64 [5]   reward of the last step: [loads cut, productions cut, action cost (+ illegal-action
                                        penalties), distance to the initial topology, line usage] (ppn_set_reward)      */
  PPN_F_ILLEGAL_COUNTS,    /* i32 [3]   IllegalActionException contents as counts: broken-line reconnections, on-cooldown
                                        line switches, on-cooldown substations (game.py:650-753)                         */
  PPN_F_ACTION_SWITCHES,   /* i32 [2]   node switches, line-status switches of the action as the caller sees it after the
                                        step: repaired in place, zeroed when the whole action was rejected (game.py:813)  */
  PPN_F_LINE_EVENTS,       /* u8 [nl]   PPN_EV_* bits: what happened to each line during the last step (the restart of an
                                        episode that ended is not part of it)                                            */
  PPN_F_SOLVE_OUTCOME,     /* i32 [1]   PPN_SOLVE_*: outcome of the last solve of the last step's cascade                 */
  PPN_F_N_STEPS,           /* i32 [1]   Game.step calls this environment has EXECUTED since ppn_reset: an environment that is over
                                        and waits for its restart does not step (throughput = sum of these / time)         */
  PPN_F_RETURN,            /* f64 [1]   sum of the five reward components over the steps executed since ppn_reset (PPN_F_N_STEPS
                                        of them): what a Runner accumulates as cumulative reward (runner.py:120-127)        */
  PPN_F_DEAD,              /* u8 [1]    0 playing; 1 over: the next step skips it until it is restarted; 2 over, its restart is
                                        owed by the next ppn_step(auto_reset = 2) (never seen after ppn_sync); 3 over, and
                                        PPN_RESTART_ATTEMPTS restarts in a row diverged as well (see ppn_process_game_over) */
  PPN_F_EPOCH,             /* i32 [1]   Game.epoch (game.py:333, 767): 1 after ppn_reset, + 1 for EVERY restart attempt of
                                        process_game_over -- the reference increments it on each recursive call too            */
  PPN_F_STEP_REPORT,       /* f64 [3]   (done, flag, sum of the five reward components) of the last step in ONE row: what a single
                                        controller gathers from every shard per step (<= 24 B per environment) -- one
                                        stream-ordered device copy instead of three reads + a pack (libppn 0.2)                  */
  PPN_F_COUNT
} ppn_field;

/* ---- lifetime ------------------------------------------------------------------------------------ */
/* Replaces Grid.__init__ + Game.__init__ parameter plumbing (grid.py:40-95, game.py:255-340). */
int ppn_create(const ppn_case* c, const ppn_rules* r, int32_t batch, int32_t device, ppn_engine** out);
int ppn_destroy(ppn_engine* e);
const char* ppn_last_error(const ppn_engine* e);     /* e may be NULL: error of the last failed ppn_create */

/* Coefficients of the five-component reward the reference's shipped environments use
 * (CustomRewardSignal, parameters/default14/reward_signal.py:8-43; default118: the same with constant = 118).
 * ppn_create installs the default14 values scaled by `constant = number of substations`; custom reward classes stay on
 * the host (they see the Observation), this removes the per-step download of the observation for the shipped one. */
typedef struct ppn_reward_params {
  double line_usage;                   /* x sum((ampere / limit)^2)                      reward_signal.py:15, 110-111 */
  double distance_initial_grid;        /* x number of elements not on their initial node reward_signal.py:17, 103-104 */
  double number_loads_cut;             /* x isolated loads                               reward_signal.py:19, 95-96   */
  double number_prods_cut;             /* x isolated productions                         reward_signal.py:20, 99-100  */
  double loadflow_exception;           /* DivergingLoadflowException -> component 3      reward_signal.py:25, 50      */
  double illegal_broken_line_switch;   /* per illegal reconnection of a broken line      reward_signal.py:29, 62-66   */
  double illegal_oncooldown_line_switch;        /*                                       reward_signal.py:30, 68-73   */
  double illegal_oncooldown_substation_switch;  /*                                       reward_signal.py:31, 75-81   */
  double too_many_productions_cut;     /* TooManyProductionsCut -> component 1           reward_signal.py:34, 89      */
  double too_many_consumptions_cut;    /* TooManyConsumptionsCut -> component 0          reward_signal.py:35, 91      */
  double too_much_activated_elements;  /* action beyond the activation maxima            reward_signal.py:39, 58-59   */
  double number_line_switches;         /* action cost per line-status switch             reward_signal.py:42, 137-138 */
  double number_node_switches;         /* action cost per node switch                    reward_signal.py:43          */
} ppn_reward_params;
/* RewardSignal.compute_reward on the device (RunEnv.step computes it from the observation, environment.py:866-874). */
int ppn_set_reward(ppn_engine* e, const ppn_reward_params* p);

/* ---- data ---------------------------------------------------------------------------------------- */
/* Thermal limits in A (reference: Chronic.get_imaps() of the FIRST chronic, game.py:301-304). */
int ppn_set_thermal_limits(ppn_engine* e, const double* limits /* [nl] */);
/* Upload chronic `slot` (0..n_slots-1; slots must be loaded in order). Replaces Chronic.__init__. */
int ppn_load_chronic(ppn_engine* e, int32_t slot, const ppn_chronic* c);
/* RESTART MEMO (libppn 0.3, round 6; off unless enabled here or with PPN_RESTART_MEMO=1).  Game.process_game_over of an episode that
 * ended at chronic position (chronic, timestep) -- reset_grid, the next timestep, the cascade from the flat start, again while the
 * restarted grid diverges too (pypownet/game.py:762-797) -- does not depend on anything else the ended episode left, except the
 * soft-overflow counters reset_grid does not clear (survey quirk q3).  With the memo on, the first restart from a position is
 * computed and kept (a snapshot of the environment's rows, ~18 KB on IEEE-118), every later restart from the same position of an
 * environment whose counters are all below n_timesteps_consecutive_soft_overflow_breaks copies it; the cumulative solve / iteration
 * counters (PPN_F_N_SOLVES, PPN_F_N_ITERS) and the epoch move by what the computed restart added, so every field reads as if the
 * restart had been computed (tests: check_restart_memo, bit for bit against an engine without it, and the oracle lock-steps run
 * with PPN_RESTART_MEMO=1).  Served where ppn_step restarts with auto_reset = 2 (the deferred restart) and inside ppn_step_observe,
 * ppn_rollout_policy and the step server of an asynchronous session (ppn_step_observe also saves what it computes; the other two
 * only serve); ppn_rollout(auto_reset = 1) plays through the same work-queue kernel and serves too; ppn_step(auto_reset = 1) computes every restart.  Not used with PPN_LOOP_RANDOM.  max_bytes: memory the snapshots may take (<= 0: 1 GiB); snapshots are dropped when chronics or thermal limits
 * change.  bench.py's headline keeps it OFF: every restart of the timed region is a computed one, as the reference's is. */
int ppn_restart_memo(ppn_engine* e, int32_t enable, int64_t max_bytes);
/* 0 snapshots held, 1 restarts served from a snapshot, 2 restarts that were not eligible (a soft-overflow counter at its threshold),
 * 3 snapshot capacity, 4 bytes per snapshot; -1 when the memo is off */
int64_t ppn_restart_memo_stat(ppn_engine* e, int32_t which);

/* ---- game ---------------------------------------------------------------------------------------- */
/* Game.__init__ tail for the listed environments (env_ids NULL = all): initial topology, case voltages,
 * chronic slot / first timestep row t0 (t0[i] = row loaded first; reference always 0), then one cascade
 * solve (game.py:339-340). */
int ppn_reset(ppn_engine* e, const int32_t* env_ids, int32_t n, const int32_t* chronic_slot, const int32_t* t0);
/* Game.step for every environment that is not done (game.py:799-885).  actions: u8 [batch x action_len],
 * host pointer (actions_on_device=0) or device pointer (1).  simulate!=0: Game.simulate (game.py:887-943):
 * the step runs on a scratch copy of the state and results are read with ppn_read(..., from_simulation=1).
 * auto_reset = 1: environments that end the step done are passed through process_game_over (game.py:762-780)
 * in the same call; PPN_F_DONE/FLAG still report the step's outcome.
 * auto_reset = 2: the same outcome, scheduled differently -- the restart of an environment that ends is DEFERRED to the next
 * ppn_step(auto_reset = 2) launch, where it runs right before that environment's step (the restart then is the head of a short
 * chain, the first step of a fresh episode, instead of the tail of the longest chain of the launch, a cascade that ended in a
 * diverging solve).  Anything that looks at the state in between (ppn_sync, ppn_read of a non-report field, ppn_write,
 * ppn_read_observation, a step in another mode, ...) settles the owed restarts first, so callers observe exactly what
 * auto_reset = 1 would have shown them.  Report fields (read without settling: a restart does not touch them): PPN_F_DONE, FLAG,
 * ILLEGAL, ILLEGAL_COUNTS, ACTION_SWITCHES, REWARD, CASCADE_DEPTH, LINE_EVENTS, SOLVE_OUTCOME, N_STEPS, RETURN. */
int ppn_step(ppn_engine* e, const uint8_t* actions, int32_t actions_on_device, int32_t simulate,
             int32_t auto_reset);
/* RunEnv.step as the reference returns it (environment.py:848-874: the step AND the observation): ppn_step followed by
 * ppn_read_observation(layout, as_f32, ..., to_host = 0) in ONE launch -- every environment's workgroup writes its row of
 * Observation.as_array() (layouts and element types of ppn_read_observation) into obs_device right behind its step, so the gather
 * costs no launch of its own and rides in the part of the launch where most of the machine waits for the longest cascade.  Same
 * rows, bit for bit, as the two calls.  obs_device: DEVICE memory, bytes >= batch x ppn_observation_length(layout) x element size.
 * auto_reset 0 or 1 (1: the row of an environment that ended shows the restarted episode, as after ppn_step(auto_reset = 1));
 * a deferred restart (2) is refused: PPN_E_INVALID. */
int ppn_step_observe(ppn_engine* e, const uint8_t* actions, int32_t actions_on_device, int32_t auto_reset,
                     int32_t layout, int32_t as_f32, void* obs_device, size_t bytes);
/* Open-loop rollout: n_steps consecutive Game.step calls of every environment in ONE launch, for callers whose actions do not
 * depend on the observations in between -- the reference's Runner.loop under a DoNothing agent (pypownet/runner.py:105-131,
 * agent.py:40-57), a recorded action file replayed (agent.py ActIOnManager), a planned switching sequence being evaluated.
 * actions: u8 [n_steps x batch x action_len] (per_step_actions = 1: environment b plays actions[s][b] at step s) or
 * [batch x action_len] (per_step_actions = 0: the same matrix at every step -- only meaningful for the do-nothing action, a
 * switch replayed every step toggles back and forth).  Same outcome, bit for bit, as n_steps calls of
 * ppn_step(actions[s], ..., auto_reset): the report fields (PPN_F_DONE, FLAG, REWARD ...) are those of the LAST step,
 * PPN_F_N_STEPS / PPN_F_RETURN accumulate over all of them.  What differs is the schedule: no environment waits for another
 * one between its steps, so a launch no longer lasts n_steps times its longest cascade (measurement).
 * auto_reset as in ppn_step (0: an environment that ends sits out the remaining steps).
 * ONE exception to "same outcome": an environment that is ALREADY over when the rollout starts (right after ppn_reset on a
 * grid that diverges, or after steps with auto_reset = 0) is restarted BEFORE its first step and plays all n_steps steps;
 * under n_steps calls of ppn_step(auto_reset != 0) it sits out the first call and is restarted in that call's post-pass
 * (n_steps - 1 steps).  Call ppn_process_game_over first and the two agree again (tests: check_rollout_equals_steps does,
 * check_rollout_dead_at_start pins the exception).
 * Since round 6 auto_reset = 1 plays through ppn_rollout_policy's work-queue kernel (items (step, environment) handed to whichever
 * workgroup is free, XCD-affine) with the action rows in place of a policy: same trajectories, and the launch no longer ends with
 * the environment whose steps add up to the longest chain (19.5 M env-steps/s against 15.8 M on the bench workload). */
int ppn_rollout(ppn_engine* e, const uint8_t* actions, int32_t actions_on_device, int32_t n_steps,
                int32_t per_step_actions, int32_t auto_reset);
/* ---- closed-loop stepping without the batch barrier: policies that live on the device (libppn 0.2) -----------------------------
 * A closed-loop agent decides from what the last step left.  When the decision rule itself runs on the device, no environment has
 * to wait for the rest of the batch between its steps: ppn_rollout_policy plays n_steps closed-loop steps of every environment in
 * ONE launch -- per step: the policy turns the environment's own state into an action, then Game.step (fused auto-reset: an
 * episode that ends is restarted right behind its last step, so the next decision sees the fresh episode, as an agent behind
 * RunEnv.step / process_game_over does, runner.py:81-96).  The steps are handed out one at a time, in environment order, to as
 * many workgroups as the GPU holds: an environment's step s + 1 goes to whichever workgroup is free once its step s is complete.
 * Same trajectories, bit for bit, as n_steps rounds of { ppn_policy_actions; ppn_step(actions, on device, auto_reset = 1) }
 * (tests: check_policy_rollout_equals_stepping); what differs is the schedule -- a launch of 4096 environments no longer lasts
 * n_steps times its longest cascade.  The built-in policies (a user policy is one more `case` in policy_action, ppn_game.inc):
 *   PPN_POLICY_DO_NOTHING    the reference's DoNothing agent (agent.py:40-57)
 *   PPN_POLICY_LINE_RELIEF   a toy operator: reconnect the first line that is out, reconnectable (timesteps_before_lines_
 *                            reconnectable = 0) and not cooling down; else, if the most loaded line carries more than
 *                            params[0] x its thermal limit and may be actioned, open it; else do nothing.  One line switch at
 *                            most: legal by construction.
 * ppn_policy_actions writes the policy's choice for the CURRENT state of every environment into a caller-owned DEVICE buffer
 * u8 [batch x action_len] (restarts owed by a deferred auto-reset are settled first: the policy looks at the state). */
#define PPN_POLICY_DO_NOTHING 0
#define PPN_POLICY_LINE_RELIEF 1
int ppn_policy_actions(ppn_engine* e, int32_t policy, const double* params, int32_t n_params, uint8_t* actions_out_device);
/* (ppn_rollout_policy's kernel is the one whose first version faulted on the GPU only -- tools/ubench/README_gpu_only_failures.md: the
 * cause is a code-generation problem that was pinned, not found.  Its gates: test_gpu_policy_rollout_equals_stepping in the `-m gpu`
 * suite and tools/ubench/gpu_only_failure_repro.sh; run both after any change of compiler or of the kernels' loop structure.) */
int ppn_rollout_policy(ppn_engine* e, int32_t policy, const double* params, int32_t n_params, int32_t n_steps);
/* ---- asynchronous stepping for EXTERNAL policies: send / recv (libppn 0.3) ------------------------------------------------------
 * The reference's consumers are agents that LOOK at the observation and then act (pypownet/runner.py:72-103: obs -> agent.act ->
 * env.step; agent.py:268-311).  Stepped synchronously, a batch advances at the pace of its slowest environment: every ppn_step lasts
 * as long as the longest cascade of the batch (one environment in 500 runs 20 Newton iterations where the median runs 3).  An
 * asynchronous session removes that barrier for policies that live OUTSIDE the engine (a torch module, a host program):
 *
 *     ppn_async_start(e, &cfg);                         a step server becomes resident on the GPU
 *     ppn_send(e, all_env_ids, batch, actions, ...);    every environment gets its first action
 *     loop:  ppn_recv(e, min_ready, ...) -> the ids of >= min_ready environments whose step is complete (their report rows are in the
 *                                            caller's device buffer, their observation rows are being gathered on the session's
 *                                            stream); stragglers keep running
 *            policy(observation rows of those ids) -> actions
 *            ppn_send(e, those ids, n, actions, ...)
 *     ppn_async_stop(e);
 *
 * Every environment is on its own clock; per environment the trajectory is bit for bit that of ppn_step_observe(auto_reset = 1)
 * called with the same actions in the same order (tests: check_async_equals_stepping) -- environments never interact.
 *   step semantics   Game.step with the fused restart (auto_reset = 1): the observation row of an environment whose episode ended
 *                    shows the restarted episode, its report row (done, flag, reward sum) the step that ended it -- what an agent
 *                    behind RunEnv.step / process_game_over sees (runner.py:81-96).
 *   ownership        obs_device / report_device are DEVICE buffers of the caller, [batch x row]: report row `env` is rewritten by
 *                    every step of environment `env`; observation row `env` is written by a gather kernel that the ppn_recv which
 *                    returns `env` queues on ppn_async_stream(e) -- complete for everything queued on that stream afterwards (the
 *                    policy), a reader elsewhere synchronises with that stream first.  Both are stable until the ppn_send that
 *                    sends `env` again.  An environment may be in flight once: sending it again before it was received is PPN_E_STATE.
 *   streams          the server runs on a stream of its own; ppn_send's work is queued on ppn_async_stream(e) (a non-blocking HIP
 *                    stream): device-side action / id buffers handed to ppn_send must be complete ON THAT STREAM (run the policy
 *                    on it, or make it wait for the policy's event) and must stay untouched until the enqueue kernel has read
 *                    them (stream order again).  ppn_recv is a host-side wait on a completion ring in pinned memory: it
 *                    synchronises nothing.  Anything that waits for the WHOLE device inside a session (hipDeviceSynchronize /
 *                    torch.cuda.synchronize, hipFree, the first launch of a kernel whose module is not loaded yet) waits for the
 *                    resident server, i.e. for its idle timeout: correct, but it costs that long -- warm the policy up before
 *                    the session and keep such calls out of the loop.
 *   other calls      any other entry point of this header called during a session first SETTLES it: waits for the steps in flight,
 *                    stops the server, does its work on a quiet engine; completions not yet received stay receivable and the next
 *                    ppn_send starts the server again.  (Correct, not fast: keep reads of state out of the loop.)
 *   liveness         the server leaves by itself when nothing has been published for cfg.idle_timeout_ms (default 100): a host that
 *                    died leaves no kernel spinning.  A host that was merely slow loses nothing: the next ppn_send / ppn_recv finds
 *                    the server gone, re-publishes the steps it had not started and launches it again. */
typedef struct ppn_async_config {
  int32_t struct_size;        /* = sizeof(ppn_async_config) */
  int32_t layout;             /* observation layout as in ppn_read_observation: 0 full, 1 minimalist, 2 AC minimalist */
  int32_t as_f32;             /* rows as float32 instead of float64 */
  int32_t workgroups;         /* resident server workgroups; 0 = four per CU (every SIMD keeps half its registers: room for the policy's multi-wave kernels) */
  int32_t idle_timeout_ms;    /* 0 = 100 */
  int32_t reserved;
  void* obs_device;           /* [batch x ppn_observation_length(layout)] rows, or NULL: no observation is written */
  size_t obs_bytes;
  double* report_device;      /* [batch x 3] (done, flag, reward sum) rows = PPN_F_STEP_REPORT, or NULL */
} ppn_async_config;
int ppn_async_start(ppn_engine* e, const ppn_async_config* cfg);
/* Publishes one step each for the n listed environments.  env_ids: HOST int32 [n] (what ppn_recv returned, or any order of one's own).
 * actions: u8 rows, host (actions_on_device = 0) or device memory; rows_by_env = 0: [n x action_len], row i belongs to env_ids[i];
 * rows_by_env = 1: a [batch x action_len] matrix indexed by environment (a policy that writes its choices in place). */
int ppn_send(ppn_engine* e, const int32_t* env_ids, int32_t n, const uint8_t* actions, int32_t actions_on_device, int32_t rows_by_env);
/* Waits until at least min(min_ready, steps in flight) steps are complete (timeout_ms < 0: for as long as it takes; on a timeout
 * fewer are returned, possibly none) and hands out at most max_n of them, in completion order: env_ids_host [max_n] receives the
 * environment indices, *n_out their number; env_ids_device (may be NULL) receives the same int32 values through an asynchronous copy
 * on ppn_async_stream(e) -- for a policy that gathers its observation rows on that stream. */
int ppn_recv(ppn_engine* e, int32_t min_ready, int32_t max_n, int32_t timeout_ms, int32_t* env_ids_host, int32_t* n_out,
             int32_t* env_ids_device);
/* Ends the session: waits for the steps in flight, stops the server.  Completions not yet received are dropped. */
int ppn_async_stop(ppn_engine* e);
/* hipStream_t of ppn_send's device work (NULL outside a session). */
void* ppn_async_stream(ppn_engine* e);
/* 0 steps in flight (sent, not yet received), 1 resident server workgroups, 2 times the server had to be started again after it
 * had left on its idle timeout, 3 steps re-published by those restarts */
int64_t ppn_async_stat(const ppn_engine* e, int32_t which);

/* Topology-action search (the reference's search agents call RunEnv.simulate once per candidate,
 * pypownet/agent.py:161-325): candidate c forks the CURRENT state of environment env_ids[c] and plays
 * Game.simulate(actions[c]) on it (game.py:887-943); any number of candidates per environment, one kernel launch.
 * actions: u8 [n x action_len] (host or device), env_ids: host int32 [n].  Results are read with
 * ppn_read(..., from_simulation = 2): n rows per field, in candidate order (PPN_F_REWARD, PPN_F_FLAG, PPN_F_OBSERVATION...). */
int ppn_simulate_candidates(ppn_engine* e, const uint8_t* actions, int32_t actions_on_device, const int32_t* env_ids,
                            int32_t n);
/* Game.process_game_over for every environment whose done flag is set (game.py:762-797).  env_mask (host,
 * u8[batch], may be NULL) additionally forces the listed LIVE environments through it, as a caller of
 * RunEnv.process_game_over() may do at any time (reference tests/common_assets.py:51).
 * The reference repeats the restart for as long as the restarted grid diverges as well (game.py:776-780 recurses without a
 * bound, i.e. until Python's recursion limit ends the process).  One pass of the engine stops after PPN_RESTART_ATTEMPTS
 * and leaves such an environment over with PPN_F_DEAD = 3: call again to go on (pypownet_amd.game.Game does, up to the
 * reference's ~1000 frames), or keep stepping with auto_reset -- every ppn_step(auto_reset != 0) launch takes up the restart of
 * its DEAD = 3 environments again before their step.  Such an environment executes no step meanwhile (PPN_F_N_STEPS) and its
 * report fields (PPN_F_DONE = 1, FLAG, CASCADE_DEPTH, REWARD) stay those of the step that ended its episode: a consumer that
 * counts episode ends per launch masks them with "PPN_F_N_STEPS moved" (bench.py does) or with PPN_F_DEAD != 3. */
#define PPN_RESTART_ATTEMPTS 64
int ppn_process_game_over(ppn_engine* e, const uint8_t* env_mask);
/* Game.is_action_valid (game.py:755-760): valid[b] = 1/0. */
int ppn_is_action_valid(ppn_engine* e, const uint8_t* actions, uint8_t* valid);

/* ---- pure solve boundary --------------------------------------------------------------------------- */
/* Replaces runpf(mpc, ppopt, '', '') / rundcpf (grid.py:227-229) for the whole batch on the CURRENT state
 * (bus types are re-derived as Grid._synchronize_bus_types does).  Results: PPN_F_VM/VA/PG/QG/PF../SUCCESS. */
int ppn_runpf_batch(ppn_engine* e);

/* The same seam in the reference's own currency: MATPOWER-format arrays in, arrays out --
 *     output, success = runpf(self.mpc, self.loadflow_options, '', '')          (pypownet/grid.py:226-229; rundcpf when
 *                                                                                 rules.mode = PPN_MODE_DC)
 * for n <= batch independent mpc's in one launch.  Every mpc is the case the engine was created for in one of its states:
 *   bus    [2nS x bus_cols]     rows in case order (row i + nS = the '666'-twin of row i, checked).  Read: PD, QD (the load
 *                               of a substation sits on the row of the busbar it is wired to, the other row holds 0),
 *                               VM, VA (warm start; VA in degrees).  BUS_TYPE is NOT read: the reference derives it from
 *                               the topology right before every call (Grid._synchronize_bus_types, grid.py:141-176,
 *                               called at grid.py:252) and so does the engine -- the derived types are written to bus_out.
 *   gen    [nP x gen_cols]      GEN_BUS = the substation id or its '666'-twin (which busbar the production is on), PG, QG,
 *                               VG, GEN_STATUS (<= 0: off).
 *   branch [nl x branch_cols]   F_BUS / T_BUS = substation ids or twins, BR_STATUS.  The electrical columns (r, x, b, tap,
 *                               shift) must be the case's -- they are checked, not read.
 * Output (caller-owned, same shapes; branch_out always has 17 columns like PYPOWER's padded result): copies of the input
 * with what runpf replaces -- bus VM / VA of the buses in service (+ the derived BUS_TYPE), gen PG / QG (0 for productions
 * that are off or isolated), branch PF QF PT QT (0 for lines out of service).  success[i] = runpf's second return value;
 * outcome[i] (may be NULL) = PPN_SOLVE_*: PPN_SOLVE_NOT_CONNEXE is the case where the reference's call raises and
 * Grid maps it to DivergingLoadflowException('The grid is not connexe') (grid.py:228-231) -- the outputs then are the inputs.
 * The state of environments 0..n-1 of the engine is overwritten (use an engine of its own for pure solves).
 * Host pointers only; the call synchronises. */
typedef struct ppn_mpc_batch {
  int32_t struct_size;                         /* = sizeof(ppn_mpc_batch) of the header the caller was built against: the library refuses
                                                  (PPN_E_INVALID) a struct of another layout instead of reading its fields shifted
                                                  (libppn 0.2; the 0.1 struct began with `n`) */
  int32_t n;
  int32_t bus_cols, gen_cols, branch_cols;     /* columns of the input arrays: >= 10 (MATPOWER writes 13), >= 8, >= 11 */
  int32_t bus_rows, gen_rows, branch_rows;     /* rows of ONE case in the arrays below: must be 2 nS, nP, nl of the engine's case --
                                                  an mpc with isolated buses dropped or without the '666'-twin rows is refused
                                                  (PPN_E_INVALID), not read past its end */
  const double* bus;                           /* [n x 2nS x bus_cols] */
  const double* gen;                           /* [n x nP x gen_cols] */
  const double* branch;                        /* [n x nl x branch_cols] */
  double* bus_out;                             /* [n x 2nS x bus_cols] */
  double* gen_out;                             /* [n x nP x gen_cols] */
  double* branch_out;                          /* [n x nl x 17] */
  uint8_t* success;                            /* [n] */
  int32_t* outcome;                            /* [n] or NULL */
} ppn_mpc_batch;
int ppn_runpf_arrays(ppn_engine* e, const ppn_mpc_batch* io);

/* ---- state access ---------------------------------------------------------------------------------- */
size_t ppn_field_bytes(const ppn_engine* e, ppn_field f);      /* Observation layouts of the reference (environment.py:406-531): layout 0 = Observation.as_array() (PPN_F_OBSERVATION),
 * 1 = MinimalistObservation.as_array(), 2 = MinimalistACObservation.as_array() -- both prefixes of the full array, gathered
 * directly at their own row stride; as_f32 != 0 emits float32 instead of float64 (3-4x fewer bytes
 * to move for policies that need no more).  dst: [rows x ppn_observation_length(layout)], rows as in ppn_read. */
int ppn_read_observation(ppn_engine* e, int32_t layout, int32_t as_f32, void* dst, size_t bytes, int32_t to_host,
                         int32_t from_simulation);
int32_t ppn_observation_length(const ppn_engine* e, int32_t layout);
/* bytes of one environment's field */
int ppn_read(ppn_engine* e, ppn_field f, void* dst, size_t bytes, int32_t to_host, int32_t from_simulation);
int ppn_write(ppn_engine* e, ppn_field f, const void* src, size_t bytes);   /* whole batch, host pointer */
int ppn_sync(ppn_engine* e);        /* settles the restarts a deferred auto-reset owes (ppn_step), then waits for the stream */
int ppn_wait(ppn_engine* e);        /* waits for the engine's stream only */
/* HIP stream the engine launches on (void* = hipStream_t), for callers that time with HIP events. */
void* ppn_stream(ppn_engine* e);
/* Average / last device time of the dominant kernel measured with HIP events on the engine stream.  Under two-capacity stepping
 * one "launch" is the whole step: schedule pre-pass + launch order + small-storage launch + large-storage launch, one bracket. */
int ppn_kernel_time(ppn_engine* e, int32_t reset, double* total_ms, int64_t* launches);

/* ---- introspection --------------------------------------------------------------------------------- */
int32_t ppn_dim(const ppn_engine* e, int32_t which);   /* 0 nS, 1 nP, 2 nL, 3 nl, 4 action_len, 5 obs_len,
                                                          6 batch, 7 lds_bytes, 8 max_active_buses,
                                                          9 lu_capacity, 10 n_chronic_slots, 11 base LU fill,
                                                          12-14 capacities of the elimination schedule (filled 2x2
                                                          block entries, Schur pair records, triple records), 15 Q-plane
                                                          capacity of the Newton storage, 16 environments resident per CU
                                                          (hipOccupancyMaxActiveBlocksPerMultiprocessor of the step kernel),
                                                          17 / 18 two-capacity stepping: pattern capacity and LDS bytes of the
                                                          small-storage launch (0: off -- see rules.lu_capacity),
                                                          19 kernel form the LAST step launch took: 0 one workgroup per environment
                                                          (K_STEP), 1 persistent (K_STEP_PERSIST), 2 step + observation (K_STEP_OBS),
                                                          3 rollout; + 4 when it was stepped in two capacity classes */
const char* ppn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PPN_H */
