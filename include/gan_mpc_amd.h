/*
 * gan_mpc_amd.h -- C ABI of libgan_mpc_amd.so: the MI355X (gfx950) implementation of the
 * GAN-MPC inner loop of returaj/gan_mpc.
 *
 * The reference has no FFI: its boundary is a Python object protocol.  Each entry
 * point below names the reference function(s) whose arithmetic it replaces (paths relative to the
 * reference repository root); the gan_mpc_amd Python package rebuilds the reference's Python protocol on top of
 * these calls through ctypes (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *  - Every buffer is caller-owned DEVICE memory, fp32, row-major, unless stated otherwise.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only enqueue
 *    work; nothing synchronises unless stated.
 *  - Return value: 0 on success, a negative GMPC_E* code on failure; gmpc_last_error() returns a
 *    thread-local description of the last failure.  No C++ exception crosses this ABI.
 *  - A gmpc_ctx belongs to one GPU and is thread-compatible (one caller at a time).
 *  - There is no CPU fallback: without a HIP device every compute entry point fails.
 *  - Ordering contract: gmpc_bilevel_grad / gmpc_upper_loss differentiate the solution the ctx holds
 *    after a COMPLETED gmpc_ilqr_solve of the same batch size.  gmpc_set_params, gmpc_rollout_cost,
 *    gmpc_lqr_backward(_after_rollout) and a failed or new gmpc_ilqr_solve overwrite parts of that
 *    state and therefore drop it: a later gmpc_bilevel_grad / gmpc_upper_loss fails with GMPC_EINVAL
 *    ("must precede") instead of differentiating a stale linearisation.
 *
 * Parameter layouts (flat fp32 vectors, flax Dense order: kernel (in,out) row-major, then bias):
 *   dyn    : for l in 0..dyn_layers-1:  W_l[dims[l]][dims[l+1]], b_l[dims[l+1]]
 *            dims[0] = n+m, dims[last] = n          (reference dynamics/nn.py:27-34)
 *   cost   : same, dims[0] = n, dims[last] = fout   (reference cost/nn.py:23-29)
 *   mpc_w  : 3 raw weights (action, state, terminal) (reference gan/runner.py:41, cost_model.py:37)
 *   critic : Wx[n][4F], Wh[F][4F], b[4F] (gate order i,f,g,o), then the head's Dense layers
 *            head_dims[0] = F, head_dims[last] = 1   (reference critic/nn.py:28-42)
 */
#ifndef GAN_MPC_AMD_H
#define GAN_MPC_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define GMPC_MAX_LAYERS 8
#define GMPC_PROF_SLOTS 10

enum {
  GMPC_OK = 0,
  GMPC_EINVAL = -1,      /* bad argument / unsupported shape */
  GMPC_ENODEV = -2,      /* no usable HIP device */
  GMPC_EHIP = -3,        /* a HIP runtime call failed */
  GMPC_ENOMEM = -4
};

typedef struct gmpc_ctx gmpc_ctx;

typedef struct gmpc_shape {
  int n;                /* state size (xc size; the MLP dynamics has an empty carry) */
  int m;                /* action size */
  int T;                /* horizon H */
  int dyn_layers;       /* number of Dense layers of the dynamics MLP */
  int dyn_dims[GMPC_MAX_LAYERS + 1];
  int cost_layers;
  int cost_dims[GMPC_MAX_LAYERS + 1];
  int lstm_features;    /* F */
  int head_layers;      /* Dense layers after the LSTM (>= 1, last one has 1 output) */
  int head_dims[GMPC_MAX_LAYERS + 1];
  /* LSTM dynamics variant (reference dynamics/nn.py:37-57; 0 = the MLP variant, the yaml default):
   * xc = [x (x_size), c (F), h (F)], n = x_size + 2F, the cell runs on [x, u] and dyn_dims describes the
   * relu tail h' -> x: dyn_dims[0] = F, dyn_dims[last] = x_size.  Goals, desired sequences, the critic and
   * the expert model keep x_size columns; the cost MLP takes the whole xc (cost_dims[0] = n).
   * dyn vector: Wx[(x_size+m)][4F] | Wh[F][4F] | b[4F] (gates i,f,g,o) | the tail's Dense layers. */
  int dyn_lstm_features;
  int x_size;           /* 0 or n for the MLP dynamics */
} gmpc_shape;

/* trajax iLQR keyword set, reference policy/eval.py:10-20 */
typedef struct gmpc_ilqr_opts {
  int maxiter;
  float grad_norm_threshold;
  float relative_grad_norm_threshold;
  float obj_step_threshold;
  float inputs_step_threshold;
  int make_psd;          /* must be 0 (the reference never sets it) */
  float psd_delta;
  float alpha_0;
  float alpha_min;
} gmpc_ilqr_opts;

const char* gmpc_last_error(void);
const char* gmpc_version(void);

/* Sizes of the flat parameter vectors for a shape: which = 0 dyn, 1 cost, 2 critic. */
long gmpc_param_count(const gmpc_shape* shape, int which);

/* Flat layout of a parameter vector, leaf by leaf, so that a caller in any language can pack a flax
 * checkpoint (replaces the pytree plumbing of policy/eval.py:56-62, gan/js_policy.py:32-39 and the
 * Python-only packing of gan_mpc_amd/params.py).  Element (r, c) of a leaf lives at
 * flat[offset + r * ld + c]; names are flax tree paths ("params/Dense_0/kernel", critic:
 * "params/ScanOptimizedLSTMCell_0/ii/kernel" ... -- the per-gate LSTM kernels are column blocks of the
 * concatenated Wx / Wh, hence ld = 4F there).
 *   which = 0 dyn, 1 cost, 2 critic (the vectors gmpc_set_params / the critic calls take),
 *           3 the host package's training vector [mpc_weights | cost_params | dynamics_params | critic_params]
 * Writes at most max_leaves entries (leaves may be NULL with max_leaves 0 to size the buffer) and
 * returns the number of leaves, or a negative GMPC_E* code. */
typedef struct gmpc_leaf {
  char name[64];
  long offset;
  int rows, cols, ld;
} gmpc_leaf;
int gmpc_pack_layout(const gmpc_shape* shape, int which, gmpc_leaf* leaves, int max_leaves);

/* One context per GPU.  Allocates the workspace for up to max_batch trajectories
 * (and 2*max_batch critic sequences). */
int gmpc_create(const gmpc_shape* shape, int max_batch, int device, gmpc_ctx** out);
int gmpc_destroy(gmpc_ctx* ctx);

/* Bind the model parameters used by the trajectory kernels (device pointers; they must stay valid
 * and unchanged until the next gmpc_set_params).  Builds the transposed weight copies the backward
 * chains read.  Replaces the params-dict unwrapping of policy/eval.py:64-73. */
int gmpc_set_params(gmpc_ctx* ctx, const float* mpc_w, const float* dyn, const float* cost,
                    void* stream);

/* a1-a4,a7: X = rollout(dynamics, U, x0); costs[t] = get_cost(X[t], pad(U)[t], t).
 * Replaces trajax rollout/evaluate as called at policy/optimizers.py:24-31 with
 * dynamics/nn.py:27-34, cost/cost_model.py:20-42, cost/nn.py:23-29.
 *   x0 [B][n], U [B][T][m], goal [B][T+1][n]  ->  X [B][T+1][n], costs [B][T+1]            */
int gmpc_rollout_cost(gmpc_ctx* ctx, int B, const float* x0, const float* U, const float* goal,
                      float* X, float* costs, void* stream);

/* a4 as the model protocol calls it (base.py:4-9, cost/cost_model.py:33-42 get_cost(xc, u, t, ...)): the
 * cost of B independent (x, u) pairs outside a rollout.  terminal == 0: the staging branch with
 * goal_row [B][n] = goal_X[t]; terminal != 0: the terminal branch w2 |MLP(x)|^2 of an arbitrary state
 * (u, goal_row may be NULL).  x [B][n], u [B][m] -> cost [B]. */
int gmpc_get_cost(gmpc_ctx* ctx, int B, const float* x, const float* u, const float* goal_row,
                  int terminal, float* cost, void* stream);

/* a1-a2 as the model protocol calls it (base.py:15-22, dynamics/dynamics_model.py:45-48 predict(xc, u, t,
 * params)): next_x [B][n] = dynamics(x, u) for B independent pairs (a horizon-1 pass of the rollout
 * kernel; drops a held iLQR solution, see the ordering contract). */
int gmpc_predict(gmpc_ctx* ctx, int B, const float* x, const float* u, float* next_x, void* stream);

/* One iLQR backward pass at an arbitrary trajectory (X, U): linearise the dynamics (the relu sign
 * masks are recomputed from (X, U)), quadratise the cost, run the time-varying LQR (Riccati)
 * recursion and the adjoint recursion.
 * Replaces trajax linearize/quadratize/tvlqr/adjoint inside trajax ilqr (policy/optimizers.py:19,55).
 *   -> K [B][T][m][n], k [B][T][m], grad [B][T][m], adjoints [B][T+1][n]; any output may be NULL.
 *   AB (optional) [B][T][n][n+m]: rows of [A_t | B_t] = d f/d[x,u]. */
int gmpc_lqr_backward(gmpc_ctx* ctx, int B, const float* X, const float* U, const float* goal,
                      float* K, float* k, float* grad, float* adjoints, float* AB, void* stream);

/* Same as gmpc_lqr_backward, but reuses the relu sign masks that the immediately preceding
 * gmpc_rollout_cost of this ctx produced for exactly this (X, U): the fused rollout + backward
 * "step" of one iLQR iteration (what bench.py times). */
int gmpc_lqr_backward_after_rollout(gmpc_ctx* ctx, int B, const float* X, const float* U,
                                    const float* goal, float* K, float* k, float* grad,
                                    float* adjoints, float* AB, void* stream);

/* a6: full iLQR solve (policy/optimizers.py:10-21 -> trajax ilqr), one independent solve per
 * trajectory (the reference's jax.vmap axis, policy/base.py:122-125).
 *   U_init [B][T][m] -> X [B][T+1][n], U [B][T][m], obj [B], grad [B][T][m], adjoints [B][T+1][n],
 *   iterations [B] (int32).  Synchronises the stream before returning. */
int gmpc_ilqr_solve(gmpc_ctx* ctx, int B, const float* x0, const float* U_init, const float* goal,
                    const gmpc_ilqr_opts* opts, float* X, float* U, float* obj, float* grad,
                    float* adjoints, int* iterations, void* stream);

/* a8-a11 (+a13/a16): upper-level loss and its bilevel gradient at the iLQR solution held by the ctx
 * after gmpc_ilqr_solve (policy/optimizers.py:61-73,78-105), per trajectory; the batch mean of
 * policy/base.py:126-127 is left to the caller (it is where the multi-GPU all-reduce goes).
 *   loss_kind 0: L2 (norm/l2_policy.py:12-18), desired [B][T+1][n]
 *   loss_kind 1: JS generator (gan/js_policy.py:60-68) with critic params `critic`
 *   sign: +1 reproduces the reference as written, -1 the implicit-function gradient.
 *   -> loss [B]; grad_sum [3 + cost_count]: SUM over the batch of d/d(mpc_w, cost params). */
int gmpc_bilevel_grad(gmpc_ctx* ctx, int B, int loss_kind, const float* desired,
                      const float* critic, float sign, float* loss, float* grad_sum, void* stream);

/* a13/a16 only: the upper-level loss [B] at the solution held by the ctx, without the gradient
 * (test-loss evaluation, norm/cost_trainer.py:13-21). */
int gmpc_upper_loss(gmpc_ctx* ctx, int B, int loss_kind, const float* desired, const float* critic,
                    float* loss, void* stream);

/* a19: Polyak blend out = factor*prev + (1-factor)*cur over a flat range (norm/cost_trainer.py:88-92).
 * out may alias prev or cur. */
int gmpc_polyak(gmpc_ctx* ctx, long count, const float* prev, const float* cur, double factor,
                float* out, void* stream);

/* a14-a15: critic BCE loss and gradient (gan/js_policy.py:41-58, critic/nn.py:28-42).
 *   xseq [Bc][T+1][n], label [Bc] (+1 / -1), critic params
 *   -> loss_sum [1] (SUM over the batch of -log p), grad_sum [critic_count] (SUM over the batch). */
int gmpc_critic_loss_grad(gmpc_ctx* ctx, int Bc, const float* xseq, const float* label,
                          const float* critic, float* loss_sum, float* grad_sum, void* stream);

/* a14/a16: critic scores and (optionally) d score / d xseq.
 *   -> score [Bc]; dxseq [Bc][T+1][n] or NULL. */
int gmpc_critic_score_vjp(gmpc_ctx* ctx, int Bc, const float* xseq, const float* critic,
                          float* score, float* dxseq, void* stream);

/* a18: optax.chain(clip_by_global_norm(max_norm), adam(lr)) on one contiguous trainable range
 * (gan/runner.py:51-63).  grad is scaled by grad_scale first (1/B for a batch sum).
 * step = 1-based update count.  params, m, v [count] are updated in place.  The hyper-parameters
 * are doubles: 1-b1, 1-b2 and the bias corrections are formed in double and rounded once to fp32,
 * as optax does with its Python-float hyper-parameters. */
int gmpc_adam_clip_step(gmpc_ctx* ctx, long count, float* params, const float* grad, float* m,
                        float* v, float grad_scale, int step, double lr, double max_norm, double b1,
                        double b2, double eps, void* stream);

/* e: the ONE exchange of an optimiser step -- the batch mean at policy/base.py:126-127 (cost /
 * generator step) and gan/js_policy.py:55 (critic step) -- for callers without torch.distributed: an
 * in-place SUM over ranks of the packed fp32 buffer [loss_sum | grad_sum | sample count] with RCCL over
 * xGMI (ncclAllReduce, ncclFloat32, ncclSum), enqueued on `stream`; the caller then divides by the
 * reduced count (or folds 1/count into gmpc_adam_clip_step's grad_scale).  One process per GPU:
 * rank 0 calls gmpc_comm_unique_id and hands the 128 bytes to the other ranks by its own means (file,
 * socket, MPI), every rank calls gmpc_comm_init(ctx, world_size, rank, id) -- collective -- once.
 * A ctx without gmpc_comm_init is a world of one and gmpc_allreduce_grads is a no-op.  RCCL is bound
 * at run time (dlopen): inside a PyTorch process the copy torch loaded is used.  The Python package
 * uses torch.distributed (backend "nccl" = the same RCCL) for this exchange, gan_mpc_amd/parallel.py. */
int gmpc_comm_unique_id(char* id128);
int gmpc_comm_init(gmpc_ctx* ctx, int world_size, int rank, const char* id128);
int gmpc_comm_world(gmpc_ctx* ctx, int* world_size, int* rank);
int gmpc_allreduce_grads(gmpc_ctx* ctx, float* packed, long count, void* stream);

/* N2: the expert sequence model that produces goal_xseq / init_useq for every solve
 * (policy/eval.py:87-107 get_goal_states_init_actions; expert/expert_model.py:60-91;
 * expert/nn.py:10-61).  lstm_features > 0: LSTMCell variant (x -> LSTM(F) -> y), == 0: StackedMLPCell
 * variant (y = relu(Dense(x)), head_dims[0] = that hidden width).  Both heads are relu MLPs
 * y -> ... -> n (state, residual: next_x = head + x) and y -> ... -> m (action, tanh).
 * Flat parameter layout `expert` (flax order, kernel (in,out) then bias):
 *   LSTM: Wx[n][4F] | Wh[F][4F] | b[4F] (gates i,f,g,o)   or   MLP: W0[n][h] | b0[h]
 *   then the state head's layers, then the action head's layers.
 * history [B][hist+1][n] (hist >= 1 teacher-forced rows, then the current state) ->
 * goal [B][T+1][x_size] (row 0 = current state), init_U [B][T][m].  x_size, m and the head widths up to 1024
 * (the C4 / C5 state sizes), F <= 128, the MLP variant's first width <= 512. */
typedef struct gmpc_expert_shape {
  int lstm_features;
  int head_layers;                           /* dense layers per head (>= 1) */
  int head_dims_x[GMPC_MAX_LAYERS + 1];      /* y width, hidden..., n */
  int head_dims_u[GMPC_MAX_LAYERS + 1];      /* y width, hidden..., m */
} gmpc_expert_shape;
int gmpc_expert_rollout(gmpc_ctx* ctx, int B, int hist, const gmpc_expert_shape* es,
                        const float* expert, const float* history, float* goal, float* init_U,
                        void* stream);
long gmpc_expert_param_count(int n, const gmpc_expert_shape* es);

/* N3: dynamics-model regression, norm/dynamics_trainer.py:14-47 (predict_loss) and
 * :62-86 (batch mean + value_and_grad) with utils.py:230-240 (discounted_sum).  For each of the B
 * sequences: x_in_t = teacher_forcing ? xseq[t] : pred_{t-1} (x_in_0 = xseq[0]),
 * pred_t = dynamics(x_in_t, useq[t]), loss = sum_t discount^t |pred_t - next_xseq[t]|^2.
 *   xseq, next_xseq [B][S][n], useq [B][S][m], 1 <= S <= T, B <= max_batch
 *   -> loss_sum [1] (sum over the batch), grad_sum [dynamics parameter count] in the flat flax
 *      order of gmpc_set_params' dyn vector (sum over the batch: divide by the global batch size).
 * Uses the dynamics parameters bound by gmpc_set_params (n + m up to 1088: C4 / C5 included).  LSTM variant:
 * x columns only (xseq, next_xseq [B][S][x_size]), the carry starts at zero and is fed back even under teacher
 * forcing (dynamics_trainer.py:24-33); grad_sum in the layout Wx | Wh | b | tail. */
int gmpc_dynamics_loss_grad(gmpc_ctx* ctx, int B, int S, const float* xseq, const float* useq,
                            const float* next_xseq, double discount, int teacher_forcing,
                            float* loss_sum, float* grad_sum, void* stream);

/* Building block of the large-state (n > 64) Riccati path, exported for its unit test: batched
 * C[b] = alpha * X[b]^T Y[b] + beta * C[b] on the fp32 matrix cores; X[b] is K x M, Y[b] K x N,
 * C[b] M x N, row-major, densely packed per batch element; Y must be followed by >= 8 readable rows
 * of N floats. */
int gmpc_bgemm_tn(gmpc_ctx* ctx, int batch, int M, int N, int K, const float* X, const float* Y,
                  float* C, float alpha, float beta, void* stream);

/* Number of candidate rollouts (trajectory, step size) the line searches of the last gmpc_ilqr_solve
 * evaluated -- the work count behind bench.py's secondary roofline.  Synchronises the whole device
 * (hipDeviceSynchronize). */
long gmpc_linesearch_candidates(gmpc_ctx* ctx);

/* Counters of the line searches of the last gmpc_ilqr_solve, `n` <= 64 values: out[k], k = 0..15 = line searches
 * that accepted the step alpha_0 / 2^k (trajax line_search_ddp as called from policy/optimizers.py:19), out[16] =
 * line searches that ran out of step sizes, out[17] = the deepest halving accepted since the solve began, out[24 + r] =
 * candidate rollouts of speculative round r.  Diagnostic (bench.py reports it); synchronises the whole device
 * (hipDeviceSynchronize: streams created non-blocking included). */
int gmpc_linesearch_stats(gmpc_ctx* ctx, long* out, int n);

/* Stream overlap hook.  `hip_event` (a hipEvent_t, or NULL to clear) is recorded on the backward pass's stream
 * behind the Jacobian chain of gmpc_lqr_backward(_after_rollout) / of every iteration of gmpc_ilqr_solve.  Order of
 * the small-state pass since round 3: terminal quadratisation, Jacobian chain, EVENT, Riccati sweep (the terminal
 * quadratisation needs X only and runs first, so that the sweep is the launch right behind the chain;
 * GMPC_TERMINAL_FIRST=0 restores chain, event, terminal, sweep).  Large-state path: the event is recorded at the top
 * of the pass, before the terminal quadratisation and the step-major pipeline.  With GMPC_LIN_SPLIT=<percent> the
 * chain's ragged last round is a launch of its own and the event sits between the two launches -- the waiter then
 * starts while the last Jacobians are still being written, so it must not read them (bench.py's critic step does not);
 * off by default (measured, no gain).  A caller that runs independent work on a second stream -- the critic step of
 * the GAN loop: reference gan/runner.py:120-168 has no data dependence between it and the policy's backward pass --
 * makes that stream wait for the event: the work then shares the chip with the Riccati sweep (two wavefronts per
 * trajectory: most wave slots and registers are free) instead of taking workgroup slots from the matrix-core-bound
 * Jacobian chain.  The event must outlive its use. */
int gmpc_set_linearize_event(gmpc_ctx* ctx, void* hip_event);

/* Optional per-kernel timing with HIP events recorded on the launch stream around each kernel
 * (bench.py's roofline leg).  Slots: 0 rollout, 1 linearize, 2 terminal, 3 riccati, 4 linesearch,
 * 5 lstm_fwd, 6 head, 7 lstm_bwd, 8 wgrad (all weight-gradient GEMMs of one critic call), 9 adam.
 * gmpc_profile_read waits for the recorded events, returns the summed milliseconds and the number
 * of launches of that slot since the last read, and resets the slot. */
int gmpc_profile_enable(gmpc_ctx* ctx, int on);
int gmpc_profile_read(gmpc_ctx* ctx, int slot, double* total_ms, int* count);
/* Name of the kernel the slot's last launch ran on, as it appears in a rocprofv3 kernel trace (slot 1, the
 * Jacobian chain: the instantiation the shape selected, e.g. "k_linearize_regs<6, 100, 8, false>"); "" for slots
 * that always run the same kernel.  The string lives in the context (copied when the chain is launched): valid until
 * the context's next backward pass or gmpc_destroy. */
const char* gmpc_profile_kernel_name(gmpc_ctx* ctx, int slot);

/* Device pointers into the ctx's solution of the last gmpc_ilqr_solve / gmpc_bilevel_grad (valid
 * until the next such call): 0 X, 1 U, 2 H = A^-1 B, 3 dX, 4 Bvec, 5 AB, 6 K, 7 k, 11 d loss / d X.
 * Buffer 5 holds [B][T][n][n+m] for n <= 64 and ONE step's [B][n][n+m] (the last one processed,
 * t = 0) for n > 64.  gmpc_debug_buffer_count returns the number of floats allocated behind the
 * pointer (for max_batch trajectories); a reader must not go past it.
 * Used by the parity tests and by EvalMPC.get_optimal_values' `lqr` slot. */
const float* gmpc_debug_buffer(gmpc_ctx* ctx, int which);
long gmpc_debug_buffer_count(gmpc_ctx* ctx, int which);

#ifdef __cplusplus
}
#endif
#endif
