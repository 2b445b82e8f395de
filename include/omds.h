/*
 * omds.h -- C-ABI of the MI355X-native MPPI rollout + DS-modulation hot path.
 *
 * The reference (epfl-lasa/OptimalModulationDS) is pure Python on PyTorch-CPU and has NO
 * FFI / plugin boundary of its own; its "interface" for this path is the Python class API of
 * python_scripts/ds_mppi/functions/{MPPI,policy,cost,LinDS}.py and
 * python_scripts/mlp_learn/sdf/robot_sdf.py.  Each entry point below names the reference
 * method(s) it replaces (file:line relative to python_scripts/).  A maintainer binds these with
 * ctypes; optimalmodulationds_amd/_lib.py is that binding.
 *
 * Conventions
 *   - every function returns an omds_status (0 = OK); omds_last_error() gives the message;
 *     nothing aborts the process and there is no CPU fallback: without a usable HIP device
 *     omds_create() fails with OMDS_ERR_HIP.
 *   - all host arrays are C-contiguous fp32 / int32 in the reference's axis order; the caller
 *     owns host buffers, the library owns device buffers for the life of the context.
 *   - one context per device; calls on a context are not re-entrant; every call enqueues on the
 *     context's HIP stream and synchronises before returning unless stated otherwise.
 */
#ifndef OMDS_H
#define OMDS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMDS_API __attribute__((visibility("default")))

typedef struct omds_ctx omds_ctx;

typedef enum {
    OMDS_OK = 0,
    OMDS_ERR_INVALID_ARG = 1,
    OMDS_ERR_HIP = 2,
    OMDS_ERR_RCCL = 3,
    OMDS_ERR_NOT_INITIALISED = 4,
    OMDS_ERR_UNSUPPORTED = 5
} omds_status;

#define OMDS_MAX_DOF 7
#define OMDS_ACT_RELU 0
#define OMDS_ACT_TANH 1

/* MPPI.__init__ arguments (ds_mppi/functions/MPPI.py:22-66). */
typedef struct {
    int32_t n_dof;         /* n: joints (<= OMDS_MAX_DOF)                                   */
    int32_t n_traj;        /* N: rollouts held by THIS context (the shard, in multi-GPU)   */
    int32_t horizon;       /* H: dt_H                                                      */
    int32_t n_kernel_max;  /* TensorPolicyMPPI.N_KERNEL_MAX (policy.py:18), 50             */
    int32_t max_obs;       /* capacity of the obstacle buffer                              */
    int32_t n_closest;     /* k: n_closest_obs                                             */
    int32_t device;        /* HIP device ordinal                                           */
    int32_t flags;         /* OMDS_FLAG_* bits, 0 = defaults                               */
} omds_config;

/* omds_config.flags: the generic step of omds_propagate -- five launches (k_pass1, k_topk, k_pass2, k_modulate,
 * k_rollout_layer1), the only one for n_dof other than 2 and 7 -- instead of the two-launch k_pass1 + k_tail step. */
#define OMDS_FLAG_UNFUSED_STEP 1
/* Keep scenes with few obstacles (O <= 32) on the two-launch k_pass1 + k_tail step instead of the one-launch fused
 * small-scene step (k_step_small) that omds_propagate picks for them by itself.                                      */
#define OMDS_FLAG_TWO_KERNEL_STEP 2
/* the all-fp32 step keeps its tail with a forward of its own (k_tail) instead of taking pass 2's forward from pass 1 (k_pass1 in
 * its emitting mode + k_tail_sel: same bits, one forward less); for A/B runs and the tests that compare the two            */
#define OMDS_FLAG_TAIL_FORWARD 4
/* k_pass1 multiplies every k-chunk of every layer instead of compacting every tile to the hidden units that fire in it (the exact
 * zero-skip of ReLU networks: same bits either way); for A/B runs and the tests that compare the two                              */
#define OMDS_FLAG_DENSE_PASS1 8

/* The constants the reference hard-codes inside propagate() (MPPI.py:117-217,277) and
 * LinDS (LinDS.py:9), as parameters; omds_default_params() fills the reference values.   */
typedef struct {
    float dt;              /* integration step of the rollouts (MPPI.py:221)               */
    float dst_thr;         /* subtracted from the network distance (MPPI.py:117)           */
    float lin_thr;         /* LinDS.lin_thr (LinDS.py:9)                                   */
    float lvel[5];         /* generalized_sigmoid (y_min,y_max,x0,x1,k) of l_vel  (:132)   */
    float ln[5];           /* ... of l_n   (:143-153)                                      */
    float ltau[5];         /* ... of l_tau (:155)                                          */
    float goal_act_cut;    /* goal activation hard cut, 0.5 (:194)                         */
    float norm_clamp;      /* velocities with norm <= this are not normalised, 0.5 (:212)  */
    float coll_slow;       /* in-collision slow-down factor 0.1 (:215)                     */
    float coll_repulse;    /* in-collision repulsion gain 0.1 (:216)                       */
    float softmax_k;       /* -10: weights of the k closest gradients (:277)               */
    float rbf_p;           /* Policy.p, RBF norm order (policy.py:41), 2                   */
    uint32_t ignored_links;/* bit c set: link c is masked in pass 1 (MPPI.py:241)          */
    uint32_t variant;      /* OMDS_VARIANT_* bits; 0 = MPPI.py, see below                  */
    uint32_t cost_terms;   /* OMDS_COST_* bits summed by Cost.evaluate_costs; default all  */
} omds_params;

/* ds_mppi/functions/MPPI_toy.py (the 2-D toy drivers' variant of MPPI.py) differs from MPPI.py in constants
 * (all of them fields above) and in two behaviours:                                                        */
#define OMDS_VARIANT_KVAL_TIMES_ACT 1u /* kernel_val_all holds phi * activation (MPPI_toy.py:178-179)       */
#define OMDS_VARIANT_NO_BASE_MASK   2u /* update mask without the rollout-0 term (MPPI_toy.py:318-321)      */
/* Cost.evaluate_costs terms (cost.py:14-21); cost_toy.py:14-18 sums GOAL | COLLISION | STAGNATION only.    */
#define OMDS_COST_GOAL         1u
#define OMDS_COST_COLLISION    2u
#define OMDS_COST_JOINT_LIMITS 4u
#define OMDS_COST_STAGNATION   8u
#define OMDS_COST_FK          16u
#define OMDS_COST_ALL         31u

OMDS_API void omds_default_params(omds_params* p);

/* MPPI.__init__ tensor allocation (MPPI.py:37-66); no warm-up propagates are run. */
OMDS_API int omds_create(const omds_config* cfg, omds_ctx** out);
OMDS_API void omds_destroy(omds_ctx* ctx);
/* Message of the last failure on ctx (ctx == NULL: last omds_create failure of this thread). */
OMDS_API const char* omds_last_error(const omds_ctx* ctx);

/* RobotSdfCollisionNet.load_weights + model preparation (robot_sdf.py:31-51,
 * frankaPlanner.py:43-51).  n_linear Linear layers; dims[n_linear+1] = {3(n+3), hidden...,
 * C}; W[i] is [dims[i+1], dims[i]] row-major like torch, b[i] is [dims[i+1]].  act =
 * OMDS_ACT_*; out_div = 100 when C == 9 (cm -> m, MPPI.py:236-237) else 1.  dims[0] may also
 * be 3(n+2): the toy networks take planar obstacle points (in_channels = DOF+2,
 * scripts/standaloneToy2d.py:33); the z column of the obstacle array is then ignored.
 * Widths: hidden layers of up to 256 units (every network the reference ships: 256 and 128) run on the fused LDS-resident MFMA
 * kernels (narrower ones zero-padded).  MLPRegression itself is width-agnostic (network_macros_mod.py:96-135): a network with a
 * hidden layer of 257 .. 4096 units takes the unfused path of csrc/wide_kernels.hip -- the reference's own op sequence on exact-fp32
 * MFMA GEMMs with the activations materialised in HBM; no screening, no skip concatenations there.  Same results contract.     */
OMDS_API int omds_set_mlp(omds_ctx* ctx, int n_linear, const int32_t* dims, const float* const* W,
                          const float* const* b, int act, float out_div);
/* The same for MLPRegression(..., skips=[...]) (network_macros_mod.py:113-146): behind the activations of
 * the Linear layers named in skip_after[n_skips] (indices into the flat sequence of Linear layers, 0 ..
 * n_linear-2) the encoded input [x, sin x, cos x] is concatenated, so the next Linear layer is 3(n+3)
 * wider than the previous one's output: W[i] is [out_dims[i], in_dims[i]] with in_dims[0] = 3(n+3) and
 * in_dims[i] = out_dims[i-1] (+ 3(n+3) behind a concatenation, its columns last, as torch.cat((y, x_nerf))
 * orders them).  out_dims[i] + 3(n+3) <= 256 for those layers (the reference narrows them by 3(n+3) itself).
 * n_skips = 0 is omds_set_mlp.  Skip-connection networks are screened like plain ones (the screening kernel ORs the
 * encoded input into the last two k-chunks of the layer behind a concatenation).                                        */
OMDS_API int omds_set_mlp_ex(omds_ctx* ctx, int n_linear, const int32_t* in_dims, const int32_t* out_dims,
                             const float* const* W, const float* const* b, int act, float out_div,
                             int n_skips, const int32_t* skip_after);

/* MPPI.update_obstacles (MPPI.py:347-350): xyzr is [O,4] spheres (x,y,z,r).  n_obs may exceed config.max_obs: the obstacle
 * buffers then grow (to twice n_obs) inside the context -- handle, network, samples, communicator and screening state stay. */
OMDS_API int omds_set_obstacles(omds_ctx* ctx, const float* xyzr, int n_obs);
/* LinDS(q_goal) / MPPI.reset_DS / switch_DS_idx (LinDS.py:7-10, MPPI.py:76-84). */
OMDS_API int omds_set_ds(omds_ctx* ctx, const float* q_goal);
/* MPPI_toy's nominal DS (MPPI_toy.py:89-91): velocity = (q - q_goal) @ A, A [n,n] row-major, not
 * normalised.  A == NULL switches back to LinDS.                                            */
OMDS_API int omds_set_ds_matrix(omds_ctx* ctx, const float* q_goal, const float* A);
/* SEDS nominal DS (ds_mppi/functions/SEDS.py:8-74): a Gaussian mixture regression x -> velocity on x = q - q_goal with
 * G components.  The caller passes the quantities SEDS.__init__ / GMR derive from the .mat file (so that a caller who
 * derives them like the reference -- torch.inverse / torch.det in float32 -- gets the reference's numbers):
 *   mu_in [G][n]      Mu[:n, j]                          b [G][n]          Mu[n:, j]
 *   sigma_inv [G][n][n]  inverse(Sigma[:n,:n,j])         A [G][n][n]       Sigma[n:,:n,j] @ inverse(Sigma[:n,:n,j])
 *   prior [G]         Priors[j]                          den [G]           sqrt(2 pi^n |det Sigma[:n,:n,j]| + 1e-100)
 * velocity (SEDS.get_velocity): beta_j = clamp(nan_to_num(prior_j N_j / sum), 1e-8), y = sum_j beta_j (b_j + A_j (x - mu_j));
 * farther than lin_thr from the goal: y / |y|, or -x / |x| where |y| < seds_thr.  G = 0 switches back to LinDS.            */
OMDS_API int omds_set_ds_seds(omds_ctx* ctx, const float* q_goal, int n_gauss, const float* mu_in, const float* b,
                              const float* sigma_inv, const float* A, const float* prior, const float* den,
                              float lin_thr, float seds_thr);
OMDS_API int omds_set_params(omds_ctx* ctx, const omds_params* p);
/* Cost(q_f, dh_params) + the q_min/q_max attributes (cost.py:5-12): dh_params [n+1,4]
 * rows (d, theta, a, alpha); q_min/q_max [n].                                             */
OMDS_API int omds_set_cost(omds_ctx* ctx, const float* dh_params, const float* q_min, const float* q_max);

/* TensorPolicyMPPI.sample_policy (policy.py:51-74) with injected samples (parity runs):
 * mu [N,K,n], sigma [N,K], alpha [N,K,n] = the first K kernels of mu_tmp/sigma_tmp/alpha_tmp. */
OMDS_API int omds_set_policy_samples(omds_ctx* ctx, const float* mu, const float* sigma, const float* alpha,
                                     int n_kernels);
/* TensorPolicyMPPI.sample_policy on the device (Philox4x32-10 + Box-Muller): theta_tmp =
 * N(0, theta_s) + theta_c; mean arrays are [K,n], [K], [K,n].  rollout_offset = global index
 * of this shard's rollout 0 (the global rollout 0 gets alpha_tmp = alpha_c, policy.py:74). */
OMDS_API int omds_sample_policy(omds_ctx* ctx, const float* mu_c, const float* sigma_c, const float* alpha_c,
                                float mu_s, float sigma_s, float alpha_s, int n_kernels, uint64_t seed,
                                int64_t rollout_offset);
/* Read back the sampled tensors in the reference layout (first K kernels). NULL = skip. */
OMDS_API int omds_get_policy_samples(omds_ctx* ctx, float* mu, float* sigma, float* alpha);

/* MPPI.propagate (MPPI.py:97-224).  q_cur is [n] (broadcast to all rollouts, per_rollout = 0)
 * or [N,n] (per-rollout start states, per_rollout = 1: used for teacher-forced parity tests). */
OMDS_API int omds_propagate(omds_ctx* ctx, const float* q_cur, int per_rollout);
/* The 5-tuple returned by propagate() plus its side outputs, reference layouts:
 * all_traj [N,H,n], closest_dist_all [N,H], kernel_val_all [N,H,K], dot_products [N,H],
 * kernel_activations [N,H], qdot [N,n], normal [N,H,n] (= norm_basis[..., 0]).  NULL = skip. */
OMDS_API int omds_get_rollouts(omds_ctx* ctx, float* all_traj, float* closest_dist_all, float* kernel_val_all,
                               float* dot_products, float* kernel_activations, float* qdot, float* normal);

/* The same outputs for `count` chosen rollouts t[count] only: all_traj [count,H,n], ..., qdot [count,n], normal [count,H,n] (NULL =
 * skip).  What the reference's planner loop reads per iteration is a few rollouts -- the best one for its FK payload, the one a new
 * kernel centre came from (frankaPlanner.py:147-168: closests_dist_all[i, h], norm_basis[i, h], all_traj[best_idx]) -- so a driver
 * need not move the N x H tensors across PCIe at all.                                                                     */
OMDS_API int omds_get_rollout_rows(omds_ctx* ctx, const int32_t* t, int count, float* all_traj, float* closest_dist_all,
                                   float* kernel_val_all, float* dot_products, float* kernel_activations, float* qdot, float* normal);

/* MPPI.distance_repulsion_nn on an arbitrary batch (MPPI.py:227-282): q [B,n], B <= N.
 * Outputs (NULL = skip): distance [B], nn_grad [B,n], mindist [B,O] (pass-1 matrix),
 * closest_idx [B,k] (ascending distance).  Also serves update_kernel_normal_bases (:284-304). */
OMDS_API int omds_dist_grad(omds_ctx* ctx, const float* q, int batch, float* distance, float* nn_grad,
                            float* mindist, int32_t* closest_idx);
/* MLPRegression.forward on raw rows (network_macros_mod.py:137-146) and the vjp of the
 * arg-min output (robot_sdf.py:153-158): x [B,n+3]; y [B,C], grad [B,n+3], min_idx [B]. B <= N*k. */
OMDS_API int omds_mlp_forward_vjp(omds_ctx* ctx, const float* x, int batch, float* y, float* grad,
                                  int32_t* min_idx);
/* RobotSdfCollisionNet.compute_signed_distance_wgrad with a column list (mlp_learn/sdf/robot_sdf.py:68-100): the Jacobian columns of
 * the listed raw outputs, one backward per column as the reference's loop of .backward() calls does.  x [B,n+3]; cols [n_cols],
 * 0 <= cols[k] < C, 1 <= n_cols <= 16; y [B,C] or NULL; jac [B, n+3, n_cols] (the reference's grads[:, :, k]). B <= N*k.   */
OMDS_API int omds_mlp_jacobian(omds_ctx* ctx, const float* x, int batch, const int32_t* cols, int n_cols, float* y,
                               float* jac);

/* MPPI.get_cost -> Cost.evaluate_costs (MPPI.py:315-317, cost.py:13-22). cost_out [N] or NULL. */
OMDS_API int omds_cost(omds_ctx* ctx, float* cost_out);
/* Cost.evaluate_costs on caller-supplied tensors (cost.py:13-22 evaluates exactly what it is given): all_traj
 * [B,H,n], closest_dist_all [B,H], B <= n_traj; cost_out [B].  Evaluated on the device; the context's own rollouts and
 * their cost stay untouched.                                                                                   */
OMDS_API int omds_cost_eval(omds_ctx* ctx, const float* all_traj, const float* closest_dist_all, int batch,
                            float* cost_out);
/* MPPI.shift_policy_means + TensorPolicyMPPI.update_policy (MPPI.py:331-345,
 * policy.py:88-113), single-shard form.  mu_c/sigma_c/alpha_c: in/out host means of the first K
 * kernels; mask_out [K] (1 = updated); weights_out [N] (normalised MPPI weights) or NULL.   */
OMDS_API int omds_weighted_update(omds_ctx* ctx, float rate, float ker_thr, float* mu_c, float* sigma_c,
                                  float* alpha_c, int32_t* mask_out, float* weights_out);
/* The same update on caller-supplied tensors (MPPI.shift_policy_means reads self.cost, self.kernel_val_all and
 * self.kernel_activations, MPPI.py:333-341), the way omds_cost_eval serves Cost.evaluate_costs: cost [N], kernel_val_all [N,H,K],
 * kernel_activations [N,H] in the reference layouts (K = the kernel count of the policy samples the context holds; they are the
 * theta_tmp of the update).  The context's own rollouts, cost and running maxima stay untouched.  Teacher-forced parity: the
 * reference's own rollout tensors in, its updated means out.                                                              */
OMDS_API int omds_weighted_update_eval(omds_ctx* ctx, const float* cost, const float* kernel_val_all, const float* kernel_activations,
                                       float rate, float ker_thr, float* mu_c, float* sigma_c, float* alpha_c, int32_t* mask_out,
                                       float* weights_out);
/* MPPI.get_qdot (MPPI.py:319-329): mode 0 = 'best', 1 = 'weighted'; out [n]. */
OMDS_API int omds_get_qdot(omds_ctx* ctx, int mode, float* out);

/* TensorPolicyMPPI.check_traj_for_kernels (policy.py:153-175) on the device-resident rollouts of
 * the last omds_propagate: candidate kernel centres = states (t, h) with closest_dist < thr_dist
 * and dot_product < thr_dot whose largest RBF value w.r.t. the K existing kernels (mu_c [K,n],
 * sigma_c [K], norm order = params.rbf_p) is < thr_kernel (K == 0: every close state).  Output
 * in the reference's boolean-mask order (rollout-major, then horizon): cand_q [cap,n], cand_th
 * [cap,2] = (t, h); *count = number found (may exceed cap, only min(count, cap) rows are written;
 * cap <= N*H).                                                                              */
OMDS_API int omds_kernel_candidates(omds_ctx* ctx, float thr_dist, float thr_kernel, float thr_dot, const float* mu_c,
                                    const float* sigma_c, int n_kernels, int cap, float* cand_q, int32_t* cand_th,
                                    int32_t* count);

/* Multi-GPU (new work; the reference is single-process): rollouts shard across one process per GPU,
 * every rank holds one context with N_local rollouts, and the only exchange is the cost-weighted update
 * (MPPI.shift_policy_means / get_qdot, MPPI.py:319-345 + policy.py:88-113).
 *
 * Native path (comm.hip): a RCCL communicator per context; the update runs on the context stream on device
 * buffers -- all-reduce SUM of [sum cost, N_local] (8 bytes) -> global beta; all-reduce SUM of the packed partial
 * sums (<= 3.4 KB); all-gather of (min cost, qdot of the arg-min) only when qdot_best is asked for.
 *   omds_comm_unique_id   : rank 0 creates the 128-byte id (ncclGetUniqueId); the launcher ships it to the other
 *                           ranks by whatever bootstrap it has (bench.py: a gloo broadcast).  No context needed;
 *                           omds_comm_last_error() holds the message of a failure.
 *   omds_comm_init_rank   : collective over all ranks (ncclCommInitRank on the context's device).  The shard of
 *                           rank 0 owns the global rollout 0 (policy.py:74): pass rollout_offset = rank * N_local
 *                           to omds_sample_policy.
 *   omds_weighted_update_sharded : collective.  mu_c/sigma_c/alpha_c in/out (identical on every rank afterwards),
 *                           mask_out [K]; qdot_weighted [n] = get_qdot('weighted') or NULL; qdot_best [n] =
 *                           get_qdot('best') over all shards or NULL (NULL skips the MINLOC gather);
 *                           n_total_out = number of rollouts over all shards or NULL.  Without a communicator it
 *                           is the single-shard update.  Failures of RCCL return OMDS_ERR_RCCL.
 *   omds_comm_probe       : OMDS_OK when RCCL can be loaded in this process (no device touched, no communicator made),
 *                           OMDS_ERR_RCCL otherwise (message: omds_comm_last_error).  A launcher calls it on EVERY rank and
 *                           lets the ranks agree before any of them enters omds_comm_init_rank: a rank that cannot load
 *                           RCCL would otherwise leave the others waiting inside ncclCommInitRank.                    */
#define OMDS_COMM_ID_BYTES 128
OMDS_API int omds_comm_probe(void);
OMDS_API int omds_comm_unique_id(uint8_t* out128);
OMDS_API const char* omds_comm_last_error(void);
OMDS_API int omds_comm_init_rank(omds_ctx* ctx, const uint8_t* id128, int rank, int world);
OMDS_API int omds_comm_destroy(omds_ctx* ctx);
OMDS_API int omds_comm_info(const omds_ctx* ctx, int32_t* rank, int32_t* world);
/* 1 while the context owns a communicator (a single-rank one included), else 0. */
OMDS_API int omds_comm_active(const omds_ctx* ctx);
OMDS_API int omds_weighted_update_sharded(omds_ctx* ctx, float rate, float ker_thr, float* mu_c, float* sigma_c,
                                          float* alpha_c, int32_t* mask_out, float* qdot_weighted, float* qdot_best,
                                          float* n_total_out);
/* Host-mediated form of the same exchange, for launchers without RCCL (tests/test_dist_gloo.py runs it over gloo,
 * world size 2, on CPU-side reductions): the library produces this shard's partial sums in two phases and the host
 * (optimalmodulationds_amd/dist.py) all-reduces them:
 *   omds_cost_sum    -> out2 = [sum_t cost, N_local]                 (all-reduce SUM, 8 bytes)
 *   omds_local_sums  -> packed buffer of omds_red_count() floats for the GLOBAL beta:
 *        [0] sum w' | sum w' mu (K*n) | sum w' sigma (K) | sum w' alpha (K*n) |
 *        sum_t max_h(phi*act) (K) | sum_h phi of GLOBAL rollout 0 (K; include_rollout0 != 0
 *        only on the shard that owns it) | sum w' qdot (n)            (all-reduce SUM)
 *        | min cost of the shard, qdot of its arg-min (1+n)          (all-gather, MINLOC)
 *   omds_apply_update: pure host arithmetic on the reduced buffer (masks MPPI.py:336-342 +
 *        TensorPolicyMPPI.update_policy policy.py:88-113); needs no context / no GPU.
 *        variant = omds_params.variant (OMDS_VARIANT_NO_BASE_MASK drops the rollout-0 mask). */
OMDS_API int omds_cost_sum(omds_ctx* ctx, float* out2);
OMDS_API int omds_red_count(const omds_ctx* ctx);
OMDS_API int omds_local_sums(omds_ctx* ctx, float sum_cost, float n_total, int include_rollout0, float* red_out);
OMDS_API int omds_apply_update(int n_kernels, int n_dof, int horizon, const float* red, float n_total, float rate,
                               float ker_thr, uint32_t variant, float* mu_c, float* sigma_c, float* alpha_c,
                               int32_t* mask_out);

/* Screening of pass 1 (screen_kernel.hip).  The N*O first-pass evaluations of MPPI.distance_repulsion_nn only feed the
 * sort that picks the k closest obstacles (MPPI.py:245-253); omds_propagate may therefore evaluate them in fp16 and
 * re-evaluate in fp32 only the candidates {o : Da(o) <= tau}, tau = k-th smallest Da + 1.25 eps (and every o whose Da is
 * not finite).  They contain the fp32 top-k (ties included) whenever the pairs that are NOT re-evaluated have a screening
 * error Da - D <= eps and the exact k-th smallest candidate stays eps below tau.  The distances and gradients a step uses
 * always come from the fp32 pass 2, and omds_dist_grad always uses the fp32 pass 1.
 * The identity with the all-fp32 step is therefore CONDITIONAL on eps, and eps is a measured quantity, not an a-priori one.
 * THE CONTRACT OF THE DEFAULT.  A context runs the ALL-FP32 step unless the caller opts in (mode 1 | 2 below, or OMDS_SCREEN=1|2 in the
 * environment): by default every network evaluation of omds_propagate is the reference's arithmetic on every pair, unconditionally.
 * Screening is an opt-in 5x: its results are the all-fp32 step's bit for bit in every test and soak run of this repository, but on
 * the strength of a MEASURED bound.  There is no usable a-priori one: propagating the f16 rounding of inputs, weights and activations
 * (2^-11 relative each) through sum |W| per layer gives |Da - D| <= 455 m for the shipped Franka network on |q| <= 2.9, |p| <= 1.5 m
 * (first order with sampled activation magnitudes: 58-81 m; planar 7-DoF: 5.8e3) against a measured eps of 0.016 m -- the norm
 * bound ignores the cancellation a trained network lives on, and is 3e4 times too large to select anything
 * (tools/studies/screen_apriori_bound.py).
 * What is measured, and when:
 *   - calibration: eps = 6 x the largest |Da - D| over ~3e5 pairs (states uniform in the joint box + states of the last
 *     propagate's rollouts, against the current obstacles), at the first screened propagate after omds_set_mlp, after
 *     omds_set_obstacles when the scene differs from the calibrated one (another count or radius, a sphere moved by more
 *     than 0.1), after a change of params.ignored_links, and on request (eps < 0 below).
 *     LATENCY: a calibration runs inside the omds_propagate that needs it -- one fp32 and one f16 pass over the batch plus two
 *     host re-sorts of the f16 weight pack by how often the hidden units fire (one in the calibration, one behind the first
 *     accepted propagate) -- about 21 ms on the first screened propagate against 5.8 ms for an ordinary 1024 x 32 iteration
 *     (tools/studies/reorder_cost.py), and the propagate after the second re-sort carries a sweep (+1 ms).  A translating
 *     scene does not recalibrate; a caller who SWAPS scenes at rate and cannot take the spike fixes the bound (eps > 0) or
 *     switches screening off for those iterations;
 *   - every step of every propagate: |Da - D| of every candidate; Da - D of the AUDIT rows, a pseudo-random 1-in-`one_in`
 *     sample (another one every step) of the pairs that are not candidates, re-evaluated in fp32 by one launch at the end
 *     of the horizon loop (k_audit); the slack of every rollout (tau - exact k-th smallest >= eps).
 *   - every 32nd screened propagate (omds_set_screening_sweep): a SWEEP -- all N x O pairs of the propagate's last horizon
 *     step (soak runs: of EVERY horizon step) in fp32 beside all their screening values, max |Da - D| over every one of them,
 *     and the distribution of Da - D over the pairs that were not candidates (omds_screen_sweep_hist): the quantity the
 *     selection rule's assumption is about, counted exhaustively instead of sampled.  profiles/r04_screen_error_hist.txt
 *     holds that distribution over > 1e11 pairs (tools/sweep_soak.py).
 * A propagate is accepted only while all these maxima stay <= eps / 2 and no slack check failed; otherwise it is redone with the
 * fp32 pass 1 (its results are then the fp32 ones by construction) and eps is widened; three fallbacks in a row suspend
 * screening until the next calibration.  A row outside the audit sample whose error exceeds eps can still go unseen in
 * one propagate: at run time the identity is measured on a sample, not proven.  What the sample stands on: the exhaustive count of
 * tools/sweep_soak.py -- every horizon step of 5 183 propagates swept, 1.93e11 pairs over ten scene / network legs, none of the
 * 1.87e11 unevaluated pairs above eps / 2, the worst at 0.197 eps (profiles/r04_screen_error_hist.txt; the round-5 build: 2.4e11 more pairs, none above eps / 2, worst 0.182 eps,
 * profiles/r05_screen_error_hist.txt; the survival function of
 * (Da - D) / eps falls by half a decade or more per 1/128).
 * mode: -1 the library's default = off (env OMDS_SCREEN=0|1|2 changes the default of contexts left at -1), 0 off, 1 on, 2 on where it pays
 * (ReLU / tanh networks, with or without skip concatenations, n_traj * n_obs >= 65536 and n_obs >= 4 n_closest; else the fp32 step).  eps > 0 sets the bound in place of a calibration (never recalibrated; the run-time checks
 * still widen it when they must); eps == 0 changes the mode only; eps < 0 discards the calibration (measured again at the next screened propagate).
 * omds_set_screening_audit: one_in = 0 (no audit sample) or a power of two; default 128 (EXPERIMENTS.md C 4.1b has the measured cost per rate).
 * omds_screen_stats: active, eps in use, largest candidate error seen since the last calibration, mean candidates per
 * (rollout, step) since the last omds_prof_reset, fp32 fallbacks since creation (NULL = skip).
 * omds_screen_audit_stats: one_in, mean audit rows per (rollout, step) since the last omds_prof_reset, largest audit error
 * seen since the last calibration, suspended flag, calibrations run since creation.                                   */
OMDS_API int omds_set_screening(omds_ctx* ctx, int mode, float eps);
OMDS_API int omds_set_screening_audit(omds_ctx* ctx, int one_in);
/* every = 0: no sweeps; default 32.  all_steps = 0: a sweeping propagate checks its last horizon step (0.5 % of the run at the
 * default period); 1: every horizon step (a soak / qualification mode: each step then also runs the fp32 pass 1, so the propagate
 * is slower than the unscreened one).  Stats: the setting, sweeps (swept steps) run since creation, largest |Da - D| a sweep saw
 * since the last calibration (NULL = skip).
 * omds_screen_sweep_hist: what all sweeps since creation (or the last reset != 0) have counted, OMDS_SWEEP_HIST_WORDS 64-bit words:
 *   [0] pairs swept  [1] of them NOT candidates (never re-evaluated by the step: the population the bound eps is about)
 *   [2] non-candidates with Da - D > eps / 2 (the acceptance margin)  [3] with Da - D > eps (a possible miss)
 *   [4] non-candidates with a non-finite difference  [5] largest Da - D over the non-candidates (float bits)
 *   [6] largest |Da - D| over all pairs (float bits)  [7] steps swept
 *   [8 .. 8 + L)     non-candidates with Da - D >= 0 by magnitude: bin b holds 2^(b - L) <= |x| < 2^(b + 1 - L) (bin 0 also zero
 *                    and everything smaller, bin L - 1 everything >= 1/2), L = OMDS_SWEEP_HIST_LOG_BINS
 *   [8 + L .. 8 + 2L)   the same for Da - D < 0 (the harmless side: the pair is even farther than its screening value said)
 *   [8 + 2L .. 8 + 2L + R)  non-candidates with Da - D > 0 by (Da - D) / eps in R = OMDS_SWEEP_HIST_RATIO_BINS linear bins over [0, 1)
 *                    (the last bin also holds everything >= 1), eps = the bound in use when the step was swept.            */
#define OMDS_SWEEP_HIST_LOG_BINS 32
#define OMDS_SWEEP_HIST_RATIO_BINS 128
#define OMDS_SWEEP_HIST_WORDS (8 + 2 * OMDS_SWEEP_HIST_LOG_BINS + OMDS_SWEEP_HIST_RATIO_BINS)
OMDS_API int omds_set_screening_sweep(omds_ctx* ctx, int every, int all_steps);
OMDS_API int omds_screen_sweep_stats(omds_ctx* ctx, int32_t* every, int64_t* sweeps, float* sweep_max_err);
OMDS_API int omds_screen_sweep_hist(omds_ctx* ctx, uint64_t* words, int n_words, int reset);
OMDS_API int omds_screen_audit_stats(omds_ctx* ctx, int32_t* one_in, double* audit_rows_per_rollout_step, float* audit_max_err,
                                     int32_t* suspended, int64_t* calibrations);
/* What tripped the fp32 fallbacks since creation (NULL = skip): a measured error above eps / 2 (candidates, audit sample or sweep);
 * a rollout whose exact k-th smallest candidate came within eps of tau (slack guard); a step whose candidate list outgrew the
 * per-entry buffers (32 candidates per rollout on average: near-flat distance fields); and how often three in a row (or a
 * non-finite error) suspended screening until the next calibration.                                                      */
OMDS_API int omds_screen_fallback_stats(omds_ctx* ctx, int64_t* by_error, int64_t* by_slack, int64_t* by_overflow, int64_t* suspensions);
/* Unit order of the screening network.  k_screen does not issue the MFMAs of a k-chunk (16 hidden units) whose activations are zero
 * for all 32 pairs of a wave -- exact: such a chunk adds nothing to any output.  Which units fire is a property of the trained
 * weights (of the shipped Franka network's 1024 hidden units, 300 fire for no pair of the shelf scene), so every calibration
 * sorts the hidden units of the fp16 pack by how often they fire on a uniform sample of 256 wave-sized blocks of (state, obstacle)
 * pairs (one k_exact launch, 6-7 ms; once on its own batch, once more on the states the first accepted propagate reaches),
 * which puts the silent ones into whole chunks.  ReLU networks without skip
 * concatenations; the fp32 kernels do not use this pack, so no returned number depends on the order.
 * reorders: packs rebuilt since creation; never_fired [n_levels <= 9] (NULL = skip): per hidden level, units that fired in no
 * sampled row at the last reorder.                                                                                          */
OMDS_API int omds_screen_order_stats(omds_ctx* ctx, int64_t* reorders, int32_t* never_fired, int n_levels);
/* The EXACT ZERO-SKIP of the fp32 pass 1 (k_pass1; ReLU networks without skip concatenations).  A hidden unit whose activation is
 * exactly zero in every row of a 64-row tile adds fmaf(0, w, acc) = acc to every chain of the next layer, so leaving it out changes
 * no bit -- as long as the units that ARE multiplied keep their ascending order.  k_pass1 therefore stores every hidden level of a
 * tile COMPACTED to the units that fire in that tile (rank order = unit order) and multiplies only those, fetching the weights by
 * unit; on the Franka shelf task 164 / 195 / 146 / 130 of the 256 units of the four levels fire in a tile.  Nothing is presumed
 * about which units fire: the result is the reference's chain bit for bit for every input (tests/test_gpu_sparse.py;
 * OMDS_FLAG_DENSE_PASS1 switches the compaction off).
 * active: 1 when the installed network qualifies; mean_units[L] / mean_chunks[L], L = 0 .. n_levels-1: firing units per tile of hidden
 * level L and k-chunks multiplied over it (of 8 positions; the last hidden level: of 16), averaged over the `tiles` tiles since
 * omds_set_mlp (NULL = skip).                                                                                                      */
OMDS_API int omds_pass1_skip_stats(omds_ctx* ctx, int32_t* active, double* mean_units, double* mean_chunks, int n_levels, int64_t* tiles);
/* (The two test hooks that damage the screening inputs / force a tile shape are NOT part of this library: they are declared in
 * include/omds_test.h and exported by libomds_hip_test.so only.)                                                           */
/* Diagnostic: the fp16 screening network alone on q [B,n] -> mindist [B,O] (the values the candidate selection sees). */
OMDS_API int omds_screen_mindist(omds_ctx* ctx, const float* q, int batch, float* mindist);
OMDS_API int omds_screen_stats(omds_ctx* ctx, int32_t* active, float* eps, float* max_err_seen,
                               double* cand_per_rollout_step, int64_t* fallbacks);

/* SDF training (mlp_learn/train_sdf.py:96-151): full-batch regression of the distance network on a data set
 * (x [B, d] raw inputs = joint angles then the obstacle point, y [B, C] link distances) -- per epoch one forward through
 * [x, sin x, cos x] -> Linear + act ... -> Linear (network_macros_mod.py:137-146), F.mse_loss(reduction = 'mean'), the backward
 * through every layer and one torch.optim.Adam step (single-tensor arithmetic: lerp of exp_avg, bias-corrected step size,
 * denom = sqrt(exp_avg_sq) / sqrt(bc2) + eps), all on the device in exact-fp32 MFMA GEMMs (csrc/train.hip).  The scheduler
 * (ReduceLROnPlateau), validation metrics and the checkpoint dictionary of the reference stay on the host
 * (tools/train_sdf_hip.py).  A trainer is its own object (no omds_ctx needed); dims[n_linear + 1] = {3 d, hidden ..., C}; W[i] is
 * [dims[i+1], dims[i]] row-major like torch.  omds_trainer_set_weights also resets the optimizer state and step count.
 * omds_trainer_step returns the loss BEFORE the update (what train_sdf.py prints as train loss); omds_trainer_eval runs the
 * forward + loss with the current weights, no update, on the training set (which = 0) or on the validation set of
 * omds_trainer_set_val_data (which = 1: train_sdf.py:84-86, 117-121; no weight copy, the optimizer state untouched); pred_out
 * [B, C] or NULL.  omds_trainer_get / set_optimizer_state: torch.optim.Adam's exp_avg (m) and exp_avg_sq (v) of every weight and
 * bias and the step count -- what train_sdf.py:130-138 saves as optimizer.state_dict() and a resumed run restores (arrays shaped
 * like the weights; get: a NULL array or entry is skipped).                                                               */
typedef struct omds_trainer omds_trainer;
OMDS_API int omds_trainer_create(int device, int n_linear, const int32_t* dims, int act, omds_trainer** out);
OMDS_API void omds_trainer_destroy(omds_trainer* tr);
OMDS_API const char* omds_trainer_last_error(const omds_trainer* tr);
OMDS_API int omds_trainer_set_weights(omds_trainer* tr, const float* const* W, const float* const* b);
OMDS_API int omds_trainer_get_weights(omds_trainer* tr, float* const* W, float* const* b);
OMDS_API int omds_trainer_set_data(omds_trainer* tr, const float* x, const float* y, int batch);
OMDS_API int omds_trainer_step(omds_trainer* tr, float lr, float beta1, float beta2, float eps, float* loss_out);
OMDS_API int omds_trainer_set_val_data(omds_trainer* tr, const float* x, const float* y, int batch);
OMDS_API int omds_trainer_eval(omds_trainer* tr, int which, float* mse_out, float* pred_out);
OMDS_API int omds_trainer_get_optimizer_state(omds_trainer* tr, float* const* mW, float* const* mb, float* const* vW, float* const* vb,
                                              int64_t* step);
OMDS_API int omds_trainer_set_optimizer_state(omds_trainer* tr, const float* const* mW, const float* const* mb, const float* const* vW,
                                              const float* const* vb, int64_t step);

/* Measurement: when enabled, launches of the dominant kernel (k_pass1) are bracketed by HIP events on the
 * context stream -- every launch for on == 1, every on-th launch for on > 1 (an event record between two
 * kernels idles the GPU for ~6 us, so throughput runs sample) -- and omds_prof_read returns the summed elapsed
 * ms and the number of bracketed launches since the last omds_prof_reset.                                  */
OMDS_API int omds_prof_enable(omds_ctx* ctx, int on);
OMDS_API int omds_prof_reset(omds_ctx* ctx);
OMDS_API int omds_prof_read(omds_ctx* ctx, double* pass1_ms, int64_t* pass1_launches, int64_t* pass1_rows);
/* Same, with the algorithmic FLOPs of the bracketed launches and the kernel's name. */
OMDS_API int omds_prof_read_ex(omds_ctx* ctx, double* ms, int64_t* launches, double* flops, const char** kernel);
/* Asynchronous form used by bench loops: propagate + cost + weighted update enqueued without
 * host round trips of rollout data; omds_sync waits for the stream.                       */
OMDS_API int omds_sync(omds_ctx* ctx);
/* Number of HIP devices this process can see (0 on a box without a GPU: not an error).  Launchers size their rank count by it
 * (bench.py --gpus N refuses when N exceeds it); no context needed.                                                        */
OMDS_API int omds_device_count(int32_t* count);
OMDS_API int omds_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OMDS_H */
