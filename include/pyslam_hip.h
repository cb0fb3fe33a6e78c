/*
 * pyslam_hip.h -- C ABI of the MI355X-native Gauss-Newton / LM core.
 *
 * The reference (utiasSTARS/pyslam) has no native or FFI boundary: its hot
 * path is pure Python.  This header defines the
 * boundary a maintainer would bind instead; every entry point names the
 * reference code it replaces (paths relative to the reference repository).
 * The binding itself (ctypes) is shown in INTEGRATION.md and implemented in
 * pyslam_amd/_native.py.
 *
 * Conventions
 *   - plain C types only; the table pointers in ps_problem_desc are HOST
 *     pointers unless ps_problem_desc.flags says they are resident in HBM
 *     (PS_DESC_DEVICE_PARAMS / PS_DESC_DEVICE_TABLES); either way
 *     ps_problem_create copies them into the handle's own layout (records
 *     sorted by landmark and by pose), the handle owns that memory and the
 *     caller keeps ownership of its arrays.  ps_set_params / ps_get_params and
 *     the pose part of ps_get_dx take host OR device pointers;
 *   - every function returns 0 on success, <0 on error (message through
 *     ps_last_error()); nothing throws across the ABI;
 *   - all launches go to ONE HIP stream per handle (the `stream` argument of
 *     ps_problem_create, a hipStream_t; NULL = a stream the handle creates);
 *     functions that return host scalars synchronise that stream, the
 *     others only enqueue;
 *   - arithmetic is fp64 throughout (the reference is float64 numpy).
 *
 * Table layout: see pyslam_amd/lowering.py (LoweredProblem).
 */
#ifndef PYSLAM_HIP_H
#define PYSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ps_problem ps_problem;

typedef struct ps_problem_desc {
    int32_t dof;                 /* 6: SE(3) poses, 3: SE(2) poses                           */
    int32_t num_poses;           /* rows of `poses` (12 doubles SE3 / 6 doubles SE2 each)     */
    const double*  poses;        /* [R row-major | t]                                         */
    const int32_t* pose_rid;     /* index in the reduced system, -1 = held constant           */
    int32_t num_points;
    const double*  points;       /* (num_points, 3)                                           */
    const int32_t* point_vid;    /* variable-landmark index, -1 = held constant               */

    /* stereo reprojection blocks (reference residuals/reprojection_residual.py:13-37,
       reprojection_motion_only_residual.py:35-113 lowered onto constant points) */
    int64_t num_obs;
    const int32_t* obs_pose;
    const int32_t* obs_point;
    const double*  obs_uvd;      /* (num_obs, 3)                                              */
    const int32_t* obs_grp;      /* row of obs_groups                                         */
    int32_t num_cams;       const double* cams;        /* (num_cams, 5) cu cv fu fv b  (b = -1: RGB-D camera, third coordinate is z) */
    int32_t num_stiff3;     const double* stiff3;      /* (num_stiff3, 9) 3x3 row-major       */
    int32_t num_obs_groups; const double* obs_groups;  /* (n, 4) cam, stiff, loss id, loss k  */

    /* pose-pose edges (reference residuals/pose_to_pose_residual.py:12-32) */
    int64_t num_edges;
    const int32_t* e_i;
    const int32_t* e_j;
    const double*  e_Tobs_inv;   /* T_2_1_obs^-1, packed like poses                           */
    const int32_t* e_grp;        /* row of edge_groups                                        */
    /* unary pose priors (reference residuals/pose_residual.py:12-27) */
    int64_t num_priors;
    const int32_t* u_i;
    const double*  u_Tobs_inv;
    const int32_t* u_grp;
    int32_t num_stiffd;      const double* stiffd;       /* (n, dof*dof) row-major            */
    int32_t num_edge_groups; const double* edge_groups;  /* (n, 3) stiff, loss id, loss k     */

    /* multi-GPU: pose pairs (reduced indices, i<j) that other landmark shards
       couple, so every rank builds the SAME block pattern for the all-reduce */
    int64_t num_extra_pairs;
    const int32_t* extra_pair_i;
    const int32_t* extra_pair_j;

    /* 0 = every pointer above is a host pointer.  PS_DESC_DEVICE_PARAMS: `poses` and `points` are device pointers (on the
       device the handle is created on) and are copied device to device.  PS_DESC_DEVICE_TABLES: EVERY other non-NULL table
       pointer above is a device pointer; the index and measurement tables come to the host once for the structure pass
       (sorting by landmark / pose, pair lists), the caller needs no host copy of them. */
    uint32_t flags;
} ps_problem_desc;

enum { PS_DESC_DEVICE_PARAMS = 1u, PS_DESC_DEVICE_TABLES = 2u };

/* sizes a binding needs to allocate result buffers */
typedef struct ps_problem_info {
    int32_t dof, num_poses, num_reduced, num_points, num_var_points;
    int64_t num_obs, num_edges, num_priors;
    int64_t reduced_nnzb;        /* blocks in the reduced (Schur) system, both triangles      */
    int64_t num_pairs;           /* off-diagonal Schur contributions (upper triangle)         */
    int64_t reduce_count;        /* doubles in the all-reduce payload [upper(S) | g | cost | flag] */
    int64_t device_bytes;        /* HBM held by the handle                                    */
    int64_t cg_restarts;         /* restarts of the pipelined CG so far: its recurrences broke down (rounding drift out of
                                    range(V^T) on the singular folded system) and the solve went on
                                    from the true residual                                    */
    int64_t cg_kernel_launches;  /* kernels enqueued for iterations of the reduced solve so far (1 per iteration for the
                                    folded CG, 3 or 4 for the explicit two-level PCG, 2 for the classic PCG)            */
    int64_t ldi_solves;          /* reduced solves preconditioned with the lagged dense inverse (option "lagged_inverse")   */
    int64_t ldi_fallbacks;       /* ... that gave the inverse up after "ldi_cap" iterations and ran the standard solver       */
    int64_t ldi_seeds;           /* inverses seeded on the side stream (after a standard solve)                             */
    int64_t xcg_fused_solves;    /* explicit two-level PCG solves set up in the one-launch form (option "xcg_fused")       */
    int64_t xcg_fused_fallbacks; /* ... repeated in the three-launch form after a breakdown of the single-reduction recurrences */
    int64_t cg_persist_solves;   /* folded CG solves run in ONE launch (option "cg_persist", csrc/ps_k_cg_persist.h)          */
    int64_t cg_persist_failures; /* ... whose in-launch exchange timed out: solved again with the launch-per-iteration kernels,
                                    and the one-launch form is not used on the handle any more                              */
    int64_t cg_persist_refused;  /* solves that wanted a one-launch form and ran launch by launch instead because its grid
                                    could not be resident at that moment: one-launch solves of OTHER handles of this process
                                    held the compute units (a form that can never be resident on the handle's stream -- device
                                    size, CU mask, occupancy -- is not even tried and does not count)                        */
    int32_t persist_cus;         /* compute units the solver's stream may use as the core found them (device count, the stream's
                                    CU mask, ROC_GLOBAL_CU_MASK; 0: unknown, e.g. HSA_CU_MASK is set: no one-launch form)     */
    int32_t persist_cus_needed;  /* units the one-launch form of this handle needs resident (0: the system does not fit the form) */
    int64_t landmark_passes_taken_over; /* linearisations that found their landmark pass done: it ran in the previous call's tail
                                    (or in ps_eval_cost) and summed that cost on its way -- options "expect_next", "fuse_cost"   */
    int64_t xcg_persist4_solves; /* of cg_persist_solves: those the four-wave form of the explicit PCG ran (k_xcg_persist4: one wave
                                    per SIMD, its rows of the coarse inverse in registers; option "xcg_persist4")              */
} ps_problem_info;

enum { PS_NUM_STAGES = 12 };
/* stage ids for ps_get_stage_times (ALLREDUCE / PACK: the landmark-sharded iteration driven by the core itself,
   ps_set_collective -- both all-reduces, and k_shard_pack + k_shard_unpack; CG_KERNEL: the launch(es) that run the CG
   iterations of the reduced solve alone -- k_cg_persist / k_xcg_persist or the per-iteration kernels -- without the set-up
   kernels and the recovery that PCG also spans; recorded at profiling level 1 like SCHUR) */
enum { PS_ST_LANDMARK = 0, PS_ST_POSE = 1, PS_ST_SCHUR = 2, PS_ST_EDGES = 3, PS_ST_PCG = 4,
       PS_ST_BACKSUB = 5, PS_ST_UPDATE = 6, PS_ST_COST = 7, PS_ST_TOTAL = 8, PS_ST_CG_KERNEL = 9,
       PS_ST_ALLREDUCE = 10, PS_ST_PACK = 11 };

const char* ps_last_error(void);
int ps_device_count(void);
/* Pay the process-wide one-off costs now (code-object load at the first kernel launch, allocator set-up at the first
   hipMalloc) instead of inside the first solve.  Idempotent; the binding calls it when it first finds a device. */
int ps_warm_up(void);

/* Lower the tables to HBM and precompute the iteration-invariant structure
   (landmark / pose segment lists, Schur pair lists, block pattern).
   Replaces the per-iteration Python bookkeeping of pyslam/problem.py:294-329.
   From 200 000 observations up the observation tables and the Schur pair list are built BY THE DEVICE from the caller's
   columns (csrc/ps_host_build.h; with PS_DESC_DEVICE_TABLES the columns are read in place), bit-identical to the host
   builder that smaller problems use.  Environment, read once here: PS_CREATE_DEVICE = 0 host builder / 1 by size
   (default) / 2 always; PS_SCHUR_MODE, PS_SCHUR_TILE_KB, PS_SCHUR_TILE_MIN_MB, PS_CREATE_KEYS64: list variants the
   parity tests hold against each other.
   THREADS: the host builder (below 200 000 observations, pose graphs, more than 255 observation groups) splits its
   counting sorts over up to 16 std::threads, all joined before this call returns; no other entry point of this header
   creates a host thread, and none is alive between calls.  (SURVEY section 8b asked for "no internal host threads": true
   of the iteration path; at create time the device build is the single-threaded route.) */
int ps_problem_create(const ps_problem_desc* desc, void* stream, ps_problem** out);
int ps_problem_destroy(ps_problem* h);
int ps_get_info(ps_problem* h, ps_problem_info* info);

/* sum_blocks sum_i rho(r_i) at the current parameters -- pyslam/problem.py:110-128
   (include_all_constant=1) or the cost term of problem.py:334,358
   (include_all_constant=0: blocks whose parameters are all constant are skipped).
   SIDE EFFECT (option "fuse_cost" = 1, the default, include_all_constant = 1, every observation on a variable landmark with
   at most 16 observations): the cost is summed by the landmark pass itself, which REWRITES Z, C^-1 and c at the current
   parameters (a linearisation that follows at this point finds its landmark pass done).  If the parameters have moved since the
   last ps_linearize those buffers then belong to the new point, and the staged entry points that read them -- ps_backsub,
   ps_gn_finish, ps_gn_solve_finish[_enqueue], ps_get_landmark_factors -- fail with "no longer belong to the last ps_linearize"
   until ps_linearize is called again.  The same holds after a whole-iteration call (ps_gn_iteration, ps_solve) that expected a
   successor: its tail has run the successor's landmark pass. */
int ps_eval_cost(ps_problem* h, int include_all_constant, double* cost);

/* Residuals + Jacobians + IRLS weights + J^T J assembly + landmark elimination:
   builds the reduced system S dx_p = g in HBM.  Replaces pyslam/problem.py:279-360
   (block.evaluate, loss.weight, sparse.bmat, HT.dot(HT.T), -HT.dot(e)).
   lambda: Marquardt damping added as lambda*diag(J^T J); 0 = the reference's GN. */
int ps_linearize(ps_problem* h, double lambda);

/* Exchange buffer of the landmark-sharded (multi-GPU) iteration: the contiguous payload
   [upper block triangle of S incl. diagonal | g | cost (2) | failure flag] a caller all-reduces (sum)
   between ps_linearize and ps_solve_reduced.  ps_shard_pack gathers it from the reduced system (S is
   exactly symmetric: only one triangle travels), ps_shard_unpack scatters the sum back, mirrors the upper
   blocks into the lower triangle and raises the landmark-failure status on EVERY rank if any shard
   reported a non-positive-definite H_ll (ranks fail together, none is left waiting in a collective).
   Enqueue-only, on the handle's stream.  SURVEY.md section 8e. */
int ps_reduce_buffer(ps_problem* h, void** dev_ptr, int64_t* count);
int ps_shard_pack(ps_problem* h);
int ps_shard_unpack(ps_problem* h);

/* Block-Jacobi PCG on the reduced system; replaces the splinalg.spsolve call of
   pyslam/problem.py:186 together with ps_backsub.  Stops when the preconditioned residual
   norm sqrt(r^T M^-1 r) has dropped by `tol` (scale-invariant; relres_out reports it).
   tol <= 0 (here and wherever this header takes a `pcg_tol` / `tol` of the reduced solve): the DEFAULT -- 1e-12 where the
   reduced system is the Schur complement of a bundle adjustment (variable landmarks), 1e-14 for pose graphs (no landmarks:
   priors of stiffness ~1e6 beside loop closures of ~1 leave cond(M^-1 S) at 1e4-1e5 after the two-level preconditioner, and
   error <= cond x relative residual): what meets |dx - dx_spsolve| <= 1e-8 |dx| without the caller knowing the
   condition number.  pyslam_amd.Options.pcg_tol = None passes 0. */
int ps_solve_reduced(ps_problem* h, double tol, int max_iters, int* iters_out, double* relres_out);

/* Landmark back-substitution dx_l = Hll^-1 (b_l - W^T dx_p). */
int ps_backsub(ps_problem* h);

/* dx in device order [reduced poses (dof each) | variable points (3 each)];
   either pointer may be NULL.  ps_step_norm2 returns ||dx||^2. */
int ps_get_dx(ps_problem* h, double* dx_pose, double* dx_point);
int ps_step_norm2(ps_problem* h, double* norm2);

/* T <- exp(step*xi) T, p <- p + step*dp -- pyslam/problem.py:155-156, 400-409. */
int ps_apply_update(ps_problem* h, double step);

/* best_params snapshot / restore of pyslam/problem.py:163-175. */
int ps_snapshot_params(ps_problem* h);
int ps_restore_params(ps_problem* h);

/* either pointer may be NULL; each may be a host or a device pointer (a torch caller keeps its parameters resident) */
int ps_get_params(ps_problem* h, double* poses, double* points);
int ps_set_params(ps_problem* h, const double* poses, const double* points);

/* One whole Gauss-Newton / LM iteration with a single host synchronisation:
   linearize -> PCG -> back-substitution -> update -> post-step cost.
   cost_out = cost after the step (linesearch != 0, pyslam/problem.py:362-398 with its
   always-full step) or cost at the linearisation point (linesearch == 0, problem.py:192). */
int ps_gn_iteration(ps_problem* h, double lambda, double pcg_tol, int pcg_max_iters,
                    int linesearch, double* cost_out, double* dx_norm_out,
                    int* pcg_iters_out, double* pcg_relres_out);

/* The loop of Problem.solve (reference pyslam/problem.py:130-178) for a problem of ONE variable SE(3) pose observing
   constant landmarks -- config 5, built per frame by pipelines/sparse.py:153-161 -- in ONE launch and one
   synchronisation: the start cost, then iterations until `iterations > max_iters`, ||dx|| < min_update_norm, cost <
   min_cost or the non-decreasing-step rules stop it (best parameters kept and restored as the reference does).
   Options carry the reference's names; linesearch != 0: an iteration's cost is the cost after its (always full) step,
   else the cost of its linearisation point (problem.py:188-192).  cost_history receives the reference's
   _cost_history (at most max_iters + 2 entries), the pose is left updated and, if pose12_out is not NULL, also returned
   ([R row-major | t]: no second copy from the device).
   Returns 0 = solved, 1 = not this kind of problem or history too long for `cap` (nothing done: iterate with
   ps_gn_iteration), <0 = error. */
typedef struct ps_solve_options {
    int32_t max_iters, allow_nondecreasing_steps, max_nondecreasing_steps, linesearch;
    double min_update_norm, min_cost, min_cost_decrease, lm_lambda;
} ps_solve_options;
int ps_motion_only_solve(ps_problem* h, const ps_solve_options* options, double* cost_history, int32_t cap,
                         int32_t* n_history, int32_t* iterations, double* last_dx_norm, double* pose12_out);

/* Start of a NEW solve on a live handle: drop every piece of state the solver carries from one whole-iteration call to
   the next (lagged coarse factors / inverses and their tags, the lagged dense inverse of S, launch-count predictions, the
   cost history), waiting for side-stream work in flight first.  Tables, parameters and options stay.  The next
   ps_gn_iteration behaves exactly like the first one of a freshly created handle -- what Problem.solve (reference
   pyslam/problem.py:130-141: every solve starts from scratch) calls before its first iteration, so that a solve is a
   function of (parameters, options) and not of what the handle did before. */
int ps_reset_solver_state(ps_problem* h);

/* Hash of the HIP sources the loaded library was compiled from (first 16 hex digits of the SHA-256 over include/pyslam_hip.h
   and pyslam_amd/csrc/, the -DPS_BUILD_SHA of __graft_entry__.build(); "unknown" for a hand build): bench.py reports THIS and
   refuses to run on a library that does not match the sources on disk. */
const char* ps_build_sha(void);

/* The loop of Problem.solve (reference pyslam/problem.py:130-178) on a resident problem, in one call: ps_reset_solver_state,
   the start cost (its pass is enqueued in front of the first iteration: no call and no synchronisation of its own), then
   whole iterations (each exactly ps_gn_iteration) until `iterations > max_iters`, ||dx|| < min_update_norm, cost < min_cost or
   the non-decreasing-step rules stop it, the best parameters kept by ps_snapshot_params / ps_restore_params as the reference
   keeps best_params.  cost_history receives the reference's _cost_history (at most max_iters + 2 entries); pcg_iters,
   pcg_relres and iter_ms (host wall clock of every iteration call, ms) receive one entry per iteration, each may be NULL.
   The same statements as pyslam_amd/problem.py: device_solve (which calls this when the device offers it), without the
   interpreter between two iterations.  Returns 0 = solved, 1 = not offered for this handle (landmark-sharded: the caller
   loops with ps_gn_iteration) or `cap` too small, <0 = error. */
int ps_solve(ps_problem* h, const ps_solve_options* options, double pcg_tol, int pcg_max_iters, double* cost_history,
             int32_t cap, int32_t* n_history, int32_t* iterations, double* last_dx_norm, int32_t* pcg_iters,
             double* pcg_relres, double* iter_ms);

/* Second half of an iteration for a landmark-sharded (multi-GPU) caller, after
   ps_linearize -> all-reduce -> ps_solve_reduced: back-substitution, update, cost, ONE
   synchronisation.  Returns this shard's cost and ||dx_pose||^2, ||dx_point||^2 separately
   (poses are replicated, landmarks are not). */
int ps_gn_finish(ps_problem* h, int linesearch, double* cost_out, double* dx_pose_norm2,
                 double* dx_point_norm2);

/* ps_solve_reduced + ps_gn_finish with ONE synchronisation (the tail is enqueued behind the CG
   launches and gated on the device-side convergence flag): the sharded caller's second half. */
int ps_gn_solve_finish(ps_problem* h, double pcg_tol, int pcg_max_iters, int linesearch, double* cost_out,
                       double* dx_pose_norm2, double* dx_point_norm2, int* pcg_iters_out,
                       double* pcg_relres_out);

/* Fully asynchronous variant for the multi-GPU driver: ps_gn_solve_finish_enqueue enqueues the
   reduced solve and the (convergence-gated) tail and leaves this shard's {cost, ||dx_point||^2}
   in the 2-double device buffer of ps_shard_buffer; the caller all-reduces that buffer on the
   same stream and calls ps_gn_result (the only synchronisation).  If *done == 0 the CG needed
   more launches: call enqueue again with first = 0 (it returns 1 on the final, ungated pass). */
int ps_shard_buffer(ps_problem* h, void** dev_ptr);

/* Native collective: hand the core RCCL's ncclAllReduce entry point and an ncclComm_t (created by
   the binding, one rank per GPU).  ps_gn_iteration then runs the whole landmark-sharded iteration
   -- both all-reduces included -- on the handle's stream with a single host synchronisation. */
int ps_set_collective(ps_problem* h, void* nccl_all_reduce_fn, void* nccl_comm);
/* The exchange of the partial reduced systems as an ALL-GATHER OF SEGMENTS inside the core's one-call sharded iteration (round 6;
   needs ps_set_collective first -- the second, two-scalar exchange stays an all-reduce).  With landmarks sharded by first observing
   pose a rank's partial system is non-zero on one band segment of S (C4 on 8 ranks: 3.2 of 23.5 MB): every rank sends
   [tail words | its elements of the packed buffer of ps_reduce_buffer] once (ncclAllGather, `maxlen` doubles per rank) and adds up
   what it receives itself -- every destination element over its contributing ranks in RANK ORDER, the same on every rank, so the
   replicated solve stays bit-identical across ranks.  The plan is the caller's (pyslam_amd/distributed.py: segment_plan, held
   against a dense sum by the CPU tests): `mine` = this rank's element positions (n_mine); `dst` (n_dst) = every position some
   rank touches, `src_ptr` (n_dst + 1) / `src_off` = for each of them the offsets rank * maxlen + 3 + k into the gathered buffer,
   ascending rank.  The three tail words (cost (2), failure flag) are summed over all ranks.  Arrays are copied.
   nccl_all_gather_fn == NULL: back to the sum all-reduce. */
int ps_set_segment_exchange(ps_problem* h, void* nccl_all_gather_fn, int32_t world, int32_t rank, int64_t maxlen,
                            int64_t n_mine, const int64_t* mine, int64_t n_dst, const int64_t* dst, const int64_t* src_ptr,
                            const int64_t* src_off);
int ps_gn_solve_finish_enqueue(ps_problem* h, double pcg_tol, int pcg_max_iters, int linesearch, int first);
int ps_gn_result(ps_problem* h, int* done, double* shard2, double* dx_pose_norm2,
                 int* pcg_iters_out, double* pcg_relres_out);

/* Covariance by columns -- pyslam/problem.py:196-216 (compute_covariance inverts the whole sparse
   precision matrix; get_covariance_block slices it).  ps_covariance_begin linearises at the current
   parameters (lambda = 0) and prepares the reduced solver.  ps_covariance_column then solves
   H x = e_k, e_k the unit vector of component `comp` of reduced pose `index` (kind 0) or of variable
   landmark `index` (kind 1), by the hot path's own Schur elimination, CG and back-substitution:
   x = column k of the covariance, left in the dx buffers (read it with ps_get_dx).  Scales to any
   problem the iteration handles; no dense n x n matrix is ever formed. */
int ps_covariance_begin(ps_problem* h);
int ps_covariance_column(ps_problem* h, int kind, int index, int comp, double tol, int max_iters,
                         int* iters_out, double* relres_out);

/* Parity / debug taps (device -> host). */
int ps_get_reduced_system(ps_problem* h, int32_t* row_ptr, int32_t* col_idx,
                          double* vals, double* g);       /* BSR, dof x dof blocks */
int ps_get_landmark_factors(ps_problem* h, double* cinv /* (nv,6) */, double* c /* (nv,3) */);
int ps_debug_reproj_blocks(ps_problem* h, double* r /* (N,3) */, double* jpose /* (N,18) */,
                           double* jpoint /* (N,9) */);  /* IRLS-scaled, original obs order */
/* The pose-factor kernel's own blocks (IRLS-scaled, as they enter J~^T J~): r~ (F, dof), J~_1 (F, dof, dof) and
   J~_2 (F, dof, dof), F = num_edges + num_priors in table order (edges first).  Edges: the reference's
   PoseToPoseResidual.evaluate (pyslam/residuals/pose_to_pose_residual.py:12-32: r = S log(T_2 T_1^-1 T_obs^-1),
   J_1 = -S Ad(T_2 T_1^-1), J_2 = S); priors: PoseResidual.evaluate (pose_residual.py:12-27: r = S log(T T_obs^-1),
   J = S in j2, j1 = 0).  Runs the production kernel with its tap open; S and g are not touched. */
int ps_debug_factor_blocks(ps_problem* h, double* r, double* j1, double* j2);
/* FNV-1a (64 bit) of the structure tables ps_problem_create left in HBM, in a fixed order: point slots, lobs, lorig,
   lm_ptr, lm_point, pose_of_rid, pitems, pitem_ptr, pobs, pairs, pair work items (XCD order), combine items, combine
   tasks, row_ptr, col_idx, diag_slot (16 words; 0 for a table the problem does not have).  What holds the device
   structure build against the host builder bit for bit (tests/test_gpu_create.py); no reference counterpart. */
int ps_debug_table_checksums(ps_problem* h, uint64_t* out, int capacity, int* count);

/* Tuning knobs (defaults in brackets):
     "pcg_variant"        [1] fused single-launch-per-iteration CG on the block-Jacobi scaled system; 0 = classic two-launch PCG
     "coarse_groups"      [-1 auto] hat-function intervals of the two-level preconditioner, 0 = off
     "coarse_basis"       [1] coarse unknowns are body-frame twists (P_iq = w L_i^T Ad(T_i)); 0 = hats in scaled coordinates
     "coarse_lag"         [1] whole-iteration calls build the two-level system with the previous iteration's coarse factor
     "direct_max_unknowns"[90] reduced systems up to this size are solved by a dense Cholesky instead of CG (0 = never)
     "fused_motion_only"  [1] problems without variable landmarks / pose factors: one launch per iteration
     "cg_explicit"        [1] long sparse chains: apply the two-level preconditioner (k_xcg_*) instead of folding it in
     "coarse_lag_x"       [1] ... and with the previous iteration's basis and X = P L_c^-T too: three set-up launches (k_rows_setup)
     "coarse_refresh_every" [1] explicit two-level PCG: only every k-th lagged set-up takes the newest coarse inverse and starts the
                              next side-stream factorisation (landmark-sharded runs whose iteration is shorter than that factorisation)
     "coarse_auto_hold"   [1] explicit two-level PCG: while the solve has settled (last iteration changed the cost by < 1e-4
                              relative) keep the lagged coarse inverse, for at most 3 set-ups in a row (no assembly, no factorisation);
                              and for as long as the caller linearises at the point (same start cost, same lambda) the inverse in use
                              was formed from.  A call whose lambda is more than a factor of four from the newest inverse's (or zero against non-zero)
                              factors its own A_c (no lag).
     "coarse_adaptive_hold" [1] explicit two-level PCG: keep the lagged coarse inverse (no assembly, no side-stream factorisation) while the
                              last solve with it took at most 3 iterations more than the first one did (at most 8 set-ups in a row)
     "xcg_restrict_fused" [1] explicit two-level PCG: three launches per iteration (restriction in the SpMV epilogue, t by recurrence)
                              instead of four
     "lagged_inverse"     [1] reduced systems of 91 .. "ldi_max_unknowns" [2048; up to 3328: pays from ~7 iterations per solve on] unknowns (folded CG, one GPU, whole-iteration calls):
                              precondition the CG with a dense fp32 inverse of the PREVIOUS iteration's S, kept current on the side
                              stream by one Newton-Schulz step per iteration (two fp32 MFMA GEMMs) and seeded from the two-level
                              operator of the last standard solve; tried while the last step changed the cost by at most
                              "ldi_cost_tol" [0.05] relative, given up (standard solver + re-seed) after "ldi_cap" [12] iterations.
                              It only preconditions: the solution is the current system's at pcg_tol either way.
     "ldi_direct"         [-1] ... seeded by a DIRECT fp64 factorisation of S on a stream of its own instead of Newton-Schulz
                              (-1: pose graphs from the start, any problem after a rejected Newton-Schulz seed; 0 never; 1 always);
                              usable a fixed 2 / 4 / 6 calls later (n <= 400 / 800 / 1 536)
     "ldi_seed_lag" [1], "ldi_seed_steps" [3], "ldi_refresh_its" [7]: schedule of the Newton-Schulz seed / refresh
     "xcg_fused"          [1] explicit two-level PCG: ONE launch per iteration (single-reduction recurrences; the restriction, the coarse
                              product for the nodes a workgroup needs, the prolongation and the SpMV in one kernel, the row's matrix
                              blocks requested before the scalar phase) up to 2 048 poses and 2 048 coarse unknowns, TWO beyond (scalars,
                              t and y = A_c^-1 t once, in k_xcg_f2_coarse); 2: always two; 0: three launches per iteration.
                              A breakdown of the recurrences repeats the solve in the three-launch form
     "band_chol"          [1] explicit two-level PCG: banded factorisation + band substitutions for the coarse inverse when A_c
                              has at most 7 block off-diagonals (chain-like problems); 0: always the dense factorisation
     "solve_horizon"      [-1 unknown] how many more whole-iteration calls the caller's stopping rule allows if the step about to be
                              taken turns out non-decreasing (reference problem.py:163-178); side work that pays back only over
                              several later calls (the lagged inverse's seed) is not started with fewer than three to come
     "expect_next"        [0] the caller will ask for another whole iteration after the coming one unless a stopping rule on ||dx|| or
                              the cost fires (ps_solve sets it for its own loop): with "fuse_cost" the coming call's tail runs the NEXT
                              iteration's landmark pass in place of its cost pass -- every observation evaluated once per iteration;
                              the next call's linearisation takes the pass over if the parameters have not moved since.  Does not
                              invalidate what was computed ahead (every other option does)
     "fuse_cost"          [1] the cost of all blocks summed by the packed landmark pass itself (tails that expect a successor, the start
                              cost of ps_solve, ps_eval_cost) or by a cost-only pass in the same structure: the same number bit for bit;
                              2: in the tails only; 0: the grid-stride cost pass of rounds 1-4 everywhere (another summation order).
                              Needs every observation on a variable landmark with at most 16 observations, else 0 is what runs
     "cg_persist"         [1] the folded two-level CG in ONE launch (csrc/ps_k_cg_persist.h) where the augmented system fits (<= 2 048
                              unknowns, <= 1 024 tasks); 0: one launch per CG iteration.  "cg_persist_spin" [200000]: passes over the
                              in-launch exchange before a workgroup gives up -- or one second, whichever comes first -- (then the solve is
                              repeated launch by launch and the form is not used on the handle any more:
                              ps_problem_info.cg_persist_failures).  The form is only used when its whole grid can be resident on the
                              compute units the handle's stream may use (device count, stream CU mask, occupancy of the kernel:
                              ps_problem_info.persist_cus / persist_cus_needed) and while one-launch solves of other handles of the
                              process leave them free (cg_persist_refused)
     "cg_pipelined"       [0] the one-launch folded CG with pipelined recurrences (the dot products formed while the exchange is in
                              flight): 2 = always (-2.8 % at C3; costs CG iterations and digits on ill-conditioned systems), 1 = only
                              when the previous reduced solve of the handle took at most 32 iterations, 0 = Chronopoulos-Gear
     "xcg_persist"        [1] the explicit two-level PCG of bundle adjustments (long rows, one workgroup per compute unit of the stream: up to 2 048 poses on a whole MI355X) in ONE
                              launch per solve (csrc/ps_k_xcg_persist.h): matrix in registers / LDS, w, partials and records exchanged
                              in-launch; 0: one launch per iteration (k_xcg_fused1).  Time-outs as "cg_persist"
     "xcg_persist4"       [1] that launch as FOUR waves per workgroup (one per SIMD: 512 registers per lane) with the workgroup's rows of
                              the coarse inverse kept in registers (csrc/ps_k_xcg_persist4.h), where the coarse level fits (at most
                              640 coarse unknowns, 44 rows of y per workgroup); 0: always the eight-wave kernel.  Same bits either way
     "pose_xcd"           [1] the pose pass's work items in eight contiguous ranges, one per XCD (0: item = workgroup); speed only
     "lm_packed" [1], "band_part" [1], "band_part_chunk" [0 auto], "sync_refactor" [1], "hold_across_steps" [1]: round-5 kernels and
                              schedules against their predecessors (DESIGN.md sections 0 and 3)
     "cg_force_restart"   [0] tests: end the first pass of a synchronous reduced solve at 1e-4 and restart from the true residual
     "cg_lds", "profile_every", "big_chol", "cg_margin", "pcg_chunk", "cg_split_min_rows", "cg_explicit_min_rows": implementation switches (see ps_set_option in csrc/ps_abi_solver.h)
     "cg_ablate", "schur_ablate", "lm_ablate": timing experiments only (results are wrong under ablation) */
int ps_set_option(ps_problem* h, const char* name, double value);

/* hipEvent stage timers on the handle's stream (the reference has no tracing). */
int ps_set_profiling(ps_problem* h, int enabled);
int ps_get_stage_times(ps_problem* h, double* ms /* PS_NUM_STAGES */, int64_t* counts, int reset);

/* Host-evaluated generic path (user-defined Python residual blocks): dense
   J (m x n, row-major) and r are uploaded, the device forms J^T J, -J^T r and
   solves by Cholesky; optionally returns the inverse (compute_covariance,
   pyslam/problem.py:196-203). */
int ps_dense_normal_solve(const double* J, const double* r, int32_t m, int32_t n,
                          double* dx, double* covariance /* n*n or NULL */);

/* The same path beyond the dense solver's size: J (m x n) and its transpose as CSR in HBM, Jacobi-preconditioned CG
   on J^T J dx = rhs without forming J^T J; rhs = -J^T r (a Gauss-Newton step; `rhs` NULL) or the caller's `rhs`
   (a unit vector: one covariance column; `r` may then be NULL).  Replaces scipy's sparse LU (pyslam/problem.py:186)
   for problems with user-defined blocks / losses / parameters and more than 2048 unknowns.  Stops at a relative
   preconditioned residual of `tol` or after max_iters iterations (the step is returned either way). */
int ps_sparse_normal_solve(int32_t m, int32_t n, const int32_t* j_row_ptr, const int32_t* j_col, const double* j_val,
                           const int32_t* jt_row_ptr, const int32_t* jt_col, const double* jt_val,
                           const double* r, const double* rhs, double tol, int32_t max_iters,
                           double* dx, int32_t* iters_out, double* relres_out);

/* The same system solved DIRECTLY (round 5): J^T J formed dense on the device, blocked multi-workgroup Cholesky, block
 * substitutions, `refine_steps` refinement steps on the residual of the original system.  n <= 8192.  What the host-evaluated path
 * uses between the single-workgroup dense solve (n <= 2048) and the CG above; stands in for scipy.sparse.linalg.spsolve
 * (reference pyslam/problem.py:186).  relres_out: ||rhs - J^T J dx|| / ||rhs||. */
int ps_sparse_normal_direct(int32_t m, int32_t n, const int32_t* j_row_ptr, const int32_t* j_col, const double* j_val,
                            const int32_t* jt_row_ptr, const int32_t* jt_col, const double* jt_val,
                            const double* r, const double* rhs, int32_t refine_steps, double* dx, double* relres_out);

/* The banded factorisation kernels of the explicit two-level PCG's coarse level on their own (csrc/ps_k_band.h: one workgroup
 * walking the block columns; csrc/ps_k_bandpart.h: the partitioned, parallel form): inverse of a symmetric positive definite
 * matrix with `bw` (1..7) block off-diagonals of dof x dof blocks.  `a`: dense, row-major, (ncb dof)^2 doubles on the host, lower
 * triangle read.  chunk_nodes < 0: serial walk; 0: partitioned, automatic chunk size; > 0: interior nodes per chunk (partitioned
 * needs 2 bw - 1 <= 7).  ainv_out: fp32, what the preconditioner keeps.  elapsed_us (or NULL): GPU time of the launches.
 * Stands in for the coarse-level part of scipy.sparse.linalg.spsolve (reference pyslam/problem.py:186); test and measurement entry. */
int ps_debug_band_inverse(const double* a, int32_t ncb, int32_t dof, int32_t bw, int32_t chunk_nodes, float* ainv_out,
                          double* elapsed_us);
/* Stress entry behind DESIGN.md section 3 "side stream" (round 6; test and measurement infrastructure): the coarse level's
 * factorisation kernels run `launches` times on a stream of their own -- lowest priority if `lowprio` -- while, if `aggressor`,
 * streaming copy kernels keep an ordinary stream busy; every output is compared bit for bit with one produced on an idle device.
 * mode 0: serial band walk (k_band_chol + k_band_inverse_rl); 1: partitioned band factorisation (BandPart); 2: dense LDS-resident
 * k_coarse_chol + k_xcg_ainv (ncb dof <= 90); 3: the same with its matrices in global scratch; + 8: the input is produced on the
 * same stream by a copy kernel in front of every factorisation (the buffer holds 2 A before); + 16: with an event recorded in
 * between.  `a` as ps_debug_band_inverse.
 * -> n_diff: launches whose fp32 inverse differs from the reference in any bit; n_pivot: launches that reported a non-positive
 * pivot on this positive definite input.  A kernel that is a function of its input gives 0 / 0 whatever runs beside it. */
int ps_debug_factor_stress(const double* a, int32_t ncb, int32_t dof, int32_t bw, int32_t mode, int32_t launches, int32_t lowprio,
                           int32_t aggressor, int32_t* n_diff, int32_t* n_pivot);

/* Frame-to-frame RANSAC, the step before the motion-only solve in the reference's sparse VO pipeline
   (pyslam/pipelines/sparse.py:148-150).  Stateless; host pointers in, host pointers out.
   ps_ransac_transforms   -- compute_transform_fast (pyslam/pipelines/ransac.py:13-67): `batch` rigid
                             alignments p_2 ~ C p_1 + r of n-point sets (pts: batch x n x 3), 4x4 out.
   ps_ransac_cost         -- FrameToFrameRANSAC.compute_ransac_cost (:153-165): inlier masks
                             (num_hyp x num_pts bytes) and counts of given 4x4 transforms; cam5 = cu cv fu fv b.
   ps_ransac_frame_to_frame -- perform_ransac (:113-151) without the random draw: hypotheses from the
                             caller's sample indices (num_hyp x set_size), scoring of all points, first
                             hypothesis with the most inliers, its transform and inlier mask. */
int ps_ransac_transforms(const double* pts_1, const double* pts_2, int32_t batch, int32_t n, double* T_out);
int ps_ransac_cost(const double* T, int32_t num_hyp, const double* pts_1, const double* obs_2, int32_t num_pts,
                   const double* cam5, double thresh, uint8_t* masks, int32_t* counts);
int ps_ransac_frame_to_frame(const double* pts_1, const double* pts_2, const double* obs_2, int32_t num_pts,
                             const int32_t* sample_idx, int32_t num_hyp, int32_t set_size, const double* cam5,
                             double thresh, double* T_all, int32_t* counts, int32_t* best_index,
                             int32_t* best_count, double* T_best, uint8_t* best_mask);

/* Dense photometric alignment: one SE(3) pose, one residual per reference pixel --
   PhotometricResidualSE3 (pyslam/residuals/photometric_residual.py:38-161) inside Problem's Gauss-Newton
   iteration with element-wise IRLS (pyslam/problem.py:279-360), as the dense VO pipeline runs it per pyramid
   level (pyslam/pipelines/dense.py:157-194).  The pixel tables are what the residual's constructor
   precomputes (:44-81); they are copied to the device at create time. */
typedef struct ps_photo ps_photo;
typedef struct ps_photo_desc {
    int32_t num_pixels;
    const double* pt_ref;        /* num_pixels x 3: triangulated reference points (:80) */
    const double* im_ref;        /* num_pixels: reference intensities */
    const double* im_jac;        /* num_pixels x 2: reference image gradient (dI/du, dI/dv) */
    const double* tri_jac_d;     /* num_pixels x 3: d point / d depth (column 2 of the triangulation Jacobian, :116) */
    int32_t height, width;       /* tracking image */
    const double* im_track;      /* height x width, row-major */
    double cam[5];               /* cu cv fu fv b */
    int32_t cam_type;            /* 0 stereo (u, v, disparity), 1 RGB-D (u, v, depth) */
    int32_t cam_w, cam_h;        /* validity bounds of the camera model (is_valid_measurement) */
    double intensity_covar, depth_covar;   /* stiffness^-2 (:57-58) */
    int32_t loss_id;             /* 0 L2, 1 L1, 2 Cauchy, 3 Huber, 4 Tukey, 5 t-distribution */
    double loss_k;
} ps_photo_desc;

int ps_photometric_create(const ps_photo_desc* desc, void* hip_stream, ps_photo** out);
int ps_photometric_destroy(ps_photo* h);
/* pose = T_track_ref as 12 doubles: R row-major, then t */
int ps_photometric_set_pose(ps_photo* h, const double* pose12);
int ps_photometric_get_pose(ps_photo* h, double* pose12);
/* sum of loss(r) over the valid pixels, and their number (evaluate() compresses invalid pixels away, :106) */
int ps_photometric_eval_cost(ps_photo* h, double* cost, int64_t* num_valid);
/* H = J~^T J~ (6 x 6, row-major), b = -J~^T r~, cost at the current pose (parity / debugging) */
int ps_photometric_normal_equations(ps_photo* h, double* H36, double* b6, double* cost, int64_t* num_valid);
/* One Gauss-Newton iteration: dx = H^-1 b in [translation; rotation] order, then
     split_params == 0: T <- exp(dx) T          (one SE3 parameter, liegroups perturb)
     split_params == 1: R <- exp(dx[3:6]) R, t += dx[0:3]   (separate SO3 / translation parameters, dense.py:185)
   cost: after the step when linesearch != 0 (Problem's degenerate line search always lands on the full step,
   pyslam/problem.py:362-398), else the cost of the linearisation point (:188-192). */
int ps_photometric_iteration(ps_photo* h, int32_t split_params, int32_t linesearch, double* dx6, double* cost);

#ifdef __cplusplus
}
#endif
#endif /* PYSLAM_HIP_H */
