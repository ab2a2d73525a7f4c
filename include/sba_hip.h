/*
 * sba_hip.h -- C ABI of libsba_hip.so, the MI355X (gfx950) sparse bundle-adjustment engine.
 *
 * The reference (JohnsonLabJanelia/laserCalib) is pure Python: there is no FFI to mirror, so each
 * entry point below names the reference *Python* interface it stands in for.  The host-side mirror
 * (lasercalib_amd/pySBA.py) binds these with ctypes; INTEGRATION.md shows the binding a reference
 * maintainer would add.
 *
 * Conventions: extern "C", POD structs, plain pointers + sizes, no exceptions cross the boundary.
 * Every function returns 0 on success or a negative sba_status; sba_last_error() gives the text.
 * All array pointers are CALLER-OWNED HOST memory unless a name ends in _dev.  Host arrays at the
 * boundary are float64 / int64 exactly as the reference's numpy arrays are (pySBA.py:28-59),
 * whatever dtype the device computes in.
 *
 * Camera row layout (11 doubles): [rotvec(3), t(3), f, k1, k2, cx, cy]      (pySBA.py:31-35)
 *   (13 doubles with SBA_CAM_RADIAL_TANGENTIAL: [rotvec(3), t(3), f, k1, k2, p1, p2, cx, cy]; P below = 11 or 13)
 * Parameter vector x:  [cams.ravel() (P*C), points.ravel() (3*N)]            (pySBA.py:138)
 * Residual vector:     interleaved [u0,v0,u1,v1,...] in the caller's observation order (pySBA.py:101)
 */
#ifndef SBA_HIP_H
#define SBA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBA_ABI_VERSION 2
#define SBA_CAM_PARAMS 11            /* the reference's camera row (pySBA.py:31-35)                      */
#define SBA_CAM_PARAMS_TANGENTIAL 13 /* [rvec(3), t(3), f, k1, k2, p1, p2, cx, cy]: extension, see below */

/* Camera model of a handle.  SBA_CAM_RADIAL is the reference's (pySBA.py:76-89: one focal length, two radial terms).
 * SBA_CAM_RADIAL_TANGENTIAL adds the two tangential (decentering) coefficients p1, p2 in OpenCV's convention
 *   x' = x d + 2 p1 x y + p2 (r2 + 2 x^2),  y' = y d + p1 (r2 + 2 y^2) + 2 p2 x y,  d = 1 + k1 r2 + k2 r2^2
 * -- BASELINE.json configs[4]; the reference itself has no tangential term (its exporter writes zeros for p1, p2,
 * lasercalib/convert_params.py:110), so this model is an extension that cannot be pinned to reference output. */
typedef enum { SBA_CAM_RADIAL = 0, SBA_CAM_RADIAL_TANGENTIAL = 1 } sba_cam_model;

typedef enum {
  SBA_OK = 0,
  SBA_ERR_INVALID = -1,      /* bad argument / shape / index out of range           */
  SBA_ERR_NO_DEVICE = -2,    /* no usable gfx950 device / HIP runtime error at init */
  SBA_ERR_HIP = -3,          /* a HIP call failed                                   */
  SBA_ERR_NONFINITE = -4,    /* residuals are not finite at the initial point (scipy raises
                                ValueError there: scipy/optimize/_lsq/least_squares.py:844-845) */
  SBA_ERR_STATE = -5,        /* call out of order (e.g. solve before upload)        */
  SBA_ERR_UNSUPPORTED = -6
} sba_status;

typedef enum { SBA_F64 = 0, SBA_F32 = 1 } sba_dtype;

/* Which parameters are free -- selects the reference solver variant being replaced. */
typedef enum {
  SBA_MODE_FULL = 0,       /* PySBA.bundleAdjust            (pySBA.py:132-147) cameras + points     */
  SBA_MODE_POINTS_ONLY = 1,/* PySBA.bundleAdjust_nocam      (pySBA.py:237-250) cameras held fixed   */
  SBA_MODE_SHARED_INTR = 2,/* PySBA.bundleAdjust_sharedcam  (pySBA.py:297-325) f,k1,k2 shared      */
  SBA_MODE_CAMS_ONLY_SQ = 3,/* PySBA.bundle_adjustment_camonly (pySBA.py:151-173): cameras free, points fixed,
                               residual = w*(pixel error)^2, x_scale = 1                               */
  SBA_MODE_TRANSFORM_SQ = 4 /* PySBA.bundleAdjust_transform_points_3d (pySBA.py:176-205): one 3x4 affine on all
                               points, cameras fixed, residual = w*(pixel error)^2; points_out receives the
                               transformed points, sba_get_transform the 12 parameters               */
} sba_mode;

typedef struct sba_handle sba_handle;

typedef struct {
  int32_t n_cams;        /* C */
  int32_t n_points;      /* N (points owned by THIS handle; a multi-GPU job gives each rank a slice) */
  int64_t n_obs;         /* M */
  int32_t dtype;         /* sba_dtype: arithmetic type of the per-observation math               */
  int32_t device;        /* HIP device ordinal                                                    */
  void*   stream;        /* hipStream_t to run on (see use_stream)                                */
  int32_t use_stream;    /* 0: ignore `stream`, create a private non-blocking stream.
                            1: run on `stream` exactly as given -- NULL then means the legacy default
                               stream (what torch.cuda.current_stream() is unless the caller changed it),
                               so that the caller's collectives are ordered with the engine's kernels */
  int32_t cam_model;     /* sba_cam_model: 0 = 11-parameter rows (the reference), 1 = 13-parameter rows            */
  int32_t reserved[2];
} sba_problem_desc;

typedef struct {
  double  ftol, xtol, gtol;   /* scipy.optimize.least_squares semantics (defaults 1e-8 there;
                                 the reference passes ftol only, pySBA.py:141)                    */
  int64_t max_nfev;           /* <=0: 100*n like scipy (scipy/optimize/_lsq/trf.py:437-438)       */
  int32_t mode;               /* sba_mode                                                         */
  int32_t verbose;            /* 0 silent, 1 summary, 2 per-iteration table (pySBA.py:141 uses 2) */
  int32_t max_iter;           /* <=0: unlimited; bench uses it to run exactly K LM iterations     */
  int32_t always_relinearize; /* bench only: rebuild the normal equations even after a rejection  */
  double  lambda0;            /* <=0: library default                                             */
  int32_t reserved[4];        /* reserved[0] != 0: time each kernel class of every iteration with HIP
                                 events on the solve stream (read back with sba_get_kernel_profile)   */
} sba_lm_opts;

typedef struct {
  double  cost;            /* 0.5*sum r^2 at the returned parameters            */
  double  initial_cost;
  double  optimality;      /* ||J^T r||_inf at the returned parameters          */
  double  step_norm;       /* last trial step 2-norm                            */
  double  lambda;          /* final damping                                     */
  int64_t nfev, njev;      /* residual / Jacobian evaluations, scipy counting   */
  int32_t iterations;      /* LM trial steps taken (accepted + rejected)        */
  int32_t accepted;        /* accepted steps                                    */
  int32_t status;          /* 0 max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 ftol+xtol (scipy codes) */
  int32_t reserved;        /* diagnostic: reduced camera systems the fp32 engine had to factor a second time in f64
                              (its f32-lane Cholesky refused them: non-positive or vanishing pivot); 0 for the fp64 engine */
  double  seconds_total;   /* wall time inside sba_solve_lm                     */
  double  seconds_device;  /* HIP-event time of the iteration loop              */
} sba_lm_report;

/* One row of the per-iteration log (same columns scipy prints with verbose=2,
 * scipy/optimize/_lsq/common.py:545-563). */
typedef struct {
  int32_t iteration; int32_t accepted;
  int64_t nfev;
  double  cost, cost_reduction, step_norm, optimality, lambda, rho;
} sba_lm_iter_log;

/* ---------------------------------------------------------------- library / device */
int         sba_abi_version(void);
int         sba_device_count(void);                 /* usable HIP devices, 0 if none */
const char* sba_last_error(const sba_handle* h);    /* h may be NULL: last create error */

/* ---------------------------------------------------------------- stateless model calls
 * sba_rotate  <-> PySBA.rotate(points, rot_vecs)      (pySBA.py:61-73)   gathered rows, n each
 * sba_project <-> PySBA.project(points, cameraArray)  (pySBA.py:76-89)
 */
int sba_rotate(int device, int dtype, int64_t n, const double* points /*n*3*/,
               const double* rot_vecs /*n*3*/, double* out /*n*3*/);
int sba_project(int device, int dtype, int64_t n, const double* points /*n*3*/,
                const double* cam_rows /*n*11*/, double* uv_out /*n*2*/);
/* the same for either camera model: cam_rows is n x 11 (SBA_CAM_RADIAL) or n x 13 (SBA_CAM_RADIAL_TANGENTIAL) */
int sba_project_model(int device, int dtype, int cam_model, int64_t n, const double* points /*n*3*/,
                      const double* cam_rows /*n*(11|13)*/, double* uv_out /*n*2*/);

/* ---------------------------------------------------------------- problem handle
 * sba_create/sba_upload <-> PySBA.__init__ state (pySBA.py:28-59): observation list + initial x.
 * weights may be NULL (reference default = ones, pySBA.py:56-58).
 * Observations need not be sorted; the library groups them by point internally and returns
 * residuals in the caller's order.
 */
int sba_create(const sba_problem_desc* desc, sba_handle** out);
int sba_upload(sba_handle* h, const double* cams /*C*11*/, const double* points /*N*3*/,
               const double* uv /*M*2*/, const int64_t* cam_idx /*M*/, const int64_t* pt_idx /*M*/,
               const double* weights /*M or NULL*/);
int sba_set_params(sba_handle* h, const double* x /*11C+3N*/);
int sba_get_params(sba_handle* h, double* cams_out /*C*11*/, double* points_out /*N*3*/);
int sba_destroy(sba_handle* h);

/* J^T r blocks of the last linearization (after sba_solve_lm / sba_lm_finish: at the returned point).
 * gc_out: 11*C camera part (THIS handle's observations only), gp_out: 3*N point part.  Either may be NULL. */
int sba_get_gradient(sba_handle* h, double* gc_out, double* gp_out);
/* The 3x4 affine (row-major, 12 doubles) found by the last SBA_MODE_TRANSFORM_SQ solve. */
int sba_get_transform(sba_handle* h, double* theta12_out);

/* sba_residual <-> PySBA.fun(params, ...) (pySBA.py:92-101). x==NULL: use the handle's parameters. */
int sba_residual(sba_handle* h, const double* x, double* r_out /*2M or NULL*/, double* cost_out);

/* Analytic replacement of scipy's 3-point finite-difference Jacobian (scipy/optimize/_numdiff.py:
 * 628-705) for PySBA.fun: per-observation blocks, rows (u,v): Jc[M][2][11], Jp[M][2][3], in the
 * caller's observation order.  Column placement = PySBA.bundle_adjustment_sparsity (pySBA.py:103-118). */
int sba_residual_jacobian(sba_handle* h, const double* x, double* r_out /*2M or NULL*/,
                          double* Jc_out /*M*22*/, double* Jp_out /*M*6*/);

/* sba_solve_lm <-> PySBA.bundleAdjust / _nocam / _sharedcam: the whole Levenberg-Marquardt loop
 * (analytic Jacobian, Schur complement, dense reduced camera solve) runs on the device.
 * log may be NULL; otherwise up to log_capacity rows are written and *log_rows is set. */
int sba_solve_lm(sba_handle* h, const sba_lm_opts* opts, double* cams_out /*C*11*/,
                 double* points_out /*N*3*/, sba_lm_report* report,
                 sba_lm_iter_log* log, int32_t log_capacity, int32_t* log_rows);

/* ---------------------------------------------------------------- phase-level LM (multi-GPU)
 * One rank = one handle holding a slice of the points and all their observations; cameras are
 * replicated.  The caller all-reduces (SUM) the exchange buffer between sba_lm_form_reduced and
 * sba_lm_solve_trial, and all-gathers the per-rank scalars between sba_lm_solve_trial and
 * sba_lm_decide.  Exchange layout (float64, n = 11*C):  [S n*n | rhs n | diagU n | gc n | cost 1].
 * exchange_dev / scalars_dev are DEVICE pointers owned by the caller (e.g. torch tensors), used on
 * the handle's stream.
 */
#define SBA_LM_NSCALARS 8
int64_t sba_lm_exchange_size(const sba_handle* h);            /* number of doubles */
int sba_lm_begin(sba_handle* h, const sba_lm_opts* opts);
int sba_lm_linearize(sba_handle* h);
int sba_lm_form_reduced(sba_handle* h, double* exchange_dev);
int sba_lm_solve_trial(sba_handle* h, const double* exchange_dev, double* scalars_dev /*8*/);
int sba_lm_decide(sba_handle* h, const double* scalars_all_dev /*n_ranks*8*/, int32_t n_ranks,
                  int32_t* status_out /* -1 = continue */, int32_t* accepted_out,
                  sba_lm_iter_log* row_out /* may be NULL */);
/* Non-blocking variant: sba_lm_decide_async only enqueues the decision (every later LM launch turns into
 * a no-op on the device once the solve has terminated, and re-linearization after a rejected step is skipped on the
 * device), so a caller may enqueue several iterations back to back and call sba_lm_poll (one stream sync) now and then.
 * status_out: -1 while running, else the scipy status code.
 * The decision may be enqueued LATER than this call: on the fp32 one-group path it rides in the prologue of the next
 * sba_lm_form_reduced's kernel (or is flushed by sba_lm_poll / sba_lm_finish).  scalars_all_dev must therefore stay
 * allocated and unchanged until the next sba_lm_form_reduced, sba_lm_poll or sba_lm_finish on this handle has been
 * CALLED (stream order takes care of the rest). */
int sba_lm_decide_async(sba_handle* h, const double* scalars_all_dev /* NULL when n_ranks == 1 */, int32_t n_ranks);
int sba_lm_poll(sba_handle* h, int32_t* status_out, int32_t* iterations_out);
int sba_lm_finish(sba_handle* h, double* cams_out, double* points_out, sba_lm_report* report);
/* The iteration loop of sba_solve_lm alone, on a solve begun with sba_lm_begin: iterations are enqueued in batches and polled
 * until the device-side state terminates (tolerances or max_iter of the options given to sba_lm_begin); single-rank handles run
 * linearise -> reduced system -> solve + trial -> decide, handles with a communicator the sharded loop with its exchanges.
 * sba_solve_lm = sba_lm_begin + sba_lm_run + sba_lm_finish.  (bench.py times exactly the K iterations with it: the initial
 * cost evaluation of sba_lm_begin and the gradient + result download of sba_lm_finish are per-solve work, not per-step.) */
int sba_lm_run(sba_handle* h, int32_t* status_out, int32_t* iterations_out);
/* Camera step delta_c (n_cams * params doubles) of the last sba_lm_solve_trial: the solution of the damped reduced camera
 * system the exchange buffer described.  A test hook -- the reduced system may be ANY symmetric positive definite matrix the
 * caller wrote into the exchange buffer, which is how tests/test_gpu_cholesky.py checks every factorisation kernel on its own. */
int sba_lm_get_step(sba_handle* h, double* delta_c_out);
/* Rows of the per-iteration log collected so far (filled in by sba_lm_poll / sba_lm_finish). */
int sba_lm_get_log(sba_handle* h, sba_lm_iter_log* log, int32_t log_capacity, int32_t* log_rows);

/* ---------------------------------------------------------------- opt-in extensions; off by default
 * sba_set_fixed_points: the reference accepts `points3Dfixed` and never uses it (pySBA.py:28,55), so all 7 degrees of freedom of
 *     the similarity gauge float.  fixed_mask[p] != 0 holds point p at its uploaded coordinates: it drops out of the unknowns (no
 *     3x3 block, no step, not counted in x / gradient norms) while its observations keep constraining the cameras.  NULL clears.
 * sba_set_robust_loss: the objective of scipy.optimize.least_squares(loss='huber' | 'soft_l1' | 'cauchy', f_scale), on every
 *     residual component (z = (f / f_scale)^2; huber: rho(z) = z for z <= 1, 2 sqrt(z) - 1 beyond; soft_l1: 2 (sqrt(1 + z) - 1);
 *     cauchy: ln(1 + z); cost = 0.5 f_scale^2 sum rho -- scipy/optimize/_lsq/least_squares.py:189-226), minimised by
 *     iteratively re-weighted Gauss-Newton steps (rows scaled by sqrt(rho')); the reference calls least_squares with the
 *     default linear loss (pySBA.py:141).  sba_residual keeps returning the plain residual vector;
 *     reported costs are the robust ones.  Applies to modes FULL / POINTS_ONLY / SHARED_INTR.
 * Both persist on the handle until changed. */
typedef enum { SBA_LOSS_LINEAR = 0, SBA_LOSS_HUBER = 1, SBA_LOSS_SOFT_L1 = 2, SBA_LOSS_CAUCHY = 3 } sba_loss;
int sba_set_fixed_points(sba_handle* h, const uint8_t* fixed_mask /*N bytes or NULL*/);
int sba_set_robust_loss(sba_handle* h, int32_t loss /*sba_loss*/, double f_scale);

/* ---------------------------------------------------------------- multi-GPU inside the library (RCCL over xGMI)
 * One process per GPU, one handle per process holding a contiguous slice of the points and all their observations
 * (cameras replicated).  After sba_comm_init the handle's sba_solve_lm runs the sharded loop itself: per LM trial ONE
 * ncclAllReduce of the reduced camera system, upper triangle only -- n(n+1)/2 + 3n + 1 doubles, n = P*C -- and ONE
 * ncclAllGather of 8 scalars per rank, both enqueued on the handle's stream between its kernels; every rank solves the
 * same camera system and takes the same accept / reject / terminate decision.  No Python, no torch.distributed in a step.
 * librccl.so is bound with dlopen on first use (single-GPU callers never load it).
 *   sba_comm_get_unique_id: rank 0 creates the 128-byte ncclUniqueId; the caller carries it to the other ranks
 *                           (any out-of-band channel: a file, MPI, torch.distributed.broadcast_object_list ...).
 *   sba_comm_init:          collective over all ranks (ncclCommInitRank on the handle's device).  n_ranks == 1 is allowed
 *                           and exercises the same code path on one GPU.
 * The report of a multi-rank sba_solve_lm holds whole-job cost / optimality / nfev; cams_out is identical on every rank,
 * points_out and sba_residual are the rank's own slice.  The squared-error variants (modes 3, 4) are single-GPU. */
#define SBA_COMM_ID_BYTES 128
int sba_comm_get_unique_id(uint8_t* id_out /*SBA_COMM_ID_BYTES*/);
int sba_comm_init(sba_handle* h, const uint8_t* id /*SBA_COMM_ID_BYTES*/, int32_t rank, int32_t n_ranks);
/* The same sharded loop without RCCL on its critical path: a one-shot exchange through peer-mapped device buffers.  Every rank
 * exports one "area" (sba_ipc_export: hipIpcGetMemHandle of uncached device memory sized for n_ranks), the caller carries the
 * 64-byte handles between the processes (any channel: the Python host uses torch.distributed / gloo), and every rank maps all
 * of them (sba_ipc_attach).  Per LM trial a rank writes its packed reduced system, later its 8 trial scalars, into its own area,
 * raises a flag, waits in a one-wave kernel (bounded: 5 s by default, SBA_IPC_TIMEOUT_S in the environment of sba_ipc_export
 * changes it; then the solve returns SBA_ERR_STATE on that rank and the handle takes no further exchange) for the peers'
 * flags and adds the n_ranks copies in rank order: the same bits on every rank, two or three small launches per exchange
 * instead of an RCCL collective.  Works between processes sharing ONE device (how tests/test_gpu_ipc.py runs the in-library
 * sharded loop on a one-GPU box), between handles of ONE process (ranks as threads: the library finds same-process areas
 * in a table instead of opening their handles; every rank's stream then needs a hardware queue of its own -- a gate kernel must
 * never be queued in front of the publish it waits for: GPU_MAX_HW_QUEUES >= 2 x the ranks of the process, set before HIP starts) and between peer GPUs of one node (EXPERIMENTAL: never run across devices,
 * see DESIGN.md section 6).  Exclusive with sba_comm_init, both ways.  The area is uncached device memory; where that
 * cannot be allocated sba_ipc_export fails (SBA_IPC_ALLOW_CACHED=1 accepts cached memory, valid on one device only).
 * A caller must not destroy a handle while a peer may still be reading its area (barrier first). */
#define SBA_IPC_HANDLE_BYTES 64
int sba_ipc_export(sba_handle* h, int32_t n_ranks, uint8_t* handle_out /*SBA_IPC_HANDLE_BYTES*/);
int sba_ipc_attach(sba_handle* h, int32_t rank, int32_t n_ranks, const uint8_t* handles_all /*n_ranks * SBA_IPC_HANDLE_BYTES, rank order*/);

/* ---------------------------------------------------------------- measurement hooks (bench.py)
 * Runs `reps` launches of one named kernel on the current parameters and returns the mean launch
 * duration in microseconds measured with HIP events on the handle's stream.
 * names: "residual", "resjac" (materialising), "linearize_points", "linearize_cams", "schur",
 *        "backsub".  */
int sba_time_kernel(sba_handle* h, const char* name, int32_t reps, double* mean_us_out);

/* In-loop timing collected while opts.reserved[0] != 0: total microseconds and launch counts per kernel
 * class, slots: 0 linearize_points, 1 linearize_cams (+its reduce), 2 schur, 3 schur_reduce+pack,
 * 4 cholesky_solve, 5 backsub_trial.  Arrays of SBA_PROFILE_SLOTS entries. */
#define SBA_PROFILE_SLOTS 6
int sba_get_kernel_profile(sba_handle* h, double* total_us_out, int64_t* count_out);

#ifdef __cplusplus
}
#endif
#endif /* SBA_HIP_H */
