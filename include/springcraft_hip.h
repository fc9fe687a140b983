/*
 * springcraft_hip.h — C ABI of libspringcraft_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for springcraft's one hot path
 *     C-alpha coordinates -> contact scan -> Kirchhoff / 3x3-block Hessian -> symmetric eigensolve
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference's src/springcraft/).  The reference is pure Python/NumPy and has no FFI; the
 * binding a maintainer would add is a ctypes stub (shown in INTEGRATION.md and implemented
 * in springcraft_amd/_hip.py).
 *
 * Conventions
 *   - plain pointers + sizes only; no C++ / torch types cross this boundary;
 *   - every function returns an int status (SC_OK == 0); no exceptions cross the boundary;
 *     sc_last_error(ctx) returns a human-readable message for the last failure on ctx;
 *   - "host" entry points take host pointers and do their own transfers (they synchronise, and
 *     return the errors of their own call);
 *     "sc_dev_*" and sc_batch_plan_assemble_f64 take device pointers valid on the context's device
 *     and only enqueue work on the context's stream: descriptor tables travel through a pinned
 *     staging arena of the context, and what a solve can only find out on the device (a NaN / Inf
 *     entry in an input matrix, a failed QL iteration) is reported by the next sc_ctx_synchronize.
 *     One exception: a two-stage solve of a latency-bound batch (batch * n / 128 <= 2800) waits for
 *     its persistent bulge chase, whose control block decides whether the chase is complete;
 *   - all matrices are float64; Kirchhoff is (n,n), Hessian (3n,3n), C order, exactly as
 *     numpy returns them in the reference (interaction.py:48,107-109);
 *   - eigenvectors are returned "rows = modes": V[i*n + c] is component c of mode i, the
 *     layout of `eig_vectors` in nma.py:63.
 */
#ifndef SPRINGCRAFT_HIP_H
#define SPRINGCRAFT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------------------- */
#define SC_OK 0
#define SC_ERR_INVALID_ARG 1   /* bad shape / null pointer / bad enum (-> ValueError)        */
#define SC_ERR_INDEX 2         /* patch index out of range (-> IndexError, forcefield.py:953) */
#define SC_ERR_SELF_PAIR 3     /* contact_pair_on with i == j (-> ValueError, interaction.py:210) */
#define SC_ERR_NO_DEVICE 4     /* no HIP device / wrong architecture                          */
#define SC_ERR_HIP 5           /* a HIP runtime call failed                                   */
#define SC_ERR_NOMEM 6         /* device allocation failed                                    */
#define SC_ERR_NOCONV 7        /* eigensolver failed to converge                              */

/* ---- force-field descriptor ---------------------------------------------------------
 * Device-side restatement of ForceField.force_constant (forcefield.py:67-94) for the
 * force fields whose constants depend only on the distance:
 *   SC_FF_INVARIANT       gamma = 1                       (forcefield.py:264-289)
 *   SC_FF_HINSEN          d = max(sqrt(d2), 2.9); d < 4 ? 860 d - 2390 : 1.28e6 d^-6
 *                                                          (forcefield.py:292-330)
 *   SC_FF_PARAMETER_FREE  gamma = 1 / d2                  (forcefield.py:333-366)
 * TabulatedForceField has its own descriptor (SC_FF_TABULATED below).  Anything else (user subclasses,
 * nested patches) goes through the *_from_pairs entry points: the library returns the ordered pair list + squared distances, the host evaluates
 * force_constant() and hands gamma[k] back.
 */
#define SC_FF_INVARIANT 0
#define SC_FF_HINSEN 1
#define SC_FF_PARAMETER_FREE 2
#define SC_FF_TABULATED 3

/* SC_FF_TABULATED: device-side restatement of TabulatedForceField (forcefield.py:369-533).  gamma(i,j,d2) is
 * looked up in one of three float32 tables [type_i][type_j][bin] (C order (20,20,n_bins), the reference keeps
 * them in float32, forcefield.py:889-891):  `bonded` when the two atoms are consecutive C-alpha of one chain
 * (bonded_next[min(i,j)] and |i-j| == 1, forcefield.py:470-473,504-506), else `intra_chain` when
 * chain[i] == chain[j], else `inter_chain`; bin = number of squared edges < d2
 * (np.searchsorted(edges**2, d2), forcefield.py:521).  All pointers are HOST pointers. */
typedef struct sc_tab_desc {
  int32_t n_bins;             /* >= 1 */
  int32_t reserved;
  const double* edges_sq;     /* (n_bins,) squared right bin edges; may be NULL when n_bins == 1 */
  const float* bonded;        /* (20,20,n_bins) */
  const float* intra_chain;   /* (20,20,n_bins) */
  const float* inter_chain;   /* (20,20,n_bins) */
  const int32_t* atom_type;   /* (n_atoms,) amino-acid index 0..19 (alphabetical by one-letter code) */
  const int32_t* chain;       /* (n_atoms,) integer chain label */
  const uint8_t* bonded_next; /* (n_atoms,) 1: atom i is peptide-bonded to atom i+1 */
} sc_tab_desc;

typedef struct sc_ff_desc {
  int32_t kind;       /* SC_FF_* */
  int32_t has_cutoff; /* 0: cutoff_distance is None -> every i != j is a contact (interaction.py:151-153) */
  double cutoff;      /* cutoff_distance in Angstrom (informational) */
  double cutoff_sq;   /* cutoff_distance**2 evaluated by the host in float64 (interaction.py:166) */
  const sc_tab_desc* tab; /* SC_FF_TABULATED only, else NULL */
} sc_ff_desc;

/* ---- contact patches (ForceField.contact_shutdown / contact_pair_off / contact_pair_on,
 * forcefield.py:96-110; applied in the order of _patch_adjacency_matrix, interaction.py:193-213).
 * All pointers are host pointers and may be NULL when the matching count is 0.
 * on_force_constants (PatchedForceField, forcefield.py:183-226): NULL -> switched-on pairs
 * use the base force field's constant; otherwise one constant per pair_on row (a value of
 * exactly -1 means "no override", the sentinel of forcefield.py:213-224).
 * base_cutoff_masks_gamma: 1 -> pairs farther than the cutoff that are in contact only because
 * of pair_on get gamma = 0 unless overridden (PatchedForceField.force_constant,
 * forcefield.py:184-196); 0 -> the base force constant is evaluated at any distance. */
typedef struct sc_patch_desc {
  int64_t n_shutdown;
  const int64_t* shutdown; /* (n_shutdown,) atom indices */
  int64_t n_pair_off;
  const int64_t* pair_off; /* (n_pair_off, 2) */
  int64_t n_pair_on;
  const int64_t* pair_on;  /* (n_pair_on, 2) */
  const double* on_force_constants; /* (n_pair_on,) or NULL */
  int32_t base_cutoff_masks_gamma;
  int32_t reserved;
} sc_patch_desc;

/* ---- context ---------------------------------------------------------------------------
 * One context = one device + one stream + a cached device workspace.  Contexts are not
 * thread-safe; use one per host thread (the reference is single-threaded Python). */
typedef struct sc_ctx sc_ctx;

int sc_ctx_create(int device, sc_ctx** out);
/* Same, but enqueue on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int sc_ctx_create_on_stream(int device, void* hip_stream, sc_ctx** out);
void sc_ctx_destroy(sc_ctx* ctx);
const char* sc_last_error(sc_ctx* ctx);
/* Page-locked host memory for results (no reference counterpart: plumbing of the boundary).  The eigenvectors of one
 * N = 2000 ANM are 288 MB; copied into fresh pageable memory they take 17-25 ms, into page-locked memory 5 ms.  The Python
 * host (springcraft_amd/_hip.py:host_array) builds the NumPy arrays it returns on such blocks and keeps a bounded pool of
 * the ones whose arrays are gone.  Any host pointer is accepted by the entry points above; these blocks are merely faster.
 * sc_host_alloc: SC_ERR_NOMEM when the runtime refuses (the caller then uses ordinary memory), SC_ERR_INVALID_ARG for 0 bytes. */
int sc_host_alloc(size_t bytes, void** out);
int sc_host_free(void* p);

/* Block until everything enqueued on the context's stream has finished.  Returns SC_ERR_NOCONV (LinAlgError in the
 * Python layer, what np.linalg.eigh raises at nma.py:61) if a device-pointer eigensolve enqueued since the last call
 * met a matrix with a NaN / Inf entry -- that matrix is solved as the zero matrix and its eigenvalues are returned as
 * NaN, the other matrices of its batch are unaffected -- or a tridiagonal QL iteration that did not converge; the
 * condition is reported once. */
int sc_ctx_synchronize(sc_ctx* ctx);
/* Library / device identification for logs: fills `buf` with e.g. "gfx950 AMD Instinct MI355X, 256 CUs". */
int sc_device_info(sc_ctx* ctx, char* buf, size_t buflen);

/* ---- contact scan (replaces interaction.py:149-178: adjacency + np.where) ---------------
 * counts[i] = number of contacts of atom i (row sums of the adjacency matrix, int64, exact);
 * *n_pairs = total number of directed pairs k. */
int sc_contacts(sc_ctx* ctx, const double* coord, int64_t n_atoms, const sc_ff_desc* ff,
                const sc_patch_desc* patch, int64_t* counts, int64_t* n_pairs);

/* Ordered pair list: pairs is (k,2) int64, sorted by i then j, holding (i,j) and (j,i)
 * (np.where order, interaction.py:177-178).  sq_dist (k,) may be NULL; it is the reference's
 * (dx*dx + dy*dy) + dz*dz with separately rounded products (interaction.py:184).
 * `capacity` is the number of rows the caller allocated; fails with SC_ERR_INVALID_ARG when
 * the scan finds more. */
int sc_pairs(sc_ctx* ctx, const double* coord, int64_t n_atoms, const sc_ff_desc* ff,
             const sc_patch_desc* patch, int64_t capacity, int64_t* pairs, double* sq_dist,
             int64_t* n_pairs);

/* ---- assembly (replaces compute_kirchhoff interaction.py:14-54 and compute_hessian :57-111)
 * inv_sqrt_mass: NULL, or (n_atoms,) 1/sqrt(m_i): the matrix is multiplied element-wise by
 * outer(w, w) as GNM.kirchhoff / ANM.hessian do (gnm.py:85-87,104-105; anm.py:89-94,112-113). */
int sc_kirchhoff_f64(sc_ctx* ctx, const double* coord, int64_t n_atoms, const sc_ff_desc* ff,
                     const sc_patch_desc* patch, const double* inv_sqrt_mass, double* kirchhoff);
int sc_hessian_f64(sc_ctx* ctx, const double* coord, int64_t n_atoms, const sc_ff_desc* ff,
                   const sc_patch_desc* patch, const double* inv_sqrt_mass, double* hessian);

/* Host-callback path: the caller supplies the ordered pair list and gamma[k] (any Python
 * ForceField.force_constant).  Asymmetric gamma is honoured exactly as the reference does:
 * off-diagonal (i,j) from gamma(i,j), diagonal = -sum over the first index (interaction.py:52,104). */
int sc_kirchhoff_from_pairs_f64(sc_ctx* ctx, int64_t n_atoms, const int64_t* pairs, int64_t k,
                                const double* gamma, double* kirchhoff);
int sc_hessian_from_pairs_f64(sc_ctx* ctx, const double* coord, int64_t n_atoms,
                              const int64_t* pairs, int64_t k, const double* gamma,
                              double* hessian);

/* ---- dense symmetric eigensolve (replaces np.linalg.eigh at nma.py:61) -------------------
 * a: (n,n) symmetric, only the lower triangle is read (UPLO='L', numpy's default); not modified.
 * w: (n,) ascending eigenvalues.  v: NULL (values only) or (n,n), rows = modes (nma.py:63). */
int sc_eigh_f64(sc_ctx* ctx, const double* a, int64_t n, double* w, double* v);

/* Partial spectrum (no reference counterpart: np.linalg.eigh always returns all n pairs; this is the path of
 * BASELINE config 5, "lowest 100 modes only").  Eigenvalues with ascending index il..iu (0-based, inclusive):
 * w: (m,), v: NULL or (m,n) rows = modes, m = iu - il + 1.  Bisection + inverse iteration on the tridiagonal
 * matrix, back-transformation of the selected vectors only. */
int sc_eigh_range_f64(sc_ctx* ctx, const double* a, int64_t n, int64_t il, int64_t iu, double* w, double* v);

/* Hermitian pseudo-inverse, replaces np.linalg.pinv(M, hermitian=True, rcond=1e-6) in the covariance / matrix
 * properties (anm.py:114-117,132-136; gnm.py:107-110,125-131): eigendecomposition on the device, then
 * (U * s) U^T with s_i = 1/w_i where |w_i| > rcond * max|w| and 0 elsewhere (one f64-MFMA GEMM).
 * a: (n,n) symmetric, lower triangle read, not modified; out: (n,n). */
int sc_pinvh_f64(sc_ctx* ctx, const double* a, int64_t n, double rcond, double* out);

/* Fused: coordinates -> Hessian (device) -> eigenpairs, no host round trip of the matrix.
 * Replaces ANM(coord, ff).eigen() (anm.py:150-167 -> nma.py:29-63) for built-in force fields. */
int sc_anm_eigen_f64(sc_ctx* ctx, const double* coord, int64_t n_atoms, const sc_ff_desc* ff,
                     const sc_patch_desc* patch, const double* inv_sqrt_mass, double* w, double* v);
int sc_gnm_eigen_f64(sc_ctx* ctx, const double* coord, int64_t n_atoms, const sc_ff_desc* ff,
                     const sc_patch_desc* patch, const double* inv_sqrt_mass, double* w, double* v);
/* Same, partial spectrum il..iu of the 3n x 3n Hessian (w: (m,), v: NULL or (m, 3n)). */
int sc_anm_eigen_range_f64(sc_ctx* ctx, const double* coord, int64_t n_atoms, const sc_ff_desc* ff,
                           const sc_patch_desc* patch, const double* inv_sqrt_mass, int64_t il, int64_t iu,
                           double* w, double* v);

/* ---- device-resident / batched entry points (bench + multi-structure sharding) ------------
 * All pointers are device pointers on the context's device.  Work is enqueued on the context's
 * stream; nothing synchronises (see the conventions at the top for the one exception and for
 * how errors found on the device are reported). */

/* d_coord: (batch, n_atoms, 3) f64.  d_matrix: (batch, dim*n_atoms, dim*n_atoms) f64, dim = 1
 * (Kirchhoff) or 3 (Hessian).  No patches on this path. d_inv_sqrt_mass: NULL or (batch, n_atoms). */
int sc_dev_kirchhoff_f64(sc_ctx* ctx, const double* d_coord, int64_t n_atoms, int64_t batch,
                         const sc_ff_desc* ff, const double* d_inv_sqrt_mass, double* d_matrix);
int sc_dev_hessian_f64(sc_ctx* ctx, const double* d_coord, int64_t n_atoms, int64_t batch,
                       const sc_ff_desc* ff, const double* d_inv_sqrt_mass, double* d_matrix);

/* ---- batches of DIFFERENT structures: sizes, force fields (incl. SC_FF_TABULATED) and patches per structure --------
 * The reference models one arbitrary structure per object (anm.py:62-63, gnm.py:58-59) with any force field
 * (forcefield.py:117-261 PatchedForceField, :369-533 TabulatedForceField) and optional mass weighting
 * (anm.py:89-94,112-113).  A plan stages everything but the coordinates once: every structure gets a slot of one common
 * matrix order (`order`, 0 = dim * the largest atom count), so that ONE sc_dev_eigh_f64(ctx, d_matrix, order, count, ..)
 * solves the whole batch.  A slot holds diag(M, D): the structure's matrix M in its leading dim * n_atoms rows / columns
 * and a diagonal pad D whose entries lie above every eigenvalue of M (between 2 and 4 times its largest absolute row
 * sum).  The blocks never mix, so of the slot's ascending eigenpairs the first dim * n_atoms are the structure's, with
 * its eigenvector components in the leading dim * n_atoms columns.
 * All descriptor pointers are HOST pointers and need not outlive sc_batch_plan_create.  For SC_FF_TABULATED the
 * per-atom arrays of ff->tab are the structure's own; parameter tables are uploaded once per distinct host array. */
typedef struct sc_structure_desc {
  int64_t n_atoms;
  const sc_ff_desc* ff;
  const sc_patch_desc* patch; /* NULL: none */
} sc_structure_desc;
typedef struct sc_batch_plan sc_batch_plan;
int sc_batch_plan_create(sc_ctx* ctx, int dim, const sc_structure_desc* structures, int64_t count, int64_t order,
                         sc_batch_plan** out);
/* d_coord: (sum n_atoms, 3) f64, the structures back to back.  d_inv_sqrt_mass: NULL or (sum n_atoms,).
 * d_matrix: (count, order, order) f64.  Enqueued on the context's stream, nothing synchronises. */
int sc_batch_plan_assemble_f64(sc_batch_plan* plan, const double* d_coord, const double* d_inv_sqrt_mass,
                               double* d_matrix);
/* Host-callback force fields for a whole batch: any Python ForceField.force_constant (forcefield.py:67-94; the use
 * doc/advanced.rst:23-70 documents and tests/test_interaction.py:92-116 exercises), which compute_kirchhoff /
 * compute_hessian evaluate on the ordered pair list (interaction.py:49,96).  The plan's descriptors then only drive
 * the contact scan (cutoff + adjacency patches, interaction.py:149-178):
 *   1. sc_batch_plan_contacts: n_pairs[count] (host) = directed contacts per structure;
 *   2. sc_batch_plan_pairs: pair lists of ALL structures in one launch, back to back -- per structure in np.where order
 *      (sorted by i then j, both directions, interaction.py:177-178), LOCAL atom indices; pair_off[count + 1] (host) =
 *      first row of every structure; sq_dist (may be NULL) as in sc_pairs; `capacity` rows were allocated;
 *   3. the caller evaluates gamma[k] per structure;
 *   4. sc_batch_plan_fill_from_pairs_f64: all padded slots from pairs + gamma in one pass over the batch.  Asymmetric
 *      gamma is honoured as the reference does: element (i, j) from gamma(i, j), diagonal (blocks) = minus the sums
 *      over the FIRST index (interaction.py:50-52,103-104); d_inv_sqrt_mass as in sc_batch_plan_assemble_f64.
 * pairs / sq_dist / gamma / pair_off / n_pairs are HOST pointers, d_coord / d_matrix device pointers; all three calls
 * synchronise the context's stream (the constants come from the host in between anyway). */
int sc_batch_plan_contacts(sc_batch_plan* plan, const double* d_coord, int64_t* n_pairs);
int sc_batch_plan_pairs(sc_batch_plan* plan, const double* d_coord, int64_t capacity, int64_t* pairs, double* sq_dist,
                        int64_t* pair_off);
int sc_batch_plan_fill_from_pairs_f64(sc_batch_plan* plan, const double* d_coord, const int64_t* pairs,
                                      const int64_t* pair_off, const double* gamma, const double* d_inv_sqrt_mass,
                                      double* d_matrix);
int64_t sc_batch_plan_order(const sc_batch_plan* plan);
/* Must be destroyed before its context. */
void sc_batch_plan_destroy(sc_batch_plan* plan);

/* Batched eigensolve of `batch` independent (n,n) symmetric matrices.
 * d_a: (batch,n,n), lower triangle read, DESTROYED (used as workspace).
 * d_w: (batch,n).  d_v: NULL or (batch,n,n) rows = modes. */
int sc_dev_eigh_f64(sc_ctx* ctx, double* d_a, int64_t n, int64_t batch, double* d_w, double* d_v);

/* Partial spectrum of `batch` matrices: d_w (batch, m), d_v NULL or (batch, m, n). d_a is destroyed. */
int sc_dev_eigh_range_f64(sc_ctx* ctx, double* d_a, int64_t n, int64_t batch, int64_t il, int64_t iu,
                          double* d_w, double* d_v);

/* Tridiagonalisation path of the eigensolver: -1 automatic (default: two-stage when n >= 512 and
 * batch * n^2 >= max(2e7, 1e4 n), else one-stage), 0 always one-stage, 1 two-stage whenever n >= 256.  Both give the
 * same eigenpairs to rounding (|dw| ~ 1e-14 |w|max); the choice only affects speed. */
int sc_ctx_set_two_stage(sc_ctx* ctx, int mode);

/* Bytes of device workspace sc_dev_eigh_f64 will hold for (n, batch) (allocated lazily, cached). */
int64_t sc_eigh_workspace_bytes(int64_t n, int64_t batch, int want_vectors);

/* Per-phase device timings (ms, HIP events on the context's stream) of the most recent
 * sc_dev_eigh_f64 when profiling was enabled with sc_ctx_set_profiling(ctx, 1):
 * out[0]=tridiagonalisation, out[1]=tridiagonal eigensolver, out[2]=back-transformation, and
 *   one-stage path (out[5] == 0): out[3]=SYMV kernels only (sum), out[4]=SYR2K kernels only (sum);
 *   two-stage path (out[5] > 0):  out[3]=stage 1 (band reduction), out[4]=stage 2 (bulge chasing),
 *                                 out[5]=the fused stage-2 back-transformation kernel alone. */
int sc_ctx_set_profiling(sc_ctx* ctx, int enabled);
int sc_last_eigh_timings(sc_ctx* ctx, double* out6);
/* Summed device time (ms) of one kernel group of the most recent profiled eigensolve, by name.  Two-stage path:
 * "panel_qr", "symm" (X = A22 V), "syr2k" (trailing update), "bulge", "dia_tfactor", "dc" (tridiagonal divide & conquer),
 * "dc_gemm" (its merge GEMMs), "bt2" (stage-2 back-transformation), "bt1_w" / "bt1_update" (the two GEMMs of the stage-1
 * back-transformation).  One name is not a time: "dc_gemm_gflop" = 1e9 flops those merge GEMMs executed (2 m n k summed on
 * the device over their records: the sizes depend on the deflation and exist nowhere else).
 * Unknown names (or phases the last solve did not run): SC_ERR_INVALID_ARG, *ms = 0. */
int sc_last_eigh_phase_ms(sc_ctx* ctx, const char* name, double* ms);

/* Event counters of the context since it was created (monitoring; nothing in the reference corresponds).  Since round 6
 * the persistent kernels' outcomes are collected on the device while solves are being enqueued; this call waits for the
 * context's stream and reads them (as sc_ctx_synchronize does).  Names:
 *   "chase_launches"   persistent bulge chases started (two-stage path): one launch for the whole stage
 *   "chase_pair_launches"   of those, how many ran in the pair form (two sweeps per workgroup through LDS, the form for
 *                      batches that are bound by memory traffic; smaller ones take one sweep per workgroup)
 *   "chase_timeouts"   of those, how many ran into the bound of an inter-workgroup wait -- expected to stay 0; the
 *                      solve is still finished correctly, on the device, by the take-over launch behind the chase
 *                      ("chase_resumed" counts every such take-over, "chase_incomplete" a chase that ended without a
 *                      flag but with sweeps left), and the context stops using the persistent form
 *   "chase_sweeps"     sweeps the persistent chases finished themselves
 *   "stepwise_chases"  bulge chases that ran as per-wavefront launches from the start
 *   "chase_xcd_min" / "chase_xcd_max"   workgroups per XCD in the most recent persistent chase
 *   "chase_wait_matrix" / "_sweep" / "_task"   where the last timed-out wait stood (-1: never)
 *   "chase_pair_fallbacks"   pair launches the device refused (dynamic LDS) and that were re-issued in the one-sweep form
 *   "xcd_count"        XCDs of the device as a probe launch saw them (0: not probed yet)
 *   "gemm3_launches"   launches the role-split persistent GEMM (k_gemm3) took
 *   "symm3_launches"   launches of the band reduction's symmetric product X = A22 V as one role-split kernel (k_symm3)
 *   "resident_launches" / "resident_takeovers"   one-stage reductions of one matrix done by the single launch that keeps
 *                      its rows in LDS (k_sytrd_resident), and those of them its take-over kernel had to do instead
 *                      ("resident_rollcall_failures": because its workgroups were not all resident in time;
 *                      "resident_lost_waits": because a wait between them ran into its bound; "resident_lost_at" = step << 32 | workgroup
 *                      of that wait, resp. the workgroups that had arrived when the roll call was given up)
 *   "panel_coop_launches"   panel factorisations by the cooperative kernel (k_panel_coop: several workgroups of one launch)
 *   "panel_coop_timeouts"   panels (per matrix) the cooperative kernel gave up on -- a wait between its workgroups ran
 *                      into its bound; expected to stay 0 -- and the take-over launch behind it factored instead: the
 *                      solve is correct, the context keeps to the chunked panel launches afterwards (until round 5 the
 *                      solve returned SC_ERR_NOCONV)
 * Unknown names: SC_ERR_INVALID_ARG, *value = 0. */
int sc_ctx_get_counter(sc_ctx* ctx, const char* name, int64_t* value);

/* ---- device-resident eigenpairs and their consumers ----------------------------------
 * An sc_modes object holds all n eigenvalues and eigenvectors of one model's Kirchhoff (dim 1) or Hessian
 * (dim 3) matrix in device memory, so that the quantities the reference derives from nma.eigen
 * (nma.py:66-105 frequencies, :108-184 mean_square_fluctuation, :187-230 bfactor, :233-359 dcc, :476-524 prs)
 * are computed without solving again and without moving the (n, n) eigenvector matrix over PCIe.
 * It belongs to the context it was created with and must be destroyed before that context. */
typedef struct sc_modes sc_modes;

/* Assemble the ANM Hessian (dim 3) / GNM Kirchhoff matrix (dim 1) from coordinates and solve it, all on device
 * (same arguments as sc_anm_eigen_f64 / sc_gnm_eigen_f64). */
int sc_modes_from_coord(sc_ctx* ctx, const double* coord, int64_t n_atoms, int dim, const sc_ff_desc* ff,
                        const sc_patch_desc* patch, const double* inv_sqrt_mass, sc_modes** out);
/* Solve a host matrix (n, n) (lower triangle read, as numpy.linalg.eigh at nma.py:61); n must be a multiple of dim. */
int sc_modes_from_matrix(sc_ctx* ctx, const double* a, int64_t n, int dim, sc_modes** out);
void sc_modes_destroy(sc_modes* modes);
int64_t sc_modes_order(const sc_modes* modes);
/* Copy out: w (n) ascending, v (n, n) rows = modes (either may be NULL). */
int sc_modes_get(sc_modes* modes, double* w, double* v);
/* out (n / dim): sum over the k listed modes of v^2 / w, summed over the dim components of every atom. */
int sc_modes_msf(sc_modes* modes, const int64_t* mode_idx, int64_t k, double* out);
/* out (n / dim, n / dim): sum over the listed modes of <v_a, v_b> / w; norm != 0 divides by sqrt(c_aa c_bb). */
int sc_modes_dcc(sc_modes* modes, const int64_t* mode_idx, int64_t k, int norm, double* out);
/* ANM only. out (n / 3, n / 3) row-major: sums of the squared 3x3 blocks of pinv(H, rcond) (numpy hermitian
 * rule); norm != 0 divides row a by out[a, a]. */
int sc_modes_prs(sc_modes* modes, double rcond, int norm, double* out);

#ifdef __cplusplus
}
#endif
#endif /* SPRINGCRAFT_HIP_H */
