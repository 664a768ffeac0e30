/*
 * liblynxhip -- C ABI of the MI355X (gfx950) beam-tracking hot path.
 *
 * This is the drop-in boundary for jank324/lynx's `Segment.track()` path.  The reference
 * has no FFI of its own (it is pure Python on jax.numpy); the functions below are what a
 * binding for that path replaces, and each one cites the reference interface it stands
 * for (paths relative to the reference repository).  Plain pointers and sizes only; every
 * `d_*` pointer is device memory obtained from lynx_buf_alloc.
 *
 * Status convention: every function returns 0 on success and a negative lynx_status on
 * failure; lynx_last_error() returns the message of the last failure on that context.
 */
#ifndef LYNX_HIP_H
#define LYNX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lynx_ctx lynx_ctx;
typedef struct lynx_lattice lynx_lattice;

enum lynx_status {
  LYNX_OK = 0,
  LYNX_ERR_INVALID = -1, /* bad argument / shape mismatch */
  LYNX_ERR_HIP = -2,     /* HIP runtime error            */
  LYNX_ERR_RCCL = -3,    /* RCCL error                   */
  LYNX_ERR_NOMEM = -4,
  /* a beam reached a cavity with energy <= 0 (or NaN): the reference's `assert Ei > 0` (lynx/accelerator/cavity.py:260).
   * Found on the device while the cavities' whole-batch predicates are evaluated (no read-back per call) and reported
   * by the next call that waits for the GPU anyway: lynx_sync, lynx_buf_d2h.  The Python layer raises AssertionError. */
  LYNX_ERR_ENERGY = -5
};

enum lynx_dtype { LYNX_F32 = 0, LYNX_F64 = 1 };

/* Element kinds on the path (reference: the element modules under lynx/accelerator/). */
enum lynx_kind {
  LYNX_KIND_IDENTITY = 0,   /* Marker (marker.py:32-35), inactive BPM (bpm.py:43-46)   */
  LYNX_KIND_DRIFT = 1,      /* drift.py:44-62               params [L]                 */
  LYNX_KIND_QUADRUPOLE = 2, /* quadrupole.py:66-80          params [L,k1,tilt,mx,my]   */
  LYNX_KIND_DIPOLE = 3,     /* dipole.py:112-181, rbend.py  params [L,angle,e1,e2,tilt,fint,fintx,gap] */
  LYNX_KIND_HCOR = 4,       /* horizontal_corrector.py:52-67 params [L,angle]          */
  LYNX_KIND_VCOR = 5,       /* vertical_corrector.py:52-66  params [L,angle]           */
  LYNX_KIND_CAVITY = 6,     /* cavity.py:97-325             params [L,V,phase_deg,f]   */
  LYNX_KIND_CUSTOM = 7,     /* custom_transfer_map.py:87-88 params [49 map entries]    */
  /* the public helpers of lynx/track_methods.py, exposed as lynx_amd.track_methods */
  LYNX_KIND_BASE_RMATRIX = 8, /* track_methods.py:37-105    params [L,k1,hx,tilt]      */
  LYNX_KIND_ROTATION = 9,     /* track_methods.py:14-34     params [angle]             */
  LYNX_KIND_MISALIGNMENT = 10, /* track_methods.py:108-122  params [mx,my,sign] (sign -1: R_entry, +1: R_exit) */
  LYNX_KIND_SOLENOID = 11,    /* solenoid.py:61-105         params [L,k,mx,my]         */
  LYNX_KIND_UNDULATOR = 12    /* undulator.py:48-60         params [L]                 */
};

/*
 * Whole-batch predicates.  The reference evaluates these as Python `if any(...)` over the
 * whole batch (so they are host decisions there too); the host evaluates them once per
 * lattice and hands them to the kernels as flags.
 */
#define LYNX_FLAG_TILT 1        /* quadrupole: any(tilt != 0)          track_methods.py:101 */
#define LYNX_FLAG_MISALIGNED 2  /* quadrupole: !all(misalignment == 0) quadrupole.py:75    */
#define LYNX_FLAG_THICK 4       /* dipole: any(length != 0)            dipole.py:119        */
#define LYNX_FLAG_CAV_BETA 8    /* cavity: any(V != 0 & E != 0)        cavity.py:290        */
#define LYNX_FLAG_CAV_GAIN 16   /* cavity: any(E + dE > 0)             cavity.py:128        */
#define LYNX_FLAG_CAV_T5XX 32   /* cavity: any(dE > 0)                 cavity.py:164        */

/* One lattice element.  Sample b reads its parameters at
 * pool[param_offset + b * batch_stride + j]; batch_stride == 0 means "same for all samples". */
typedef struct lynx_elem {
  int32_t kind;
  int32_t flags;
  int32_t param_offset;
  int32_t batch_stride;
} lynx_elem;

/* A step of the tracking program = what `Segment.track` calls a "todo"
 * (segment.py:344-354): a maximal run of skippable elements whose maps are composed
 * first (segment.py:329-336), or one active cavity (cavity.py:81-246). */
enum lynx_step_kind { LYNX_STEP_RUN = 0, LYNX_STEP_CAVITY = 1 };
typedef struct lynx_step {
  int32_t kind;  /* lynx_step_kind */
  int32_t first; /* first element index */
  int32_t last;  /* one past the last element index (first+1 for a cavity) */
  int32_t flags; /* cavity steps: copy of the element's LYNX_FLAG_CAV_*; runs: LYNX_STEP_FLAG_RAW */
} lynx_step;
/* Run step: take the first element's map as it is instead of left-multiplying it onto the
 * identity.  `Element.track` applies `transfer_map()` directly (element.py:72,84), while
 * `Segment.transfer_map` starts from eye(7) (segment.py:331-335); the two differ only in
 * signed zeros and in how a NaN entry spreads (NaN * 0).  Cavity steps are always raw
 * (cavity.py:113-121). */
#define LYNX_STEP_FLAG_RAW 64
/* Run step that holds exactly one identity element (an active BPM, bpm.py:48-58): while the particles
 * pass it, the streaming kernel adds up their x and y -- the BPM's reading, `stack([mu_x, mu_y])` of
 * the beam that ENTERS it -- without splitting the pass there.  At most LYNX_MAX_OBSERVERS per
 * program; their sums come back through `d_observations` of lynx_track_particles. */
#define LYNX_STEP_FLAG_OBSERVE 128
#define LYNX_MAX_OBSERVERS 8

/* flags of lynx_track_particles */
#define LYNX_TRACK_MOMENTS 1     /* also accumulate the output-beam moments (fused epilogue) */
#define LYNX_TRACK_TWO_KERNEL 2  /* build+compose in its own launch instead of the fused prologue */
/* d_p_in is [N][7]: one incoming beam shared by every sample of the lattice batch.  The
 * reference's `ParticleBeam.broadcast` repeats the particles physically
 * (particle_beam.py:838-843) before a parameter scan; a beam broadcast lazily is read once
 * per sample out of the caches instead, which halves the HBM traffic of the pass. */
#define LYNX_TRACK_SHARED_INPUT 4
/* Apply every step on its own.  By default a run that is followed by an active cavity is applied
 * together with it: one 7x7 application with T_cav . T_run (composed per sample, like the maps
 * of a run are) plus the two rows of T_run that give the s and delta entering the cavity --
 * the same algebra as segment.py:344-354 / cavity.py:113-161, rounded differently. */
#define LYNX_TRACK_SEQUENTIAL_STEPS 8
/* Like LYNX_TRACK_MOMENTS, but accumulate the whole 6x6 covariance (21 products per particle)
 * instead of the second moments the reference's ParticleBeam exposes as properties
 * (particle_beam.py:736-836: the six variances, sigma_xx' and sigma_yy' -- 8 products). */
#define LYNX_TRACK_COVARIANCE 16

/* Layout of one sample's moment record (float64 regardless of the particle dtype):
 * [0..6] mean of the 7 coordinates, [7..27] upper triangle (row-major, i<=j<6) of the
 * BIASED 6x6 covariance, [28..33] reserved (0), [34] 1 if the whole triangle was accumulated,
 * 0 if only the property set was (then every other entry of [7..27] is NaN), [35] number of
 * particles. */
#define LYNX_MOMENT_STRIDE 36

typedef struct lynx_device_info_t {
  char name[64];
  char arch[32];
  int32_t compute_units;
  int32_t lds_bytes_per_cu;
  int64_t hbm_bytes;
} lynx_device_info_t;

/* ---- diagnostics ---------------------------------------------------------------- */
const char* lynx_version(void);
int lynx_device_count(int* count);
int lynx_ctx_create(int device, lynx_ctx** out);
int lynx_ctx_destroy(lynx_ctx* ctx);
const char* lynx_last_error(lynx_ctx* ctx);
int lynx_device_info(lynx_ctx* ctx, lynx_device_info_t* out);
int lynx_sync(lynx_ctx* ctx);
/* HIP-event timer on the context's stream (the stream every kernel below runs on). */
int lynx_timer_start(lynx_ctx* ctx);
int lynx_timer_stop(lynx_ctx* ctx, float* elapsed_ms);
/* Per-launch timing of the streaming kernel (k_track) with HIP events recorded on the
 * context's stream immediately around every launch between begin and end.  This is the
 * figure bench.py's roofline uses; it matches rocprofv3's per-kernel duration. */
int lynx_profile_begin(lynx_ctx* ctx);
int lynx_profile_end(lynx_ctx* ctx, double* total_ms, int64_t* launches);
/* every launch of the profile that lynx_profile_end closed last, in launch order: up to `capacity` durations in
 * milliseconds into ms_out, their number into *launches (bench.py: the per-launch curve of the timed steps) */
int lynx_profile_launches(lynx_ctx* ctx, double* ms_out, int64_t capacity, int64_t* launches);
/* ... and every lynx_gather_moments of that profile: its duration on the stream it ran on (with more than one rank
 * this includes the wait for the slowest rank) */
int lynx_profile_gathers(lynx_ctx* ctx, double* ms_out, int64_t capacity, int64_t* gathers);
/* The launch-plan switches (environment variables LYNX_XPOSE, LYNX_UNROLL, ... -- listed with what each selects at
 * `struct Knobs` in lynx_amd/csrc/lynx_hip.hip) are read ONCE, by lynx_ctx_create; a tracking call reads none.  This
 * reads them again: for tests and A/B scripts that change the environment of a live context.                       */
int lynx_ctx_reload_knobs(lynx_ctx* ctx);

/* Calibration: average time of a plain 16-byte-per-lane copy of `bytes` (read + write),
 * i.e. the practical HBM ceiling of this GPU for a stream shaped like a tracking pass.
 * vec_per_thread: 16-byte vectors each thread copies (a workgroup owns one contiguous block of
 * 256 * vec_per_thread vectors); 0 = grid-stride loop; 100 + n = n per thread with non-temporal
 * stores.  1 is the fastest shape on MI355X (6.2 TB/s, with either kind of store).             */
int lynx_diag_copy(lynx_ctx* ctx, void* d_dst, const void* d_src, size_t bytes, int repeats,
                   int vec_per_thread, float* avg_ms);

/* ---- device buffers (reference: jax.Array storage behind `ParticleBeam.particles`,
 *      particle_beam.py:24-45; the Python side owns the handles) ----------------------- */
int lynx_buf_alloc(lynx_ctx* ctx, size_t bytes, void** d_out);
/* ... for a result the host will read back (what `track` returns of a ParameterBeam: mu, cov -- element.py:71-82; a
 * moment record): small blocks are host memory the GPU writes through, reading them back is a wait and a memcpy */
int lynx_buf_alloc_result(lynx_ctx* ctx, size_t bytes, void** d_out);
int lynx_buf_free(lynx_ctx* ctx, void* d_ptr);
int lynx_buf_h2d(lynx_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int lynx_buf_d2h(lynx_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int lynx_buf_d2d(lynx_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
int lynx_buf_memset(lynx_ctx* ctx, void* d_dst, int value, size_t bytes);
int lynx_pool_trim(lynx_ctx* ctx); /* return cached blocks to the driver */

/* ---- lattice program (reference: the `Segment.elements` list and every element's
 *      parameter arrays, segment.py:40-54; partition into steps, segment.py:344-351) --- */
int lynx_lattice_create(lynx_ctx* ctx, int dtype, int64_t batch, int32_t n_elems,
                        const lynx_elem* elems, int32_t n_steps, const lynx_step* steps,
                        const void* pool, int64_t pool_count, lynx_lattice** out);
/* overwrite pool[offset : offset+count] (element parameters changed, e.g. `quad.k1 = ...`) */
int lynx_lattice_update_params(lynx_lattice* lat, int64_t offset, int64_t count, const void* host);
/* overwrite element / step flags (whole-batch predicates that depend on the beam energy) */
int lynx_lattice_set_flags(lynx_lattice* lat, const int32_t* elem_flags, const int32_t* step_flags);
int lynx_lattice_destroy(lynx_lattice* lat);

/* ---- the hot path ----------------------------------------------------------------- */

/* Build every element's 7x7 map from the batched parameters and compose each step's map
 * (reference: Element.transfer_map of every kind, track_methods.py:14-122,
 * Segment.transfer_map segment.py:329-338, Cavity._cavity_rmatrix cavity.py:248-325).
 *   d_energy_in  [B]                      beam energy entering the lattice
 *   d_steps_out  [B][n_steps][64]         per step: 49 map entries + 8 cavity coefficients
 *   d_energy_out [B] or NULL              beam energy leaving the lattice                  */
int lynx_build_compose(lynx_ctx* ctx, lynx_lattice* lat, const void* d_energy_in,
                       void* d_steps_out, void* d_energy_out);

/* Segment.track on a ParticleBeam (reference: segment.py:340-356 -> element.py:83-92
 * `particles @ tm^T`, cavity.py:141-161,219-226).
 *   d_p_in / d_p_out [B][N][7]  (may alias; d_p_in [N][7] with LYNX_TRACK_SHARED_INPUT)
 *   d_moments_out    [B][36] float64 or NULL (needs LYNX_TRACK_MOMENTS)
 *   d_observations   [B][n_observers][2] float64 or NULL: mean x and mean y of the beam entering
 *                    each LYNX_STEP_FLAG_OBSERVE step, in lattice order (BPM.reading, bpm.py:48-54);
 *                    required when the program has such steps                                 */
int lynx_track_particles(lynx_ctx* ctx, lynx_lattice* lat, int64_t n_particles,
                         const void* d_energy_in, const void* d_p_in, void* d_p_out,
                         void* d_energy_out, double* d_moments_out, int flags, double* d_observations);

/* The allocating form: `Segment.track` returns a NEW beam and leaves the incoming one alone (element.py:75-92), so the
 * blocks of the outgoing beam come from the library's own pool -- one call where the caller of lynx_track_particles
 * makes up to five.  d_out[4]: the particles [B][N][7]; the outgoing energy [B] (only if want_energy_out, else NULL);
 * the moment records [B][36] float64 (only with LYNX_TRACK_MOMENTS / _COVARIANCE); the observations
 * [B][n_observers][2] float64 (only if the program has observer steps).  The caller owns them (lynx_buf_free).
 * On failure nothing is allocated and all four are NULL.                                                          */
int lynx_track_particles_new(lynx_ctx* ctx, lynx_lattice* lat, int64_t n_particles, const void* d_energy_in,
                             const void* d_p_in, int flags, int32_t want_energy_out, void** d_out);

/* Segment.track on a ParameterBeam (reference: element.py:71-82 mu'=T mu, cov'=T cov T^T;
 * cavity.py:134-140,202-218).  d_mu [B][7], d_cov [B][7][7] (in/out may alias).            */
int lynx_track_moments(lynx_ctx* ctx, lynx_lattice* lat, const void* d_energy_in,
                       const void* d_mu_in, const void* d_cov_in, void* d_mu_out,
                       void* d_cov_out, void* d_energy_out);

/* Reverse pass of lynx_track_particles: gradient of a scalar function L of the outgoing
 * beam's moment record with respect to every element parameter and the incoming energy
 * (the reference only claims differentiability, setup.py:14-17,
 * tests/test_differentiable.py).
 *   d_moments_fwd  [B][36] float64  the forward moment record (lynx_track_particles)
 *   d_grad_moments [B][36] float64  dL/d(record): [0..6] d/dmean, [7..27] d/dcov (upper
 *                                   triangle, each off-diagonal entry counted once)
 *   d_grad_params  [B][E][8]        dL/d(parameter j of element e), parameter order of the
 *                                   element kind; unused slots 0; custom maps: 0
 *   d_grad_energy_in [B]            dL/d(incoming energy)
 *   d_grad_p_in    [B][N][7] or NULL dL/d(incoming particle coordinates) (the reference's
 *                                   tests/test_differentiable.py:75-91 differentiates through those)
 *   d_grad_observations [B][n_observers][2] float64 or NULL: dL/d(reading) of the program's observer steps (active
 *                                   BPMs, bpm.py:48-58: the reading is the mean x, y of the beam that enters the BPM)
 * Limit of this version: n_steps <= 64 (every 4th per-particle state is parked in a
 * fixed-size private array during the forward sweep).                                       */
int lynx_track_particles_backward(lynx_ctx* ctx, lynx_lattice* lat, int64_t n_particles,
                                  const void* d_energy_in, const void* d_p_in,
                                  const double* d_moments_fwd, const double* d_grad_moments,
                                  void* d_grad_params, void* d_grad_energy_in, void* d_grad_p_in,
                                  const double* d_grad_observations);

/* Reverse pass of lynx_track_moments (ParameterBeam): gradient of a scalar function of the
 * outgoing (mu, cov) with respect to every element parameter, the incoming energy and the
 * incoming mu and cov (the reference's tests/test_differentiable.py:54-72 makes those the
 * leaves).
 *   d_mu_bar [B][7], d_cov_bar [B][7][7]   dL/d(outgoing mu), dL/d(outgoing cov), entry by entry
 *   d_grad_params [B][E][8], d_grad_energy_in [B]   as for lynx_track_particles_backward
 *   d_grad_mu_in [B][7], d_grad_cov_in [B][7][7]    dL/d(incoming mu), dL/d(incoming cov)      */
int lynx_track_moments_backward(lynx_ctx* ctx, lynx_lattice* lat, const void* d_energy_in,
                                const void* d_mu_in, const void* d_cov_in, const void* d_mu_bar,
                                const void* d_cov_bar, void* d_grad_params, void* d_grad_energy_in,
                                void* d_grad_mu_in, void* d_grad_cov_in);

/* Moment read-out of an existing ParticleBeam (reference: particle_beam.py:736-836,
 * one fused pass instead of 14 separate reductions).  d_moments_out [B][36] float64;
 * covariance != 0: the whole 6x6 covariance, else the property set (see LYNX_MOMENT_STRIDE). */
int lynx_moments(lynx_ctx* ctx, int dtype, int64_t batch, int64_t n_particles, const void* d_p,
                 double* d_moments_out, int32_t covariance);

/* Screen read-out of a ParticleBeam: per-sample 2-D histogram of (x, y) with the bin edges the
 * caller supplies (reference: screen.py:196-213, `jnp.histogramdd` with `pixel_bin_edges`
 * :107-120, then `flipud(hist.T)`).  Bin of a value v: last edge <= v, the right-most edge
 * included (numpy.histogramdd).  d_image [B][ny][nx] int32 counts, row 0 = highest y; the
 * call clears it first.  d_xedges [nx+1], d_yedges [ny+1] in the particle dtype.            */
int lynx_histogram2d(lynx_ctx* ctx, int dtype, int64_t batch, int64_t n_particles, const void* d_p,
                     const void* d_xedges, const void* d_yedges, int32_t nx, int32_t ny, int32_t* d_image);

/* Screen read-out of a ParameterBeam: density of the bivariate normal of (x, y) on a pixel grid
 * (reference: screen.py:160-195).  d_xs [nx], d_ys [ny] pixel coordinates, d_mu [B][7],
 * d_cov [B][7][7]; d_image [B][nx][ny] with the x axis flipped, as the reference returns it. */
int lynx_gaussian_image(lynx_ctx* ctx, int dtype, int64_t batch, const void* d_mu, const void* d_cov,
                        const void* d_xs, const void* d_ys, int32_t nx, int32_t ny, void* d_image);

/* Test hook: the float32 sine / cosine the cavity kick uses on the device (cavity.py:141-161
 * calls cos per particle), scalar (packed = 0) or two-per-lane (packed = 1) code path.        */
int lynx_diag_phase_trig(lynx_ctx* ctx, int64_t n, const float* d_x, int32_t packed, float* d_sin,
                         float* d_cos);

/* Aperture (reference: aperture.py:69-108): particles with |x| < x_max and |y| < y_max
 * (rectangular) or x^2/x_max^2 + y^2/y_max^2 <= 1 (elliptical) survive.
 *   d_x_max, d_y_max [B] (param_stride 1) or one value for all samples (param_stride 0)
 *   d_mask [B][N] uint8, d_counts / d_offsets [B][ceil(N/1024)], d_totals [B] survivors     */
int lynx_aperture_mask(lynx_ctx* ctx, int dtype, int64_t batch, int64_t n_particles, const void* d_p,
                       const void* d_x_max, const void* d_y_max, int32_t param_stride, int32_t elliptical,
                       unsigned char* d_mask, int32_t* d_counts, int64_t* d_offsets, int64_t* d_totals);
/* One sample: d_kept [total][7] survivors and d_lost [N - total][7] lost particles, each in
 * their original order (what `particles[mask]` returns), from the mask and offsets above.    */
int lynx_aperture_compact(lynx_ctx* ctx, int dtype, int64_t n_particles, const void* d_p,
                          const unsigned char* d_mask, const int64_t* d_offsets, void* d_kept,
                          void* d_lost);

/* Seeded synthetic beam, generated in HBM: uncorrelated 6-D Gaussian per sample, 7th
 * coordinate 1 (shape of ParticleBeam.from_parameters, particle_beam.py:144-170; not the
 * same random stream).  mu[6], sigma[6] are host arrays.                                   */
int lynx_fill_gaussian(lynx_ctx* ctx, int dtype, int64_t batch, int64_t n_particles,
                       const double* mu, const double* sigma, uint64_t seed, void* d_p);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (no reference counterpart: the
 *      reference is single-process; batch samples are independent, so only the
 *      per-sample moment records are exchanged) --------------------------------------- */
#define LYNX_UNIQUE_ID_BYTES 128
int lynx_comm_unique_id(char* id_out /* LYNX_UNIQUE_ID_BYTES */);
int lynx_comm_init(lynx_ctx* ctx, int n_ranks, int rank, const char* id);
int lynx_comm_destroy(lynx_ctx* ctx);
/* what the communicator actually is: RCCL version code (ncclGetVersion), ranks, this rank; ranks = 0 when there is none */
int lynx_comm_info(lynx_ctx* ctx, int32_t* rccl_version, int32_t* n_ranks, int32_t* rank);
/* all-gather `count` float64 per rank: d_recv [n_ranks][count].  Asynchronous: it starts when everything issued
 * so far has run -- with more than one rank on a communication stream of its own, underneath the next tracking
 * call (nothing on the main stream waits for it; environment LYNX_GATHER_OVERLAP=0: in line on the main stream,
 * also the default of a one-rank communicator); lynx_buf_d2h and lynx_sync wait for it either way.  Both blocks
 * may be handed to lynx_buf_free at any time: the library keeps them out of its allocator until the gather is done. */
int lynx_gather_moments(lynx_ctx* ctx, const double* d_send, double* d_recv, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* LYNX_HIP_H */
