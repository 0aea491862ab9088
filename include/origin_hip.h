/*
 * origin_hip.h -- C ABI of liborigin_hip.so, the MI355X (gfx950) implementation of
 * ORIGIN's dense hot path.
 *
 * The reference (musevlt/origin) has no FFI: its hot path is four Python functions in
 * muse_origin/lib_origin.py that Step.run bodies in muse_origin/steps.py resolve by
 * name (steps.py:19-41).  This ABI is what those functions are re-implemented on; each
 * entry point cites the reference interface it replaces.  The Python binding is
 * origin_amd/_capi.py (ctypes); INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success or a negative ORIGIN_E_* code; the message of
 *     the last failure on the calling thread is origin_last_error().  Nothing aborts.
 *   - cubes are C-order (Nz, Ny, Nx), x fastest, float32; masks / profile indices uint8.
 *   - pointers named d_* are DEVICE pointers obtained from origin_malloc (or any HIP
 *     allocation on the context's device); h_* are host pointers.  The library never
 *     keeps a caller pointer after the call returns, except plan objects that own
 *     private device copies of what they were given.
 *   - all work is enqueued on the context's stream; calls that return host values
 *     synchronise, the others are asynchronous (origin_sync to wait).
 *   - one host thread per context; a context is not re-entrant.
 */
#ifndef ORIGIN_HIP_H
#define ORIGIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORIGIN_OK 0
#define ORIGIN_E_ARG (-1)      /* bad argument / unsupported size            */
#define ORIGIN_E_NOMEM (-2)    /* device allocation failed                    */
#define ORIGIN_E_HIP (-3)      /* a HIP runtime call or kernel launch failed  */
#define ORIGIN_E_NODEVICE (-4) /* no usable GPU                               */
#define ORIGIN_E_STATE (-5)    /* call order / plan mismatch                  */

typedef struct origin_ctx origin_ctx;
typedef struct origin_glr_plan origin_glr_plan;

/* ---- context, memory, timing ------------------------------------------------------ */
const char *origin_last_error(void);
int origin_abi_version(void);
int origin_device_count(int *count);
int origin_ctx_create(int device, origin_ctx **out);
int origin_ctx_destroy(origin_ctx *ctx);
int origin_sync(origin_ctx *ctx); /* main stream and, if work is pending there, the aux stream */
/* Work enqueued through an *_async entry point runs on the context's low-priority auxiliary
 * stream, ordered after everything enqueued on the main stream before the call.  Its inputs
 * must stay untouched and its outputs unread until origin_aux_join (the main stream waits for
 * it; no host synchronisation) or origin_sync. */
int origin_aux_join(origin_ctx *ctx);
int origin_device_name(origin_ctx *ctx, char *buf, int buflen);
int origin_mem_info(origin_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
/* the hipStream_t of the context, as an opaque pointer (for interop / RCCL) */
int origin_stream(origin_ctx *ctx, void **stream);

/* origin_free keeps blocks of at least 1 MiB (up to ORIGIN_ALLOC_CACHE_GB, default 96; 0 = off) for
 * the next origin_malloc of their size: the steps of the reference return fresh arrays, and a
 * hipMalloc of a 5 GB cube takes ~40 ms.  Reuse is ordered by the context's streams; origin_mem_info
 * counts the kept blocks as free (they are released when an allocation would fail). */
int origin_malloc(origin_ctx *ctx, size_t bytes, void **d_ptr);
int origin_free(origin_ctx *ctx, void *d_ptr);
int origin_memset(origin_ctx *ctx, void *d_ptr, int byte, size_t bytes);
int origin_h2d(origin_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int origin_d2h(origin_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* Device float32 -> host float64, n elements: the widening the reference's float64 interface
 * needs (cubes leaving through the function seam, steps.py store_cube), chunked through pinned
 * staging and widened by the host worker pool while the next chunk is in flight. */
int origin_d2h_f32_as_f64(origin_ctx *ctx, double *h_dst, const float *d_src, size_t n);
/* The other direction: host float64 -> device float32 (round to nearest even, as
 * ndarray.astype(float32)), narrowed by the host worker pool into pinned staging. */
int origin_h2d_f64_as_f32(origin_ctx *ctx, float *d_dst, const double *h_src, size_t n);
int origin_d2d(origin_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* strided 3-D box copy of `elem`-byte elements between (Nz, Ny, Nx)-shaped arrays; kind:
 * 0 = host->device, 1 = device->host, 2 = device->device.  Pitches in elements.  Used
 * for tile upload/download and halo strips. */
int origin_copy_box(origin_ctx *ctx, int kind, void *dst, long dst_pitch_y, long dst_pitch_z,
                    const void *src, long src_pitch_y, long src_pitch_z, int nz, int ny,
                    int nx, int elem);
/* Spaxel-list moves (csrc/columns.hip): packed[z][i] = cube[z][idx[i]] and back, for `elem`-byte
 * elements (1 or 4) of an (Nz, S) cube and a contiguous (Nz, n) buffer; idx: n int32 spaxel
 * indices on the device, each < S.  What the tiled path exchanges when PCA areas (irregular sets of
 * spaxels, steps.py:492-569) are handed to ranks as wholes: the spaxels a rank's GLR reads around
 * its own (lib_origin.py:1027-1043) are a list, not a rectangle.  Asynchronous on the stream. */
int origin_gather_columns(origin_ctx *ctx, const void *d_cube, int Nz, long S, const int *d_idx,
                          long n, int elem, void *d_packed);
int origin_scatter_columns(origin_ctx *ctx, void *d_cube, int Nz, long S, const int *d_idx, long n,
                           int elem, const void *d_packed);

/* ---- purity threshold reductions -----------------------------------
 * Device side of Compute_threshold_purity (lib_origin.py:1391-1479, called by
 * ComputePurityThreshold.run, steps.py:874-890): the default threshold list needs the
 * per-spaxel maxima of the local-maximum cubes (:1437-1442), the purity curve the number of
 * voxels above each threshold (:1444-1452).  d_keep: uint8 [Ny*Nx] or NULL, 0 = spaxel
 * excluded (cube_local_min * (segmap == 0), :1430). */
int origin_zmax_map(origin_ctx *ctx, const float *d_cube, const uint8_t *d_keep, int Nz, long S,
                    float *d_map /* [S] */);
/* h_counts[t] = #{ voxels : (double) value > h_thr[t] }, thresholds in any order, nthr <= 1024 */
int origin_count_above(origin_ctx *ctx, const float *d_cube, const uint8_t *d_keep, int Nz, long S,
                       int nthr, const double *h_thr, long *h_counts);

/* ---- detection thresholding (the consumer of the local-maximum cubes) -------------------
 * Detection.run, steps.py:956-974 (and det_correl_min, :935-939):
 *     z, y, x = np.where(cube > thr);  T_GLR = cube[z, y, x];  profile = cube_profile[z, y, x]
 * as an ordered stream compaction on the device: positions in C order (z slowest, x fastest, the
 * order np.where returns), compared in float64 (NaN is never above).  d_aux: uint8 cube gathered at
 * the same positions (cube_profile) or NULL.  *h_count receives the TOTAL number of voxels above
 * thr; the first min(cap, total) of them are written to d_z / d_y / d_x (int32) and, where given,
 * d_val (float32) and d_auxout (uint8) -- call again with a larger cap when total > cap; cap = 0
 * only counts.  d_cube must be 16-byte aligned.  Synchronises the stream. */
int origin_where_above(origin_ctx *ctx, const float *d_cube, const uint8_t *d_aux, int Nz, int Ny,
                       int Nx, double thr, long cap, int *d_z, int *d_y, int *d_x, float *d_val,
                       uint8_t *d_auxout, long *h_count);

/* ---- FITS data units ------------------------------------------------
 * Step.dump / Step.load (steps.py:301-352) write and reload every cube / image of a step as a
 * FITS image extension through mpdaf (Cube.write(convert_float32=False) -> float64; lazy
 * reload in DataObj.__get__, steps.py:131-160).  A FITS data unit is the array in big-endian
 * byte order, IEEE-754 for BITPIX < 0.  These two calls convert between a device array and
 * the bytes of a data unit (widening / narrowing included), replacing the host-side dtype and
 * byte-order pass of astropy.io.fits; origin_amd/fitsio.py writes the headers around them.
 * element types: 0 = float32, 1 = uint8, 2 = int32, 3 = float64.
 * bitpix: -64, -32, 8, 16, 32, 64.  d_dst of encode / d_src of decode: n * |bitpix| / 8 bytes,
 * 8-byte aligned.  Asynchronous on the context's stream. */
int origin_fits_encode(origin_ctx *ctx, const void *d_src, int src_type, long n, int bitpix,
                       void *d_dst);
int origin_fits_decode(origin_ctx *ctx, const void *d_src, int bitpix, long n, int dst_type,
                       void *d_dst);
/* The same conversions streamed to / from an open file descriptor at its current offset
 * (sequential write() / read()): conversion, PCIe copy through two pinned buffers and file
 * I/O overlap chunk by chunk.  The caller writes the header blocks and the padding of the
 * data unit to a multiple of 2880 bytes.  Both return when the file / the device array is
 * complete. */
int origin_fits_write_data(origin_ctx *ctx, const void *d_src, int src_type, long n, int bitpix,
                           int fd);
int origin_fits_read_data(origin_ctx *ctx, int fd, int bitpix, long n, int dst_type, void *d_dst);

/* ---- inter-GPU exchange (one process per GPU; RCCL over xGMI on the context's stream) ----
 * The reference has no multi-GPU path (its only parallelism is the joblib pool of
 * lib_origin.py:1150-1160); these serve the spatial tiling of origin_amd/multigpu.py.
 * librccl.so is opened on first use.  origin_comm_unique_id is called on ONE rank and the
 * ORIGIN_COMM_ID_BYTES blob handed to every rank by any host channel; origin_comm_create is
 * collective.  allreduce / exchange are asynchronous on the context's stream. */
#define ORIGIN_COMM_ID_BYTES 128
typedef struct origin_comm origin_comm;
int origin_comm_unique_id(char *id /* [ORIGIN_COMM_ID_BYTES] */);
int origin_comm_create(origin_ctx *ctx, const char *id, int rank, int world, origin_comm **out);
int origin_comm_destroy(origin_comm *comm);
/* in-place sum over all ranks of n float64 (per-channel sum / count, steps.py:442) */
int origin_comm_allreduce_f64(origin_comm *comm, double *d_buf, long n);
/* grouped point-to-point: contiguous device buffers, sizes in bytes; every rank posts all
 * its receives and sends in one group (halo strips of cube_faint before the GLR) */
int origin_comm_exchange(origin_comm *comm, int nsend, const int *send_peer,
                         const void *const *d_send, const long *send_bytes, int nrecv,
                         const int *recv_peer, void *const *d_recv, const long *recv_bytes);

/* HIP-event timers on the context's stream: slot in [0, 64). */
int origin_timer_start(origin_ctx *ctx, int slot);
int origin_timer_stop(origin_ctx *ctx, int slot);
int origin_timer_ms(origin_ctx *ctx, int slot, float *ms); /* synchronises on the stop */

/* Built-in per-kernel-class profiler: when enabled, every kernel launch of the library is
 * bracketed by HIP events on the context's stream; totals are kept per class
 * (id in [0, origin_prof_count())).  origin_prof_get synchronises on pending events. */
int origin_prof_enable(origin_ctx *ctx, int on);
int origin_prof_reset(origin_ctx *ctx);
int origin_prof_count(void);
int origin_prof_get(origin_ctx *ctx, int id, const char **name, double *total_ms,
                    long *launches);

/* ---- A. DCT continuum + standardisation -------------------------------------------
 * Replaces dct_residual (lib_origin.py:150-240) and the dense lines of
 * Preprocessing.run (steps.py:431-450, :463-465). */

/* Per-spaxel fit of order+1 DCT-II atoms (lib_origin.py:127-146): weighted LSQ
 * D (D^T S^-1 D)^-1 D^T S^-1 s for spaxels without masked voxels (:226-235), plain
 * D D^T s otherwise or when approx != 0 (:191-194, :237).  d_coef: float64 [(order+1)][S]. */
int origin_dct_fit(origin_ctx *ctx, const float *d_raw, const float *d_var,
                   const uint8_t *d_mask, int Nz, int Ny, int Nx, int order, int approx,
                   double *d_coef);
/* origin_dct_fit followed by origin_dct_resid_sums (below), as one call: the per-channel sums
 * of raw over the unmasked spaxels are taken inside the fit's moments pass, which reads raw and
 * mask anyway, so the separate 5 B/voxel plane pass disappears (masked voxels are visited once
 * more, mask bytes only, in the spaxel groups that have any).  Same d_coef, d_zsum, d_zcnt as
 * the two calls (summation order differs: agreement to a few ulp of float64). */
int origin_dct_fit_sums(origin_ctx *ctx, const float *d_raw, const float *d_var,
                        const uint8_t *d_mask, int Nz, int Ny, int Nx, int order, int approx,
                        double *d_coef, double *d_zsum, double *d_zcnt);
/* Continuum cube D c (what dct_residual returns), float32. */
int origin_dct_continuum(origin_ctx *ctx, const double *d_coef, int Nz, int Ny, int Nx,
                         int order, float *d_cont);
/* Per-channel sum and count over unmasked spaxels of (raw - cont): the two halves of
 * nanmean(data, axis=(1,2)) (steps.py:434-442).  d_zsum, d_zcnt: float64 [Nz].  Across
 * GPUs these are all-reduced by the host before origin_dct_standardize. */
int origin_dct_resid_sums(origin_ctx *ctx, const float *d_raw, const uint8_t *d_mask,
                          const double *d_coef, int Nz, int Ny, int Nx, int order,
                          double *d_zsum, double *d_zcnt);
/* cube_std = (raw - cont - zsum/zcnt) / sqrt(var), 0 where masked; cont_dct = cont / sqrt(var)
 * (steps.py:439-446, :463); optional per-spaxel images (may be NULL): ima_std = mean_z
 * cube_std (:450), ima_dct = mean_z cont_dct (:465), o2 = mean_z cube_std^2 (O2test,
 * lib_origin.py:957-974).  d_cont_dct may be NULL. */
int origin_dct_standardize(origin_ctx *ctx, const float *d_raw, const float *d_var,
                           const uint8_t *d_mask, const double *d_coef,
                           const double *d_zsum, const double *d_zcnt, int Nz, int Ny,
                           int Nx, int order, float *d_cube_std, float *d_cont_dct,
                           float *d_ima_std, float *d_ima_dct, double *d_o2);
/* cont_dct (and ima_dct, may be NULL) alone, for callers that pass NULL for both above: nothing
 * downstream of the O2 map waits for the continuum cube, so it can be enqueued after the O2
 * map has been fetched and run while the host fits the thresholds (lib_origin.py:977-1024).
 * Same values as origin_dct_standardize gives. */
int origin_dct_cont_std(origin_ctx *ctx, const float *d_var, const double *d_coef, int Nz, int Ny,
                        int Nx, int order, float *d_cont_dct, float *d_ima_dct);
/* the same on the auxiliary stream (see origin_aux_join): nothing downstream of the O2 map needs
 * cont_dct, so the pass can hide behind the greedy PCA's latency-bound kernels */
int origin_dct_cont_std_async(origin_ctx *ctx, const float *d_var, const double *d_coef, int Nz,
                              int Ny, int Nx, int order, float *d_cont_dct, float *d_ima_dct);

/* ---- B. O2 test and greedy PCA ---------------------------------------------------- */

/* O2test (lib_origin.py:957-974): out[s] = mean_z cube[z, s]^2, float64 [S]. */
int origin_o2(origin_ctx *ctx, const float *d_cube, int Nz, long S, double *d_out);

/* Compute_GreedyPCA_area / Compute_GreedyPCA (lib_origin.py:769-821, :848-954): the whole
 * greedy loop for `na` areas: d_X (float32 (Nz, S), S = Ny*Nx) is cube_std, d_F receives
 * cube_faint; d_X == d_F or d_X == NULL means in place.  All areas advance in lock step,
 * control flow on the device, cube kept in coefficient form F = X - U C until the end.
 *   d_spx      int32 device: concatenated flat spaxel indices s = y*Nx + x of the areas, each
 *              in the column order of cube[:, areamap == i]; h_spx_off: host int64 [na+1].
 *   d_test0    float64 device [S]: O2 test per spaxel (testO2, lib :840) -- not modified.
 *   h_thr      host float64 [na]: thresholds (thresO2).
 *   d_mapO2    int32 device [S]: out, iterations per spaxel (zero outside the areas).
 *   h_nstop    out: number of areas stopped by itermax (:902-905); h_iters (may be NULL):
 *              lock-step iterations executed; h_trace (may be NULL): per iteration
 *              (areas iterating, total nuisance spaxels), int64 [2*trace_cap].
 * Per iteration (lib :899-949): nuisance/background selection (:889-917, including the
 * filtered-index quirk of :908-917), b = mean background (:917), Xp = X - b(b^T X)
 * (:920-923; the division by sum(b^2) at :924 only rescales Xp), G = Xp^T Xp with
 * v_mfma_f64_16x16x4_f64 and its leading eigenvector (repeated squaring on the f64 matrix
 * cores up to 96 columns; above, Lanczos with the matrix resident in the registers and LDS of
 * one CU, converged to a Ritz residual of 1e-14 and verified against G) in place of svds(k=1)
 * (:940), u = Xp v/|Xp v|, F -= u u^T F and the new O2 test (:943-946). */
int origin_pca_run(origin_ctx *ctx, const float *d_X, float *d_F, int Nz, long S, int na,
                   const int *d_spx,
                   const long *h_spx_off, const double *d_test0, const double *h_thr,
                   double noise_pop, int itermax, int *d_mapO2, int *h_nstop, int *h_iters,
                   long *h_trace, int trace_cap);
/* Tail hook of the greedy PCA: `hook(user, n_active, areas)` is called ONCE per run, from inside
 * origin_pca_run / origin_pca_run_into on the calling thread, once at most max_active areas have
 * been iterating for three iterations in a row without their nuisance count falling below 0.6 of
 * what it was at the first of them (stragglers; a run that is about to end does not call).  Before the call the areas that have
 * finished -- they never iterate again -- are written to d_F (the pass they would have had at the
 * end, on the context's stream); `areas` lists the indices of those that go on.  The caller may
 * enqueue the next stage for everything that does not depend on them (origin_glr_run_rows with
 * ORIGIN_GLR_SIDE).  hook = NULL removes it.  Results of the PCA are unchanged. */
int origin_pca_set_tail_hook(origin_ctx *ctx, void (*hook)(void *user, int n_active, const int *areas),
                             void *user, int max_active);
/* The same with cube_faint written into a box of a larger cube: d_F is the box's first element,
 * spaxel y * out_nx + x of the (Nz, S) input goes to d_F[z * out_pz + y * out_py + x].  The tiled
 * path (origin_amd/multigpu.py) points it at the interior of the halo-extended tile the GLR
 * reads, so that no copy stands between the greedy PCA and the halo exchange.  d_X must be a
 * separate contiguous cube.  out_nx == 0: identical to origin_pca_run. */
int origin_pca_run_into(origin_ctx *ctx, const float *d_X, float *d_F, int Nz, long S, int na,
                        const int *d_spx, const long *h_spx_off, const double *d_test0,
                        const double *h_thr, double noise_pop, int itermax, int *d_mapO2,
                        int *h_nstop, int *h_iters, long *h_trace, int trace_cap, int out_nx,
                        long out_py, long out_pz);

/* Pieces of the above exposed for unit tests.  gram: G_a = X_a^T X_a for `nmat` float64
 * matrices X_a ([Nz][ld_a] at d_Xp + xp_off[a], ld_a a multiple of 16); the caller lists the
 * 32x32 upper-triangle tiles (tile_i <= tile_j, matrix tile_a); g_total = sum ld_a^2.
 * eig: leading eigenvector (unit norm) of symmetric PSD matrices G_a (n_a x n_a, row stride
 * ld_a); scratch rows: q_off[a] into a buffer of q_total = sum origin_pca_eig_qrows()*ld_a
 * doubles; d_info (may be NULL): (eigenvalue, Ritz residual, restarts) per matrix. */
int origin_pca_gram(origin_ctx *ctx, const double *d_Xp, const long *d_xp_off, const long *d_ld,
                    int Nz, int ntiles, const int *d_tile_i, const int *d_tile_j,
                    const int *d_tile_a, long g_total, double *d_G, const long *d_g_off);
int origin_pca_eig(origin_ctx *ctx, const double *d_G, const long *d_g_off, const long *d_ld,
                   const long *d_n, int nmat, long q_total, const long *d_q_off, double *d_v,
                   const long *d_v_off, double *d_info);
int origin_pca_eig_qrows(void);

/* Host-only helper of Compute_PCA_threshold (lib_origin.py:999-1002): data > 0, sigma clip
 * (median / std, <= maxiters iterations), np.histogram(bins='fd', density=True).  h_hist gets
 * *nbins values, h_edges *nbins + 1; cap_bins = capacity of h_hist.  No GPU involved. */
int origin_o2_histogram(const double *h_data, long n, double sigclip, int maxiters,
                        double *h_hist, double *h_edges, long cap_bins, long *nbins,
                        long *nkept);

/* the same for `na` areas (values of area a at h_data[h_off[a] .. h_off[a+1])) on host threads;
 * results of area a start at a * (cap_bins + 1) in h_hist / h_edges. */
int origin_o2_histogram_batch(const double *h_data, const long *h_off, int na, double sigclip,
                              int maxiters, double *h_hist, double *h_edges, long cap_bins,
                              long *h_nbins);

/* Gaussian fit of one histogram half (lib_origin.py:1014-1018: astropy LevMarLSQFitter on
 * Gaussian1D = MINPACK lmder, analytic Jacobian, xtol 1e-7, maxfev 100): h_p[3] = (amplitude,
 * mean, stddev), start values in, solution out; *info = MINPACK's termination code. */
int origin_gauss_fit(const double *h_x, const double *h_y, long m, double *h_p, int *info,
                     int *nfev);

/* Threshold of every area from the histograms origin_o2_histogram_batch left (same layout):
 * mode, half-maximum width, bin centres left of mode + fwhm/2, Gaussian fit, thresO2 = mean -
 * stddev * coef with coef = norm.ppf(pfa) (lib_origin.py:1004-1022).  h_res[a] = (thresO2,
 * mean, stddev); h_status[a] = 0 ok, 1 = histogram maximum in the first bin (the reference's
 * argmin over an empty slice raises), 2 = fewer than three bins to fit.  Host threads only. */
int origin_o2_threshold_batch(const double *h_hist, const double *h_edges, const long *h_nbins,
                              int na, long cap_bins, double coef, double *h_res, int *h_status);

/* ComputePCAThreshold.run for every area in one pass over the host worker pool (steps.py:610-631,
 * lib_origin.py:977-1024): gather the area's O2 values from the map (h_map float64 [S]; h_idx =
 * concatenated flat spaxel indices, area a at [h_off[a], h_off[a+1])) into h_data (the
 * reference's testO2, same layout), then origin_o2_histogram and the threshold fit of
 * origin_o2_threshold_batch.  Outputs as those two; h_status[a] = 3 if the histogram failed. */
int origin_o2_areas_fit(const double *h_map, const int *h_idx, const long *h_off, int na,
                        double sigclip, int maxiters, double coef, double *h_data, double *h_hist,
                        double *h_edges, long cap_bins, long *h_nbins, double *h_res,
                        int *h_status);

/* ---- C. GLR correlation ------------------------------------------------------------
 * Replaces Correlation_GLR_test (lib_origin.py:1070-1217) and the dense lines of
 * ComputeTGLR.run (steps.py:781-793).
 *
 * A plan owns device copies of the zero-mean PSFs k_fz = PSF_fz - mean (lib :1033-1034),
 * the optional field weights, the prepared profiles (trimmed / normalised / mean
 * subtracted by the host exactly as lib :1155-1165) and the derived normalisation
 * tables.  h_psf: float64 [nfields][Nz][P][P] (P odd, as the reference holds them);
 * h_weights: float64 [nfields][Ny][Nx] or NULL (weights=None); h_taps: float64
 * concatenated taps of the K prepared profiles, h_tap_off: int [K+1]. */
int origin_glr_plan_create(origin_ctx *ctx, int Nz, int Ny, int Nx, int nfields, int P,
                           const double *h_psf, const double *h_weights, int K,
                           const double *h_taps, const int *h_tap_off,
                           origin_glr_plan **out);
int origin_glr_plan_destroy(origin_glr_plan *plan);
/* bytes of device memory the plan holds */
/* Arithmetic of the GLR stages when the plan is eligible for the matrix cores: 1 (default) =
 * two-term f16 split of data and taps, three MFMAs per product, fp32 accumulation (~3e-7 of
 * sum |p x| from float64: fp32 class); 2 = bf16 operands, one MFMA per product (BASELINE config 4
 * "bf16 GLR"; |dT| ~1e-2); 0 = fp32 FMA chain (~1e-7).  Eligible: the spatial stage for odd PSF
 * sizes 5..25 (one field, or a mosaic of weighted fields: per-field accumulation); the spectral
 * stage for profile half widths <= 32 and K <= 26 -- through the border-class table without
 * weight maps, through the plan's norm cube with them (that cube, [Nz + 96][Ny][Nx] float32, is
 * allocated and filled by the plan's first run): the FOLD form where the cube is smooth along z
 * (origin_glr_plan_fold_eps; precision 1 or 2), else a second Toeplitz product (precision 1 only).
 * A stage that is not eligible runs the fp32 kernels; plans where neither stage is (a field
 * smaller than the PSF without weight maps, longer / more profiles) always report 0.  get
 * returns the plan's setting. */
int origin_glr_plan_set_precision(origin_glr_plan *plan, int precision);
int origin_glr_plan_get_precision(origin_glr_plan *plan, int *precision);
int origin_glr_plan_bytes(origin_glr_plan *plan, size_t *bytes);
/* Matrix-core instructions (v_mfma_f32_32x32x16_{f16,bf16}: 32768 flop each) ONE run of the plan
 * issues in its spatial / spectral stage -- what rocprofv3's SQ_INSTS_MFMA counts per launch
 * (partial edge tiles included); 0 for a stage that runs the fp32 kernels or the weighted forms.
 * bench.py prices `roofline.executed` with it (measurement). */
int origin_glr_plan_mfma_count(origin_glr_plan *plan, long *spatial, long *spectral);
/* FOLD of the matrix-core spectral stage (plans without weight maps): away from the cube's first
 * and last 32 channels 1/sqrt(den_k[z]) = a_k s(z) (1 + e_k(z)) with a_k = 1/sqrt(sum p_k^2) and s
 * a factor of the spaxel's border class alone; *eps = the plan's max |e| (measured on its own
 * tables at creation; +inf for plans the stage does not serve).  Where eps <= 2e-6 (and K <= 24)
 * the stage compares the profiles through accumulators that carry a_k and applies s(z) to the
 * maximum and the minimum (*active = 1): correl and correl_min carry a relative error <= eps, the
 * profile index is that of a maximum up to the same eps.
 * Plans with weight maps (an explicit norm cube): den_k[z, s] = norm[z, s] sum p_k^2 (1 + e)^2; eps
 * is measured by the plan's FIRST run on the norm cube it makes (+inf and inactive before), and
 * where it passes the stage is the same FOLD kernel with rsq(norm) of each voxel behind the loop
 * instead of a second Toeplitz product (the 32 channels at either end keep the two-product
 * kernel); bf16 plans included.
 * ORIGIN_GLR_NO_FOLD=1 in the environment runs the exact form everywhere. */
int origin_glr_plan_fold_eps(origin_glr_plan *plan, float *eps, int *active);
/* The same arithmetic without a plan or a device (host only: num_cu compute units, `terms` = 3
 * for the f16 split, 1 for bf16; n_narrow = profiles of half width <= 16): lets the CPU tests
 * check bench.py's `executed` against the counter passes committed under profiles/. */
int origin_glr_mfma_count_model(int num_cu, int terms, int K, int n_narrow, int Nz, int Ny, int Nx,
                                int P, long *spatial, long *spectral);

/* A GLR run in ROW BANDS: the same results as origin_glr_run, written band by band -- so that the
 * bands whose input is final can start while the greedy PCA still iterates over its last areas
 * (origin_pca_set_tail_hook).  A band is the rows [y0, y1) of the field, y0 a multiple of 64, y1 a
 * multiple of 64 or Ny; its spatial stage reads cube rows [y0 - P/2, y1 + P/2).  flags:
 * ORIGIN_GLR_FIRST on the first band of a run (zeroes the pad channels of the work cube, on the main
 * stream), ORIGIN_GLR_SIDE to enqueue the band on the context's side stream (all compute units but
 * ORIGIN_GLR_SIDE_RESERVE, default an eighth of them; it starts behind everything the main stream was given before
 * the call).  origin_glr_run_finish makes the main stream wait for the side bands and writes the
 * maps (NULL: none).  Only plans whose two stages run the table kernels on the matrix cores
 * (one field, no weight maps, precision 1 or 2, PSF 5..25, profile half widths <= 32, K <= 26):
 * origin_glr_rows_supported; others return ORIGIN_E_STATE. */
#define ORIGIN_GLR_FIRST 1
#define ORIGIN_GLR_SIDE 2
int origin_glr_rows_supported(origin_glr_plan *plan, int *ok);
int origin_glr_run_rows(origin_ctx *ctx, origin_glr_plan *plan, const float *d_cube,
                        const uint8_t *d_mask, float *d_work, float *d_correl, uint8_t *d_profile,
                        float *d_correl_min, int y0, int y1, int flags);
/* The same for a RECTANGLE of the field: rows [y0, y1), columns [x0, x1), each a multiple of 64 or
 * the field's end (origin_glr_run_rows = all columns).  A tile of a tiled field runs the rectangles
 * that stay clear of its halo while the halo strips are on their way (origin_amd/multigpu.py).
 * With x0 > 0 or x1 < Nx a wave of the spectral stage holds 32 columns of ONE row (rows are not
 * multiples of 32 long in general): results agree with origin_glr_run to rounding, not bit for bit. */
int origin_glr_run_rect(origin_ctx *ctx, origin_glr_plan *plan, const float *d_cube,
                        const uint8_t *d_mask, float *d_work, float *d_correl, uint8_t *d_profile,
                        float *d_correl_min, int y0, int y1, int x0, int x1, int flags);
int origin_glr_run_finish(origin_ctx *ctx, origin_glr_plan *plan, float *d_work, float *d_maxmap,
                          float *d_minmap);

/* correl = max_k T_k, profile = first argmax_k, correl_min = min_k T_k (lib :1205-1212).
 * If d_mask != NULL the ComputeTGLR glue is fused: correl[mask] = 0, profile[mask] = 0
 * (steps.py:781, :788) before maxmap = amax_z correl, minmap = amin_z correl_min (:792-793).
 * d_maxmap / d_minmap may be NULL.  d_work: float32 workspace of
 * origin_glr_work_elems(plan) elements. */
int origin_glr_work_elems(origin_glr_plan *plan, size_t *elems);
int origin_glr_run(origin_ctx *ctx, origin_glr_plan *plan, const float *d_cube,
                   const uint8_t *d_mask, float *d_work, float *d_correl,
                   uint8_t *d_profile, float *d_correl_min, float *d_maxmap,
                   float *d_minmap);

/* ---- D. local maxima ---------------------------------------------------------------
 * compute_local_max (lib_origin.py:1220-1256): size^3 maximum_filter (scipy 'reflect'
 * border == clamped window for a max), keep voxels equal to their window maximum and
 * not masked, zero elsewhere; same on -correl_min. */
int origin_local_max(origin_ctx *ctx, const float *d_correl, const float *d_correl_min,
                     const uint8_t *d_mask, int Nz, int Ny, int Nx, int size,
                     float *d_local_max, float *d_local_min);

/* Sparse form of the same pass (size 3, Nx % 4 == 0): the two cubes are > 98 % zeros, and the
 * consumers of cube_local_max / cube_local_min (Compute_threshold_purity lib_origin.py:1391-1479,
 * Detection.run steps.py:935-974) only ever count or pick the non-zero voxels.  Instead of two
 * dense float32 cubes (8 of the pass's 17 B per voxel) the pass appends (linear index z*Ny*Nx +
 * y*Nx + x, value) of every non-zero output to per-wave segments: segment w of `seg_cap` entries
 * starts at w * seg_cap in idx / val, counts[w] (maxima of correl) and counts[nseg + w] (maxima of
 * -correl_min) say how many it holds.  A count above seg_cap means the segment overflowed (entries
 * beyond it were dropped): use the dense form.  origin_local_max_sparse_plan gives nseg and seg_cap
 * for a shape (nseg 0: no sparse form for it); order inside and across segments is the march order
 * of the waves, not np.where's -- consumers that need an order sort what they keep.  The values are
 * bit for bit those of origin_local_max. */
int origin_local_max_sparse_plan(origin_ctx *ctx, int Nz, int Ny, int Nx, long *nseg, int *seg_cap);
int origin_local_max_sparse(origin_ctx *ctx, const float *d_correl, const float *d_correl_min,
                            const uint8_t *d_mask, int Nz, int Ny, int Nx, long nseg, int seg_cap,
                            long long *d_idx_max, float *d_val_max, long long *d_idx_min,
                            float *d_val_min, int *d_counts);
/* consumers: the dense cube of n voxels (zeros + entries); counts above each of nthr thresholds
 * (origin_count_above's contract; keep: uint8 per spaxel, 0 = excluded, or NULL); the entries
 * above a threshold in no particular order (the first `cap`; *h_count = how many there are; aux:
 * a uint8 cube gathered at the hits, or NULL); per-spaxel maximum over z, 0 where a column has no
 * positive entry (the dense cube's columns always hold zeros). */
int origin_sparse_to_dense(origin_ctx *ctx, const long long *d_idx, const float *d_val,
                           const int *d_counts, long nseg, int seg_cap, float *d_dense, long n);
int origin_sparse_count_above(origin_ctx *ctx, const long long *d_idx, const float *d_val,
                              const int *d_counts, long nseg, int seg_cap, const uint8_t *d_keep,
                              long S, int nthr, const double *h_thr, long *h_counts);
int origin_sparse_where_above(origin_ctx *ctx, const long long *d_idx, const float *d_val,
                              const int *d_counts, long nseg, int seg_cap, double threshold,
                              const uint8_t *d_aux, long cap, long long *d_out_idx,
                              float *d_out_val, uint8_t *d_out_aux, long *h_count);
int origin_sparse_zmax_map(origin_ctx *ctx, const long long *d_idx, const float *d_val,
                           const int *d_counts, long nseg, int seg_cap, const uint8_t *d_keep,
                           long S, float *d_map);

#ifdef __cplusplus
}
#endif
#endif /* ORIGIN_HIP_H */
