/*
 * msm_amd.h -- C ABI of the MI355X multi-scalar-multiplication engine (libmsm_amd.so).
 *
 * Drop-in boundary for the reference's WASM export surface (one set per curve x group):
 *
 *   reference export (wasmcurves/src/build_multiexp.js)            this library
 *   ------------------------------------------------------------   -------------------------------------
 *   g1m_multiexp(pBases,pScalars,scalarSize,n,pr)        :251-257   msm_<curve>_g1m_multiexp
 *   g1m_multiexpAffine(pBases,pScalars,scalarSize,n,pr)  :251-257   msm_<curve>_g1m_multiexpAffine
 *   g1m_multiexp_chunk(...,startBit,chunkSize,pr)        :96-104    msm_<curve>_g1m_multiexp_chunk
 *   g1m_multiexpAffine_chunk(...)                        :96-104    msm_<curve>_g1m_multiexpAffine_chunk
 *   g2m_* (same four, Fp2 coordinates)                              msm_<curve>_g2m_*
 *   (wired in wasmcurves/src/build_curve_jacobian_a0.js:1426-1430; <curve> = bls12381 | bn128)
 *
 * Data formats are the reference's linear-memory images:
 *   field element : n8 bytes little-endian, Montgomery form (value * 2^(8*n8) mod q), < q
 *                   n8 = 48 (bls12381) / 32 (bn128); an Fp2 element is c0 || c1
 *   affine point  : x || y            ; infinity = all-zero bytes
 *   Jacobian point: X || Y || Z       ; infinity <=> Z == 0
 *   scalars       : n * scalar_size bytes, plain little-endian integers (not Montgomery, not
 *                   reduced mod r; any value < 2^(8*scalar_size)), 1 <= scalar_size <= 64; any n < 2^32 for the
 *                   synchronous calls (inputs beyond 2^26 points are processed in slices), n <= 2^27 per asynchronous call
 *   output        : one Jacobian point, ALWAYS normalised: (x, y, R) or (0, R, 0) for infinity,
 *                   i.e. byte-equal to g?m_normalize() of the reference's result
 *                   (build_curve_jacobian_a0.js:940-973).
 *
 * Pointers are host pointers unless the call carries MSM_AMD_DEVICE_PTRS.  The caller owns every
 * buffer; nothing is retained after return.  The reference functions return void and trap on
 * bad memory; these return 0 on success or a negative MSM_AMD_E* code (msm_amd_last_error()
 * gives the text).  n == 0 yields infinity, as build_multiexp.js:283-287 / :119-125.
 * A context may be shared by threads: the multiexp entry points serialise submission and collection inside the context and
 * overlap the GPU work of concurrent callers (up to `slots` calls at a time; further callers wait).  One context per GPU.
 * msm_amd_set_option and msm_amd_last_error are per context, not per thread.
 */
#ifndef MSM_AMD_H
#define MSM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msm_amd_ctx msm_amd_ctx;
typedef struct msm_amd_bases msm_amd_bases; /* prepared bases (msm_amd_prepare_bases) */

enum { MSM_AMD_BLS12381 = 0, MSM_AMD_BN128 = 1 };

/* flags */
enum {
    MSM_AMD_BASES_JACOBIAN = 1, /* bases are X||Y||Z (g?m_multiexp); default affine x||y (g?m_multiexpAffine) */
    MSM_AMD_DEVICE_PTRS = 2,    /* bases and scalars already live in this context's GPU memory              */
    MSM_AMD_FFT_INVERSE = 8,    /* msm_amd_fft: the inverse transform (g?m_ifft / frm_ifft)                                */
    MSM_AMD_DEVICE_BASES = 4,   /* only the bases are resident on the GPU (msm_amd_dev_alloc + msm_amd_h2d once: the fixed
                                   SRS of a prover); the scalars are a host buffer copied over PCIe by every call    */
    MSM_AMD_BASES_IN_SUBGROUP = 16 /* the caller vouches that every base lies in the order-r subgroup (what
                                   msm_amd_batch_in_group_affine checks; true of any SRS).  Allows the engine's GLV schedule
                                   (BLS12-381 G1, 180000 <= n <= 2^18), which reduces scalars mod r and uses the endomorphism: correct in
                                   the subgroup only.  Without the flag the result is the reference's for ANY points of the curve. */
};

/* error codes */
enum {
    MSM_AMD_OK = 0,
    MSM_AMD_EINVAL = -1,   /* bad argument (null pointer, scalar_size out of range, bad curve/group, misaligned device pointer) */
    MSM_AMD_EHIP = -2,     /* HIP runtime failure (no GPU, launch failure, a failed copy) */
    MSM_AMD_ENOMEM = -3,   /* out of memory, host or device (the error text names the allocation: scratch of a call, a prepared table) */
    MSM_AMD_EPEER = -4     /* multi-GPU call: another rank of the group failed its part (this rank's part was fine) */
};

/* ---- context ------------------------------------------------------------------------------- */
int msm_amd_create(int device_id, msm_amd_ctx **out_ctx);
void msm_amd_destroy(msm_amd_ctx *ctx);
const char *msm_amd_last_error(const msm_amd_ctx *ctx);
/* Optional tuning: window size c in bits (0 = automatic). */
int msm_amd_set_window_bits(msm_amd_ctx *ctx, uint32_t c);
/* Options of a context (msm_amd_set_option; 0 = automatic unless stated).  Environment: MSM_AMD_C, MSM_AMD_SEG, MSM_AMD_REDUCE_L
 * (= window_bits, segment, reduce_l at create), MSM_AMD_DIST_TIMEOUT_MS, MSM_AMD_RCCL_LIB (the collective library of a multi-GPU
 * group, default librccl.so.1), MSM_AMD_NO_CALIBRATE (keep the streams in creation order), MSM_AMD_DEBUG_TAIL (print segment / long
 * bucket / hot bucket counts of every collected call); experiments only: MSM_AMD_S_PRIORITY / MSM_AMD_A_PRIORITY (HIP priority of the sort /
 * accumulation stream at create; measured: no effect).
 *
 *   -- what a host may want to set -------------------------------------------------------------------------------------------------
 *   slots            calls in flight per context, 1..8 (default 2); each slot owns its scratch: HBM per slot below
 *   window_bits      window size c in bits, <= 21 (also msm_amd_set_window_bits); at msm_amd_prepare_bases: the table's window size
 *   slice            points per pass of a synchronous call (default 2^26); larger inputs are processed slice by slice
 *   glv              0 never | 1 (default) where it pays AND the call carries MSM_AMD_BASES_IN_SUBGROUP | 2 whenever it applies (the option
 *                    itself being the caller's word that the bases lie in the order-r subgroup): the GLV schedule of BLS12-381 G1
 *   even_windows     default 1: when c does not divide the scalar width the windows are c and c - 1 bits wide and tile the width that
 *                    32-byte scalars are EXPECTED to use (the group order's: 255 / 254 bits) exactly, with one more window for the bits
 *                    above, which reduced scalars leave empty; scalars that do use those bits give the same result and switch the next
 *                    16 calls of the context to windows over all 256 bits.  0 = windows of c bits only (the round-2 layout)
 *   narrow           default 1: the windows follow the OBSERVED width of 32-byte scalars.  Every call reports its widest scalar; the maximum over
 *                    the context's last eight calls, when below the group order's width (128-bit challenges, 64-bit values, a 253-bit field),
 *                    is the width the next calls' windows tile -- fewer windows, nothing carried into a single bucket: 1.2 instead of 2.1 ms
 *                    at n = 2^20 for 64-bit scalars.  Exact whatever follows: a few bits more land in an overflow window; a wider scalar is
 *                    detected on the GPU and the call is run again over the declared width inside its wait (that one call pays twice; the
 *                    hint follows from the next call on; streams that alternate widths keep the wide layout).  0 = never; setting the
 *                    option forgets what was observed
 *   zero_copy        default 1: the window partials are written by the reduction kernel straight into pinned host memory
 *   dist_gw          window groups of a call of a multi-GPU group (0 = planned per call, else a divisor of the world size); must be the
 *                    same on every rank (checked in the exchange: MSM_AMD_EINVAL on every rank otherwise)
 *   dist_timeout_ms  how long an exchange of a multi-GPU group waits for its peers (default 300000; 0 = for ever); after that the
 *                    communicator is aborted and the group is dead (msm_amd_dist_wait)
 *
 *   -- tests and before / after measurements (results are identical whatever they are set to) ---------------------------------------
 *   segment          max entries one lane adds before a long bucket is cut into segments for the tail kernels (tests force splitting)
 *   hot              default 1: a bucket beyond 32 x the segment length is cut into short segments folded by 32 workgroups; 0 = ordinary segments
 *   items            the accumulation's work list (lanes per bucket in proportion to its population): 0 where a launch is short of filling
 *                    the chip, 1 wherever the populations allow, 2 never;   items_t: entries per lane of the work list (0 = per launch)
 *   split2           default 1: launches with few, long buckets cut them into slices; 0 = one lane per bucket always
 *   splitk           lanes per bucket of a sliced launch (0 = per launch, 1..64 forced);   splitk_lanes: the lane count it aims at (default 327680)
 *   sort_big         fine sort of spans beyond one workgroup: 0 chunk-parallel kernels where such spans are expected, 1 always, 2 never
 *   sort_geom        partition geometry: 0 round 4's by size, 1 round 3's (256 coarse bins, 9216-entry spans), 2 = 512 bins wherever the
 *                    window allows, 3 = 2 + the 1024-lane fine sort
 *   reduce_l         buckets per lane of the bucket reduction (a power of two)
 *   reduce_planes    the bucket reduction as plain sums whose weights the host's Horner pass applies (k_bucket_planes): 1 (default) where it
 *                    pays, 2 wherever it can, 0 never (k_bucket_reduce: running sums + a double-and-add per lane)
 *   defer_reduce     1 (default): a G1 call that is part of a stream of calls (others in flight at its submission, or lately) with a full-size
 *                    accumulation has its reduction queued by the next submission on the context (in the form that hides behind that call's
 *                    accumulation) or by the wait for its own ticket (nothing followed: the short lone form) -- so such a call only
 *                    completes once one of the two has happened: wait for tickets, do not poll for them.  0: every reduction is queued at
 *                    submission, in the form the calls in flight at that moment suggest.  2: G2 calls are deferred too (measured: slower)
 *
 *   -- experiments that were measured and left off, kept for re-measurement ---------------------------------------------------------
 *   split (1 = reduce a lone call's windows in two overlapping groups: slower), reduce_pair (1 = the G1 reduction on lane pairs: no gain),
 *   acc_variant (build of the G1 accumulation: 1 two interleaved multiply-add chains, 2 one chain), acc_stream (1 the slot's stream, 2
 *   stream A), tile_mode (scalars per partition tile: 1 = 1024, 2 = 4096), chunk_mode (chunk-parallel fine sort: 1 = 4096-entry chunks on
 *   256 lanes, 2 = 8192 on 1024)
 *
 * HBM a context holds beside the caller's inputs: per slot, the scratch of its largest call so far -- at n = 2^20 (BLS12-381 G1, 16-bit
 * windows): 64 MiB partition entries + 64 MiB index lists + 119 MiB buckets + ~0.3 GB for the segment lists of skewed inputs (sized for the
 * worst case: every entry in a hot bucket: up to 2^20 + entries / segment segments of one bucket image each; 0.6 GB for G2) + 6 MiB of
 * counters; at n = 2^26 (20-bit windows) ~13 GiB.  Per context: ~70 MB of kernel scratch per hardware queue (GPU_MAX_HW_QUEUES, default
 * 4), reserved by msm_amd_create -- which fails with MSM_AMD_ENOMEM when the device cannot hold it: an exhausted device then yields
 * error codes, never an abort.  Prepared tables are extra (msm_amd_plan_prepared says how much). */
int msm_amd_set_option(msm_amd_ctx *ctx, const char *name, uint32_t value);
/* Diagnostic: out[11*11]; out[i*11+j] = 1 when streams i and j (the context's A, S, B0, C0, B1, C1, B2, C2, B3, C3; 10 = the null stream) share a
 * hardware queue (their kernels serialise).  HIP maps streams onto hardware queues in creation order across the whole process, and
 * the engine's throughput depends on that mapping. */
int msm_amd_debug_stream_sharing(msm_amd_ctx *ctx, uint8_t *out);

/* ---- generic entry points -------------------------------------------------------------------- */
/* R = sum_i scalar_i * base_i                                 (build_multiexp.js:251-371) */
int msm_amd_multiexp(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *bases, const void *scalars,
                     uint32_t scalar_size, uint32_t n, uint8_t *out);
/* R = sum_i getChunk(scalar_i, start_bit, chunk_size) * base_i (build_multiexp.js:96-249, :25-94) */
int msm_amd_multiexp_chunk(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *bases,
                           const void *scalars, uint32_t scalar_size, uint32_t n, uint32_t start_bit,
                           uint32_t chunk_size, uint8_t *out);

/* ---- pipelined form ------------------------------------------------------------------------------ */
/* msm_amd_multiexp_async queues the MSM and returns a ticket; msm_amd_wait blocks until it is done and
 * writes the result.  Up to `slots` tickets (default 2, option "slots" 1..8) may be outstanding per context:
 * the engine overlaps the bucket reductions of older calls (one HIP stream per slot) with the sort and bucket
 * accumulation of the newest one (each slot has its own scratch buffers).  Inputs must stay valid and unchanged until the ticket has been waited for.
 * msm_amd_multiexp(...) == msm_amd_multiexp_async(...) + msm_amd_wait(...).
 * When every slot is taken a submission waits as long as at least one of them is held by a SYNCHRONOUS call of another thread (that
 * thread will free it); if all hold asynchronous tickets only the caller could free one, and the call fails with MSM_AMD_EINVAL
 * ("wait for an older ticket first").  Calls may be issued from several threads on one context.            */
int msm_amd_multiexp_async(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *bases,
                           const void *scalars, uint32_t scalar_size, uint32_t n, uint64_t *ticket);
int msm_amd_wait(msm_amd_ctx *ctx, uint64_t ticket, uint8_t *out);

/* ---- prepared bases: the fixed points of a prover (its SRS), set up once ------------------------------------------------------
 * The reference is called with the same bases over and over (wasmcurves/benchmarks/multiexp.js:18-41; a Groth16 / PLONK prover's
 * SRS); what it recomputes per call that depends on the bases alone can be kept.  msm_amd_prepare_bases builds, on the GPU, the
 * table 2^(start bit of window w) * P_i for every window w (W x n affine points in HBM: 1.25 GB for n = 2^20 BLS12-381 G1 points,
 * 13 windows).  A call on the handle then sorts the digit of EVERY window of every scalar into ONE bucket set -- entry (i, w)
 * addresses table point w * n + i -- so buckets, bucket reduction and the window combine shrink from W windows' worth to one, and
 * the windows can be wider: n = 2^20 runs 13 windows over 2^19 buckets, 13 instead of 16 bucket additions per pair.
 * Same result bytes as msm_amd_multiexp on the same points (the normalised Jacobian point).
 *   scalar_bits  the table serves scalars below 2^scalar_bits; 0 = the bit length of the curve's group order (255 / 254: any scalar
 *                reduced mod r).  A call that meets a wider scalar fails with MSM_AMD_EINVAL -- nothing is truncated silently.
 *   flags        msm_amd_prepare_bases: MSM_AMD_BASES_JACOBIAN, MSM_AMD_DEVICE_PTRS (bases resident on this GPU);
 *                msm_amd_multiexp_prepared: MSM_AMD_DEVICE_PTRS (scalars resident).
 *   n            a call may use any prefix of the prepared points.
 * The asynchronous form returns a ticket for msm_amd_wait.  A handle is released with msm_amd_release_bases once no call on it is
 * in flight; it belongs to the context that prepared it (msm_amd_destroy frees the tables of handles never released -- the handles
 * are invalid afterwards).  Option "window_bits" (>= 3) at prepare time forces the table's window size. */
int msm_amd_prepare_bases(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *bases, uint32_t n,
                          uint32_t scalar_bits, msm_amd_bases **out);
int msm_amd_release_bases(msm_amd_ctx *ctx, msm_amd_bases *bases);
/* out6: [0] points, [1] scalar_bits, [2] c, [3] windows W, [4] signed windows (c bits; the other W - [4] are c - 1 bits, unsigned),
 * [5] buckets = 2^(c-1); *table_bytes (may be NULL): HBM held by the table */
int msm_amd_bases_info(const msm_amd_bases *bases, uint32_t *out6, uint64_t *table_bytes);
/* The same figures for a table that does not exist yet -- what msm_amd_prepare_bases would choose for n points and scalars below
 * 2^scalar_bits (0 = the group order's width), window_bits = 0 or the forced window size -- so that a host can budget the HBM
 * (W x n points: 12.5 x the bases at n = 2^20) before it prepares anything.  Needs neither a GPU nor a context. */
int msm_amd_plan_prepared(int curve, int group, uint32_t n, uint32_t scalar_bits, uint32_t window_bits, uint32_t *out6,
                          uint64_t *table_bytes);
int msm_amd_multiexp_prepared(msm_amd_ctx *ctx, msm_amd_bases *bases, uint32_t flags, const void *scalars,
                              uint32_t scalar_size, uint32_t n, uint8_t *out);
int msm_amd_multiexp_prepared_async(msm_amd_ctx *ctx, msm_amd_bases *bases, uint32_t flags, const void *scalars,
                                    uint32_t scalar_size, uint32_t n, uint64_t *ticket);

/* ---- multi-GPU: one shard of the Pippenger windows -------------------------------------------- */
/* Computes the contribution of windows {w : w % nshards == shard} (already weighted by 2^(c*w)),
 * written as an (un-normalised) Jacobian point.  Summing the nshards partials gives the MSM.      */
int msm_amd_multiexp_shard(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *bases,
                           const void *scalars, uint32_t scalar_size, uint32_t n, uint32_t shard, uint32_t nshards,
                           uint8_t *out_partial);
/* pipelined form of the same (collect with msm_amd_wait; the ticket yields the un-normalised partial) */
int msm_amd_multiexp_shard_async(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *bases,
                                 const void *scalars, uint32_t scalar_size, uint32_t n, uint32_t shard,
                                 uint32_t nshards, uint64_t *ticket);
/* out = normalise(sum of `count` Jacobian partials) -- host only, no GPU needed */
int msm_amd_combine(int curve, int group, const uint8_t *partials, uint32_t count, uint8_t *out);

/* ---- multi-GPU: the exchange itself, RCCL over xGMI behind the C ABI (one process per GPU) ------------------------------
 * The reference offers two seams and a call is split along both: its per-window export fnName_chunk (build_multiexp.js:96-104, :469),
 * which its external caller fans out to workers and recombines on the main thread, and the MSM's linearity over slices of the points
 * (what that caller does above 2^22 points; :251-371 is one slice).  Here the workers are the GPUs of one node: the `world` ranks form
 * gp point slices x gw window groups (gp * gw = world, chosen per call from the per-rank cost model; option "dist_gw" forces gw), rank
 * r = p * gw + w computes the windows {k : k % gw == w} of slice p.
 *   rank 0:      msm_amd_dist_unique_id(id)           -> 128 bytes, handed to the other ranks by whatever channel the host has
 *   every rank:  msm_amd_dist_init(ctx, rank, world, id)   (collective: returns when all ranks have joined)
 *                msm_amd_multiexp_dist(ctx, ...)      -> every rank passes ALL n points and scalars (resident or host memory) and works
 *                                                        on its part of them; ONE ncclAllGather of world x 304 bytes (status + Jacobian
 *                                                        partial), the partials added on the host: the same normalised result on every rank
 *                msm_amd_dist_plan(ctx, ..., n, 0, 0, &gp, &gw, &first, &count) + msm_amd_multiexp_dist_local(ctx, ..., bases_local, scalars_local, count, n)
 *                                                     -> the same for a rank that holds only ITS slice [first, first + count) of the n
 *                                                        points (6 GiB of bases at n = 2^26 become 0.75 GiB per GPU at gp = 8)
 *                msm_amd_dist_destroy(ctx)            (also done by msm_amd_destroy)
 * The asynchronous pair keeps several calls in flight per rank (up to `slots`).  Every rank must SUBMIT the same calls in the same
 * order; the exchanges then happen in that order on every rank.  The tickets may be waited for in ANY order and from any threads: a
 * wait for a later ticket performs the exchanges of the earlier ones on the way and parks their results for whoever asks for them
 * (every ticket must still be waited for, or the group left with msm_amd_dist_destroy, which collects the rest).  A call whose part cannot
 * be queued or fails on one rank still completes on all: that rank gets its own error, the others MSM_AMD_EPEER.  The ranks' headers are
 * compared in every exchange: ranks that planned different splits of the call, or different window layouts inside one point slice
 * (options that differ between ranks), get MSM_AMD_EINVAL on EVERY rank instead of a sum of partials that do not belong together.
 * Nobody is left waiting in a collective: an exchange is bounded by option "dist_timeout_ms" (default 5 minutes); when the peers do not
 * arrive, or the collective fails locally, the communicator is aborted (ncclCommAbort) and the group is DEAD on this rank -- the call,
 * every pending ticket and every later call of the group return MSM_AMD_EPEER at once, plain calls on the context keep working, and
 * msm_amd_dist_destroy + a new msm_amd_dist_init on every rank bring the group back.
 * RCCL is dlopen'ed on first use (MSM_AMD_RCCL_LIB names another library with the same entry points: the tests' shared-memory
 * stand-in, tests/host/fake_rccl.c).  With more than four ranks export GPU_MAX_HW_QUEUES=8 before the process
 * starts (read by the HIP runtime at start-up; msm_amd_dist_init warns when it is missing).                                    */
int msm_amd_dist_unique_id(uint8_t *out128);
int msm_amd_dist_init(msm_amd_ctx *ctx, int rank, int world, const uint8_t *id128);
int msm_amd_dist_destroy(msm_amd_ctx *ctx);
int msm_amd_dist_info(const msm_amd_ctx *ctx, int *rank, int *world); /* world = 0: not in a group */
/* *count = the ranks the communicator itself reports (ncclCommCount) -- what a launcher prints to show that the collective library saw
 * every rank; *dead (may be NULL) = 1 once the group has been aborted (see msm_amd_dist_wait).  MSM_AMD_EINVAL outside a group. */
int msm_amd_dist_comm_count(const msm_amd_ctx *ctx, int *count, int *dead);
/* the split of a call of n points over a group of `world` ranks and the slice of the points rank `rank` = p * gw + w takes
 * (world = 0: this context's own group and rank).  With world != 0 ctx may be NULL (no GPU needed: a launcher can place the
 * slices before any context exists); a context contributes its "dist_gw" option. */
int msm_amd_dist_plan(const msm_amd_ctx *ctx, int curve, int group, uint32_t scalar_size, uint32_t n, uint32_t world, uint32_t rank,
                      uint32_t *gp, uint32_t *gw, uint32_t *first, uint32_t *count);
int msm_amd_multiexp_dist(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *bases, const void *scalars,
                          uint32_t scalar_size, uint32_t n, uint8_t *out);
int msm_amd_multiexp_dist_async(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *bases,
                                const void *scalars, uint32_t scalar_size, uint32_t n, uint64_t *ticket);
/* bases_local / scalars_local: the n_local points of this rank's slice; n_local must equal the `count` of msm_amd_dist_plan for
 * n_total (checked: the split depends on the context's options, which must not change between plan and call) */
int msm_amd_multiexp_dist_local(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *bases_local,
                                const void *scalars_local, uint32_t scalar_size, uint32_t n_local, uint32_t n_total, uint8_t *out);
int msm_amd_multiexp_dist_local_async(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *bases_local,
                                      const void *scalars_local, uint32_t scalar_size, uint32_t n_local, uint32_t n_total,
                                      uint64_t *ticket);
/* prepared bases in a group: the points cut along the points only, every rank holding the PREPARED table of its own slice (the slices
 * need not be equal); this rank's part is a whole msm_amd_multiexp_prepared over its n_local points.  Collect the asynchronous form with
 * msm_amd_dist_wait(ctx, curve, group of the table, ticket, out). */
int msm_amd_multiexp_dist_prepared(msm_amd_ctx *ctx, msm_amd_bases *local_table, uint32_t flags, const void *scalars_local,
                                   uint32_t scalar_size, uint32_t n_local, uint8_t *out);
int msm_amd_multiexp_dist_prepared_async(msm_amd_ctx *ctx, msm_amd_bases *local_table, uint32_t flags, const void *scalars_local,
                                         uint32_t scalar_size, uint32_t n_local, uint64_t *ticket);
int msm_amd_dist_wait(msm_amd_ctx *ctx, int curve, int group, uint64_t ticket, uint8_t *out);

/* ---- reference-named wrappers (host pointers) ------------------------------------------------- */
#define MSM_AMD_DECL(curve, g)                                                                                       \
    int msm_##curve##_##g##_multiexp(msm_amd_ctx *ctx, const uint8_t *bases, const uint8_t *scalars,               \
                                     uint32_t scalar_size, uint32_t n, uint8_t *out);                              \
    int msm_##curve##_##g##_multiexpAffine(msm_amd_ctx *ctx, const uint8_t *bases, const uint8_t *scalars,         \
                                           uint32_t scalar_size, uint32_t n, uint8_t *out);                        \
    int msm_##curve##_##g##_multiexp_chunk(msm_amd_ctx *ctx, const uint8_t *bases, const uint8_t *scalars,         \
                                           uint32_t scalar_size, uint32_t n, uint32_t start_bit,                   \
                                           uint32_t chunk_size, uint8_t *out);                                     \
    int msm_##curve##_##g##_multiexpAffine_chunk(msm_amd_ctx *ctx, const uint8_t *bases, const uint8_t *scalars,   \
                                                 uint32_t scalar_size, uint32_t n, uint32_t start_bit,             \
                                                 uint32_t chunk_size, uint8_t *out);
MSM_AMD_DECL(bls12381, g1m)
MSM_AMD_DECL(bls12381, g2m)
MSM_AMD_DECL(bn128, g1m)
MSM_AMD_DECL(bn128, g2m)
#undef MSM_AMD_DECL

/* ---- comparison / conversion helpers the reference's harness uses (host, no GPU) --------------- */
/* g?m_normalize (build_curve_jacobian_a0.js:940-973), g?m_toAffine (:977-1011), g?m_eq (:173-238) */
int msm_amd_normalize(int curve, int group, const uint8_t *jac_in, uint8_t *jac_out);
int msm_amd_to_affine(int curve, int group, const uint8_t *jac_in, uint8_t *aff_out);
int msm_amd_eq(int curve, int group, const uint8_t *jac_a, const uint8_t *jac_b); /* 1 equal, 0 not, <0 error */
/* f1m_toMontgomery / f1m_fromMontgomery (build_f1m.js:1089-1110) on `count` base-field elements */
int msm_amd_f_to_montgomery(int curve, const uint8_t *in, uint32_t count, uint8_t *out);
int msm_amd_f_from_montgomery(int curve, const uint8_t *in, uint32_t count, uint8_t *out);

/* ---- batch conversions either side of the MSM, on the GPU ----------------------- */
/* g?m_batchToAffine   (build_curve_jacobian_a0.js:1040-1125)  n Jacobian -> n affine; Z == 0 -> (0, 0)
 * g?m_batchToJacobian (:1014-1038, :1418)                     n affine -> n Jacobian (x, y, R); (0,0) -> (0, R, 0)
 * g?m_batchLEMtoU     (:1206-1231, :1413)  affine LE-Montgomery -> x || y big-endian, non-Montgomery; infinity = 0x40 00 ...
 * g?m_batchUtoLEM     (:1233-1258, :1415)  the inverse
 * flags: MSM_AMD_DEVICE_PTRS when `in` and `out` are GPU buffers (16-byte aligned).                                */
int msm_amd_batch_to_affine(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *jac_in, uint32_t n, void *aff_out);
int msm_amd_batch_to_jacobian(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *aff_in, uint32_t n, void *jac_out);
int msm_amd_batch_lem_to_u(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *aff_in, uint32_t n, void *u_out);
int msm_amd_batch_u_to_lem(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *u_in, uint32_t n, void *aff_out);
/* g?m_batchLEMtoC / g?m_batchCtoLEM (build_curve_jacobian_a0.js:1168-1205, 1265-1328, 1414-1416): n affine points <-> n compressed
 * points of one field element each (x big-endian plain; first byte: 0x80 = y is the "negative" root, 0x40 = infinity).  Infinity is
 * the affine (0, 0): the reference's LEMtoC tests it with the Jacobian _isZero, i.e. looks past its affine input.  CtoLEM returns
 * MSM_AMD_EINVAL (and writes infinity) where x^3 + b is not a square; the reference does not terminate on such input.          */
int msm_amd_batch_lem_to_c(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *aff_in, uint32_t n, void *c_out);
int msm_amd_batch_c_to_lem(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *c_in, uint32_t n, void *aff_out);
/* g?m_inCurveAffine (:1330-1353) over a batch: out_flags[i] = 1 iff y^2 == x^3 + b for affine point i ((0,0) gives 0, as
 * in the reference).  out_flags is a host buffer of n bytes unless MSM_AMD_DEVICE_PTRS is set.                           */
int msm_amd_batch_in_curve_affine(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *points, uint32_t n,
                                  uint8_t *out_flags);
/* g1m_inGroupAffine / g2m_inGroupAffine (bls12381/build_bls12381.js:1494-1576, 1370-1480), batched on the GPU:
 * out_flags[i] = 1 for affine infinity, 0 for a point off the curve, else 1 iff the point lies in the order-r subgroup.
 * BLS12-381 only, as in the reference.                                                                          */
int msm_amd_batch_in_group_affine(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *points, uint32_t n,
                                  uint8_t *out_flags);

/* ---- batchApplyKey (reference: wasmcurves/src/build_applykey.js:20-76 wired at
 * bls12381/build_bls12381.js:79-92 and bn128/build_bn128.js:84-88) ------------------------------------------------------ */
/* g?m_batchApplyKey (flag MSM_AMD_BASES_JACOBIAN: Jacobian inputs) / g?m_batchApplyKeyMixed (affine inputs), on the GPU:
 * out[i] = (first * inc^i) * in[i], first32 / inc32 = Fr elements in Montgomery form (32 bytes LE, as frm_*).  out: n Jacobian
 * points, NORMALISED (the reference returns some representative; compare through g?m_normalize).  MSM_AMD_DEVICE_PTRS applies
 * to in/out.                                                                                                         */
int msm_amd_batch_apply_key(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *in, uint32_t n,
                            const uint8_t *first32, const uint8_t *inc32, void *out);
/* frm_batchApplyKey: out[i] = in[i] * first * inc^i in Fr (all Montgomery, 32 bytes LE each), byte-identical to the reference. */
int msm_amd_fr_batch_apply_key(msm_amd_ctx *ctx, int curve, uint32_t flags, const void *in, uint32_t n, const uint8_t *first32,
                               const uint8_t *inc32, void *out);

/* frm_fft / frm_ifft (group = 0: n Fr elements, Montgomery, 32 bytes each; byte-identical to the reference) and g1m_/g2m_ fft /
 * ifft (group = 1 / 2: n Jacobian points in, n NORMALISED Jacobian points out) -- wasmcurves/src/build_fft.js:178-245 (fft, ifft),
 * :247-394 (rawfft), :396-516 (fftFinal), roots of unity :33-50.  n a power of two <= 2^27; flags: MSM_AMD_FFT_INVERSE,
 * MSM_AMD_DEVICE_PTRS.  The reference transforms in place; here in -> out (they may be the same buffer).           */
int msm_amd_fft(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, const void *in, uint32_t n, void *out);

/* Building blocks of the blocked FFT (what ffjavascript drives for transforms larger than one WASM instance), build_fft.js:657-1353,
 * for group = 0 (Fr elements: 32 bytes Montgomery, byte-identical to frm_*) and 1 / 2 (Jacobian points in, NORMALISED Jacobian points
 * out, as g1m_* / g2m_*).  All transform their buffers IN PLACE, as the reference does.  first32 / inc32 / factor32: Fr elements in
 * Montgomery form; the reference's running key t = t * inc becomes first * inc^i computed by lane i.
 *   msm_amd_fft_join                    <p>_fftJoin(pBuff1, pBuff2, n, first, inc)                     :657-746
 *   msm_amd_fft_join_ext                <p>_fftJoinExt(..., totalBits); with MSM_AMD_FFT_INVERSE: <p>_fftJoinExtInv   :749-988
 *   msm_amd_prepare_lagrange_evaluation <p>_prepareLagrangeEvaluation(pBuff1, pBuff2, n, first, inc, totalBits)    :991-1112
 *   msm_amd_fft_mix                     <p>_fftMix(pBuff, n, exp): one radix-2 pass of span 2^exp      :1114-1236
 *   msm_amd_fft_final                   <p>_fftFinal(pBuff, n, factor): out[i] = factor * in[n-1-i]    :1238-1353
 *   msm_amd_fr_batch_montgomery         frm_batchToMontgomery / frm_batchFromMontgomery (build_batchconvertion.js), to_montgomery = 1 / 0 */
int msm_amd_fft_join(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, void *buff1, void *buff2, uint32_t n,
                     const uint8_t *first32, const uint8_t *inc32);
int msm_amd_fft_join_ext(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, void *buff1, void *buff2, uint32_t n,
                         const uint8_t *first32, const uint8_t *inc32, uint32_t total_bits);
int msm_amd_prepare_lagrange_evaluation(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, void *buff1, void *buff2,
                                        uint32_t n, const uint8_t *first32, const uint8_t *inc32, uint32_t total_bits);
int msm_amd_fft_mix(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, void *buff, uint32_t n, uint32_t exp);
int msm_amd_fft_final(msm_amd_ctx *ctx, int curve, int group, uint32_t flags, void *buff, uint32_t n, const uint8_t *factor32);
int msm_amd_fr_batch_montgomery(msm_amd_ctx *ctx, int curve, uint32_t flags, const void *in, uint32_t n, int to_montgomery,
                                void *out);

/* ---- GLV pre-pass for BLS12-381 G1 (reference: wasmcurves/src/build_glv.js) ------- */
/* g1m_glv_decomposeScalar (:53-146): 32-byte scalar k -> out64 = |k1| (16 bytes, then 16 zero) || |k2| (same),
 * *sign bit 0 = (k1 >= 0), bit 1 = (k2 >= 0), with k = k1 + k2*lambda (mod r).  Host only.               */
int msm_amd_glv_decompose_scalar(const uint8_t *scalar32, uint8_t *out64, uint32_t *sign);
/* g1m_glv_endomorphism (:150-163): affine (x, y) -> (beta*x, sign ? y : -y).  Host only.                     */
int msm_amd_glv_endomorphism(const uint8_t *point_affine, uint32_t sign, uint8_t *out_affine);
/* g1m_glv_preprocessEndomorphism (:178-267) on the GPU: n (affine point, 32-byte scalar) pairs -> 2n pairs whose
 * scalars fit 128 bits; msm(points_out, scalars_out, 2n) == msm(points, scalars, n).  flags: MSM_AMD_DEVICE_PTRS. */
int msm_amd_glv_preprocess(msm_amd_ctx *ctx, uint32_t flags, const void *points, const void *scalars, uint32_t n,
                           void *points_out, void *scalars_out);

/* The split the engine itself applies inside BLS12-381 G1 MSMs (not a reference export; exposed so that it can be
 * checked without a GPU): k == s * (k1 + k2*lambda) (mod r), |k1|, k2 < 2^127.  out32 = |k1| (16 bytes LE) || k2 (16 bytes);
 * *sign bit 0 = the P term is negative, bit 1 = the phi(P) term is negative.  Host only.  Option "glv"
 * (msm_amd_set_option): 0 = never split, 1 = where it pays (default: 180000 <= n <= 2^18), 2 = whenever it applies (n >= 1024; forces 16-bit windows).      */
int msm_amd_glv_split(const uint8_t *scalar32, uint8_t *out32, uint32_t *sign);

/* ---- device memory + measurement utilities (bench / tests) ------------------------------------- */
int msm_amd_dev_alloc(msm_amd_ctx *ctx, size_t bytes, void **dev_ptr);
int msm_amd_dev_free(msm_amd_ctx *ctx, void *dev_ptr);
int msm_amd_h2d(msm_amd_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int msm_amd_d2h(msm_amd_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
/* Fills dev_out with n affine points k_i * G (k_i from a counter-based PRNG of `seed`): synthetic bases. */
int msm_amd_gen_points(msm_amd_ctx *ctx, int curve, int group, uint64_t seed, uint32_t n, void *dev_out);
/* points [first, first + n) of the same sequence: a rank of a multi-GPU group generates only its slice */
int msm_amd_gen_points_range(msm_amd_ctx *ctx, int curve, int group, uint64_t seed, uint32_t first, uint32_t n, void *dev_out);
/* Per-stage GPU time of the last call on this context, HIP events on the engine's stream (ms).
 * Order: [0] total GPU, [1] digits+histogram, [2] scan+order, [3] scatter, [4] bucket accumulation,
 *        [5] bucket reduction, [6] host combine (wall), [7] bases conversion (Jacobian input only).
 * Returns the number of entries written. */
int msm_amd_get_timings(const msm_amd_ctx *ctx, float *ms, int capacity);
/* Window geometry of the last call: [0] c, [1] windows W, [2] buckets per window, [3] windows processed,
 * [4],[5] low/high word of the number of (pair, window) entries scattered into buckets */
int msm_amd_get_geometry(const msm_amd_ctx *ctx, uint32_t *out6);
/* The call-to-call state that shapes a context's next calls: [0] calls left in the wide layout (recent scalars used
 * bits beyond the group order's width), [1] calls left expecting skew (recent calls queued over-long buckets), [2] slots in the low 16 bits,
 * the mean number of entries per bucket the last call reported (what the next call of the same shape expects: sparse inputs take the
 * reduction with the short chain) in the high 16, [3] entries per
 * lane of the accumulation's work list in the call planned last (0: one lane per bucket, or slices) in the low 16 bits, the observed
 * scalar width the next calls' windows tile (0: none -- the group order's) in the high 16.  A measurement reports them. */
int msm_amd_get_hints(const msm_amd_ctx *ctx, uint32_t *out4);

#ifdef __cplusplus
}
#endif
#endif /* MSM_AMD_H */
