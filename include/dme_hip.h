/*
 * dme_hip.h — C ABI of libdme_hip.so, the MI355X (gfx950) implementation of the
 * Dirichlet-multinomial MLE hot path of maxsklar/BayesPy.
 *
 * The reference has no FFI: its boundary for this path is the Python module
 * DirichletEstimation/dirichletMultinomialEstimation.py ("DE/dme.py") called from
 * DirichletEstimation/findDirichletPrior.py ("DE/fdp.py").  Each entry point below names the
 * reference lines it replaces.  bayespy_amd/ binds these with ctypes.
 *
 * Conventions
 *   - every pointer marked [dev] is a device pointer (hipMalloc / torch tensor data_ptr()),
 *     owned by the caller; the library allocates no device memory and keeps nothing of a call's
 *     DATA.  What it does keep between calls, all of it small and none of it visible in results:
 *       per host thread  - the text of dme_last_error(); the launch structure pinned with
 *                          dme_solve_set_path(); the CU limit of dme_set_cu_limit(); 64 bytes of
 *                          pinned host memory for dme_solve's completion poll (allocated on first
 *                          use, never freed); the launch-count hint of the solver: how many
 *                          launches this thread's previous solve of the same (K, ldU) needed, which
 *                          sizes the first batch of the next one (it changes how many launches are
 *                          enqueued, never what they compute);
 *       per process      - per device: the address of dme_eval_noalloc's static scratch array and
 *                          "attribute set" flags of kernels that need more than 64 KB of LDS;
 *                          the device's CU count;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return value: DME_OK or a DME_ERR_* code; the text is in dme_last_error() (thread-local);
 *   - all launches are asynchronous on `stream` unless the entry says it synchronises;
 *   - nothing throws across the ABI.
 */
#ifndef DME_HIP_H
#define DME_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DME_ABI_VERSION 2  /* 2: overflow store, ragged U/V, bounded compact message */

#define DME_OK 0
#define DME_ERR_INVALID_ARG 1   /* bad shape / pointer / alignment                               */
#define DME_ERR_HIP 2           /* a HIP runtime call failed                                     */
#define DME_ERR_UNSUPPORTED 3   /* shape outside what the kernels are instantiated for           */
#define DME_ERR_WORKSPACE 4     /* workspace too small (see *_workspace_bytes)                   */
#define DME_ERR_ZERO_DIVISION 5 /* the reference would raise ZeroDivisionError (DE/dme.py:69-71) */
#define DME_ERR_MATH_DOMAIN 6   /* the reference would raise ValueError (math.log of x <= 0)     */
#define DME_ERR_INTERNAL 7
#define DME_PARSE_FALLBACK 100  /* dme_parse_counts: text outside the fast grammar, nothing parsed */

int dme_abi_version(void);
const char* dme_last_error(void);

/* Number of compute units of the current device (used by callers to size workspaces). */
int dme_device_cus(void);

/* ---- EXPERIMENTAL: not part of the drop-in path, may change or go ----------------------------
 * Kept for benchmarks/overlap_probe.py only, which ran the solver of one estimation on a few CUs
 * beside the statistics pass of the next; the measurement did not pay (+7 % at
 * one GPU's share of an 8-GPU job, -1.6 % on the bench) and nothing in bayespy_amd/ or bench.py
 * calls these.  No counterpart in the reference.
 *   dme_set_cu_limit: every launch of the calling thread is sized for `cus` compute units from now
 *   on (grids of one workgroup per CU, the solver's share of workgroups); 0 restores the device's own
 *   count.  dme_stream_create_masked: a stream whose kernels run only on the compute units whose
 *   bits are set in `mask` (nwords 32-bit words; bit i = CU i in the HIP runtime's numbering:
 *   hipExtStreamCreateWithCUMask); dme_stream_destroy ends it. */
int dme_set_cu_limit(int cus);
int dme_stream_create_masked(const uint32_t* mask /*[host]*/, int nwords, void** stream /*out*/);
int dme_stream_destroy(void* stream);
/* ---- end of the experimental entries --------------------------------------------------------- */

/* ---------------------------------------------------------------------------------------------
 * Extents for the histograms.  The reference lengthens U[k] / V as rows arrive (DE/dme.py:118, :122);
 * here nbins / tbins are fixed before the statistics pass.  A caller who does not know the largest
 * cell and the largest row total measures them with one streaming read of X:
 *     out[0] = max(out[0], max_ik x_ik),  out[1] = max(out[1], max_i sum_k max(x_ik, 0))
 * (device words, max-accumulated: zero them first; negative cells are ignored here and reported
 * by the statistics pass).
 * ------------------------------------------------------------------------------------------- */
int dme_bounds_dense_u16(const uint16_t* X /*[dev]*/, int64_t N, int64_t K, int64_t ld,
                         uint64_t* out /*[dev][2]*/, void* stream);
int dme_bounds_dense_i32(const int32_t* X /*[dev]*/, int64_t N, int64_t K, int64_t ld,
                         uint64_t* out /*[dev][2]*/, void* stream);
/* the same for CSR rows (largest stored value, largest sum of a row's positive values) */
int dme_bounds_csr_i32(const int64_t* rowptr /*[dev][N+1]*/, const int32_t* vals /*[dev][nnz]*/,
                       int64_t N, int64_t nnz, uint64_t* out /*[dev][2]*/, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Overflow store: where a statistics pass puts the cells whose count does not fit the dense
 * histogram (x_ik = v >= nbins).  The reference's U[k] are lists of their own lengths
 * (DE/dme.py:118), so one outlier count costs it one long list; a dense K x nbins histogram would
 * pay K times that.  With a store, hist keeps a width the caller chooses and every larger count is
 * a 16-byte record:
 *     uint64 head[2] = {used, capacity};  struct {uint32 column, value; uint64 rows;} record[capacity]
 * `used` counts every attempt; records beyond `capacity` are dropped and reported in status[1].
 * A pass appends records with rows = 1 (equal cells repeat); dme_shard_compact adds them into the
 * compact form, dme_shard_expand writes merged ones back.  No counterpart in the reference.
 *   dme_overflow_bytes(capacity)   = 16 + 16 * capacity
 *   dme_overflow_init              head = {0, capacity}
 *   dme_overflow_extents           ext[k] = max(ext[k], value) over the records (after dme_shard_extents)
 *   dme_overflow_append            dst += the records of src (status[1] += rows of those dst cannot hold)
 * ------------------------------------------------------------------------------------------- */
int64_t dme_overflow_bytes(int64_t capacity);
int dme_overflow_init(uint64_t* store /*[dev], 16-byte aligned*/, int64_t capacity, void* stream);
int dme_overflow_extents(const uint64_t* overflow /*[dev]*/, int64_t K, int64_t* ext /*[dev][K]*/,
                         void* stream);
int dme_overflow_append(uint64_t* dst /*[dev]*/, const uint64_t* src /*[dev]*/,
                        uint64_t* status /*[dev][8] or NULL*/, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Statistics build.  Replaces CompressedRowData.appendRow (DE/dme.py:113-123) applied to every
 * row, plus the CLI's mean accumulator (DE/fdp.py:87-89), in one pass over the count matrix.
 *
 * Output is the VALUE HISTOGRAM form of the statistics (all += so that row chunks / shards add):
 *     hist [k*nbins + v] += #{rows i : x_ik == v}            v in [1, nbins)   (bin 0 unused)
 *     histV[t]           += #{rows i : sum_k x_ik == t}      t in [1, tbins)   (bin 0 unused)
 *     init_acc[k]        += sum_i x_ik / (1 + sum_k x_ik)    (fp64; DE/fdp.py:88-89)
 * dme_suffix_scan turns hist/histV into the reference's U/V.  Integer results are exact.
 *
 * status[8] [dev] (uint64, += ; caller zeroes):
 *     [0] rows processed        [1] cells with count >= nbins that no overflow store took (NOT
 *                                   counted: result incomplete)
 *     [2] rows with total >= tbins (NOT counted)              [3] negative cells (i32 only)
 *     [4] max count seen        [5] max row total seen
 *     [6] CSR only: malformed rowptr - negative, decreasing, or spanning more entries than the nnz
 *         passed (the partitioned pass then counts nothing; the atomics pass skips such rows)
 *     [7] internal faults (a bounded in-kernel wait that expired; a compact all-reduce message that
 *         outgrew the capacity its caller promised): the result is incomplete
 * The caller must check status[1..3] == 0 and status[6..7] == 0 (data errors and faults are reported after
 * the fact so that the pass stays asynchronous).
 *
 * X is row-major, `ld` elements between rows (ld >= K).  Rows of 8 uint16 / 4 int32 columns or more
 * are read in 16-byte vectors whatever their pitch, alignment and K (global memory takes unaligned
 * vector loads; a row that ends inside a vector has its last vector loaded shifted): uint16 up to
 * K = 32 768, int32 up to K = 16 384 (from 512 columns on int32 rows run in chunks of 8 columns like
 * uint16 rows; counts of 65536 and more are legal and settled one by one).  Rows that do not fill a
 * wave lie side by side in it, rows of more than 8 vectors per lane are split over teams of 2, 4 or
 * 8 waves.  Narrow rows that lie back to back from a 16-byte-aligned address: uint16 K = 4 as two
 * rows per vector, K = 2 (both element types) as pair counts (see dme_stats.hip).  Beyond the column limits:
 * DME_ERR_UNSUPPORTED (use the CSR entry).
 * overflow [dev] or NULL: an overflow store (above) for the cells with count >= nbins.
 * workspace [dev], 256-byte aligned: dme_stats_workspace_bytes(N, K) bytes.
 * ------------------------------------------------------------------------------------------- */
int64_t dme_stats_workspace_bytes(int64_t N, int64_t K);

int dme_stats_dense_u16(const uint16_t* X /*[dev]*/, int64_t N, int64_t K, int64_t ld,
                        uint64_t* hist /*[dev]*/, int64_t nbins, uint64_t* histV /*[dev]*/,
                        int64_t tbins, double* init_acc /*[dev]*/, uint64_t* status /*[dev]*/,
                        uint64_t* overflow /*[dev] or NULL*/, void* workspace /*[dev]*/,
                        int64_t workspace_bytes, void* stream);

int dme_stats_dense_i32(const int32_t* X /*[dev]*/, int64_t N, int64_t K, int64_t ld,
                        uint64_t* hist /*[dev]*/, int64_t nbins, uint64_t* histV /*[dev]*/,
                        int64_t tbins, double* init_acc /*[dev]*/, uint64_t* status /*[dev]*/,
                        uint64_t* overflow /*[dev] or NULL*/, void* workspace /*[dev]*/,
                        int64_t workspace_bytes, void* stream);

/* Dense rows wider than the entries above take: compress them to CSR on the device and use
 * dme_stats_csr_i32.  dme_dense_row_nnz_* writes counts[i] = the non-zero cells of row i;
 * dme_shard_offsets(counts, N, rowptr) turns them into rowptr[N + 1] (rowptr[N] = nnz, which the caller
 * reads back to size colidx / vals); dme_dense_to_csr_* fills colidx / vals (columns ascending
 * within a row; negative cells are kept and reported by the CSR pass).  No counterpart in the
 * reference, whose appendRow takes rows of any length (DE/dme.py:113-123). */
int dme_dense_row_nnz_u16(const uint16_t* X /*[dev]*/, int64_t N, int64_t K, int64_t ld,
                          int64_t* counts /*[dev][N] out*/, void* stream);
int dme_dense_row_nnz_i32(const int32_t* X /*[dev]*/, int64_t N, int64_t K, int64_t ld,
                          int64_t* counts /*[dev][N] out*/, void* stream);
int dme_dense_to_csr_u16(const uint16_t* X /*[dev]*/, int64_t N, int64_t K, int64_t ld,
                         const int64_t* rowptr /*[dev][N+1]*/, int32_t* colidx /*[dev][nnz] out*/,
                         int32_t* vals /*[dev][nnz] out*/, void* stream);
int dme_dense_to_csr_i32(const int32_t* X /*[dev]*/, int64_t N, int64_t K, int64_t ld,
                         const int64_t* rowptr /*[dev][N+1]*/, int32_t* colidx /*[dev][nnz] out*/,
                         int32_t* vals /*[dev][nnz] out*/, void* stream);

/* CSR rows (rowptr int64[N+1], colidx int32[nnz], vals int32[nnz]); same outputs.  Entries with
 * value 0 are legal and ignored, as a zero count is in DE/dme.py:117.
 * With a workspace of dme_stats_csr_workspace_bytes(N, nnz, K) bytes (256-byte aligned; it holds
 * one 8-byte record per stored entry) the pass partitions the entries by column bucket and counts
 * each bucket in LDS; with workspace == NULL, or for small inputs (the size call returns 0), every
 * entry costs two 64-bit global atomics instead.  init_acc is summed in arrival order here. */
int64_t dme_stats_csr_workspace_bytes(int64_t N, int64_t nnz, int64_t K);
int dme_stats_csr_i32(const int64_t* rowptr /*[dev]*/, const int32_t* colidx /*[dev]*/,
                      const int32_t* vals /*[dev]*/, int64_t N, int64_t nnz, int64_t K,
                      uint64_t* hist, int64_t nbins, uint64_t* histV, int64_t tbins,
                      double* init_acc, uint64_t* status, uint64_t* overflow /*[dev] or NULL*/,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Reverse suffix scan: value histograms -> the reference's compressed statistics
 * (closed form of DE/dme.py:116-123):
 *     U[k*ldU + n] = sum_{v > n} hist[k*nbins + v]   = #{i : x_ik > n},   n in [0, nbins-1)
 *     V[n]         = sum_{t > n} histV[t]            = #{i : sum_k x_ik > n}
 *     lenU[k] = len(U[k]) = largest v with hist[k][v] > 0 (0 if none);  lenV[0] likewise.
 * U/V are written as int64 (exact) and, when Uf/Vf are non-NULL, also as fp64 for the solver.
 * Entries at n >= lenU[k] are written as 0.  ldU >= nbins-1; V/Vf hold tbins-1 entries.
 * ------------------------------------------------------------------------------------------- */
int dme_suffix_scan(const uint64_t* hist /*[dev]*/, int64_t K, int64_t nbins,
                    const uint64_t* histV /*[dev]*/, int64_t tbins, int64_t* U /*[dev]*/,
                    double* Uf /*[dev] or NULL*/, int64_t ldU, int64_t* lenU /*[dev] [K]*/,
                    int64_t* V /*[dev]*/, double* Vf /*[dev] or NULL*/,
                    int64_t* lenV /*[dev] [1]*/, void* stream);

/* RAGGED statistics: every column at its own length, as the reference's lists are (DE/dme.py:118,
 * :122).  The count histograms in compact form (dme_shard_extents [+ dme_overflow_extents] ->
 * dme_ragged_plan -> dme_shard_compact, below) are scanned in place of hist, and U / Uf take the same
 * places: U[k][n] = U[off[k] + n], n in [0, off[k+1] - off[k]).  K = 50 000 columns with one count of
 * 100 000 then cost their own ~1e5 + sum of the other columns' extents, not K x 100 001 cells.
 *   dme_ragged_plan   off[k] = ext[0] + ... + ext[k-1] (as dme_shard_offsets) and slotoff[k] = the same
 *                     over ceil(ext / DME_SEGMENT_TERMS): the solver's slots (dme_ragged_t below);
 *                     both [K + 1].  The caller reads off[K] and slotoff[K] back to size the arrays.
 *   dme_suffix_scan_ragged   compact [off[K]] -> U, Uf [off[K]], lenU[K]; histV -> V, Vf, lenV as
 *                     dme_suffix_scan; slotcol[s] = the column of slot s (with slotoff; both or neither). */
#define DME_SEGMENT_TERMS 256
int dme_ragged_plan(const int64_t* ext /*[dev][K]*/, int64_t K, int64_t* off /*[dev][K+1] out*/,
                    int64_t* slotoff /*[dev][K+1] out*/, void* stream);
int dme_suffix_scan_ragged(const uint64_t* compact /*[dev][off[K]]*/, const int64_t* off /*[dev][K+1]*/,
                           int64_t K, const uint64_t* histV /*[dev]*/, int64_t tbins,
                           int64_t* U /*[dev][off[K]]*/, double* Uf /*[dev][off[K]] or NULL*/,
                           int64_t* lenU /*[dev][K]*/, int64_t* V /*[dev]*/, double* Vf /*[dev] or NULL*/,
                           int64_t* lenV /*[dev][1]*/, const int64_t* slotoff /*[dev][K+1] or NULL*/,
                           int32_t* slotcol /*[dev][slotoff[K]] or NULL*/, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weighted rows: appendRow(row, weight) with weight != 1 (DE/dme.py:119, :123 add `weight`; used
 * with fractional weights by ConjugatePriorTools/dirichletMixtureEstimation.py:129-133).
 *     histW[k*nbins + v] += weights[i] for x_ik == v;  histVW[t] += weights[i] for row total t
 *     maxcol[k] = max(maxcol[k], x_ik), maxtot[0] likewise (len(U[k]) grows whatever the weight)
 * dme_suffix_scan_weighted ADDS the reverse suffix sums into the fp64 U/V of dme_suffix_scan
 * (Uf[k*ldU+n] += sum_{v>n} histW[k][v]) and raises lenU/lenV to maxcol/maxtot.  int32 rows only;
 * status as for the dense entries ([0] rows, [1] cells >= nbins, [2] totals >= tbins, [3] negatives).
 * ------------------------------------------------------------------------------------------- */
int dme_stats_dense_weighted_i32(const int32_t* X /*[dev]*/, const double* weights /*[dev][N]*/,
                                 int64_t N, int64_t K, int64_t ld, double* histW /*[dev]*/,
                                 int64_t nbins, double* histVW /*[dev]*/, int64_t tbins,
                                 int64_t* maxcol /*[dev][K]*/, int64_t* maxtot /*[dev][1]*/,
                                 uint64_t* status /*[dev]*/, void* stream);
/* The same pass with C weights per row (weights[i*C + c]): C histograms from one read of X -
 * the C components of the mixture EM, each of which the reference feeds row by row
 * (ConjugatePriorTools/dirichletMixtureEstimation.py:130-133 `componentCompressedData[c].appendRow(
 * row, cProbs[c])`).  histW is [C][K][nbins], histVW [C][tbins]; maxcol / maxtot / status are
 * shared (they do not depend on the weights).  1 <= C <= 64. */
int dme_stats_dense_weighted_batch_i32(const int32_t* X /*[dev]*/,
                                       const double* weights /*[dev][N][C]*/, int64_t C, int64_t N,
                                       int64_t K, int64_t ld, double* histW /*[dev]*/,
                                       int64_t nbins, double* histVW /*[dev]*/, int64_t tbins,
                                       int64_t* maxcol /*[dev][K]*/, int64_t* maxtot /*[dev][1]*/,
                                       uint64_t* status /*[dev]*/, void* stream);
/* The same pass with a workspace (dme_stats_weighted_workspace_bytes(K, C, nbins, tbins) bytes, 8-byte
 * aligned device memory, contents irrelevant before and after): every workgroup then leaves its bins as
 * one image and a second kernel adds the images in a fixed order, instead of 256 workgroups sending their
 * non-zero bins to the same few thousand addresses with fp64 atomics (1 M rows x 100 columns x 5 weights:
 * 0.37 -> see profiles/r3_weighted_em.json).  Same reference lines as above (DE/dme.py:119,123 with weight
 * != 1; CPT/dirichletMixtureEstimation.py:129-133).  workspace NULL, or smaller than asked for: the pass
 * of dme_stats_dense_weighted_batch_i32. */
int64_t dme_stats_weighted_workspace_bytes(int64_t K, int64_t C, int64_t nbins, int64_t tbins);
int dme_stats_dense_weighted_batch_ws_i32(const int32_t* X /*[dev]*/,
                                          const double* weights /*[dev][N][C]*/, int64_t C, int64_t N,
                                          int64_t K, int64_t ld, double* histW /*[dev]*/,
                                          int64_t nbins, double* histVW /*[dev]*/, int64_t tbins,
                                          int64_t* maxcol /*[dev][K]*/, int64_t* maxtot /*[dev][1]*/,
                                          uint64_t* status /*[dev]*/, void* workspace /*[dev]*/,
                                          int64_t workspace_bytes, void* stream);
int dme_suffix_scan_weighted(const double* histW, int64_t K, int64_t nbins, const double* histVW,
                             int64_t tbins, const int64_t* maxcol, const int64_t* maxtot,
                             double* Uf /*[dev]*/, int64_t ldU, int64_t* lenU /*[dev]*/,
                             double* Vf /*[dev]*/, int64_t capV, int64_t* lenV /*[dev]*/,
                             void* stream);

/* dme_suffix_scan_weighted for all C weight columns of a batch in one launch (the C component statistics
 * an E-step of CPT/dirichletMixtureEstimation.py:128-133 hands to C solves): histW [C][K][nbins], histVW
 * [C][tbins] as dme_stats_dense_weighted_batch_i32 fills them; Uf [C][K][ldU], lenU [C][K], Vf [C][capV],
 * lenV [C]. */
int dme_suffix_scan_weighted_batch(const double* histW, int64_t C, int64_t K, int64_t nbins,
                                   const double* histVW, int64_t tbins, const int64_t* maxcol,
                                   const int64_t* maxtot, double* Uf /*[dev]*/, int64_t ldU,
                                   int64_t* lenU /*[dev]*/, double* Vf /*[dev]*/, int64_t capV,
                                   int64_t* lenV /*[dev]*/, void* stream);

/* DE/fdp.py:104-108: alpha0[k] = init_acc[k] / (sum(init_acc) + 0.01) + 0.01 */
int dme_cli_init_alphas(const double* init_acc /*[dev]*/, int64_t K, double* alpha0 /*[dev]*/,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: rows shard over GPUs and the value histograms of disjoint row sets add exactly
 * (they are sums of the per-row terms of DE/dme.py:116-123; the reference itself is one process).
 * ONE integer SUM all-reduce (RCCL, issued by the caller: torch.distributed) carries the message
 *     msg = hist[nh] | histV[nv] | status[8] | slots[world][2 + K]          (64-bit words)
 * dme_shard_pack writes a rank's message: its histograms, its additive status words (status[4],
 * status[5] - the observed maxima - as 0) and, in its own slot row only, the two maxima and the
 * bit patterns of its K fp64 mean accumulators (DE/fdp.py:87-89); other ranks' slot rows are 0, so
 * the sum hands every rank all slot rows unchanged.  dme_shard_unpack, after the all-reduce, sets
 * status[4], status[5] inside msg to the maxima over ranks and writes init_acc[k] = the ranks'
 * accumulators added in rank order (bit-identical on every rank).  hist / histV / status are then
 * read in place from msg.  dme_shard_message_words = nh + nv + 8 + world*(2 + K).
 * ------------------------------------------------------------------------------------------- */
int64_t dme_shard_message_words(int64_t nh, int64_t nv, int64_t K, int64_t world);
int dme_shard_pack(const uint64_t* hist /*[dev]*/, int64_t nh, const uint64_t* histV /*[dev]*/,
                   int64_t nv, const uint64_t* status /*[dev][8]*/,
                   const double* init_acc /*[dev][K]*/, int64_t K, int64_t rank, int64_t world,
                   uint64_t* msg /*[dev]*/, void* stream);
int dme_shard_unpack(uint64_t* msg /*[dev]*/, int64_t nh, int64_t nv, int64_t K, int64_t world,
                     double* init_acc /*[dev][K] out*/, void* stream);

/* The same message in 32-bit words, for jobs of fewer than 2^32 rows in all (no bin can then exceed
 * 32 bits; SURVEY 8e "uint32 if N < 2^32"): half the bytes on the wire.
 *     msg32 = hist[nh] | histV[nv] as one word per bin | slots[world][8 + K] as (lo, hi) word pairs
 * A 32-bit SUM transports a 64-bit (lo, hi) pair unchanged only while one rank writes it, so a rank's
 * eight status words ride in its own slot row in front of its K mean accumulators; every other row of
 * its message is 0.  The all-reduce is an int32 SUM.  dme_shard_unpack32 widens the bins into hist64 /
 * histV64 (what dme_suffix_scan reads), writes status[8] - the additive words summed over the ranks in
 * 64 bits (exact for any split of the rows), status[4], status[5] as maxima - and init_acc as
 * dme_shard_unpack does.  The caller promises the row bound (every rank the same one) and checks it
 * afterwards: status[0] (the job's rows) <= the bound proves that no bin wrapped.
 * dme_shard_message_words32 = nh + nv + 2 * world * (8 + K). */
int64_t dme_shard_message_words32(int64_t nh, int64_t nv, int64_t K, int64_t world);
int dme_shard_pack32(const uint64_t* hist /*[dev]*/, int64_t nh, const uint64_t* histV /*[dev]*/,
                     int64_t nv, const uint64_t* status /*[dev][8]*/,
                     const double* init_acc /*[dev][K]*/, int64_t K, int64_t rank, int64_t world,
                     uint32_t* msg /*[dev]*/, void* stream);
int dme_shard_unpack32(const uint32_t* msg /*[dev]*/, int64_t nh, int64_t nv, int64_t K,
                       int64_t world, uint64_t* hist64 /*[dev][nh] out*/,
                       uint64_t* histV64 /*[dev][nv] out*/, uint64_t* status /*[dev][8] out*/,
                       double* init_acc /*[dev][K] out*/, void* stream);

/* Compact form of hist (BASELINE configs[3], K = 50 000: nbins is set by the largest count of any
 * column, nearly every column stops far below it; also the form the ragged scan reads).  What travels
 * are the bins 1 .. ext[k] of every column laid end to end, `compact`, passed to dme_shard_pack /
 * _pack32 in place of hist with nh = the compact capacity:
 *   dme_shard_extents  ext[k] = the largest occupied bin of column k (0: none); dme_overflow_extents
 *                      raises it by the overflow records - then a MAX all-reduce of ext over the ranks
 *                      (the caller's);
 *   dme_shard_offsets  off[k] = ext[0] + ... + ext[k-1], off[K] = the compact length;
 *   dme_shard_compact  compact[off[k] + v - 1] = hist[k*nbins + v] (+ the rows of the overflow records
 *                      of (k, v)), v in 1 .. ext[k]; the words from off[K] to `capacity` are zeroed.  A
 *                      caller that sizes `compact` from a promised bound instead of reading off[K] back
 *                      passes that bound as `capacity`: nothing is written beyond it and
 *                      off[K] > capacity raises status[7] (checked after the fact, like the other
 *                      status words);
 *   dme_shard_expand   the way back after the sum: fills hist (every bin; zeros outside 1 .. ext[k]);
 *                      cells beyond the dense bins go to `overflow` (NULL, or a store that is full:
 *                      status[1] counts their rows).
 * No counterpart in the reference (one process). */
int dme_shard_extents(const uint64_t* hist /*[dev][K*nbins]*/, int64_t K, int64_t nbins,
                      int64_t* ext /*[dev][K] out*/, void* stream);
int dme_shard_offsets(const int64_t* ext /*[dev][K]*/, int64_t K, int64_t* off /*[dev][K+1] out*/,
                      void* stream);
int dme_shard_compact(const uint64_t* hist /*[dev][K*nbins]*/, int64_t K, int64_t nbins,
                      const uint64_t* overflow /*[dev] or NULL*/, const int64_t* off /*[dev][K+1]*/,
                      uint64_t* compact /*[dev][capacity] out*/, int64_t capacity,
                      uint64_t* status /*[dev][8] or NULL*/, void* stream);
int dme_shard_expand(const uint64_t* compact /*[dev][capacity]*/, int64_t capacity, int64_t K,
                     int64_t nbins, const int64_t* off /*[dev][K+1]*/,
                     uint64_t* hist /*[dev][K*nbins] out*/, uint64_t* overflow /*[dev] or NULL*/,
                     uint64_t* status /*[dev][8] or NULL*/, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Solver.  U is dense-padded fp64: U[k][n] = Uf[k*ldU + n] for n < lenU[k]; V = Vf[0..lenV).
 * lenU [K] and lenV [1] are device arrays (as written by dme_suffix_scan).
 * workspace: dme_solver_workspace_bytes(K, ldU, capV) bytes [dev], 256-byte aligned, where
 * capV >= lenV is the allocated length of Vf.
 * ------------------------------------------------------------------------------------------- */
int64_t dme_solver_workspace_bytes(int64_t K, int64_t ldU, int64_t capV);

/* One evaluation at `alpha` [dev][K]: backs dirichLogProb (DE/dme.py:16-29), priorGradient
 * (:32-48), priorHessianConst (:52-53) and priorHessianDiag (:55-61) one-to-one.
 *     out_scalars[0] = log-probability (:29)      out_scalars[1] = hConst (:53)
 *     out_scalars[2] = sum(alpha)                  out_scalars[3] = sum_k grad_k^2 (:104)
 *     out_grad[K], out_hdiag[K]                    (either may be NULL)
 * Asynchronous. */
int dme_eval(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf, int64_t capV,
             const int64_t* lenV, int64_t K, const double* alpha, double hyperprior,
             double* out_scalars /*[dev][4]*/, double* out_grad /*[dev][K]*/,
             double* out_hdiag /*[dev][K]*/, void* workspace, int64_t workspace_bytes,
             void* stream);

/* dme_eval for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  NO counterpart in the reference: it exists for
 * those callers only.  Same arguments, kernels and results (bit for bit) as dme_eval with the
 * scalar hyperprior, minus the workspace: the solver workspace is a static device array inside the
 * library, DME_NOALLOC_SCRATCH_BYTES per GPU, held until the process ends.  Shapes whose
 * dme_solver_workspace_bytes(K, ldU, capV) exceeds that are rejected with DME_ERR_WORKSPACE (use
 * dme_eval).  Calls no hipMalloc / hipFree / allocating routine; the first call on a device looks
 * up the array's address there (hipGetSymbolAddress, which may load the library's code object and is
 * not a stream operation): make ONE eager call per device - any small shape will do - before calling
 * under HIP graph capture; later calls read the cached address with one atomic load.  All calls on one
 * GPU share the array: calls that overlap on one GPU (other streams or threads) race - serialise them.
 * Asynchronous. */
#define DME_NOALLOC_SCRATCH_BYTES (8LL << 20)
int dme_eval_noalloc(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf,
                     int64_t capV, const int64_t* lenV, int64_t K, const double* alpha,
                     double hyperprior, double* out_scalars /*[dev][4]*/,
                     double* out_grad /*[dev][K]*/, double* out_hdiag /*[dev][K]*/, void* stream);

/* getPredictedStep (DE/dme.py:67-72) and getPredictedStepAlt (:76-80) on device vectors.
 * status [dev][1] int64: 0 ok, DME_ERR_ZERO_DIVISION if the reference would divide by zero. */
int dme_predicted_step(double hconst, const double* hdiag /*[dev]*/, const double* grad /*[dev]*/,
                       int64_t K, double* step /*[dev]*/, int64_t* status /*[dev]*/, void* stream);
int dme_predicted_step_alt(double hconst, const double* hdiag, const double* grad,
                           const double* alphas, int64_t K, double* step, int64_t* status,
                           void* stream);

/* findDirichletPriors (DE/dme.py:126-190): the whole Newton loop on the device — loss, gradient,
 * Hessian, Sherman-Morrison step, strict-< acceptance (:160), log-space trial (:165-172), x0.9
 * step halving (:175-180), exits A/B/C (:145-147, :182-184, :189-190).
 *   alpha [dev][K]: in = initAlphas, out = returned priors.
 *   trace [dev][trace_cap][4] or NULL: per iteration {currentLoss, gradientSize, branch, learnRate}
 *       as logged at DE/dme.py:143 (branch: how the PREVIOUS iterate was reached: 0 init,
 *       1 Newton, 2 log-space, 3 gradient/halving).
 *   trace_alpha [dev][trace_cap][K] or NULL: the priors logged at :143.
 *   result [dev][8] int64: [0] exit reason (1 = small gradient, 2 = small learn rate,
 *       3 = max iterations), [1] iterations, [2] loss evaluations, [3] error
 *       (0 / DME_ERR_ZERO_DIVISION / DME_ERR_MATH_DOMAIN), [4] evaluation launches used,
 *       [5] the bits of final_loss (double), [6] done flag, [7] "parity-fragile" count: accept /
 *       reject decisions (DE/dme.py:160, :176) taken on |trial loss - current loss| <= ~4 ulp, i.e.
 *       on rounding noise that the reference's sequential sums may resolve the other way.
 *   result_host [host][8] or NULL: a copy of `result` in host memory, valid on return (saves the
 *       caller a device-to-host copy).
 *   final_loss [dev][1] or NULL: getTotalLoss at the returned priors.
 * The host only re-enqueues batches of (evaluate, step) launches and polls a done flag once per
 * batch; every decision is taken on the device.  SYNCHRONISES `stream` before returning.
 * Returns DME_ERR_ZERO_DIVISION / DME_ERR_MATH_DOMAIN where the reference raises. */
int dme_solve(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf, int64_t capV,
              const int64_t* lenV, int64_t K, double* alpha, int64_t iterations,
              double hyperprior, double* trace, double* trace_alpha, int64_t trace_cap,
              int64_t* result, int64_t* result_host, double* final_loss, void* workspace,
              int64_t workspace_bytes, void* stream);

/* dme_solve without the wait, for callers that keep the GPU fed (the next job's statistics pass is
 * enqueued while this solve still runs; the reference, one thread, has no counterpart).
 * dme_solve_enqueue puts launches [first, first + count) of the solve on `stream` - launch 0 is
 * preceded by the set-up - followed by a copy of result[16] to result_host (pinned host memory,
 * valid once the stream has reached it), and returns at once.  count == 0 asks for the default batch
 * (what this thread's previous solve of the same shape needed plus one spare launch; a launch that
 * finds the solve finished ends at once); *enqueued returns the number taken.  The caller later
 * synchronises the stream (or an event), reads result_host[6] (done) and, if it is 0, calls again with
 * first = the launches enqueued so far.  Once done, dme_solve_finish(result_host, K, ldU) maps
 * result_host[3] to the return code of dme_solve (DME_ERR_ZERO_DIVISION, ...) and records the launch
 * count for the next default batch.
 *   init_acc [dev][K] or NULL: start from the CLI's initial alphas computed from the mean
 *       accumulators (DE/fdp.py:104-108, the bits of dme_cli_init_alphas) instead of reading them
 *       from `alpha`, which is then output only;
 *   status [dev][8] or NULL: the statistics pass's status words, copied into result[8..16) so that
 *       the one host copy also brings the data-error counters;
 *   result [dev][16], result_host [host][16]: words 0..7 as for dme_solve. */
int dme_solve_enqueue(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf,
                      int64_t capV, const int64_t* lenV, int64_t K, double* alpha,
                      int64_t iterations, double hyperprior, const double* init_acc,
                      const uint64_t* status, double* trace, double* trace_alpha, int64_t trace_cap,
                      int64_t* result /*[dev][16]*/, int64_t* result_host /*[host][16]*/,
                      void* workspace, int64_t workspace_bytes, int64_t first, int64_t count,
                      int64_t* enqueued /*[host][1] out*/, void* stream);
int dme_solve_finish(const int64_t* result_host /*[host][16]*/, int64_t K, int64_t ldU);

/* The solver on RAGGED statistics (dme_suffix_scan_ragged): the same kernels, device functions and
 * order of operations as the dense-padded entries - a column's terms, its slots in the partial-sum
 * table and the memory both take follow its own length.  `ragged` is a HOST struct of device pointers.
 * beta == NULL and has_W == 0 select the scalar hyperprior (dme_eval / dme_solve), otherwise the
 * (Beta, W) one (dme_eval_cpt / dme_solve_cpt; hyperprior must then be 0).  Workspace:
 * dme_solver_workspace_bytes_ragged(K, ragged->nslots, capV).  dme_solve_finish takes ragged->nslots in
 * place of ldU. */
typedef struct dme_ragged {
    const int64_t* offU;    /* [dev][K+1]  U[k][n] = Uf[offU[k] + n], n < offU[k+1] - offU[k]          */
    const int64_t* slotoff; /* [dev][K+1]  column k owns the slots [slotoff[k], slotoff[k+1]) of
                                           DME_SEGMENT_TERMS terms each                                 */
    const int32_t* slotcol; /* [dev][nslots] the column of every slot                                   */
    int64_t nslots;         /* slotoff[K]                                                               */
    int64_t total;          /* offU[K]: the length of Uf                                                */
} dme_ragged_t;
int64_t dme_solver_workspace_bytes_ragged(int64_t K, int64_t nslots, int64_t capV);
int dme_eval_ragged(const double* Uf, const dme_ragged_t* ragged /*[host]*/, const int64_t* lenU,
                    const double* Vf, int64_t capV, const int64_t* lenV, int64_t K, const double* alpha,
                    double hyperprior, const double* beta /*[dev][K] or NULL*/, double W, int has_W,
                    double* out_scalars /*[dev][4]*/, double* out_grad, double* out_hdiag,
                    void* workspace, int64_t workspace_bytes, void* stream);
int dme_solve_ragged(const double* Uf, const dme_ragged_t* ragged /*[host]*/, const int64_t* lenU,
                     const double* Vf, int64_t capV, const int64_t* lenV, int64_t K, double* alpha,
                     int64_t iterations, double hyperprior, const double* beta /*[dev][K] or NULL*/,
                     double W, int has_W, double* trace, double* trace_alpha, int64_t trace_cap,
                     int64_t* result, int64_t* result_host, double* final_loss, void* workspace,
                     int64_t workspace_bytes, void* stream);
int dme_solve_enqueue_ragged(const double* Uf, const dme_ragged_t* ragged /*[host]*/,
                             const int64_t* lenU, const double* Vf, int64_t capV, const int64_t* lenV,
                             int64_t K, double* alpha, int64_t iterations, double hyperprior,
                             const double* init_acc, const uint64_t* status, double* trace,
                             double* trace_alpha, int64_t trace_cap, int64_t* result /*[dev][16]*/,
                             int64_t* result_host /*[host][16]*/, void* workspace,
                             int64_t workspace_bytes, int64_t first, int64_t count,
                             int64_t* enqueued /*[host][1] out*/, void* stream);

/* Launch structure of dme_solve / dme_solve_cpt, for the calling host thread.  The three structures
 * run the same device functions in the same order and return the same bits; the choice is by size:
 *   SINGLE  at most 64 segments of 256 terms (the reference's sample files, K = 2): the whole loop
 *           in one launch of one workgroup;
 *   FUSED   K <= 2048: one launch per evaluation (every workgroup repeats the O(K) step on an LDS
 *           copy of the state, then the workgroups share the evaluation of the new point);
 *   CHAIN   otherwise: an (evaluate, step) launch pair per evaluation.
 * DME_SOLVE_PATH_AUTO (the default) picks by size; the others pin a structure (FUSED falls back to
 * CHAIN beyond K = 2048) - for tests and timing comparisons.  No reference counterpart. */
#define DME_SOLVE_PATH_AUTO 0
#define DME_SOLVE_PATH_CHAIN 1
#define DME_SOLVE_PATH_SINGLE 2
#define DME_SOLVE_PATH_FUSED 3
int dme_solve_set_path(int path);

/* The same two entries with the (Beta, W) conjugate hyperprior of
 * ConjugatePriorTools/dirichletMultinomialEstimation.py ("CPT/dme.py") instead of the scalar one:
 *   beta [dev][K] or NULL, W used when has_W != 0.
 * dme_eval_cpt backs CPT's dirichLogProb (:20-39, prior terms included), priorGradient (:42-66),
 * priorHessianConst (:70-76), priorHessianDiag (:78-87).  dme_solve_cpt is CPT's
 * findDirichletPriors (:156-219); as there, the loss it compares drops Beta and W
 * (getTotalLoss, :109-110) while gradient and Hessian carry them.  digamma / trigamma are
 * evaluated on the device (recurrence + asymptotic series, ~1e-16). */
int dme_eval_cpt(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf,
                 int64_t capV, const int64_t* lenV, int64_t K, const double* alpha,
                 const double* beta /*[dev][K] or NULL*/, double W, int has_W,
                 double* out_scalars /*[dev][4]*/, double* out_grad, double* out_hdiag,
                 void* workspace, int64_t workspace_bytes, void* stream);
int dme_solve_cpt(const double* Uf, int64_t ldU, const int64_t* lenU, const double* Vf,
                  int64_t capV, const int64_t* lenV, int64_t K, double* alpha, int64_t iterations,
                  const double* beta /*[dev][K] or NULL*/, double W, int has_W, double* trace,
                  double* trace_alpha, int64_t trace_cap, int64_t* result, int64_t* result_host,
                  double* final_loss, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Text ingestion for the CLI (host code, no GPU): replaces the csv.reader + int() row loop of
 * DE/fdp.py:67-79 when every line holds exactly K integers ([ ]*[+-]?[0-9]+[ ]*) separated by one
 * `delimiter` byte.  dme_parse_count_rows returns the number of lines (to size `out`);
 * dme_parse_counts fills out[row*K + k] (host memory, int32) with `nthreads` threads.  Text the
 * strict grammar does not cover returns DME_PARSE_FALLBACK: the caller then uses the reference's own
 * loop, which keeps the reference's error behaviour (ValueError, ragged-row logging, ...).
 * ------------------------------------------------------------------------------------------- */
int64_t dme_parse_count_rows(const char* text /*[host]*/, int64_t nbytes);
int dme_parse_counts(const char* text /*[host]*/, int64_t nbytes, int delimiter, int64_t K,
                     int32_t* out /*[host]*/, int64_t out_rows, int64_t* n_rows /*[host]*/,
                     int nthreads);
/* The CLI's -s (DE/fdp.py:78: a row is kept when random.random() < sampleRate, and only a kept row's
 * fields are converted, :79) and -M (DE/fdp.py:85: `while sum(counts) > M: counts[drawCategory(counts)]
 * -= 1`, DE/samplingTools.py:49-58) on top of the fast parser, with the reference's draws in the
 * reference's order - the caller supplies them from Python's own `random` stream (bulk draws of the same
 * MT19937 state), so a seeded run keeps and thins the very rows the reference does.
 *   dme_parse_counts_selected  parses only the lines with keep[i] != 0 (n_keep = the text's line count),
 *                      written back to back; the other lines are skipped unparsed, as the reference
 *                      never looks at them either.  For -s without -M: the draws are one per line.
 *   dme_cap_draws_needed  n + sum_i max(0, sum(row_i) - M): the draws rows X could take at most.
 *   dme_sample_and_cap_i32  rows X [n][K] in place: row i takes u[p++] (kept iff < sample_rate) and, if
 *                      kept, one draw per decrement while its total exceeds M; kept rows move to the
 *                      front.  *n_kept rows remain, *n_used draws were taken (the caller advances its
 *                      stream by that many). */
int dme_parse_counts_selected(const char* text /*[host]*/, int64_t nbytes, int delimiter, int64_t K,
                              const uint8_t* keep /*[host][n_keep]*/, int64_t n_keep,
                              int32_t* out /*[host]*/, int64_t out_rows, int64_t* n_rows /*[host]*/,
                              int nthreads);
int64_t dme_cap_draws_needed(const int32_t* X /*[host][n][K]*/, int64_t n, int64_t K, int64_t M);
int dme_sample_and_cap_i32(int32_t* X /*[host][n][K]*/, int64_t n, int64_t K, double sample_rate,
                           int64_t M, const double* u /*[host][n_u]*/, int64_t n_u,
                           int64_t* n_kept /*[host]*/, int64_t* n_used /*[host]*/);

/* ---------------------------------------------------------------------------------------------
 * Synthetic count rows for benchmarks and tests (no reference counterpart: the reference's
 * generators are host-side samplers, DE/samplingTools.py:41-67).  Row i draws
 * x_i ~ DirichletMultinomial(totals[i], alpha) by the Polya-urn scheme (draw j copies a uniformly
 * chosen earlier draw with probability (j)/(A + j), else draws from alpha/A), counter-based RNG
 * keyed on (seed, row_offset + i).  cdf [dev][K] = inclusive prefix sums of alpha (fp64).
 * scratch: dme_synth_scratch_bytes(max_total) bytes [dev].  elem_bytes: 2 (uint16) or 4 (int32).
 * ------------------------------------------------------------------------------------------- */
int64_t dme_synth_scratch_bytes(int64_t max_total);
int dme_synth_dirmult(void* X /*[dev]*/, int elem_bytes, int64_t N, int64_t K, int64_t ld,
                      const int32_t* totals /*[dev][N]*/, int64_t max_total,
                      const double* cdf /*[dev][K]*/, uint64_t seed, int64_t row_offset,
                      void* scratch, int64_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DME_HIP_H */
