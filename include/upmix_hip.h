/*
 * upmix_hip.h - C ABI of libupmix_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the multi-band STFT centre-extraction hot path of
 * willleskowitz/upmix.  The reference has no FFI; its boundary for this path
 * is the Python call
 *     extract_center_left_right_multi_band_in_memory(L, R, sr, band_extractors)
 *         -> (center, left, right)          python-prototype/center_extraction.py:477-513
 * which fans out MultiBandExtractorAccu.process_all_blocks(L, R) per band
 * (center_extraction.py:426-472, hot loop :449-460 -> process_stereo_chunk :353-409)
 * and sums the bands in float32 (:503-511).  Everything that function computes
 * from (L, R, band parameters) is what upx_process() computes.  Band planning,
 * window design and the band-limit gain vector stay on the host and cross this
 * ABI as DATA (float arrays), exactly as the reference precomputes them in
 * MultiBandExtractorAccu.__init__ (:240-271) and _band_limit (:334-351).
 *
 * Plain C: pointers + sizes, no C++/torch/numpy types.  All functions return 0
 * on success or a negative upx_status; upx_last_error() gives the message of
 * the last failure on the calling thread.  A plan is not thread-safe; distinct
 * plans are.
 */
#ifndef UPMIX_HIP_H
#define UPMIX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum upx_status {
    UPX_OK = 0,
    UPX_ERR_INVALID = -1,     /* bad argument (maps to ValueError on the Python side) */
    UPX_ERR_UNSUPPORTED = -2, /* STFT size / overlap not covered by the kernels */
    UPX_ERR_HIP = -3,         /* HIP runtime failure */
    UPX_ERR_NO_DEVICE = -4,   /* no usable GPU */
    UPX_ERR_RCCL = -5,        /* RCCL missing or failed */
    UPX_ERR_NOMEM = -6
} upx_status;

typedef struct upx_plan upx_plan;
typedef struct upx_comm upx_comm;

/* ABI version of this header (bumped on incompatible change). */
int upx_abi_version(void);

/* Message of the last error on this thread ("" if none). */
const char* upx_last_error(void);

/* Number of visible HIP devices. */
int upx_device_count(int* count);

/* 1 if (block_size, hop) is covered by the gfx950 kernels, else 0: power-of-two sizes 64..65536, any hop in
   [1, N] with at most 64 frames overlapping one sample.  hop = N/2, N/4, N/8: band-limited bands (pass band below
   bin N/16; what the reference's planner gives every large STFT) take the two-kernel band-limited path at any size,
   other bands with N in 256..8192 the fused streaming kernel; everything else the unfused pipeline. */
int upx_supported(int32_t block_size, int32_t hop);

/*
 * Create a plan for n_bands bands on `device`.
 *   block_size[b], hop[b]      STFT size N_b and hop (center_extraction.py:250-252)
 *   w_analysis  = concat_b analysis_window[N_b]      float32 (:257)
 *   w_synthesis = concat_b synthesis_window[N_b]     float32 (:258, design :80-105)
 *   gain        = concat_b g_b[N_b/2+1]              float32: the real per-bin factor
 *                 _band_limit applies to both spectra (:334-351)
 * Arrays are copied; nothing is retained.  Replaces the per-band state built by
 * chain_bands (:518-580) / MultiBandExtractorAccu.__init__ (:240-271).
 */
int upx_plan_create(upx_plan** out, int device, int n_bands, const int32_t* block_size, const int32_t* hop,
                    const float* w_analysis, const float* w_synthesis, const float* gain);

/*
 * What upx_plan_create would SELECT for these bands, without a device: for every band the kernel(s) of the launch that
 * carries it, one line per band in `names` ("analysis|synthesis" for a band-limited group; merged bands repeat their
 * group's line).  Same arguments as upx_plan_create; n = size of `names` (UPX_ERR_INVALID when too small).
 *
 * Environment: the library reads its UPX_* tuning / test knobs (launch geometry, kernel family, chunk lengths - DESIGN.md,
 * "Knobs") at plan creation ONLY in a process that opts in with UPX_TUNING=1; without it no variable changes which kernels
 * run, how a signal is cut, or a single bit of the result.  The band sum is always the reference's ((0 + b0) + b1) + ...
 * (center_extraction.py:508-511); the experiments that once reordered it exist in -DUPX_EXPERIMENTS builds only.
 */
int upx_plan_kernel_names(int n_bands, const int32_t* block_size, const int32_t* hop, const float* w_analysis,
                          const float* w_synthesis, const float* gain, char* names, size_t n);
void upx_plan_destroy(upx_plan* plan);

/* Tuning: hop-blocks each stream walks (0 = automatic).  band = -1 sets all bands. */
int upx_plan_set_blocks_per_stream(upx_plan* plan, int band, int blocks);

/*
 * Whole-signal call on HOST buffers (H2D, kernels, D2H; blocking).
 *   stereo  interleaved float32 [T][2] (L, R)
 *   out_c/out_l/out_r  float32 [T] each: centre, left-side, right-side
 * Replaces extract_center_left_right_multi_band_in_memory (:477-513); same
 * (center, left, right) order.  Any length: a launch covers < 2^29 samples (byte offsets are 32-bit inside
 * the kernels), longer signals - and by default every long signal - go through upx_process_chunked().
 */
int upx_process(upx_plan* plan, const float* stereo, int64_t n_samples, float* out_c, float* out_l, float* out_r);
/*
 * The same, streamed: the signal is cut into chunks of about `chunk` samples (rounded to the bands' common frame
 * grid); the upload of chunk i+1, the kernels of chunk i and the download of chunk i-1 overlap, and each chunk's
 * overlap-add tail is added onto the next chunk on the device.  Device memory is O(chunk), so signals larger
 * than HBM or longer than 2^29 samples work.  upx_process() calls this for long signals (from 2 chunks of 2^22
 * samples; UPX_STREAM_CHUNK overrides the chunk length, 0 disables).  Differs from the one-shot result only by the
 * float32 association of the overlap-add at the chunk seams (same as a multi-GPU run).
 */
int upx_process_chunked(upx_plan* plan, const float* stereo, int64_t n_samples, float* out_c, float* out_l,
                        float* out_r, int64_t chunk);

/*
 * The same call on the arrays the reference's caller holds (main.py:49-50, 78-80 hand
 * extract_center_left_right_multi_band_in_memory two float64 COLUMN VIEWS of one [T][2] array; center_extraction.py:477-482
 * takes any real arrays): `left` / `right` point at sample 0 of either channel, `stride` is the distance between
 * consecutive samples of a channel in elements:
 *     stride 2 and right == left + one element   the two columns of one C-contiguous interleaved [T][2] array
 *     stride 1                                   two contiguous arrays
 * (anything else: UPX_ERR_UNSUPPORTED - gather on the host).  sample_format: UPX_SAMPLE_F32 | UPX_SAMPLE_F64.  The samples
 * go over the link as they are, chunk by chunk like upx_process, and are cast to float32 and interleaved ON THE DEVICE:
 * the same round-to-nearest cast the host would make, so the result is bit-identical to upx_process on
 * float32(interleave(left, right)) - without two strided host passes over the signal before the first byte moves.
 */
enum { UPX_SAMPLE_F32 = 0, UPX_SAMPLE_F64 = 1 };
int upx_process_lr(upx_plan* plan, const void* left, const void* right, int sample_format, int64_t stride,
                   int64_t n_samples, float* out_c, float* out_l, float* out_r);

/*
 * A batch of independent tracks through ONE plan (BASELINE configs[4]; the reference's analogue is running
 * main.py:36-80 once per file): track t is stereo[t] ([n_samples[t]][2] float32) -> out_c[t] / out_l[t] / out_r[t]
 * (float32 [n_samples[t]]).  The tracks (cut into chunks exactly as upx_process cuts a long signal) form one queue
 * of work items; the upload of item i+1, the kernels of item i and the download of item i-1 overlap across track
 * boundaries.  Each track's result is bit-identical to a upx_process call on that track alone.  Zero-length tracks
 * are skipped.
 */
int upx_process_tracks(upx_plan* plan, int32_t n_tracks, const float* const* stereo, const int64_t* n_samples,
                       float* const* out_c, float* const* out_l, float* const* out_r);

/* ---- device-resident interface (benchmarks, sharding, pipelines) -------- */
int upx_dev_alloc(upx_plan* plan, void** ptr, size_t bytes);
int upx_dev_free(upx_plan* plan, void* ptr);
int upx_dev_memset(upx_plan* plan, void* ptr, int value, size_t bytes);
/* Page-locked host memory for the buffers a caller hands to upx_process / upx_process_tracks / upx_wav_*: copies to
   and from it run at link speed, without the runtime's staging copy and without page faults of fresh pages (a call
   into fresh pageable arrays spends most of its time there).  upmix_amd keeps a pool of such blocks and returns its
   result arrays in them (the reference returns fresh NumPy arrays, center_extraction.py:503-513). */
int upx_host_alloc(upx_plan* plan, void** ptr, size_t bytes);
int upx_host_free(upx_plan* plan, void* ptr);
int upx_copy_h2d(upx_plan* plan, void* dst_dev, const void* src_host, size_t bytes);
int upx_copy_d2h(upx_plan* plan, void* dst_host, const void* src_dev, size_t bytes);
int upx_sync(upx_plan* plan);

/*
 * Run all bands on device buffers (asynchronous on the plan's stream).
 *   d_stereo  [t_in][2]  valid input samples from local sample 0 (zero beyond)
 *   own_len   samples this call owns: frames j with j*hop_b < own_len are computed
 *   t_out     length of the output planes; hop-blocks up to
 *             min(t_out, own_len + N_b - hop_b) receive band b (the part beyond
 *             own_len is the overlap-add spill used for multi-GPU seams)
 * For a whole signal: t_in = own_len = t_out = T.
 */
int upx_process_device(upx_plan* plan, const float* d_stereo, int64_t t_in, int64_t own_len, float* d_c,
                       float* d_l, float* d_r, int64_t t_out);

/*
 * Prepares the plan for upx_process_device calls of this shape: launch geometry, stream tables (uploaded), seam and
 * scratch buffers - everything the FIRST call of a shape otherwise allocates, uploads and synchronises for on the way
 * (a few tenths of a millisecond: the software part of a cold call; the rest of a cold call is the card's clock ramp).
 * Touches no buffer of the caller's.  The FIRST reserve of a plan also runs one tiny warm-up call through every kernel of
 * the plan (a few frames of silence in a temporary device allocation it frees again), so that the runtime's per-kernel set-up
 * is not paid by the caller's first step either; every reserve then walks the geometry of the shape asked for WITHOUT
 * launching.  Neither leaves a trace in the timing rings or in the band / launch reports (upx_plan_band_times_*,
 * upx_plan_band_info, upx_plan_band_fill keep describing the caller's own calls).  A later upx_process_device(plan, ., t_in,
 * own_len, ., ., ., t_out) then only enqueues kernels.  Blocking.  The reference has no counterpart (its state is built in
 * MultiBandExtractorAccu.__init__, center_extraction.py:240-271, which upx_plan_create mirrors).
 */
int upx_plan_reserve(upx_plan* plan, int64_t t_in, int64_t own_len, int64_t t_out);

/* Per-band kernel timing with HIP events on the plan's stream.  enable: 0 off, 1 on (forgets the calls recorded so far),
 * 2 resume / 3 pause without forgetting: the events cost ~2 us each (24 us of a 1.45 ms call with ten of them), so a
 * measuring loop may time every n-th call only. */
int upx_plan_enable_timing(upx_plan* plan, int enable);
/* Milliseconds of each band's kernel in the last upx_process_device call (syncs). */
int upx_plan_band_times_ms(upx_plan* plan, float* ms, int n_bands);
/* Sum over the last n_calls (<= 64) timed upx_process_device calls, with ONE synchronisation: a timed loop can
   run back to back and read its kernel times afterwards. */
int upx_plan_band_times_sum_ms(upx_plan* plan, float* ms, int n_bands, int n_calls);
/* The same per call: ms[c * n_bands + b] for the last n_calls (<= 64) timed calls, oldest first (medians). */
int upx_plan_band_times_calls_ms(upx_plan* plan, float* ms, int n_bands, int n_calls);
/* Bands on the band-limited two-kernel path (upx_zoom.h): time of the analysis launches and of the synthesis
   launches (+ the stream seam add), summed over the last n_calls timed calls.  Single-kernel bands report 0 and
   their whole time.  upx_plan_band_phase_kernel_name: phase 0 = analysis ("" for single-kernel bands), 1 = synthesis
   or the band's only kernel. */
int upx_plan_band_phase_times_sum_ms(upx_plan* plan, float* ms_analysis, float* ms_synthesis, int n_bands, int n_calls);
int upx_plan_band_phase_kernel_name(upx_plan* plan, int band, int phase, char* name, size_t n);
/* How the last upx_process_device call cut `band`'s launch group into streams: the first FRAME of every stream (frame j
   starts at sample j * hop; the first stream starts at frame -1, the pair partner of frame 0) and, last, the end frame.
   Band-limited groups: the table of the Ls/Rs streams (+ end), then the table of the centre streams (+ end).  Unfused
   groups: the first emitted block of every chunk.  n_out = entries available; at most `cap` are written.  For tests
   that aim oracle windows at stream seams, and for reports. */
int upx_plan_band_stream_starts(upx_plan* plan, int band, int32_t* starts, int32_t cap, int32_t* n_out);

/* Kernel symbol name / launch geometry of a band (for profiles and DESIGN.md). */
int upx_plan_band_info(upx_plan* plan, int band, int32_t* workgroups, int32_t* threads, int32_t* lds_bytes,
                       int32_t* blocks_per_stream);
/* How much of the chip the last upx_process_device call's launch of `band` (its merged group) filled: workgroups of the
   main kernel (fused kernel, or the band-limited synthesis: first launch pair) and the workgroup slots the chip holds of
   that kernel at once (CUs x resident workgroups per CU by registers and LDS); the same for the band-limited analysis
   (0 / 0 for single-kernel bands).  workgroups < slots: the launch leaves CUs idle (short signals, bench.py c1 / c2). */
int upx_plan_band_fill(upx_plan* plan, int band, int32_t* workgroups, int32_t* slots, int32_t* workgroups_analysis,
                       int32_t* slots_analysis);
/* Name of the kernel that carries `band`, as rocprofv3 prints it (NUL-terminated, truncated to n bytes):
   "upx_band_kernel<upx::WideCfg<13, 4>, 2>", "upx_band_kernel<upx::Cfg<10, 4, 16>, 2>",
   "upx_zoom_synthesis_kernel<upx::ZoomCfg<8, 16, 4>>" for the band-limited two-kernel path (its analysis kernel:
   upx_plan_band_phase_kernel_name), or "unfused<N>" for the multi-kernel path (upx_big_* kernels). */
int upx_plan_band_kernel_name(upx_plan* plan, int band, char* name, size_t n);

/*
 * Adjacent bands with the same STFT size, hop and windows are merged into ONE kernel launch (their
 * transforms are identical linear operators; only the per-bin gain -> mask step is per band).
 * Reports the first band of `band`'s group and the number of bands in it.  Timing and launch geometry
 * of a group are reported on its first band.  Set UPX_NO_BAND_MERGE=1 to launch every band separately.
 */
int upx_plan_band_group(upx_plan* plan, int band, int32_t* leader, int32_t* size);

/* max|x| over n floats on the device (peak normalisation of main.py:85-88). */
int upx_absmax(upx_plan* plan, const float* d_x, int64_t n, float* result);
/* x *= scale on the device (main.py:95-97). */
int upx_scale(upx_plan* plan, float* d_x, int64_t n, float scale);

/*
 * WAV in -> WAV out with the codec next to the kernels: the flow of main.py:43-160 for one file.
 * Raw PCM goes up (half the bytes of float32), is decoded on the device, upmixed, peak-normalised
 * (one global scale = peak_in / max(|Ls|,|C|,|Rs|, 1e-9), main.py:85-97) and laid out per export mode
 * (main.py:110-157) and quantised on the device, so only final 2-channel sample data comes back.
 *   pcm_in      interleaved samples, `channels` = 1 (duplicated to L = R, main.py:47-48) or 2
 *   in_format / out_format   UPX_PCM16 | UPX_PCM24 (packed 3 bytes) | UPX_PCM32 | UPX_F32
 *   mode        UPX_EXPORT_STEREO_SUM: out0 = [Ls + C/2, Rs + C/2]
 *               UPX_EXPORT_SPLIT:      out0 = [Ls, 0], out1 = [C, C], out2 = [0, Rs]
 *               UPX_EXPORT_AB:         out0 = [Ls + C + Rs, L + R]
 *   out0..2     caller buffers of n_frames * 2 samples in out_format (unused ones may be NULL)
 *   stats[3]    peak_in, overall_peak, scale_factor (the numbers main.py prints)
 * Decoding follows python-soundfile (int / 2^(bits-1)); encoding is round-to-nearest-even of
 * x * (2^(bits-1) - 1), clipped (libsndfile's byte-exact behaviour is not pinned).
 */
enum { UPX_PCM16 = 16, UPX_PCM24 = 24, UPX_PCM32 = 32, UPX_F32 = 1032 };
enum { UPX_EXPORT_STEREO_SUM = 0, UPX_EXPORT_SPLIT = 1, UPX_EXPORT_AB = 2 };
int upx_wav_pipeline(upx_plan* plan, const void* pcm_in, int in_format, int channels, int64_t n_frames, int mode,
                     int out_format, void* out0, void* out1, void* out2, double* stats);
/*
 * The same flow for ONE TIME SHARD of a file (one process per GPU; main.py:43-157 on a slice): the two
 * scalars of main.py:53-55 / :85-88 are global, so the call is split where they cross the ranks.
 *   begin   raw samples of the shard (t_in frames from its first owned one: own range + right halo) go up in chunks of
 *           UPX_WAV_CHUNK owned frames (2^22; on the shard grid); chunk c is decoded and runs through all bands while
 *           chunk c + 1 comes up (chunk seams as in upx_process_chunked: <= 1e-7 on the K-1 blocks behind a seam); the
 *           RCCL overlap-add seam if `comm` has more than one rank
 *           (upx_comm_seam_exchange with `spill`; planes hold t_out >= own_len frames, own_len + spill for a shard with
 *           a successor); then peaks[0] = max |input| and peaks[1] = max(|Ls|,|C|,|Rs|) over the OWNED frames (a NaN
 *           anywhere gives NaN, as np.max does).  The caller takes the maximum of both over the ranks.
 *   finish  planes * scale (main.py:95-97), export layout + quantisation on the device (as upx_wav_pipeline), piece by
 *           piece, each piece's final 2-channel sample data coming down into out0..2 while the next one is exported.
 * upx_wav_pipeline is begin + finish for a shard that is the whole file.
 */
int upx_wav_shard_begin(upx_plan* plan, upx_comm* comm, const void* pcm_in, int in_format, int channels, int64_t t_in,
                        int64_t own_len, int64_t t_out, int64_t spill, double* peaks);
int upx_wav_shard_finish(upx_plan* plan, double scale, int mode, int out_format, void* out0, void* out1, void* out2);
/*
 * The same two halves for a caller that streams FILE -> GPU -> FILE (multi_gpu.run_rank; main.py:43, :119-153 around the
 * device): the reads of the input file overlap the uploads, the writes of the output files overlap the downloads.
 *   open / feed / seal = begin with the shard's samples handed over in order, in pieces of any length: a piece is
 *       queued for upload as soon as it is fed (it must stay valid and unchanged until upx_wav_shard_seal returns;
 *       page-locked memory - upx_host_alloc - makes the upload asynchronous), and every chunk whose input is complete
 *       runs behind it.  seal: the RCCL seam, the peaks (as begin).
 *   finish_async = finish that returns once everything is queued: piece k (piece_frames frames each, 0 = the plan's
 *       UPX_WAV_CHUNK; *n_pieces of them) of out0..2 is valid after upx_wav_shard_wait_piece(plan, k); pieces land in order.
 */
int upx_wav_shard_open(upx_plan* plan, upx_comm* comm, int in_format, int channels, int64_t t_in, int64_t own_len,
                       int64_t t_out, int64_t spill);
int upx_wav_shard_feed(upx_plan* plan, const void* pcm, int64_t n_frames);
int upx_wav_shard_seal(upx_plan* plan, double* peaks);
int upx_wav_shard_finish_async(upx_plan* plan, double scale, int mode, int out_format, void* out0, void* out1, void* out2,
                               int64_t piece_frames, int32_t* n_pieces);
int upx_wav_shard_wait_piece(upx_plan* plan, int32_t piece);
/* Between begin and finish: the device planes of the open shard (plane length t_out, the first own_len samples owned) - for
   a seam applied by the caller (upx_seam_add_local between two shards on one device: long files on one GPU, and the
   one-GPU test of the sharded pipeline) - and the peaks of the owned range recomputed after such a seam. */
int upx_wav_shard_planes(upx_plan* plan, float** d_c, float** d_l, float** d_r, int64_t* own_len, int64_t* t_out);
int upx_wav_shard_peaks(upx_plan* plan, double* peaks);
/* Host wall-clock milliseconds of the last upx_wav_shard_begin / _finish (= upx_wav_pipeline) on this plan:
   ms3[0] = begin (samples up, chunk by chunk, while the previous chunk is decoded and runs through the bands; peaks),
   ms3[1] = the part of begin that came after the last sample had landed (the last chunk's kernels: what is not hidden),
   ms3[2] = finish (export layout + quantisation piece by piece while the previous piece goes down). */
int upx_wav_pipeline_times_ms(upx_plan* plan, float* ms3);

/*
 * Block-at-a-time streaming: MultiBandExtractorAccu.process_stereo_chunk / flush_final (center_extraction.py:353-424)
 * for a plan that holds ONE band.  The overlap-add accumulators of :269-271 live on the device as a ring [3][N]:
 *   upx_stream_chunk  one block of up to N samples per channel (shorter: zero-extended) goes up, the frame is
 *                     transformed (rfft x2, band limit, mask, irfft x3), added onto the ring, and the ring's first hop
 *                     samples come down: 2 N floats up, 3 hop floats down per call (round 3: 3 N down, OLA on the host);
 *                     same float32 additions in the same order as `accum += rec; emit accum[:hop]; shift`.
 *   upx_stream_state  the accumulators in natural order (what :411-424 returns); clear != 0 also zeroes them = flush_final.
 *   upx_stream_set_state  overwrite them (a caller that assigns to .accumC / .accumL / .accumR).
 */
int upx_stream_chunk(upx_plan* plan, const float* block_l, int32_t n_l, const float* block_r, int32_t n_r, float* out_c,
                     float* out_l, float* out_r);
int upx_stream_state(upx_plan* plan, float* acc_c, float* acc_l, float* acc_r, int clear);
int upx_stream_set_state(upx_plan* plan, const float* acc_c, const float* acc_l, const float* acc_r);

/* ---- multi-GPU seam exchange over RCCL (one process per GPU) ------------ */
#define UPX_UNIQUE_ID_BYTES 128
/* Rank 0 creates the id and shares the 128 bytes with the other ranks out of band. */
int upx_comm_unique_id(char* id_out);
int upx_comm_create(upx_comm** out, upx_plan* plan, int rank, int n_ranks, const char* id);
void upx_comm_destroy(upx_comm* comm);
/*
 * Overlap-add seam: each rank's planes hold `spill` samples past own_len that
 * belong to the next rank's head.  Packs them into seam[n_ranks][3][spill],
 * one ncclAllReduce(sum, float32), adds row rank-1 onto the head of this rank.
 * own_len == 0 ranks are not supported.
 */
int upx_comm_seam_exchange(upx_comm* comm, float* d_c, float* d_l, float* d_r, int64_t own_len, int64_t spill);
/*
 * Failure containment between ranks.  A collective has no error path of its own: a rank whose peer never enters the
 * all-reduce waits in hipStreamSynchronize for ever.  The reference fails with plain exceptions (main.py:40-41,
 * center_extraction.py:91-92); here the ranks first VOTE over their process group (upmix_amd.rendezvous.all_ok) before
 * every collective, and what a vote cannot catch - a peer that dies inside the collective - ends in an abort:
 *   upx_comm_wait     waits for the last queued seam exchange; when `timeout_s` (< 0: UPX_COMM_TIMEOUT, else
 *                     UPX_RDZV_TIMEOUT, else 600 s) runs out, or RCCL reports an asynchronous error, the communicator is
 *                     aborted and UPX_ERR_RCCL returned.  upx_wav_shard_seal calls it before it synchronises.
 *   upx_comm_abort    ncclCommAbort: the communicator's kernels leave, the communicator is gone (idempotent); every later
 *                     call on it fails.  The process is expected to exit non-zero - a fresh process is the only restart.
 *   upx_comm_reserve  allocates the seam buffer for `spill` up front (otherwise inside the first exchange).
 */
int upx_comm_wait(upx_comm* comm, double timeout_s);
int upx_comm_abort(upx_comm* comm);
int upx_comm_reserve(upx_comm* comm, int64_t spill);
/* Samples of the predecessor's spill that upx_comm_seam_exchange adds onto the head of `rank`: `spill`, except on the
   LAST rank, whose planes may end at own_len < spill (sharding.ShardGeometry.plan only keeps shards with a successor
   at least `spill` long; the rest of the spill lies past the signal's end); 0 for rank 0.  Pure arithmetic, no GPU. */
int64_t upx_comm_seam_add_len(int rank, int n_ranks, int64_t own_len, int64_t spill);
/* Test hook: run the same pack -> ncclAllReduce -> add sequence with a seam of n_rows rows, packing this rank's
   spill into row my_row and adding that same row back onto its own head (any communicator size, e.g. 1 rank). */
int upx_comm_seam_selftest(upx_comm* comm, float* d_c, float* d_l, float* d_r, int64_t own_len, int64_t spill,
                           int n_rows, int my_row);
/* Same seam arithmetic between two shards on ONE device (no RCCL): adds the
   spill of the `prev` planes onto the head of the `next` planes. */
int upx_seam_add_local(upx_plan* plan, const float* prev_c, const float* prev_l, const float* prev_r,
                       int64_t prev_own_len, float* next_c, float* next_l, float* next_r, int64_t spill);

#ifdef __cplusplus
}
#endif
#endif /* UPMIX_HIP_H */
