/* av1mi.h - C ABI of libav1mi: the MI355X-native AV1 chunk encoder that replaces the
 * `av1an` -> SVT-AV1 subprocess behind the av1-base daemon's encode boundary.
 *
 * Every entry point names the reference interface (file:line under /root/reference) it
 * replaces.  Plain C types only: no C++/torch types cross this boundary; nothing throws.
 *
 * Error convention (maps 1:1 onto the reference's `EncodeError`,
 * crates/daemon/src/encode/av1an.rs:17-30):
 *     0            -> Ok(())
 *     > 0          -> EncodeError::Av1anFailed(code)      (encoder / GPU failure; see AV1MI_E_*)
 *     < 0 (-errno) -> EncodeError::Io(io::Error::from_raw_os_error(errno))
 * `Av1anTerminated` (killed by signal) has no in-process equivalent.
 */
#ifndef AV1MI_H
#define AV1MI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AV1MI_ABI_VERSION 7

/* positive failure codes (-> Av1anFailed(code)) */
enum {
  AV1MI_OK = 0,
  AV1MI_E_INVALID_ARG = 1,   /* bad dimensions / parameters */
  AV1MI_E_NO_DEVICE = 2,     /* no MI355X visible, or HIP runtime failure at init */
  AV1MI_E_HIP = 3,           /* a HIP call or kernel failed; see av1mi_last_error() */
  AV1MI_E_OOM = 4,           /* host or device allocation failed */
  AV1MI_E_OVERFLOW = 5,      /* a tile outgrew its bitstream slot */
  AV1MI_E_FORMAT = 6,        /* input file is not a supported Y4M (420, 8/10 bit) */
  AV1MI_E_UNSUPPORTED = 7    /* valid request this build cannot serve yet (e.g. frames wider than 64 superblocks) */
};

typedef struct av1mi_ctx av1mi_ctx;

/* Operating point.  Replaces the reference's single constant
 *   SVT_PARAMS = "--crf 8 --preset 3 --film-grain 20 ... --keyint 240 --lookahead 40"
 * (crates/daemon/src/encode/av1an.rs:14) and `--pix-format yuv420p10le` (av1an.rs:90). */
typedef struct {
  uint32_t width, height;   /* luma size, even, >= 8, yuv 4:2:0.  Sizes that are not multiples of 8 are coded at the next
                               multiple of 8 (source edge-extended on the device) and signalled exactly, as any AV1 encoder does */
  uint32_t bit_depth;       /* 8 or 10 (samples: uint8_t / little-endian uint16_t) */
  uint32_t cq_level;        /* "--crf N": 0..63, mapped to base_q_idx like aom (30 -> 120) */
  uint32_t keyint;          /* "--keyint": 1 = every frame a key frame; N > 1 = a key frame every N frames of a chunk, the
                               frames between are INTER frames predicted from the previous reconstruction (one
                               reference, integer-pel full search; chunks always start with a key frame) */
  uint32_t block_log2;      /* leaf block size log2: 3 (8x8) .. 6 (64x64: 64-point luma transforms, of which AV1 codes the 32x32
                               low-frequency corner); 0 = default (5) */
  uint32_t cdf_update;      /* 1 = adaptive CDFs (default), 0 = static CDFs (disable_cdf_update) */
  uint32_t enable_cdef;     /* 1 = CDEF on (default) */
  uint32_t cdef_y_pri, cdef_y_sec, cdef_uv_pri, cdef_uv_sec, cdef_damping; /* 0s = defaults */
  uint32_t intra_mode_mask; /* bit m = luma intra mode m is a candidate (AV1 mode numbering); 0 = default {DC, V, H} */
  uint32_t film_grain;      /* "--film-grain N" (av1an.rs:14): 0 = off; N = 1..50 writes a film-grain table into every
                               frame header (2-point scaling functions, value 2N luma / N chroma, AR lag 0);
                               synthesis is decoder-side */
  uint32_t first_frame;     /* number of the chunk's first frame inside the clip (seeds grain_seed per frame) */
  uint32_t me_range;        /* inter frames: motion search range in luma samples, 8 or 16 (0 = 8) */
  uint32_t enable_lr;       /* 1 = loop restoration on luma (Wiener, 64x64 units, each unit off or one of 3 filters by SSE);
                               2 = RESTORE_SWITCHABLE: each unit off, one of the 3 Wiener filters or one of 3 self-guided
                               filters (parameter set 9, three weightings);
                               default 0: the decision needs the CDEF output, which serialises CDEF before entropy coding */
  uint32_t tile_sb;         /* tile size in 64x64 superblocks, both ways: 0 = automatic (1; 2 when the frame has more than 64
                               superblock rows or columns, e.g. 8K - AV1 allows at most 64 x 64 tiles), or force 1 / 2 */
  uint32_t deblock;         /* 1 = deblocking filter on (default 0: `loop_filter_level` = 0); the level follows the quantiser:
                               (ac_q * 20723 + 1015158) >> 18 at 8-bit scale, 4 less on key frames, all four filters alike */
  uint32_t enable_qm;       /* "--enable-qm 1" (av1an.rs:14): quantiser matrices (using_qmatrix).  The level of all three planes
                               follows the quantiser as in SVT-AV1/libaom: qm_min + base_q_idx * (qm_max + 1 - qm_min) / 256
                               (0 = steepest matrix ... 14; 15 = flat).  Default 0 */
  uint32_t qm_min, qm_max;  /* "--qm-min" / "--qm-max": 0..15, qm_min <= qm_max; av1mi_default_params sets 8 / 15 (the encoder's
                               defaults); the reference's production string uses 1 / 15 */
  uint32_t subpel;          /* inter frames: 1 = quarter-sample motion vectors (the full search's winner refined over its half- and then
                               quarter-sample neighbours) predicted with the 8-tap EIGHTTAP filter; 0 (default) = whole-sample
                               vectors, frame filter BILINEAR */
  uint32_t color_range;     /* color_config.color_range of the sequence header: 0 (default) = studio / limited range - what Y4M input and the
                               reference's pipeline (ffmpeg -> SVT-AV1, `--pix-format yuv420p10le`, av1an.rs:90) carry; 1 = full range.
                               av1mi_encode_file takes it from the Y4M header's XCOLORRANGE tag when there is one */
  uint32_t intra_angle_delta; /* 1: a directional winner of the luma mode decision (V, H, D45 .. D67) is refined over the angle deltas
                               -3 .. +3 (3 degrees each) by closed-loop SAD, chroma follows luma; 0 (default): delta 0 */
  uint32_t intra_edge_filter; /* sequence header enable_intra_edge_filter (SVT-AV1 and libaom run with it on): 1 = directional intra
                               predictions read the filtered / upsampled edges of AV1 spec 7.11.2.7 - 7.11.2.12; 0 (default) */
  uint32_t cfl;             /* 1: chroma-from-luma prediction (UV_CFL_PRED, AV1 spec 7.11.5) is a candidate for the chroma planes of key-frame
                               blocks up to 32x32 (alpha per plane by least squares over the reconstructed luma); 0 (default) */
  uint32_t tx_search;       /* 1: transform type search - intra luma blocks up to 16x16 whose residual is sparse (at most one sample in
                               eight nonzero) are coded with the identity transform (IDTX); 0 (default): the mode's default type */
  uint32_t color_primaries, transfer_characteristics, matrix_coefficients;
                            /* color_config's colour description (AV1 spec 5.5.2; CICP / ISO 23091-2 code points), written into the sequence
                               header and, for Matroska output, the track's Colour element: what the reference's ffmpeg -> SVT-AV1 pipeline
                               passes through (`--pix-format yuv420p10le`, av1an.rs:90).  All 0 (default) = no description
                               (color_description_present_flag 0); BASELINE config 5 "8K 10-bit HDR" = 9 / 16 / 9 (BT.2020 primaries, SMPTE
                               2084 PQ, BT.2020 non-constant luminance).  Values 0..255 each; the triple 1 / 13 / 0 (sRGB + identity) implies
                               4:4:4 and is refused */
  uint32_t partition_search; /* 1: content-driven partition - a node of the partition tree larger than `min_block_log2` and not larger than `block_log2`
                               splits when its four quadrants differ in activity (largest quadrant variance of the source luma > 4 x the
                               smallest + (ac_q / 16)^2), so quiet areas keep large blocks and edges / small objects get small ones; decided
                               for all frames of a chunk by one GPU pass over the source.  0 (default): every block is `block_log2` */
  uint32_t min_block_log2;  /* smallest leaf under partition_search: 3 (8x8, default when 0) .. block_log2 */
  uint32_t me_presearch;    /* inter frames: 1 = hierarchical motion search - a quarter-resolution search of +-64 luma samples per 32x32 cell finds the
                               centre the full-resolution search of +-me_range then runs around (row a13's "1/4-res ... full"); 0 (default): the
                               full-resolution search runs around the zero vector */
} av1mi_params;

typedef struct {
  uint8_t *data;            /* owned by the library (a page-locked block of its pool, or malloc); release with av1mi_free() ONLY */
  size_t size;
} av1mi_buf;

/* per-chunk report: fills the reference's hollow JobMetrics fields `fps, frames_encoded, psnr`
 * (crates/daemon/src/metrics.rs:12-30; zeros today at job_executor.rs:117-137) */
typedef struct {
  uint32_t frames;
  uint64_t bytes;
  double sse[3];            /* sum of squared error of the reconstruction, per plane */
  double psnr[3];
  /* device time per stage, milliseconds (HIP events on the encoder's stream) */
  float ms_h2d, ms_recon, ms_cdef, ms_entropy, ms_pack, ms_d2h, ms_total;
  float ms_symbolize;       /* part of ms_entropy spent in the symbolize kernel */
  uint64_t n_symbols;       /* arithmetic-coded symbols */
  uint32_t max_tile_symbols; /* longest tile: the serial chain that bounds the range-coding kernel */
  uint32_t cap_scale;       /* per-tile capacity multiplier the chunk finally ran with (1 unless a tile overflowed
                               and the chunk was re-run) */
  uint32_t chunks;          /* av1mi_encode_file: chunks the clip was split into (1 for av1mi_encode_chunk) */
  uint32_t gpus_used;       /* bit d = GPU d encoded at least one chunk of the job (av1mi_encode_chunk: the context's device) - what
                               `gpu_mask` and av1mi_plan_workers resolved to on this host */
} av1mi_report;

void av1mi_default_params(av1mi_params *p, uint32_t width, uint32_t height, uint32_t bit_depth);

/* One context = one GPU + one HIP stream = one chunk in flight.  `workers` of the reference's
 * ConcurrencyPlan (crates/daemon/src/concurrency.rs:9-18 -> `--workers`, av1an.rs:100-101)
 * becomes "number of contexts".  Thread-safe across contexts; one thread per context. */
int av1mi_ctx_create(int device_id, av1mi_ctx **out);
void av1mi_ctx_destroy(av1mi_ctx *ctx);
const char *av1mi_last_error(const av1mi_ctx *ctx);

/* Encode one scene-chunk: `n_frames` planar I420 frames, frame k at
 * `frames + k * frame_bytes` with frame_bytes = w*h*3/2*bytes_per_sample (Y then U then V) -
 * the unit av1an hands to one SVT-AV1 worker.  `frames_on_device` != 0: `frames` is a device
 * pointer (HBM-resident input).  Output: a Section-5 OBU stream (temporal delimiter + sequence
 * header + OBU_FRAME per frame); `frame_sizes` (optional, n_frames entries) receives each
 * temporal unit's size.  `recon` (optional, same layout/pointer kind as `frames`) receives the
 * decoder-identical reconstruction.  Blocking; returns the error codes above. */
int av1mi_encode_chunk(av1mi_ctx *ctx, const av1mi_params *params, const void *frames, uint32_t n_frames,
                       int frames_on_device, av1mi_buf *out, uint32_t *frame_sizes, void *recon,
                       av1mi_report *report);

void av1mi_free(void *p);

/* ---- scene-cut detection (the chunker) ----------------------------------------------------
 * Replaces the scene detection av1an performs before it hands chunks to its `--workers N` encoders
 * (`--workers`, `--temp`: crates/daemon/src/encode/av1an.rs:100-104).
 * Device: SAD of every frame's luma against its predecessor (one streaming kernel).  Host: frame t
 * starts a new scene iff  2*d(t) >= 5*mean(d over the last <= 8 in-scene frames)  and  d(t) >= 8
 * (d = mean absolute luma difference at 8-bit scale, Q8 integers; with no history yet: d(t) >= 24)
 * and the current scene already holds `min_scene_len` frames.  A frame without predecessor
 * (`prev_frame` NULL and t = 0) always starts a scene.  `state` (zero-initialised by the caller) carries
 * the history across calls so a clip can be streamed window by window; `prev_frame` is the last frame of
 * the previous window (same residency as `frames`).  Outputs (each n_frames entries, optional):
 * `sad` = luma SAD against the predecessor (0 for a frame without one), `is_cut` = 1 where a scene starts. */
typedef struct {
  uint32_t frames_since_cut;
  uint32_t hist_n;
  uint32_t hist_q8[8];
} av1mi_scene_state;

int av1mi_scene_cuts(av1mi_ctx *ctx, const av1mi_params *params, const void *frames, uint32_t n_frames,
                     int frames_on_device, const void *prev_frame, av1mi_scene_state *state,
                     uint32_t min_scene_len, uint64_t *sad, uint8_t *is_cut);

/* ---- the drop-in for `run_av1an` ---------------------------------------------------------
 * Replaces  pub fn run_av1an(params: &Av1anEncodeParams) -> Result<(), EncodeError>
 * (crates/daemon/src/encode/av1an.rs:126-139), called from JobExecutor::execute through
 * tokio::task::spawn_blocking (crates/daemon/src/job_executor.rs:279-287).
 * Field for field `Av1anEncodeParams` (av1an.rs:36-45): input_path, output_path,
 * temp_chunks_dir, concurrency.av1an_workers. */
typedef struct {
  const char *input_path;   /* Y4M (420jpeg/420p10) - container demux/decode is out of scope */
  const char *output_path;  /* IVF written atomically (tmp + rename); never left zero-length */
  const char *temp_dir;     /* caller-owned scratch (job_executor.rs:275-276); only tmp files go here */
  uint32_t workers;         /* chunks in flight = contexts, placed on the allowed GPUs by av1mi_plan_workers; 0 = default
                               (AV1MI_DEFAULT_WORKERS_PER_GPU per allowed GPU) */
  uint32_t chunk_frames;    /* frames per chunk; 0 = chunks end at detected scene cuts (av1mi_scene_cuts,
                               min scene 12 frames) and after 240 frames at the latest (the reference's
                               `--keyint 240`, av1an.rs:14) */
  int32_t gpu_mask;         /* bit i = may use GPU i; <= 0 = all visible */
  av1mi_params params;      /* width/height/bit_depth are taken from the Y4M header */
} av1mi_job;

/* frames_total: the clip's length when the input is a regular file (from its size), else the frames read so far */
typedef void (*av1mi_progress_cb)(void *user, uint32_t frames_done, uint32_t frames_total, double fps,
                                  uint64_t bytes_out);

/* What the daemon's probe pass (JobMetrics.total_frames / bitrate_kbps, crates/daemon/src/metrics.rs:12-30; zeros today, job_executor.rs:117-137) needs
 * from a Y4M input: geometry, frame rate, range tag, frame count (0 = unknown: not a regular file). */
typedef struct {
  uint32_t width, height, bit_depth, fps_num, fps_den, color_range;
  uint64_t frames;
} av1mi_clip_info;
int av1mi_probe_y4m(const char *path, av1mi_clip_info *info);

int av1mi_encode_file(const av1mi_job *job, av1mi_progress_cb cb, void *user, av1mi_report *total);

/* av1mi_encode_file keeps its idle contexts (streams + HBM workspace) and pinned host buffers for the next job of the process -
 * the daemon encodes job after job (job_executor.rs:266-437), and setting them up costs more than a short clip's encode.
 * Bounded (8 contexts per GPU, 24 GiB of host buffers); this returns everything that is idle. */
void av1mi_release_caches(void);

/* ---- chunk -> GPU placement (independent chunks, no exchange step) -----------------------------------------
 * Replaces the placement av1an does implicitly by forking `--workers N` encoder processes on one host (av1an.rs:100-101;
 * ConcurrencyPlan.av1an_workers, concurrency.rs:9-18, 67-73).  Two pure functions, so the rule can be tested without a GPU and
 * is the same wherever chunks meet devices (av1mi_encode_file's worker pool, bench.py's ranks):
 *   av1mi_chunk_owner   chunk i of a job -> owner i mod n_owners (owner = a rank of the N-GPU bench, or a GPU of one process)
 *   av1mi_plan_workers  `workers` contexts over the GPUs `gpu_mask` allows among `n_devices`: worker i -> the
 *                       av1mi_chunk_owner(i, allowed)-th allowed device; workers == 0 -> AV1MI_DEFAULT_WORKERS_PER_GPU per allowed
 *                       GPU; at most 64.  Writes min(result, cap) entries; returns the worker count, 0 if no device is allowed. */
#define AV1MI_DEFAULT_WORKERS_PER_GPU 4
uint32_t av1mi_chunk_owner(uint32_t chunk_index, uint32_t n_owners);
int av1mi_plan_workers(uint32_t workers, int32_t gpu_mask, int n_devices, int32_t *device_of_worker, uint32_t cap);

/* ---- the caller's encode segment ---------------------------------------------------------------
 * Mirror of the part of JobExecutor::execute that surrounds the encode call (crates/daemon/src/job_executor.rs:
 * 266-317 and 413-436): state -> "encoding", create `{temp_base_dir}/chunks_{id}`, run the encode, on success state ->
 * "validating" and require the output to exist and be non-empty, remove the chunks directory on every path, and fill the
 * JobMetrics fields the reference leaves at zero (metrics.rs:12-30; job_executor.rs:117-137).  Size gate, atomic
 * replacement and skip markers (job_executor.rs:319-411) stay with the caller: control plane, out of scope.
 * `state_cb` is called with the reference's stage strings ("encoding", "validating", then "size_gating" when the segment
 * hands back to the caller, or "failed"); `error` receives the reference's failure text ("Output file not found: ...",
 * "Output file is empty", "MI355X encoder failed with exit code: N", "IO error: ...").  Returns 0, or the failing code. */
typedef struct {
  const char *id;              /* Job.id */
  const char *input_path, *output_path;
  const char *temp_base_dir;   /* JobExecutor.temp_base_dir */
  uint32_t workers;            /* ConcurrencyPlan.av1an_workers */
  av1mi_params params;
} av1mi_exec_job;

typedef struct {               /* JobMetrics (metrics.rs:12-30), the fields this path can fill */
  char stage[16];
  float progress, fps, bitrate_kbps, psnr;
  uint32_t crf, workers;
  uint64_t frames_encoded, total_frames, size_in_bytes_after;
} av1mi_job_metrics;

typedef void (*av1mi_state_cb)(void *user, const char *stage, const av1mi_job_metrics *m);

int av1mi_job_execute(const av1mi_exec_job *job, av1mi_state_cb state_cb, void *user, av1mi_job_metrics *metrics,
                      char *error, size_t error_cap);

/* Layout pin of this ABI: a binding (integration/mi355x.rs, the ctypes mirror) that was written against another revision of a structure
 * corrupts memory silently - AV1MI_ABI_VERSION is bumped by hand.  Writes up to `cap` entries
 *   { sizeof(av1mi_params), sizeof(av1mi_job), sizeof(av1mi_report), sizeof(av1mi_buf), sizeof(av1mi_clip_info), sizeof(av1mi_scene_state),
 *     sizeof(av1mi_exec_job), sizeof(av1mi_job_metrics), offsetof(av1mi_job, params), offsetof(av1mi_report, ms_h2d),
 *     offsetof(av1mi_exec_job, params), offsetof(av1mi_job_metrics, frames_encoded) }
 * and returns how many there are (12); a binding asserts them against its own `size_of` / `offset_of` once at start-up
 * (integration/mi355x.rs: `check_layout`; tests/test_abi_host.py does it for the ctypes mirror). */
#define AV1MI_LAYOUT_ENTRIES 12
uint32_t av1mi_struct_sizes(uint32_t *sizes, uint32_t cap);

/* helpers shared with the host mirror / tests */
uint32_t av1mi_cq_to_qindex(uint32_t cq_level);
uint32_t av1mi_abi_version(void);
/* writes the sequence header OBU + frame header bytes the encoder will emit (for KATs) */
int av1mi_write_headers(const av1mi_params *p, uint8_t *seq_hdr, size_t *seq_len, uint8_t *frame_hdr,
                        size_t *frame_hdr_bits);

#ifdef __cplusplus
}
#endif
#endif
