// integration/mi355x.rs - the file a maintainer adds as crates/daemon/src/encode/mi355x.rs.
// SOURCE ONLY: the build image has no cargo/rustc, so this has not been compiled; INTEGRATION.md explains every line.
//! MI355X in-process encoder: drop-in for `run_av1an` (same params, same error type).
use super::av1an::{Av1anEncodeParams, EncodeError};
use std::ffi::CString;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct Av1miParams {            // include/av1mi.h: av1mi_params
    pub width: u32, pub height: u32, pub bit_depth: u32,
    pub cq_level: u32, pub keyint: u32, pub block_log2: u32, pub cdf_update: u32, pub enable_cdef: u32,
    pub cdef_y_pri: u32, pub cdef_y_sec: u32, pub cdef_uv_pri: u32, pub cdef_uv_sec: u32, pub cdef_damping: u32,
    pub intra_mode_mask: u32, pub film_grain: u32, pub first_frame: u32, pub me_range: u32, pub enable_lr: u32, pub tile_sb: u32, pub deblock: u32,
    pub enable_qm: u32, pub qm_min: u32, pub qm_max: u32, pub subpel: u32,
    pub color_range: u32,           // 0 = studio (default; a Y4M XCOLORRANGE tag wins), 1 = full
    pub intra_angle_delta: u32,     // 1 = directional intra winners refined over the angle deltas -3..+3
    pub intra_edge_filter: u32,     // 1 = enable_intra_edge_filter (filtered / upsampled prediction edges)
    pub cfl: u32,                   // 1 = chroma-from-luma prediction is a candidate (key frames, blocks up to 32x32)
    pub tx_search: u32,             // 1 = identity transform (IDTX) for sparse intra luma residuals
    pub color_primaries: u32, pub transfer_characteristics: u32, pub matrix_coefficients: u32,   // CICP colour description (0 / 0 / 0 = none; HDR10: 9 / 16 / 9)
    pub partition_search: u32,      // 1 = content-driven partition between min_block_log2 and block_log2
    pub min_block_log2: u32,        // smallest leaf under partition_search (0 = 3: 8x8)
    pub me_presearch: u32,          // 1 = quarter-resolution pre-search (+-64) before the full-resolution search
}
#[repr(C)]
pub struct Av1miJob {               // include/av1mi.h: av1mi_job  <->  Av1anEncodeParams (av1an.rs:36-45)
    pub input_path: *const c_char,  // params.input_path
    pub output_path: *const c_char, // params.output_path
    pub temp_dir: *const c_char,    // params.temp_chunks_dir (caller-owned, job_executor.rs:275-276)
    pub workers: u32,               // params.concurrency.av1an_workers (`--workers`, av1an.rs:100-101)
    pub chunk_frames: u32,
    pub gpu_mask: i32,
    pub params: Av1miParams,
}
#[repr(C)]
#[derive(Default)]
pub struct Av1miReport {            // fills JobMetrics.{fps, frames_encoded, psnr} (metrics.rs:12-30)
    pub frames: u32, pub bytes: u64, pub sse: [f64; 3], pub psnr: [f64; 3],
    pub ms_h2d: f32, pub ms_recon: f32, pub ms_cdef: f32, pub ms_entropy: f32, pub ms_pack: f32,
    pub ms_d2h: f32, pub ms_total: f32, pub ms_symbolize: f32, pub n_symbols: u64,
    pub max_tile_symbols: u32, pub cap_scale: u32, pub chunks: u32, pub gpus_used: u32,
}
type ProgressCb = Option<extern "C" fn(user: *mut c_void, done: u32, total: u32, fps: f64, bytes: u64)>;

#[link(name = "av1mi")]
extern "C" {
    fn av1mi_abi_version() -> u32;
    fn av1mi_struct_sizes(sizes: *mut u32, cap: u32) -> u32;
    fn av1mi_default_params(p: *mut Av1miParams, w: u32, h: u32, bit_depth: u32);
    fn av1mi_encode_file(job: *const Av1miJob, cb: ProgressCb, user: *mut c_void, total: *mut Av1miReport) -> c_int;
}

// Layout pin (include/av1mi.h: av1mi_struct_sizes).  Compile time: the sizes this file was written against (ABI version 7, LP64);
// run time, once: the library's own sizes and offsets - a libav1mi.so built from another revision of the header is refused
// instead of being handed structures it would read past.
pub const AV1MI_ABI_VERSION: u32 = 7;
const _: () = assert!(std::mem::size_of::<Av1miParams>() == 35 * 4);
const _: () = assert!(std::mem::size_of::<Av1miJob>() == 3 * 8 + 3 * 4 + 35 * 4);
const _: () = assert!(std::mem::size_of::<Av1miReport>() == 120);
pub fn check_layout() -> Result<(), EncodeError> {
    static ONCE: std::sync::OnceLock<bool> = std::sync::OnceLock::new();
    let ok = *ONCE.get_or_init(|| unsafe {
        let mut v = [0u32; 12];
        av1mi_abi_version() == AV1MI_ABI_VERSION && av1mi_struct_sizes(v.as_mut_ptr(), 12) == 12
            && v[0] as usize == std::mem::size_of::<Av1miParams>() && v[1] as usize == std::mem::size_of::<Av1miJob>()
            && v[2] as usize == std::mem::size_of::<Av1miReport>() && v[8] as usize == std::mem::offset_of!(Av1miJob, params)
            && v[9] as usize == std::mem::offset_of!(Av1miReport, ms_h2d)
    });
    if ok { Ok(()) } else { Err(EncodeError::Av1anFailed(7 /* AV1MI_E_UNSUPPORTED: ABI mismatch */)) }
}

/// Same contract as `run_av1an` (av1an.rs:126-139): blocks until `output_path` is complete.
pub fn run_mi355x(params: &Av1anEncodeParams, cq_level: u32) -> Result<(), EncodeError> {
    let c = |p: &std::path::Path| CString::new(p.as_os_str().as_encoded_bytes()).map_err(|e| EncodeError::Io(std::io::Error::other(e)));
    check_layout()?;
    let (i, o, t) = (c(&params.input_path)?, c(&params.output_path)?, c(&params.temp_chunks_dir)?);
    let mut p = Av1miParams::default();
    unsafe { av1mi_default_params(&mut p, 8, 8, 8) };   // geometry comes from the Y4M header
    // the reference's operating point, SVT_PARAMS (av1an.rs:14): "--crf 8 ... --film-grain 20 ... --keyint 240"
    p.cq_level = cq_level;                               // "--crf" (8 in production; 30 is the benchmark's operating point)
    p.keyint = 240;                                      // "--keyint 240": IPPP inside a chunk, chunks start at scene cuts
    p.film_grain = 20;                                   // "--film-grain 20": film-grain table in every frame header
    p.enable_qm = 1; p.qm_min = 1; p.qm_max = 15;        // "--enable-qm 1 --qm-min 1 --qm-max 15": quantiser matrices
    p.subpel = 1; p.deblock = 1; p.enable_lr = 2;        // tools SVT-AV1 has on at "--preset 3": sub-sample motion, deblocking, restoration
    let job = Av1miJob { input_path: i.as_ptr(), output_path: o.as_ptr(), temp_dir: t.as_ptr(),
                         workers: params.concurrency.av1an_workers, chunk_frames: 0 /* scene-cut chunks */, gpu_mask: 0, params: p };
    let mut rep = Av1miReport::default();
    match unsafe { av1mi_encode_file(&job, None, std::ptr::null_mut(), &mut rep) } {
        0 => Ok(()),
        rc if rc > 0 => Err(EncodeError::Av1anFailed(rc)),                                   // av1an.rs:21
        rc => Err(EncodeError::Io(std::io::Error::from_raw_os_error(-rc))),                   // av1an.rs:29
    }
}
