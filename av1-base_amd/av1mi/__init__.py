"""av1mi - Python binding of libav1mi (the MI355X AV1 chunk encoder) and the host-side mirror of
the reference's encode boundary.

Reference interface mirrored (files under /root/reference/crates/daemon/src):
  encode/av1an.rs:36-61   struct Av1anEncodeParams{input_path, output_path, temp_chunks_dir,
                          concurrency} + ::new          -> EncodeParams
  encode/av1an.rs:17-30   enum EncodeError{Av1anFailed(i32), Av1anTerminated, Io(io::Error)}
                                                         -> EncodeError / EncodeFailed / EncodeIo
  encode/av1an.rs:126-139 fn run_av1an(&params) -> Result<(), EncodeError>   -> run_mi355x
  concurrency.rs:9-18,28-89 ConcurrencyPlan + derive (av1an_workers -> `--workers`)
                                                         -> ConcurrencyPlan / derive_plan

There is NO CPU fallback: if libav1mi.so is missing this module raises at import, and without a
HIP device every call fails with AV1MI_E_NO_DEVICE.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AV1MI_LIB") or os.path.join(_HERE, "..", "libav1mi.so")   # AV1MI_LIB: another build of the same library (A/B runs on one box)

# PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so).  Two HIP runtimes in
# one process do not coexist, so when torch is installed it is imported FIRST: libav1mi.so's
# DT_NEEDED libamdhip64.so.7 then binds to the runtime already in the process.  A non-Python host
# (the daemon through its FFI) has only /opt/rocm's runtime and needs none of this.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    pass

if not os.path.exists(LIB_PATH):
    raise ImportError("libav1mi.so not built (run __graft_entry__.build() or av1-base_amd/build.py): %s" % LIB_PATH)
_lib = C.CDLL(os.path.abspath(LIB_PATH))

E_INVALID_ARG, E_NO_DEVICE, E_HIP, E_OOM, E_OVERFLOW, E_FORMAT, E_UNSUPPORTED = range(1, 8)

# every symbol include/av1mi.h declares
ABI_SYMBOLS = ["av1mi_default_params", "av1mi_ctx_create", "av1mi_ctx_destroy", "av1mi_last_error", "av1mi_encode_chunk",
               "av1mi_free", "av1mi_encode_file", "av1mi_cq_to_qindex", "av1mi_abi_version", "av1mi_write_headers", "av1mi_scene_cuts", "av1mi_job_execute", "av1mi_probe_y4m", "av1mi_chunk_owner", "av1mi_plan_workers", "av1mi_release_caches", "av1mi_struct_sizes"]


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("width", "height", "bit_depth", "cq_level", "keyint", "block_log2", "cdf_update",
                                          "enable_cdef", "cdef_y_pri", "cdef_y_sec", "cdef_uv_pri", "cdef_uv_sec",
                                          "cdef_damping")] + [("intra_mode_mask", C.c_uint32), ("film_grain", C.c_uint32), ("first_frame", C.c_uint32), ("me_range", C.c_uint32), ("enable_lr", C.c_uint32), ("tile_sb", C.c_uint32), ("deblock", C.c_uint32), ("enable_qm", C.c_uint32), ("qm_min", C.c_uint32), ("qm_max", C.c_uint32), ("subpel", C.c_uint32), ("color_range", C.c_uint32), ("intra_angle_delta", C.c_uint32), ("intra_edge_filter", C.c_uint32), ("cfl", C.c_uint32), ("tx_search", C.c_uint32),
                                                              ("color_primaries", C.c_uint32), ("transfer_characteristics", C.c_uint32), ("matrix_coefficients", C.c_uint32),
                                                              ("partition_search", C.c_uint32), ("min_block_log2", C.c_uint32), ("me_presearch", C.c_uint32)]


class Buf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("size", C.c_size_t)]


class Report(C.Structure):
    _fields_ = [("frames", C.c_uint32), ("bytes", C.c_uint64), ("sse", C.c_double * 3), ("psnr", C.c_double * 3),
                ("ms_h2d", C.c_float), ("ms_recon", C.c_float), ("ms_cdef", C.c_float), ("ms_entropy", C.c_float),
                ("ms_pack", C.c_float), ("ms_d2h", C.c_float), ("ms_total", C.c_float), ("ms_symbolize", C.c_float),
                ("n_symbols", C.c_uint64), ("max_tile_symbols", C.c_uint32), ("cap_scale", C.c_uint32), ("chunks", C.c_uint32), ("gpus_used", C.c_uint32)]


class Job(C.Structure):
    _fields_ = [("input_path", C.c_char_p), ("output_path", C.c_char_p), ("temp_dir", C.c_char_p), ("workers", C.c_uint32),
                ("chunk_frames", C.c_uint32), ("gpu_mask", C.c_int32), ("params", Params)]


class SceneState(C.Structure):
    """include/av1mi.h: av1mi_scene_state (zero-initialised = start of a clip)"""
    _fields_ = [("frames_since_cut", C.c_uint32), ("hist_n", C.c_uint32), ("hist_q8", C.c_uint32 * 8)]


class ExecJob(C.Structure):
    """include/av1mi.h: av1mi_exec_job"""
    _fields_ = [("id", C.c_char_p), ("input_path", C.c_char_p), ("output_path", C.c_char_p), ("temp_base_dir", C.c_char_p),
                ("workers", C.c_uint32), ("params", Params)]


class JobMetrics(C.Structure):
    """include/av1mi.h: av1mi_job_metrics (the reference's JobMetrics fields this path fills, metrics.rs:12-30)"""
    _fields_ = [("stage", C.c_char * 16), ("progress", C.c_float), ("fps", C.c_float), ("bitrate_kbps", C.c_float), ("psnr", C.c_float),
                ("crf", C.c_uint32), ("workers", C.c_uint32), ("frames_encoded", C.c_uint64), ("total_frames", C.c_uint64),
                ("size_in_bytes_after", C.c_uint64)]


class ClipInfo(C.Structure):
    """include/av1mi.h: av1mi_clip_info"""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bit_depth", C.c_uint32), ("fps_num", C.c_uint32), ("fps_den", C.c_uint32),
                ("color_range", C.c_uint32), ("frames", C.c_uint64)]


STATE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.POINTER(JobMetrics))
PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_uint64)

_lib.av1mi_default_params.argtypes = [C.POINTER(Params), C.c_uint32, C.c_uint32, C.c_uint32]
_lib.av1mi_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
_lib.av1mi_ctx_destroy.argtypes = [C.c_void_p]
_lib.av1mi_last_error.argtypes = [C.c_void_p]
_lib.av1mi_last_error.restype = C.c_char_p
_lib.av1mi_encode_chunk.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_int, C.POINTER(Buf),
                                    C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(Report)]
_lib.av1mi_free.argtypes = [C.c_void_p]
_lib.av1mi_scene_cuts.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(SceneState),
                                  C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)]
_lib.av1mi_encode_file.argtypes = [C.POINTER(Job), PROGRESS_CB, C.c_void_p, C.POINTER(Report)]
_lib.av1mi_job_execute.argtypes = [C.POINTER(ExecJob), STATE_CB, C.c_void_p, C.POINTER(JobMetrics), C.c_char_p, C.c_size_t]
_lib.av1mi_chunk_owner.argtypes = [C.c_uint32, C.c_uint32]
_lib.av1mi_chunk_owner.restype = C.c_uint32
_lib.av1mi_plan_workers.argtypes = [C.c_uint32, C.c_int32, C.c_int, C.POINTER(C.c_int32), C.c_uint32]
_lib.av1mi_probe_y4m.argtypes = [C.c_char_p, C.POINTER(ClipInfo)]
_lib.av1mi_cq_to_qindex.argtypes = [C.c_uint32]
_lib.av1mi_cq_to_qindex.restype = C.c_uint32
_lib.av1mi_abi_version.restype = C.c_uint32
_lib.av1mi_struct_sizes.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
_lib.av1mi_struct_sizes.restype = C.c_uint32
ABI_VERSION = 7   # include/av1mi.h: AV1MI_ABI_VERSION this mirror was written against
_lib.av1mi_write_headers.argtypes = [C.POINTER(Params), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t)]


# ------------------------------------------------------------------ error taxonomy (av1an.rs:17-30)
class EncodeError(Exception):
    """Base of the reference's `EncodeError` variants."""


class EncodeFailed(EncodeError):
    """`EncodeError::Av1anFailed(code)`: the encoder returned a non-zero code."""

    def __init__(self, code, detail=""):
        super().__init__("MI355X encoder failed with code: %d%s" % (code, (" (" + detail + ")") if detail else ""))
        self.code = code


class EncodeIo(EncodeError):
    """`EncodeError::Io(io::Error)`."""

    def __init__(self, err):
        super().__init__("IO error: %s" % os.strerror(err))
        self.errno = err


def _raise_for(rc, detail=""):
    if rc == 0:
        return
    if rc < 0:
        raise EncodeIo(-rc)
    raise EncodeFailed(rc, detail)


def default_params(width, height, bit_depth=8, **kw):
    p = Params()
    _lib.av1mi_default_params(C.byref(p), width, height, bit_depth)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def struct_sizes():
    """include/av1mi.h: av1mi_struct_sizes - the library's own sizes / offsets of the ABI structures"""
    v = (C.c_uint32 * 32)()
    n = int(_lib.av1mi_struct_sizes(v, 32))
    return list(v[:n])


def mirror_sizes():
    """the same list computed from this module's ctypes mirror (tests/test_abi_host.py compares the two; check_layout() at import
    refuses a library whose structures this mirror does not describe)"""
    return [C.sizeof(Params), C.sizeof(Job), C.sizeof(Report), C.sizeof(Buf), C.sizeof(ClipInfo), C.sizeof(SceneState), C.sizeof(ExecJob),
            C.sizeof(JobMetrics), Job.params.offset, Report.ms_h2d.offset, ExecJob.params.offset, JobMetrics.frames_encoded.offset]


def chunk_owner(chunk_index, n_owners):
    """chunk i -> owner i mod n (include/av1mi.h: av1mi_chunk_owner)"""
    return int(_lib.av1mi_chunk_owner(chunk_index, n_owners))


def chunks_of_rank(n_chunks, world, rank):
    """the chunks of an n_chunks job that rank `rank` of `world` encodes (bench.py's ranks, one GPU each)"""
    return [c for c in range(n_chunks) if chunk_owner(c, world) == rank]


def plan_workers(workers, gpu_mask, n_devices):
    """device of every worker context (include/av1mi.h: av1mi_plan_workers)"""
    buf = (C.c_int32 * 64)()
    n = _lib.av1mi_plan_workers(workers, gpu_mask, n_devices, buf, 64)
    return [int(buf[i]) for i in range(min(n, 64))]


def release_caches():
    """give back the idle contexts and pinned host buffers av1mi_encode_file keeps between jobs"""
    _lib.av1mi_release_caches()


def probe_y4m(path):
    """Geometry, frame rate, range tag and frame count of a Y4M file (include/av1mi.h: av1mi_probe_y4m)."""
    ci = ClipInfo()
    _raise_for(_lib.av1mi_probe_y4m(os.fspath(path).encode(), C.byref(ci)))
    return ci


def cq_to_qindex(cq):
    return int(_lib.av1mi_cq_to_qindex(cq))


def write_headers(params):
    seq = C.create_string_buffer(64)
    fh = C.create_string_buffer(1024)
    n = C.c_size_t(64)
    bits = C.c_size_t(0)
    _raise_for(_lib.av1mi_write_headers(C.byref(params), seq, C.byref(n), fh, C.byref(bits)))
    return seq.raw[:n.value], fh.raw[:(bits.value + 7) // 8], bits.value


class Context:
    """One GPU + one HIP stream = one chunk in flight (include/av1mi.h: av1mi_ctx)."""

    def __init__(self, device_id=0):
        h = C.c_void_p()
        _raise_for(_lib.av1mi_ctx_create(device_id, C.byref(h)), "av1mi_ctx_create")
        self._h = h

    def close(self):
        if self._h:
            _lib.av1mi_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def last_error(self):
        return _lib.av1mi_last_error(self._h).decode()

    def scene_cuts(self, params, frames, n_frames, prev_frame=None, state=None, min_scene_len=12, on_device=False):
        """Scene-cut pass over `n_frames` frames (host bytes, or device pointers with on_device=True).
        Returns ([luma SAD vs predecessor], [is_cut], state)."""
        state = state if state is not None else SceneState()
        sad = (C.c_uint64 * n_frames)()
        cut = (C.c_uint8 * n_frames)()
        if on_device:
            fptr = C.c_void_p(int(frames))
            pptr = C.c_void_p(int(prev_frame)) if prev_frame else None
        else:
            keep = bytes(frames)
            fptr = C.cast(C.c_char_p(keep), C.c_void_p)
            keep2 = bytes(prev_frame) if prev_frame is not None else None
            pptr = C.cast(C.c_char_p(keep2), C.c_void_p) if keep2 is not None else None
        rc = _lib.av1mi_scene_cuts(self._h, C.byref(params), fptr, n_frames, 1 if on_device else 0, pptr, C.byref(state),
                                   min_scene_len, sad, cut)
        _raise_for(rc, self.last_error())
        return list(sad), list(cut), state

    def encode_chunk(self, params, frames, n_frames, on_device=False, want_recon=False, recon_ptr=None, copy_out=True):
        """frames: bytes-like/numpy (host) or an int device pointer (on_device=True).
        Returns (bitstream bytes, [frame sizes], Report, recon bytes or None).  copy_out=False: the library's host buffer is
        released without being copied into a Python object (first element None; Report.bytes has the size) - what a
        throughput measurement wants: the C call has returned, the bitstream was complete in host memory."""
        import numpy as np
        out = Buf()
        sizes = (C.c_uint32 * n_frames)()
        rep = Report()
        bps = 2 if params.bit_depth > 8 else 1
        nbytes = params.width * params.height * 3 // 2 * bps * n_frames
        recon = None
        rptr = None
        if on_device:
            fptr = C.c_void_p(int(frames))
            if recon_ptr is not None:
                rptr = C.c_void_p(int(recon_ptr))
        else:
            arr = np.ascontiguousarray(np.frombuffer(frames, dtype=np.uint8) if not isinstance(frames, np.ndarray) else frames)
            if arr.nbytes != nbytes:
                raise ValueError("frames: expected %d bytes, got %d" % (nbytes, arr.nbytes))
            fptr = arr.ctypes.data_as(C.c_void_p)
            if want_recon:
                recon = np.empty(nbytes, dtype=np.uint8)
                rptr = recon.ctypes.data_as(C.c_void_p)
        rc = _lib.av1mi_encode_chunk(self._h, C.byref(params), fptr, n_frames, 1 if on_device else 0, C.byref(out), sizes, rptr,
                                     C.byref(rep))
        if rc:
            _raise_for(rc, self.last_error())
        data = C.string_at(out.data, out.size) if copy_out else None
        _lib.av1mi_free(out.data)
        return data, list(sizes), rep, recon


# ------------------------------------------------------------------ ConcurrencyPlan (concurrency.rs:9-89)
class ConcurrencyPlan:
    def __init__(self, total_cores, target_threads, av1an_workers, max_concurrent_jobs):
        self.total_cores = total_cores
        self.target_threads = target_threads
        self.av1an_workers = av1an_workers
        self.max_concurrent_jobs = max_concurrent_jobs


def derive_plan(total_cores, target_cpu_utilization=0.85, workers_override=0, max_jobs_override=0):
    """Restates ConcurrencyPlan::derive (concurrency.rs:28-61): utilisation clamped to [0.5, 1.0],
    8 workers if >= 32 cores else 4, 1 job if >= 24 cores else 2; overrides win when non-zero."""
    util = min(1.0, max(0.5, float(target_cpu_utilization)))
    target_threads = max(1, int(round(total_cores * util)))
    workers = workers_override if workers_override else (8 if total_cores >= 32 else 4)
    jobs = max_jobs_override if max_jobs_override else (1 if total_cores >= 24 else 2)
    return ConcurrencyPlan(total_cores, target_threads, workers, jobs)


# ------------------------------------------------------------------ the boundary (av1an.rs:36-139)
class EncodeParams:
    """Field for field `Av1anEncodeParams` (av1an.rs:36-45), plus the operating point that the
    reference hard-codes as SVT_PARAMS (av1an.rs:14)."""

    def __init__(self, input_path, output_path, temp_chunks_dir, concurrency, cq_level=30, chunk_frames=60, gpu_mask=0, **enc):
        self.input_path = os.fspath(input_path)
        self.output_path = os.fspath(output_path)
        self.temp_chunks_dir = os.fspath(temp_chunks_dir)
        self.concurrency = concurrency
        self.cq_level = cq_level
        self.chunk_frames = chunk_frames
        self.gpu_mask = gpu_mask
        self.enc = enc


def run_mi355x(params, progress=None):
    """Drop-in for `run_av1an(&params)`: blocks until the output file is complete; returns the
    Report on success, raises EncodeFailed / EncodeIo otherwise (never leaves a partial output)."""
    job = Job()
    job.input_path = params.input_path.encode()
    job.output_path = params.output_path.encode()
    job.temp_dir = params.temp_chunks_dir.encode()
    job.workers = params.concurrency.av1an_workers if params.concurrency else 0
    job.chunk_frames = params.chunk_frames
    job.gpu_mask = params.gpu_mask
    p = default_params(8, 8, 8, cq_level=params.cq_level, **params.enc)
    job.params = p
    rep = Report()
    cb = PROGRESS_CB(lambda u, d, t, fps, b: progress(d, t, fps, b)) if progress else C.cast(None, PROGRESS_CB)
    rc = _lib.av1mi_encode_file(C.byref(job), cb, None, C.byref(rep))
    _raise_for(rc)
    return rep


def job_execute(job_id, input_path, output_path, temp_base_dir, workers=0, cq_level=30, **enc):
    """The encode segment of JobExecutor::execute (job_executor.rs:266-317, 413-436) around the in-process encoder.
    Returns (rc, [stage strings seen], JobMetrics, error text)."""
    j = ExecJob()
    j.id = os.fspath(job_id).encode()
    j.input_path = os.fspath(input_path).encode()
    j.output_path = os.fspath(output_path).encode()
    j.temp_base_dir = os.fspath(temp_base_dir).encode()
    j.workers = workers
    j.params = default_params(8, 8, 8, cq_level=cq_level, **enc)
    stages = []
    cb = STATE_CB(lambda u, s, m: stages.append(s.decode()) if not stages or stages[-1] != s.decode() else None)
    m = JobMetrics()
    err = C.create_string_buffer(256)
    rc = _lib.av1mi_job_execute(C.byref(j), cb, None, C.byref(m), err, 256)
    return rc, stages, m, err.value.decode()


def check_layout():
    """Refuse a library whose ABI structures are not the ones this mirror describes (a field added to av1mi_params and forgotten here
    would otherwise corrupt memory silently).  Runs at import."""
    if int(_lib.av1mi_abi_version()) != ABI_VERSION:
        raise ImportError("libav1mi.so has ABI version %d, this mirror is written for %d" % (int(_lib.av1mi_abi_version()), ABI_VERSION))
    lib_sizes, mine = struct_sizes(), mirror_sizes()
    if lib_sizes != mine:
        raise ImportError("libav1mi.so structure layout %s differs from the ctypes mirror %s" % (lib_sizes, mine))


check_layout()
