// av1mi_file.cpp - av1mi_encode_file(): the in-process replacement of
//   pub fn run_av1an(params: &Av1anEncodeParams) -> Result<(), EncodeError>
// (/root/reference/crates/daemon/src/encode/av1an.rs:126-139), including the parts av1an does
// around the codec: splitting the clip into independent chunks (each starts with a key frame +
// sequence header; `--workers`/`--temp`, av1an.rs:100-104), running `workers` chunks in flight
// (here: one context per visible GPU, round-robin) and concatenating chunk streams in order
// (§8e).  Input is Y4M; output is Matroska (.mkv/.webm, the reference's job output,
// jobs.rs:187-188), IVF or a bare OBU stream by extension (input demux/decode via ffmpeg is out of scope, §8b).  Output is written to a temporary name and renamed, so the
// caller's `exists && len > 0` validation (job_executor.rs:296-317) never sees a partial file.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include "../../include/av1mi.h"

namespace {

struct Y4m {
  FILE *f = nullptr;
  uint32_t w = 0, h = 0, bd = 8, fps_n = 30, fps_d = 1;
  int color_range = -1;       // XCOLORRANGE tag: 0 LIMITED, 1 FULL, -1 absent
  size_t frame_bytes = 0;
  uint64_t total_frames = 0;  // from the file size (plain "FRAME\n" markers); 0 = unknown (pipe)
  long data_off = 0;          // file offset of the first FRAME marker
  uint64_t next_frame = 0;    // frames handed out so far (regular files: position of the next frame)
  bool regular = false;       // seekable regular file: frames can be read in parallel with pread
};

int y4m_open(const char *path, Y4m *y) {
  y->f = fopen(path, "rb");
  if (!y->f) return -errno;
  char line[512];
  size_t n = 0;
  int ch;
  while ((ch = fgetc(y->f)) != EOF && ch != '\n' && n < sizeof(line) - 1) line[n++] = (char)ch;
  line[n] = 0;
  if (strncmp(line, "YUV4MPEG2", 9) != 0) return AV1MI_E_FORMAT;
  bool ok420 = true;
  for (char *tok = strtok(line + 9, " "); tok; tok = strtok(nullptr, " ")) {
    switch (tok[0]) {
      case 'W': y->w = (uint32_t)atoi(tok + 1); break;
      case 'H': y->h = (uint32_t)atoi(tok + 1); break;
      case 'F': sscanf(tok + 1, "%u:%u", &y->fps_n, &y->fps_d); break;
      case 'X':
        if (!strcmp(tok, "XCOLORRANGE=FULL")) y->color_range = 1;
        else if (!strcmp(tok, "XCOLORRANGE=LIMITED")) y->color_range = 0;
        break;
      case 'C':
        if (strncmp(tok + 1, "420p10", 6) == 0) y->bd = 10;
        else if (strncmp(tok + 1, "420", 3) == 0 && (tok[4] == 0 || tok[4] == 'j' || tok[4] == 'm' || (tok[4] == 'p' && tok[5] == 'a'))) y->bd = 8;
        else ok420 = false;
        break;
      default: break;
    }
  }
  if (!ok420 || !y->w || !y->h) return AV1MI_E_FORMAT;
  if (!y->fps_n || !y->fps_d) { y->fps_n = 30; y->fps_d = 1; }
  y->frame_bytes = (size_t)y->w * y->h * 3 / 2 * (y->bd > 8 ? 2 : 1);
  {  // frame count of a regular file: every frame is "FRAME\n" + frame_bytes (frame parameters would only make this an over-estimate)
    const long pos = ftell(y->f);
    struct stat sb;
    if (pos >= 0 && fstat(fileno(y->f), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > pos) {
      // the parallel reader knows where every frame starts only if the markers are the plain six bytes; a clip whose first marker
      // carries frame parameters ("FRAME Ixx\n") goes through the sequential reader, frame count unknown
      char mark[6];
      const bool plain = pread(fileno(y->f), mark, 6, (off_t)pos) == 6 && memcmp(mark, "FRAME\n", 6) == 0;
      if (plain) {
        y->total_frames = (uint64_t)(sb.st_size - pos) / (6 + y->frame_bytes);
        y->data_off = pos;
        y->regular = true;
      }
    }
  }
  return 0;
}

// returns 1 frame read, 0 clean EOF, <0 error
int y4m_read_frame(Y4m *y, uint8_t *dst) {
  char hdr[128];
  size_t n = 0;
  int ch;
  while ((ch = fgetc(y->f)) != EOF && ch != '\n' && n < sizeof(hdr) - 1) hdr[n++] = (char)ch;
  if (ch == EOF && n == 0) return 0;
  hdr[n] = 0;
  if (strncmp(hdr, "FRAME", 5) != 0) return AV1MI_E_FORMAT;
  if (fread(dst, 1, y->frame_bytes, y->f) != y->frame_bytes) return AV1MI_E_FORMAT;
  return 1;
}

void put_le(uint8_t *p, uint64_t v, int n) { for (int i = 0; i < n; i++) p[i] = (uint8_t)(v >> (8 * i)); }

// ---- output containers --------------------------------------------------------------------
// The reference's job output is `{temp_output_dir}/{uuid}.mkv` (crates/daemon/src/jobs.rs:187-188), muxed by
// av1an/mkvmerge.  The container is chosen by the extension of output_path: .mkv/.webm -> Matroska (EBML header,
// Segment{Info, Tracks{V_AV1 + av1C CodecPrivate}, one Cluster of SimpleBlocks per chunk}), .obu -> bare Section-5
// OBU stream, anything else -> IVF.  Video only: the Y4M input has no audio to pass through.
struct Muxer {
  FILE *fo = nullptr;
  enum Kind { IVF, MKV, OBU } kind = IVF;
  uint32_t w = 0, h = 0, fps_n = 30, fps_d = 1;
  uint64_t frames = 0, bytes = 0;
  long seg_size_pos = 0, seg_data_pos = 0, dur_pos = 0;
  std::vector<uint8_t> cluster;     // SimpleBlocks of the open cluster
  uint64_t cluster_t0 = 0;          // its timestamp, ms
  bool cluster_open = false;

  static void ebml_id(std::vector<uint8_t> &v, uint32_t id) { for (int sh = 24; sh >= 0; sh -= 8) if (id >> sh) v.push_back((uint8_t)(id >> sh)); }
  static void ebml_size(std::vector<uint8_t> &v, uint64_t n) {  // 8-byte form: 0x01 + 56 bits
    v.push_back(0x01);
    for (int sh = 48; sh >= 0; sh -= 8) v.push_back((uint8_t)(n >> sh));
  }
  static void ebml_elem(std::vector<uint8_t> &v, uint32_t id, const std::vector<uint8_t> &payload) { ebml_id(v, id); ebml_size(v, payload.size()); v.insert(v.end(), payload.begin(), payload.end()); }
  static std::vector<uint8_t> be(uint64_t x, int n) { std::vector<uint8_t> v; for (int i = n - 1; i >= 0; i--) v.push_back((uint8_t)(x >> (8 * i))); return v; }
  static std::vector<uint8_t> str(const char *s) { return std::vector<uint8_t>(s, s + strlen(s)); }
  uint64_t ts_ms(uint64_t frame) const { return frame * 1000ull * fps_d / (fps_n ? fps_n : 30); }

  void begin(const char *path, const std::vector<uint8_t> &seq_hdr_obu) {
    const char *dot = strrchr(path, '.');
    kind = (dot && (!strcmp(dot, ".mkv") || !strcmp(dot, ".webm"))) ? MKV : ((dot && !strcmp(dot, ".obu")) ? OBU : IVF);
    if (kind == IVF) {
      uint8_t ivf[32] = { 'D', 'K', 'I', 'F', 0, 0, 32, 0, 'A', 'V', '0', '1' };
      put_le(ivf + 12, w, 2); put_le(ivf + 14, h, 2); put_le(ivf + 16, fps_n, 4); put_le(ivf + 20, fps_d, 4);
      fwrite(ivf, 1, 32, fo);
      bytes = 32;
    } else if (kind == MKV) {
      std::vector<uint8_t> hdr, e;
      ebml_elem(e, 0x4286, be(1, 1)); ebml_elem(e, 0x42F7, be(1, 1)); ebml_elem(e, 0x42F2, be(4, 1)); ebml_elem(e, 0x42F3, be(8, 1));
      ebml_elem(e, 0x4282, str(dot && !strcmp(dot, ".webm") ? "webm" : "matroska")); ebml_elem(e, 0x4287, be(4, 1)); ebml_elem(e, 0x4285, be(2, 1));
      ebml_elem(hdr, 0x1A45DFA3, e);
      ebml_id(hdr, 0x18538067);  // Segment, size patched at the end
      fwrite(hdr.data(), 1, hdr.size(), fo);
      seg_size_pos = ftell(fo);
      uint8_t unk[8] = { 0x01, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF };
      fwrite(unk, 1, 8, fo);
      seg_data_pos = ftell(fo);
      std::vector<uint8_t> info, ib;
      ebml_elem(ib, 0x2AD7B1, be(1000000, 4));              // TimestampScale: 1 ms
      ebml_elem(ib, 0x4D80, str("libav1mi")); ebml_elem(ib, 0x5741, str("libav1mi"));
      ebml_id(ib, 0x4489); ebml_size(ib, 8);                 // Duration (float64 ms), patched at the end
      const size_t dur_off = ib.size();
      for (int i = 0; i < 8; i++) ib.push_back(0);
      ebml_elem(info, 0x1549A966, ib);
      const long info_pos = ftell(fo);
      fwrite(info.data(), 1, info.size(), fo);
      dur_pos = info_pos + (long)(info.size() - ib.size() + dur_off);
      // Tracks: one video track, CodecPrivate = AV1CodecConfigurationRecord (marker/version, profile/level, flags) + configOBUs
      std::vector<uint8_t> tracks, te, vid, priv;
      priv.push_back(0x81); priv.push_back((uint8_t)((0 << 5) | 31));  // seq_profile 0, seq_level_idx 31 (as in the sequence header)
      priv.push_back((uint8_t)(((bit_depth > 8) << 6) | (1 << 3) | (1 << 2)));  // high_bitdepth, 4:2:0
      priv.push_back(0);
      priv.insert(priv.end(), seq_hdr_obu.begin(), seq_hdr_obu.end());
      ebml_elem(vid, 0xB0, be(w, 2)); ebml_elem(vid, 0xBA, be(h, 2));
      if (cp | tc | mc) {
        // Colour (Matroska): what the sequence header's colour description says, for players that read the container first
        std::vector<uint8_t> col;
        ebml_elem(col, 0x55B1, be(mc ? mc : 2, 1));            // MatrixCoefficients
        ebml_elem(col, 0x55B2, be(bit_depth, 1));              // BitsPerChannel
        ebml_elem(col, 0x55B3, be(1, 1)); ebml_elem(col, 0x55B4, be(1, 1));   // ChromaSubsamplingHorz / Vert: 4:2:0
        ebml_elem(col, 0x55B9, be(full_range ? 2 : 1, 1));     // Range: 1 = broadcast, 2 = full
        ebml_elem(col, 0x55BA, be(tc ? tc : 2, 1));            // TransferCharacteristics
        ebml_elem(col, 0x55BB, be(cp ? cp : 2, 1));            // Primaries
        ebml_elem(vid, 0x55B0, col);
      }
      ebml_elem(te, 0xD7, be(1, 1)); ebml_elem(te, 0x73C5, be(1, 1)); ebml_elem(te, 0x83, be(1, 1)); ebml_elem(te, 0x9C, be(0, 1));
      ebml_elem(te, 0x86, str("V_AV1")); ebml_elem(te, 0x63A2, priv);
      ebml_elem(te, 0x23E383, be(1000000000ull * fps_d / (fps_n ? fps_n : 30), 4));  // DefaultDuration, ns
      ebml_elem(te, 0xE0, vid);
      std::vector<uint8_t> tentry;
      ebml_elem(tentry, 0xAE, te);
      ebml_elem(tracks, 0x1654AE6B, tentry);
      fwrite(tracks.data(), 1, tracks.size(), fo);
      bytes = (uint64_t)ftell(fo);
    }
  }
  void flush_cluster() {
    if (!cluster_open) return;
    std::vector<uint8_t> c, body;
    ebml_elem(body, 0xE7, be(cluster_t0, 4));
    body.insert(body.end(), cluster.begin(), cluster.end());
    ebml_elem(c, 0x1F43B675, body);
    fwrite(c.data(), 1, c.size(), fo);
    bytes += c.size();
    cluster.clear();
    cluster_open = false;
  }
  // one temporal unit (TD + [sequence header] + frame); `key`: starts a new cluster in Matroska
  void write_frame(const uint8_t *tu, uint32_t size, bool key) {
    if (kind == IVF) {
      uint8_t fh[12];
      put_le(fh, size, 4); put_le(fh + 4, frames, 8);
      fwrite(fh, 1, 12, fo); fwrite(tu, 1, size, fo);
      bytes += 12 + size;
    } else if (kind == OBU) {
      fwrite(tu, 1, size, fo);
      bytes += size;
    } else {
      const uint64_t t = ts_ms(frames);
      if (key || !cluster_open || t - cluster_t0 > 30000) { flush_cluster(); cluster_open = true; cluster_t0 = t; }
      // SimpleBlock: track 1, int16 relative timestamp, flags (0x80 = key frame); the temporal delimiter is dropped
      const uint8_t *p = tu; uint32_t n = size;
      if (n >= 2 && p[0] == 0x12 && p[1] == 0x00) { p += 2; n -= 2; }
      ebml_id(cluster, 0xA3); ebml_size(cluster, 4 + (uint64_t)n);
      cluster.push_back(0x81);
      const int16_t rel = (int16_t)(t - cluster_t0);
      cluster.push_back((uint8_t)(rel >> 8)); cluster.push_back((uint8_t)rel);
      cluster.push_back(key ? 0x80 : 0x00);
      cluster.insert(cluster.end(), p, p + n);
    }
    frames++;
  }
  void end() {
    if (kind == IVF) {
      if (fseek(fo, 24, SEEK_SET) == 0) { uint8_t cnt[4]; put_le(cnt, frames, 4); fwrite(cnt, 1, 4, fo); }
    } else if (kind == MKV) {
      flush_cluster();
      const long endpos = ftell(fo);
      std::vector<uint8_t> sz;
      ebml_size(sz, (uint64_t)(endpos - seg_data_pos));
      if (fseek(fo, seg_size_pos, SEEK_SET) == 0) fwrite(sz.data(), 1, 8, fo);
      const double dur = (double)ts_ms(frames);
      uint64_t bits;
      memcpy(&bits, &dur, 8);
      std::vector<uint8_t> d = be(bits, 8);
      if (fseek(fo, dur_pos, SEEK_SET) == 0) fwrite(d.data(), 1, 8, fo);
      fseek(fo, endpos, SEEK_SET);
    }
  }
  uint32_t bit_depth = 8;
  uint32_t cp = 0, tc = 0, mc = 0;   // colour description of the job (av1mi_params), 0 = none
  bool full_range = false;
};

// ---- host staging (BASELINE config 5 "per-GPU multi-stream overlap"; SURVEY.md §8e limiter "PCIe H2D of source frames") -------
// Chunks are read straight into PINNED host buffers (hipHostMalloc, portable across the job's GPUs) from a small pool: a
// context's upload of its next chunk is then a true asynchronous DMA at PCIe rate that runs beside the kernels of the other
// contexts on the same GPU (several contexts per GPU by default), instead of the runtime staging pageable memory through its
// own bounce buffer.  A slot goes back to the pool as soon as its chunk is encoded.
struct PinnedPool {
  struct Slot { uint8_t *p = nullptr; size_t cap = 0; bool busy = false, pinned = false; };
  std::vector<Slot> slots;
  std::mutex mu;
  std::condition_variable cv;
  int device = 0;
  // Page-locked bytes this job may hold (AV1MI_PIN_BUDGET_MB, default 24 GiB - what the process-wide cache keeps anyway): slots
  // beyond it, and slots the runtime refuses to pin, are plain memory (the encode still works, their upload is staged by the runtime)
  size_t pin_budget = (size_t)24 << 30, pinned_bytes = 0;
  void init(int n, int dev);   // takes idle buffers from the process-wide cache
  void retire();               // gives them back
  Slot *acquire(size_t bytes) {
    std::unique_lock<std::mutex> lk(mu);
    Slot *s = nullptr;
    cv.wait(lk, [&] { for (auto &x : slots) if (!x.busy) { s = &x; return true; } return false; });
    s->busy = true;
    lk.unlock();
    if (s->cap < bytes) {
      release_mem(s);
      const size_t want = bytes + (bytes >> 3);
      bool may_pin;
      { std::lock_guard<std::mutex> lk2(mu); may_pin = pinned_bytes + want <= pin_budget; if (may_pin) pinned_bytes += want; }
      s->pinned = may_pin && hipSetDevice(device) == hipSuccess && hipHostMalloc((void **)&s->p, want, hipHostMallocPortable) == hipSuccess;
      if (!s->pinned) {
        if (may_pin) { (void)hipGetLastError(); std::lock_guard<std::mutex> lk2(mu); pinned_bytes -= want; }
        s->p = (uint8_t *)malloc(want);
      }
      s->cap = s->p ? want : 0;
    }
    return s;
  }
  void release(Slot *s) {
    { std::lock_guard<std::mutex> lk(mu); s->busy = false; }
    cv.notify_all();
  }
  void release_mem(Slot *s) {
    if (!s->p) return;
    if (s->pinned) { (void)hipHostFree(s->p); std::lock_guard<std::mutex> lk2(mu); pinned_bytes -= s->cap < pinned_bytes ? s->cap : pinned_bytes; } else free(s->p);
    s->p = nullptr; s->cap = 0; s->pinned = false;
  }
  ~PinnedPool() { retire(); }
};

// ---- process-wide caches ---------------------------------------------------------------------------------------------------
// The daemon is a long-lived process that encodes job after job (JobExecutor::execute, job_executor.rs:266-437): creating
// contexts (streams, several GB of HBM workspace each) and pinning host buffers per job cost more than a short clip's whole
// encode (measured: 4 contexts 50 ms, 6 pinned 373 MB slots ~60 ms to allocate and ~60 ms to free, against 18 ms of GPU work
// for 240 1080p frames).  Idle contexts and pinned slots are therefore kept between calls - "the per-GPU context owns its
// frame pool"  - bounded, and released by av1mi_release_caches().
struct GlobalCache {
  std::mutex mu;
  std::vector<std::pair<int, av1mi_ctx *>> ctxs;   // idle contexts (device, context)
  std::vector<PinnedPool::Slot> slots;             // idle host buffers
  size_t slot_bytes = 0;
};
GlobalCache &cache() { static GlobalCache *g = new GlobalCache(); return *g; }   // never destroyed: no HIP calls at process exit
const size_t kMaxCachedCtxPerDev = 8, kMaxCachedSlotBytes = (size_t)24 << 30;

void PinnedPool::init(int n, int dev) {
  slots.resize((size_t)n);
  device = dev;
  GlobalCache &g = cache();
  std::lock_guard<std::mutex> lk(g.mu);
  for (auto &s : slots) {
    if (g.slots.empty()) break;
    s = g.slots.back();
    s.busy = false;
    g.slot_bytes -= s.cap;
    if (s.pinned) pinned_bytes += s.cap;
    g.slots.pop_back();
  }
  if (const char *e = getenv("AV1MI_PIN_BUDGET_MB")) { const long mb = atol(e); if (mb >= 0) pin_budget = (size_t)mb << 20; }
}
void PinnedPool::retire() {
  GlobalCache &g = cache();
  for (auto &s : slots) {
    if (!s.p) continue;
    bool kept = false;
    {
      std::lock_guard<std::mutex> lk(g.mu);
      if (s.pinned && g.slot_bytes + s.cap <= kMaxCachedSlotBytes) { g.slots.push_back(s); g.slot_bytes += s.cap; kept = true; }
    }
    if (!kept) release_mem(&s);
    s.p = nullptr; s.cap = 0;
  }
  slots.clear();
}

extern "C" void av1mi_host_ctx_recycle(av1mi_ctx *c);
extern "C" void av1mi_host_expect_blocks(unsigned n);
int take_ctx(int dev, av1mi_ctx **out) {
  {
    GlobalCache &g = cache();
    std::lock_guard<std::mutex> lk(g.mu);
    for (size_t i = 0; i < g.ctxs.size(); i++)
      if (g.ctxs[i].first == dev) { *out = g.ctxs[i].second; g.ctxs.erase(g.ctxs.begin() + (long)i); return AV1MI_OK; }
  }
  return av1mi_ctx_create(dev, out);
}
void give_ctx(int dev, av1mi_ctx *c) {
  if (!c) return;
  av1mi_host_ctx_recycle(c);
  GlobalCache &g = cache();
  {
    std::lock_guard<std::mutex> lk(g.mu);
    size_t n = 0;
    for (auto &e : g.ctxs) n += e.first == dev;
    if (n < kMaxCachedCtxPerDev) { g.ctxs.emplace_back(dev, c); return; }
  }
  av1mi_ctx_destroy(c);
}

// Reads frames [first, first + count) of a Y4M into dst (tight I420, frame after frame).  Regular files: the frames sit at
// known offsets, so up to `threads` readers pread() them in parallel - one thread copies page cache to the destination at a few
// GB/s, which at 6 MB per 1080p 10-bit frame is ~1 k frames/s, a tenth of what one GPU encodes.  Pipes: sequential fread.
// Returns the number of whole frames read (short at end of file); *err receives a format / IO error.
uint32_t y4m_read_frames(Y4m *y, uint8_t *dst, uint32_t count, int threads, int *err) {
  *err = 0;
  if (!y->regular) {
    uint32_t n = 0;
    int r = 1;
    while (n < count && (r = y4m_read_frame(y, dst + (size_t)n * y->frame_bytes)) == 1) n++;
    if (r != 0 && r != 1) *err = r;
    return n;
  }
  const uint64_t left = y->total_frames > y->next_frame ? y->total_frames - y->next_frame : 0;
  uint32_t n = (uint32_t)(left < count ? left : count);
  if (n == 0) {   // a trailing partial frame is a malformed file, not a clean end
    struct stat sb;
    if (fstat(fileno(y->f), &sb) == 0 && (uint64_t)sb.st_size > (uint64_t)y->data_off + y->next_frame * (6 + y->frame_bytes)) *err = AV1MI_E_FORMAT;
    return 0;
  }
  const int fd = fileno(y->f);
  const uint64_t first = y->next_frame;
  std::atomic<uint32_t> next(0);
  std::atomic<int> bad(0);
  auto work = [&] {
    for (;;) {
      const uint32_t k = next.fetch_add(1);
      if (k >= n || bad.load()) return;
      const off_t off = (off_t)((uint64_t)y->data_off + (first + k) * (6 + y->frame_bytes));
      char mark[6];
      if (pread(fd, mark, 6, off) != 6 || memcmp(mark, "FRAME\n", 6) != 0) { bad.store(AV1MI_E_FORMAT); return; }   // frame parameters are not supported
      uint8_t *d = dst + (size_t)k * y->frame_bytes;
      size_t got = 0;
      while (got < y->frame_bytes) {
        const ssize_t r = pread(fd, d + got, y->frame_bytes - got, off + 6 + (off_t)got);
        if (r <= 0) { bad.store(r < 0 ? -errno : AV1MI_E_FORMAT); return; }
        got += (size_t)r;
      }
    }
  };
  const int nt = threads < 1 ? 1 : ((uint32_t)threads > n ? (int)n : threads);
  std::vector<std::thread> th;
  for (int i = 1; i < nt; i++) th.emplace_back(work);
  work();
  for (auto &t : th) t.join();
  if (bad.load()) { *err = bad.load(); return 0; }
  y->next_frame += n;
  // a clip whose size is not a whole number of frames: the tail shows up as an error once the whole frames are consumed
  return n;
}

struct Chunk {
  uint32_t index = 0, n_frames = 0, first_frame = 0;
  PinnedPool::Slot *slot = nullptr;   // the chunk's frames: slot->p, pinned
  av1mi_buf out = { nullptr, 0 };
  std::vector<uint32_t> sizes;
  av1mi_report rep = {};
  int rc = 0;
};

}  // namespace

// ---- chunk -> GPU placement: the two rules everything multi-GPU in this build goes through ----------------
extern "C" uint32_t av1mi_chunk_owner(uint32_t chunk_index, uint32_t n_owners) { return n_owners ? chunk_index % n_owners : 0u; }

extern "C" int av1mi_plan_workers(uint32_t workers, int32_t gpu_mask, int n_devices, int32_t *device_of_worker, uint32_t cap) {
  if (n_devices <= 0) return 0;
  int devs[64], nd = 0;
  for (int d = 0; d < n_devices && d < 32 && nd < 64; d++) if (gpu_mask <= 0 || ((gpu_mask >> d) & 1)) devs[nd++] = d;
  if (nd == 0) return 0;
  // default: AV1MI_DEFAULT_WORKERS_PER_GPU chunks in flight per allowed GPU - a chunk's P-frame chain is latency bound, chains of
  // different chunks overlap (the reference runs `--workers 8` on one host, concurrency.rs:67-73)
  uint32_t w = workers ? workers : (uint32_t)nd * AV1MI_DEFAULT_WORKERS_PER_GPU;
  if (w > 64) w = 64;
  for (uint32_t i = 0; i < w && i < cap; i++) if (device_of_worker) device_of_worker[i] = devs[av1mi_chunk_owner(i, (uint32_t)nd)];
  return (int)w;
}

extern "C" void av1mi_host_release_streams(void);
extern "C" void av1mi_host_release_buffers(void);
extern "C" void av1mi_release_caches(void) {
  GlobalCache &g = cache();
  std::vector<std::pair<int, av1mi_ctx *>> cs;
  std::vector<PinnedPool::Slot> ss;
  {
    std::lock_guard<std::mutex> lk(g.mu);
    cs.swap(g.ctxs);
    ss.swap(g.slots);
    g.slot_bytes = 0;
  }
  for (auto &e : cs) av1mi_ctx_destroy(e.second);
  for (auto &s : ss) if (s.p) { if (s.pinned) (void)hipHostFree(s.p); else free(s.p); }
  av1mi_host_release_streams();   // the stream sets destroyed contexts left in av1mi_host.cpp's pool
  av1mi_host_release_buffers();   // and the page-locked bitstream blocks av1mi_free() put back
}

extern "C" int av1mi_probe_y4m(const char *path, av1mi_clip_info *info) {
  if (!path || !info) return AV1MI_E_INVALID_ARG;
  Y4m y;
  const int rc = y4m_open(path, &y);
  if (y.f) fclose(y.f);
  if (rc) return rc;
  info->width = y.w; info->height = y.h; info->bit_depth = y.bd; info->fps_num = y.fps_n; info->fps_den = y.fps_d;
  info->color_range = y.color_range > 0 ? 1u : 0u;
  info->frames = y.total_frames;
  return AV1MI_OK;
}

extern "C" int av1mi_encode_file(const av1mi_job *job, av1mi_progress_cb cb, void *user, av1mi_report *total) {
  if (!job || !job->input_path || !job->output_path) return AV1MI_E_INVALID_ARG;
  const bool timing = getenv("AV1MI_TIMING") != nullptr;   // phase times on stderr (diagnostics)
  const auto tt0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (timing) fprintf(stderr, "[av1mi_encode_file] %8.2f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count(), what);
  };
  Y4m y;
  int rc = y4m_open(job->input_path, &y);
  if (rc) { if (y.f) fclose(y.f); return rc; }
  av1mi_params prm = job->params;
  prm.width = y.w; prm.height = y.h; prm.bit_depth = y.bd;
  if (y.color_range >= 0) prm.color_range = (uint32_t)y.color_range;  // the clip's own tag wins over the job's default
  const bool scene_mode = job->chunk_frames == 0;
  const uint32_t chunk_frames = job->chunk_frames ? job->chunk_frames : 60;
  // contexts: `workers` chunks in flight, spread round-robin over the allowed GPUs
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fclose(y.f); return AV1MI_E_NO_DEVICE; }
  int32_t dev_of[64];
  const int planned = av1mi_plan_workers(job->workers, job->gpu_mask, ndev, dev_of, 64);
  if (planned <= 0) { fclose(y.f); return AV1MI_E_NO_DEVICE; }
  const uint32_t workers = (uint32_t)planned;
  std::vector<av1mi_ctx *> ctxs(workers, nullptr);
  for (uint32_t i = 0; i < workers; i++) {
    rc = take_ctx(dev_of[i], &ctxs[i]);
    if (rc) { for (uint32_t k = 0; k < i; k++) give_ctx(dev_of[k], ctxs[k]); fclose(y.f); return rc; }
  }
  std::vector<int> ctx_dev(dev_of, dev_of + workers);
  auto give_all = [&] { for (size_t k = 0; k < ctxs.size(); k++) give_ctx(ctx_dev[k], ctxs[k]); ctxs.clear(); };
  lap("contexts created");
  av1mi_ctx *det_ctx = nullptr;  // the reader thread's own context for the scene-cut pass
  if (scene_mode) {
    rc = take_ctx(dev_of[0], &det_ctx);
    if (rc) { give_all(); fclose(y.f); return rc; }
    ctxs.push_back(det_ctx);  // returned with the others; gets no worker thread
    ctx_dev.push_back(dev_of[0]);
  }
  std::string tmp = std::string(job->output_path) + ".tmp." + std::to_string((long)getpid());
  FILE *fo = fopen(tmp.c_str(), "wb");
  if (!fo) { int e = -errno; give_all(); fclose(y.f); return e; }
  Muxer mux;
  mux.fo = fo; mux.w = y.w; mux.h = y.h; mux.fps_n = y.fps_n; mux.fps_d = y.fps_d; mux.bit_depth = y.bd;
  mux.cp = prm.color_primaries; mux.tc = prm.transfer_characteristics; mux.mc = prm.matrix_coefficients; mux.full_range = prm.color_range != 0;
  {
    uint8_t sh[64]; size_t shn = sizeof(sh);
    rc = av1mi_write_headers(&prm, sh, &shn, nullptr, nullptr);
    if (rc) { fclose(fo); unlink(tmp.c_str()); give_all(); fclose(y.f); return rc; }
    mux.begin(job->output_path, std::vector<uint8_t>(sh, sh + shn));
  }

  // pinned staging: one slot being filled by the reader, one waiting, one per worker being uploaded / encoded
  PinnedPool pool;
  pool.init((int)workers + 2, dev_of[0]);
  av1mi_host_expect_blocks(2 * workers + 2);   // finished chunks waiting for the writer keep their page-locked output blocks
  int hc = 0;
  { cpu_set_t cs; if (sched_getaffinity(0, sizeof(cs), &cs) == 0) hc = CPU_COUNT(&cs); }   // the cores this process may use, not the host's
  if (hc <= 0) hc = (int)std::thread::hardware_concurrency();
  int read_threads = hc >= 16 ? 12 : (hc >= 8 ? 6 : (hc >= 4 ? 3 : 1));
  if (const char *e = getenv("AV1MI_READ_THREADS")) { const int k = atoi(e); if (k > 0 && k <= 64) read_threads = k; }
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<Chunk *> queue;
  std::map<uint32_t, Chunk *> done;
  bool eof = false;
  int first_err = 0;
  auto worker = [&](av1mi_ctx *ctx) {
    for (;;) {
      Chunk *ck = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return !queue.empty() || eof; });
        if (queue.empty()) return;
        ck = queue.front();
        queue.pop_front();
      }
      ck->sizes.resize(ck->n_frames);
      av1mi_params cp = prm;
      cp.first_frame = prm.first_frame + ck->first_frame;
      ck->rc = av1mi_encode_chunk(ctx, &cp, ck->slot->p, ck->n_frames, 0, &ck->out, ck->sizes.data(), nullptr, &ck->rep);
      if (timing) fprintf(stderr, "[av1mi_encode_file] %8.2f ms  chunk %u encoded (h2d %.2f total %.2f ms on the device)\n",
                          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count(), ck->index, ck->rep.ms_h2d, ck->rep.ms_total);
      pool.release(ck->slot);   // (av1mi_encode_chunk returns after its upload has completed)
      ck->slot = nullptr;
      {
        std::lock_guard<std::mutex> lk(mu);
        done[ck->index] = ck;
      }
      cv_done.notify_all();
    }
  };
  std::vector<std::thread> threads;
  for (auto c : ctxs) if (c != det_ctx) threads.emplace_back(worker, c);

  const auto t0 = std::chrono::steady_clock::now();
  uint32_t n_chunks = 0, next_write = 0, frames_done = 0, frames_read = 0;
  uint64_t bytes_out = mux.bytes;
  const uint32_t keyint = prm.keyint ? prm.keyint : 1;
  av1mi_report tot = {};
  auto drain = [&](bool all) {
    for (;;) {
      Chunk *ck = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        if (all) cv_done.wait(lk, [&] { return next_write >= n_chunks || done.count(next_write); });
        if (next_write >= n_chunks || !done.count(next_write)) return;
        ck = done[next_write];
        done.erase(next_write);
      }
      if (ck->rc && !first_err) first_err = ck->rc;
      if (!ck->rc && !first_err) {
        size_t off = 0;
        for (uint32_t f = 0; f < ck->n_frames; f++) {
          mux.write_frame(ck->out.data + off, ck->sizes[f], f % keyint == 0);
          off += ck->sizes[f];
        }
        bytes_out = mux.bytes;
        frames_done += ck->n_frames;
        tot.frames += ck->n_frames; tot.bytes += ck->rep.bytes; tot.n_symbols += ck->rep.n_symbols;
        tot.gpus_used |= ck->rep.gpus_used;
        for (int p = 0; p < 3; p++) tot.sse[p] += ck->rep.sse[p];
        tot.ms_recon += ck->rep.ms_recon; tot.ms_cdef += ck->rep.ms_cdef; tot.ms_entropy += ck->rep.ms_entropy;
        tot.ms_pack += ck->rep.ms_pack; tot.ms_h2d += ck->rep.ms_h2d; tot.ms_d2h += ck->rep.ms_d2h; tot.ms_total += ck->rep.ms_total;
        if (cb) {
          double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          // total: the clip's length when the input is a regular file, else the frames read so far
          const uint32_t tot_frames = y.total_frames >= frames_read ? (uint32_t)y.total_frames : frames_read;
          cb(user, frames_done, tot_frames, sec > 0 ? frames_done / sec : 0.0, bytes_out);
        }
      }
      av1mi_free(ck->out.data);
      delete ck;
      next_write++;
    }
  };
  auto enqueue = [&](Chunk *ck) {
    // Bound the chunks held in memory: at most 2 * workers read but not yet written.  Finished chunks are written
    // by THIS thread, so while waiting it must keep draining - waiting only for "fewer outstanding" would deadlock
    // once every outstanding chunk is finished and parked in `done`.
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return n_chunks - next_write < workers * 2 || done.count(next_write); });
        if (n_chunks - next_write < workers * 2) {
          ck->index = n_chunks;
          ck->first_frame = (uint32_t)frames_read;
          frames_read += ck->n_frames;
          n_chunks++;
          queue.push_back(ck);
          break;
        }
      }
      drain(false);
    }
    cv_work.notify_one();
    drain(false);
  };
  if (!scene_mode) {
    // reader: fixed-length chunks (every chunk starts with a key frame)
    for (;;) {
      Chunk *ck = new Chunk();
      ck->slot = pool.acquire((size_t)chunk_frames * y.frame_bytes);
      if (!ck->slot->p) { pool.release(ck->slot); delete ck; if (!first_err) first_err = AV1MI_E_OOM; break; }
      int rerr = 0;
      ck->n_frames = y4m_read_frames(&y, ck->slot->p, chunk_frames, read_threads, &rerr);
      if (rerr && !first_err) first_err = rerr;  // a malformed or truncated frame (AV1MI_E_FORMAT) or an I/O error
      if (ck->n_frames == 0) { pool.release(ck->slot); delete ck; break; }
      lap("chunk read");
      enqueue(ck);
      if (first_err) break;
    }
  } else {
    // reader: chunks end at scene cuts (GPU luma-SAD pass per window of frames, include/av1mi.h: av1mi_scene_cuts)
    const uint32_t WIN = 32, MIN_SCENE = 12;
    // a chunk's frames plus the window being examined stay under 4 GiB of pinned host memory
    const uint64_t cap = ((uint64_t)4 << 30) / y.frame_bytes;
    const uint32_t max_len = (uint32_t)(cap < MIN_SCENE + WIN ? MIN_SCENE : (cap - WIN > 240 ? 240 : cap - WIN));
    const size_t slot_bytes = (size_t)(max_len + WIN) * y.frame_bytes;
    std::vector<uint8_t> prev(y.frame_bytes);
    std::vector<uint8_t> cuts(WIN);
    bool has_prev = false, oom = false;
    av1mi_scene_state st = {};
    // Windows are read straight behind the current chunk's frames in its pinned slot and examined there; only where a chunk
    // ends inside a window do the frames after the cut move to the next chunk's slot.
    Chunk *cur = new Chunk();
    cur->slot = pool.acquire(slot_bytes);
    oom = !cur->slot->p;
    while (!oom && !first_err) {
      uint8_t *win = cur->slot->p + (size_t)cur->n_frames * y.frame_bytes;
      int rerr = 0;
      uint32_t nw = y4m_read_frames(&y, win, WIN, read_threads, &rerr);
      if (rerr && !first_err) first_err = rerr;  // a malformed or truncated frame (AV1MI_E_FORMAT) or an I/O error
      if (nw == 0) break;
      const int src = av1mi_scene_cuts(det_ctx, &prm, win, nw, 0, has_prev ? prev.data() : nullptr, &st, MIN_SCENE, nullptr, cuts.data());
      if (src) { if (!first_err) first_err = src; break; }
      memcpy(prev.data(), win + (size_t)(nw - 1) * y.frame_bytes, y.frame_bytes);
      has_prev = true;
      for (uint32_t t = 0; t < nw; t++) {
        if ((cuts[t] && cur->n_frames > 0) || cur->n_frames == max_len) {
          // frame t starts the next chunk: it and the rest of the window move to a slot of their own
          Chunk *nx = new Chunk();
          nx->slot = pool.acquire(slot_bytes);
          if (!nx->slot->p) { pool.release(nx->slot); delete nx; oom = true; break; }
          memcpy(nx->slot->p, cur->slot->p + (size_t)cur->n_frames * y.frame_bytes, (size_t)(nw - t) * y.frame_bytes);
          enqueue(cur);
          cur = nx;
        }
        cur->n_frames++;   // (the frame already sits in place: the window was read behind the chunk's frames)
      }
    }
    if (oom && !first_err) first_err = AV1MI_E_OOM;
    if (cur->n_frames > 0 && !first_err) enqueue(cur); else { if (cur->slot) pool.release(cur->slot); delete cur; }
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    eof = true;
  }
  cv_work.notify_all();
  lap("input consumed");
  drain(true);
  lap("all chunks written");
  for (auto &t : threads) t.join();
  if (first_err) { for (auto c : ctxs) av1mi_ctx_destroy(c); ctxs.clear(); }   // a context that saw a failure is not reused
  give_all();
  pool.retire();
  lap("contexts and host buffers back in the cache");
  fclose(y.f);
  // patch frame count, finish atomically
  int io_err = 0;
  mux.end();
  if (fflush(fo) != 0 || ferror(fo)) io_err = -EIO;
  fclose(fo);
  if (!first_err && !io_err && frames_done == 0) first_err = AV1MI_E_FORMAT;
  if (first_err || io_err) { unlink(tmp.c_str()); return first_err ? first_err : io_err; }
  if (rename(tmp.c_str(), job->output_path) != 0) { int e = -errno; unlink(tmp.c_str()); return e; }
  if (total) {
    const double mx = (double)((1 << y.bd) - 1);
    for (int p = 0; p < 3; p++) {
      const double npx = (double)tot.frames * y.w * y.h / (p ? 4 : 1);
      tot.psnr[p] = tot.sse[p] > 0 ? 10.0 * log10(mx * mx * npx / tot.sse[p]) : 99.0;
    }
    tot.chunks = n_chunks;
    *total = tot;
  }
  return AV1MI_OK;
}
